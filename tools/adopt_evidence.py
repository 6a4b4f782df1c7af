import re, json, csv, sys, shutil
tag=sys.argv[1]
src=f'gpurun_out/{tag}/'
for f in "bench_n1.json bench_n1_config2.json bench_n1_config4.json bench_n1_fast.json bench_n1_mid.json bench_n1_pcm16.json bench_n1_verify.json full_parity.txt host_output.txt kernel_stats.csv kernel_stats_config2.csv kernel_stats_config4.csv kernel_stats_mid.csv mid_bench.txt pmc_sq.txt ragged.txt ragged_131072.txt ragged_fast.txt small_batch.txt stream_latency.txt tail_presets.txt".split():
    shutil.copy(src+f, 'profiles/r04_'+f)
shutil.copy(src+'traffic.json','profiles/traffic.json')
old=open('profiles/r04_elems_split.txt').read().split('\n')
hdr=[l for l in old if l.startswith('#')]
new=open(src+'elems_split.txt').read().rstrip('\n').split('\n')
open('profiles/r04_elems_split.txt','w').write('\n'.join(hdr+new)+'\n')
old=open('profiles/r04_tail.txt').read().split('\n')
new=open(src+'tail.txt').read().rstrip('\n').split('\n')
i=[k for k,l in enumerate(old) if l.strip()=='#'][0]
comm='\n'.join(old[i:])
def row(n):
    for l in new:
        m=re.match(r'n=\s*%d\s+exact\s+([\d.]+) ms.*?\(single launch\s+([\d.]+)\)\s+fast\s+([\d.]+) ms.*?\(single launch\s+([\d.]+)\)'%n, l)
        if m: return [float(x) for x in m.groups()]
r={n:row(n) for n in (36000,40000,65536,65537,70000,98304,131073,200000)}
a_start=comm.index("# 65 537 utterances "); a_end=comm.index("# The composite launches sit")
newread=("# 65 537 utterances %.1f ms (%.1f as one launch), 70 000: %.1f (%.1f), 98 304: %.1f (%.1f), 131 073: %.1f (%.1f: three\n"
 "# rounds); fast 70 000: %.1f (%.1f), 40 000: %.1f (%.1f: the gap between half a machine and a whole one).  (This table is\n"
 "# from the box of the round's last collection, %.1f ms for the headline batch; boxes of the pool differ by +-4 %%: an\n"
 "# earlier collection of the round had 51.8 / 56.6 / 22.6 on a 41.2-ms box.)\n#\n") % (
 r[65537][0], r[65537][1], r[70000][0], r[70000][1], r[98304][0], r[98304][1], r[131073][0], r[131073][1],
 r[70000][2], r[70000][3], r[40000][2], r[40000][3], r[65536][0])
comm=comm[:a_start]+newread+comm[a_end:]
open('profiles/r04_tail.txt','w').write('\n'.join(new)+'\n'+comm)
B={}
for f in ['bench_n1','bench_n1_config2','bench_n1_config4','bench_n1_fast','bench_n1_mid','bench_n1_pcm16']:
    d=json.load(open(f'profiles/r04_{f}.json')); fm=d.get('fast_mode') or {}
    B[f]=dict(ms=d['ms_per_step'], value=d['value'], frac=d['roofline']['frac'], fast_ms=fm.get('ms_per_step'), fast_frac=(fm.get('roofline') or {}).get('frac'), fast_value=fm.get('value'), sha=d['roofline']['kernel_source_sha'])
K={}
for f in ['kernel_stats','kernel_stats_config4','kernel_stats_mid','kernel_stats_config2']:
    rows=list(csv.DictReader(open(f'profiles/r04_{f}.csv')))
    K[f]=[round(float(x['AverageNs'])/1e6,2) for x in rows[:2]]
print(B); print(K); print(r)
f1=lambda x:'%.1f'%x
p='DESIGN.md'; s=open(p).read()
def sub(pattern, repl, count=1):
    global s
    s2,n=re.subn(pattern, repl, s, count=count, flags=re.S)
    assert n>=1, pattern
    s=s2
sub(r"\*\*[\d.]+\*\* in the committed line \(`r04_bench_n1\.json`; rocprofv3 average of 6 launches of the same command, ramp launch included: [\d.]+, `r04_kernel_stats\.csv`\) \| 588 – 626 \| \*\*7\.4 – 7\.8 %\*\* \([\d.]+\) \|",
    "**%.2f** in the committed line (`r04_bench_n1.json`; rocprofv3 average of 6 launches of the same command, ramp launch included: %.2f, `r04_kernel_stats.csv`) | 588 – 626 | **7.4 – 7.8 %%** (%.4f) |" % (B['bench_n1']['ms'], K['kernel_stats'][0], B['bench_n1']['frac']))
sub(r"\| \*\*15\.4 – 16\.6\*\*, [\d.]+ in the committed line \(rocprofv3 [\d.]+\) \| 1 520 – 1 640 \| \*\*19\.0 – 20\.4 %\*\* \([\d.]+\) \|",
    "| **15.4 – 16.6**, %.2f in the committed line (rocprofv3 %.2f) | 1 520 – 1 640 | **19.0 – 20.4 %%** (%.4f) |" % (B['bench_n1']['fast_ms'], K['kernel_stats'][1], B['bench_n1']['fast_frac']))
sub(r"\| 31\.5 – 32\.9 \([\d.]+; rocprofv3 [\d.]+\) \| \d+ \| [\d.]+ % \|",
    "| 31.5 – 32.9 (%.2f; rocprofv3 %.2f) | %d | %.1f %% |" % (B['bench_n1_mid']['ms'], K['kernel_stats_mid'][0], round(25230.3/B['bench_n1_mid']['ms']), 100*B['bench_n1_mid']['frac']))
sub(r"\| 76\.5 – 80 \(\*\*[\d.]+\*\*; rocprofv3 [\d.]+\) \| 315 – 330 \| [\d.]+ % \|",
    "| 76.5 – 80 (**%.2f**; rocprofv3 %.2f) | 315 – 330 | %.1f %% |" % (B['bench_n1_config4']['ms'], K['kernel_stats_config4'][0], 100*B['bench_n1_config4']['frac']))
sub(r"\| 23\.1 – 24\.6 \([\d.]+; rocprofv3 [\d.]+\) \| [\d ]+ \| [\d.]+ % \|",
    "| 23.1 – 24.6 (%.2f; rocprofv3 %.2f) | %s | %.1f %% |" % (B['bench_n1_config4']['fast_ms'], K['kernel_stats_config4'][1], format(round(25230.3/B['bench_n1_config4']['fast_ms']),',').replace(',',' '), 100*B['bench_n1_config4']['fast_frac']))
sub(r"\| \*\*6\.5 – 6\.7\*\* \([\d.]+\) \| \d+ \| 3\.0 % \|", "| **6.5 – 6.7** (%.2f) | 237 | 3.0 %% |" % B['bench_n1_config2']['ms'])
sub(r"\(\*\*[\d.]+\*\* in the committed line\)", "(**%.2f** in the committed line)" % B['bench_n1']['ms'])
sub(r"\| \*\*15\.4 – 16\.6 ms\*\* \([\d.]+\),", "| **15.4 – 16.6 ms** (%.2f)," % B['bench_n1']['fast_ms'])
sub(r"exact 76\.5 – 80 ms \([\d.]+\); fast 23\.1 – 24\.6 ms \([\d.]+\) = 2\.6e11 samples/s;", "exact 76.5 – 80 ms (%.2f); fast 23.1 – 24.6 ms (%.2f) = 2.6e11 samples/s;" % (B['bench_n1_config4']['ms'], B['bench_n1_config4']['fast_ms']))
rows={36000:"| 36 000 | %s (32 768 two lanes + 3 232 pipelined) | %s | %s | %s |",40000:"| 40 000 | %s | %s | %s (32 768 in 2 chunks + 7 232 in 9) | %s |",
65536:"| 65 536 | %s | %s | %s | %s |",65537:"| 65 537 | **%s** | %s | %s | %s |",70000:"| 70 000 | **%s** | %s | **%s** | %s |",98304:"| 98 304 | %s | %s | %s | %s |",131073:"| 131 073 | %s | %s | %s | %s |",200000:"| 200 000 | %s | %s | %s | %s |"}
for n,fmt in rows.items():
    label=fmt.split('|')[1]
    m=re.compile(r'^\|'+re.escape(label)+r'\|.*$', re.M).search(s); assert m, n
    s=s[:m.start()]+fmt%tuple(f1(x) for x in r[n])+s[m.end():]
i=s.index("65 537 utterances **"); j=s.index("The cut is predictable (`grail_plan_blocks`")
s=s[:i]+("65 537 utterances **%s ms** (%s as one launch), 70 000: **%s** (%s), 98 304: %s (%s), 131 073: %s (%s); fast 70 000: **%s** (%s), 40 000: %s (%s).  (The box of the committed table runs the headline batch in %s ms; a slower box of the round: 41.2 and 51.8 / 56.6 / 22.6.)  "
   % (f1(r[65537][0]),f1(r[65537][1]),f1(r[70000][0]),f1(r[70000][1]),f1(r[98304][0]),f1(r[98304][1]),f1(r[131073][0]),f1(r[131073][1]),f1(r[70000][2]),f1(r[70000][3]),f1(r[40000][2]),f1(r[40000][3]),f1(r[65536][0])))+s[j:]
i=s.index("| any batch size (§4; the box of the committed table"); j=s.index("8 presets 70 000", i)
s=s[:i]+("| any batch size (§4; the box of the committed table: %s ms for 65 536) | 36 000: %s ms exact / %s fast; 65 537: %s / %s; 70 000: %s / %s; 98 304: %s / %s; 131 073: %s / %s; 200 000: %s / %s (one launch: %s / %s; %s / %s; %s / %s; %s / %s; %s / %s; %s / %s); "
 % (f1(r[65536][0]), f1(r[36000][0]),f1(r[36000][2]), f1(r[65537][0]),f1(r[65537][2]), f1(r[70000][0]),f1(r[70000][2]), f1(r[98304][0]),f1(r[98304][2]), f1(r[131073][0]),f1(r[131073][2]), f1(r[200000][0]),f1(r[200000][2]),
    f1(r[36000][1]),f1(r[36000][3]), f1(r[65537][1]),f1(r[65537][3]), f1(r[70000][1]),f1(r[70000][3]), f1(r[98304][1]),f1(r[98304][3]), f1(r[131073][1]),f1(r[131073][3]), f1(r[200000][1]),f1(r[200000][3])))+s[j:]
open(p,'w').write(s)
p='profiles/README.md'; s=open(p).read()
sub(r"exact [\d.]+ ms/step = [\d.]+e11 samples/s, `roofline\.frac` [\d.]+", "exact %.2f ms/step = %.3fe11 samples/s, `roofline.frac` %.4f" % (B['bench_n1']['ms'], B['bench_n1']['value']/1e11, B['bench_n1']['frac']))
sub(r"`fast_mode` object [\d.]+ ms = [\d.]+e11 \([\d.]+\)", "`fast_mode` object %.2f ms = %.2fe11 (%.3f)" % (B['bench_n1']['fast_ms'], B['bench_n1']['fast_value']/1e11, B['bench_n1']['fast_frac']))
sub(r"exact [\d.]+ ms / fast [\d.]+ ms; config 4 \(eight live formants\) exact [\d.]+ / fast [\d.]+ ms;", "exact %.2f ms / fast %.2f ms; config 4 (eight live formants) exact %.2f / fast %.2f ms;" % (B['bench_n1_config2']['ms'], B['bench_n1_config2']['fast_ms'], B['bench_n1_config4']['ms'], B['bench_n1_config4']['fast_ms']))
sub(r"\(second tolerance tier: [\d.]+ ms = [\d.]+e11\)", "(second tolerance tier: %.2f ms = %.2fe11)" % (B['bench_n1_mid']['ms'], B['bench_n1_mid']['value']/1e11))
sub(r"exact headline kernel avg [\d.]+ ms over 6 launches \(ramp launch included\), fast [\d.]+; config 4 [\d.]+ / [\d.]+; tier 2 [\d.]+; config 2 [\d.]+ / [\d.]+",
    "exact headline kernel avg %.2f ms over 6 launches (ramp launch included), fast %.2f; config 4 %.2f / %.2f; tier 2 %.2f; config 2 %.2f / %.2f" % (K['kernel_stats'][0],K['kernel_stats'][1],K['kernel_stats_config4'][0],K['kernel_stats_config4'][1],K['kernel_stats_mid'][0],K['kernel_stats_config2'][0],K['kernel_stats_config2'][1]))
sub(r"kernel_source_sha [0-9a-f]{16}\)", "kernel_source_sha %s)" % B['bench_n1']['sha'])
open(p,'w').write(s)
