"""Composite launches: a batch is cut into blocks with a kernel family each (launch_plan.cpp plan_blocks), so that one
utterance more than a family holds does not cost a whole further round of it.  Exact arithmetic is mapping-invariant:
whatever the cut, every sample must equal the oracle's bit pattern.  The machine is made small with
"assume_compute_units" (1 CU = 256 lanes), so that batches of a few hundred short utterances are cut the way batches
of 70 000 are on the whole device."""
import ctypes as C

import numpy as np
import pytest

import grail_hip as G
import oracle_lib as O
from grail_hip import workload as W

pytestmark = pytest.mark.gpu
ULP = 2.0 ** -23


def _ovoices(voices):
    return [O.Voice.from_buffer_copy(bytes(v)) for v in voices]


def _ragged(n_utt, n_voices, seed=5):
    """Utterances of 1 - 6 segments of 8 - 40 ms with random blend lengths of 2^-k s: lengths differ by 10 x."""
    rng = np.random.default_rng(seed)
    segs, offs, _, _ = W.make_batch(n_utt, n_voices=n_voices, segments=6, length=0.02, blend_length=2.0 ** -6)
    segs = segs.reshape(n_utt, 6)
    keep = rng.integers(1, 7, size=n_utt)
    segs["length"] = rng.uniform(0.008, 0.04, size=segs.shape).astype(np.float32)
    segs["blend_length"] = (2.0 ** -rng.integers(5, 8, size=segs.shape)).astype(np.float32)
    flat, o = [], [0]
    for u in range(n_utt):
        flat.append(segs[u, :keep[u]])
        o.append(o[-1] + int(keep[u]))
    ids = np.arange(n_utt, dtype=np.uint32)
    return np.concatenate(flat), np.array(o, dtype=np.uint32), ids % np.uint32(n_voices), ids * np.uint32(31) + np.uint32(7)


def _device_render(ctx, segs, offs, vids, seeds, stride, pcm16=False):
    """Through the asynchronous entry points (device buffers): what bench.py and a Rust caller use."""
    n = len(offs) - 1
    batch = ctx.upload(segs, offs, vids, seeds)
    item = 2 if pcm16 else 4
    d_out = ctx.device_alloc(n * stride * item)
    d_len = ctx.device_alloc(n * 4)
    try:
        ctx.memset(d_out, 0, n * stride * item)
        if pcm16:
            batch.synthesize_pcm16_async(d_out, stride, d_len)
        else:
            batch.synthesize_async(d_out, stride, d_len)
        status = G.OK
        try:
            ctx.sync()
        except G.GrailError as e:
            status = e.status
        out = np.zeros((n, stride), dtype=np.int16 if pcm16 else np.float32)
        lens = np.zeros(n, dtype=np.uint32)
        ctx.d2h(out, d_out, out.nbytes)
        ctx.d2h(lens, d_len, lens.nbytes)
        return out, lens, status
    finally:
        ctx.device_free(d_out)
        ctx.device_free(d_len)
        batch.free()


def _bit_identical(out, lens, ref, ref_len, what):
    assert np.array_equal(lens, ref_len), f"{what}: lengths differ at {np.flatnonzero(lens != ref_len)[:8]}"
    for u in range(len(lens)):
        n = int(lens[u])
        a, b = out[u, :n].view(np.uint32), ref[u, :n].view(np.uint32)
        assert np.array_equal(a, b), f"{what}: utterance {u} first differs at sample {int(np.argmax(a != b))}"


@pytest.fixture
def small_machine(gpu_ctx):
    def set_cus(c, ragged_plan=0):
        gpu_ctx.set_option("assume_compute_units", c)
        # (these tests are about the cut by size: ragged batches keep it; the plan by the rows' lengths has its own test)
        gpu_ctx.set_option("ragged_plan", ragged_plan)
    yield set_cus
    gpu_ctx.set_option("assume_compute_units", 0)
    gpu_ctx.set_option("ragged_plan", 1)
    gpu_ctx.set_option("composite_launches", 1)
    gpu_ctx.set_option("sort_by_length", 1)
    gpu_ctx.set_option("arithmetic", 0)
    gpu_ctx.set_voices(W.single_voice())


@pytest.mark.parametrize("n_voices", [1, 8])
@pytest.mark.parametrize("cus,n_utt", [(1, 257), (1, 300), (1, 341), (1, 600), (2, 700), (1, 1100)])
def test_composite_exact_is_bit_identical_to_the_oracle(gpu_ctx, small_machine, cus, n_utt, n_voices):
    """1 CU holds 256 utterances on one lane each, 128 / 64 on two / four, 32 / 16 in pipelined workgroups: 257, 300,
    341 ... are one round of the one-lane kernel plus a rest on a wider mapping."""
    voices = W.single_voice() if n_voices == 1 else W.preset_voices(8)
    gpu_ctx.set_voices(voices)
    small_machine(cus)
    segs, offs, vids, seeds = W.make_batch(n_utt, n_voices=n_voices, length=0.03, blend_length=2.0 ** -5)
    stride = W.max_samples(length=0.03)
    out, lens, status = _device_render(gpu_ctx, segs, offs, vids, seeds, stride)
    blocks = gpu_ctx.get_option("last_launch_blocks")
    assert status == G.OK and blocks >= 2, blocks
    ref, ref_len = O.synthesize_batch(_ovoices(voices), segs, offs, vids, seeds, stride)
    _bit_identical(out, lens, ref, ref_len, f"cus={cus} n={n_utt} blocks={blocks}")
    # the cut is the one the pure planner predicts (aligned batch, default options)
    span = int(np.ceil(4 * np.float32(0.03) * 48000.0)) + 0
    plan = G.plan_blocks(n_utt, span, 0, 4 if n_voices == 1 else 8, compute_units=cus)
    assert len(plan) == blocks
    # and switching the composite launches off gives the same bits in one launch
    gpu_ctx.set_option("composite_launches", 0)
    one, one_len, _ = _device_render(gpu_ctx, segs, offs, vids, seeds, stride)
    assert gpu_ctx.get_option("last_launch_blocks") == 1
    _bit_identical(one, one_len, ref, ref_len, "single launch")


@pytest.mark.parametrize("sort", [1, 0])
def test_composite_ragged_batch_with_the_length_sorted_slot_order(gpu_ctx, small_machine, sort):
    """Ragged utterances: the host hands out launch slots longest first (sort_by_length) and the blocks are ranges of
    SLOTS; rows, lengths and seeds stay the utterance's own."""
    voices = W.preset_voices(8)
    gpu_ctx.set_voices(voices)
    small_machine(1)
    gpu_ctx.set_option("sort_by_length", sort)
    n_utt = 431
    segs, offs, vids, seeds = _ragged(n_utt, 8)
    stride = 12288
    out, lens, status = _device_render(gpu_ctx, segs, offs, vids, seeds, stride)
    assert status == G.OK and gpu_ctx.get_option("last_launch_blocks") >= 2
    ref, ref_len = O.synthesize_batch(_ovoices(voices), segs, offs, vids, seeds, stride)
    assert ref_len.max() > 5 * max(int(ref_len.min()), 1)
    _bit_identical(out, lens, ref, ref_len, f"ragged sort={sort}")
    # the one-call host form (a single row block) takes the same path
    host, host_len = gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
    assert gpu_ctx.get_option("last_launch_blocks") >= 2
    _bit_identical(host, host_len, ref, ref_len, "host form")


def test_composite_lengths_and_truncation_are_reported_across_blocks(gpu_ctx, small_machine):
    """Rows that do not fit out_stride are cut and reported whichever block they sit in; untouched tails stay."""
    voices = W.single_voice()
    gpu_ctx.set_voices(voices)
    small_machine(1)
    gpu_ctx.set_option("sort_by_length", 0)            # blocks in batch order: rows 0..255 | 256..
    n_utt = 300
    segs, offs, vids, seeds = _ragged(n_utt, 1, seed=11)
    full_ref, full_len = O.synthesize_batch(_ovoices(voices), segs, offs, vids, seeds, 12288)
    stride = int(np.sort(full_len)[n_utt // 2]) // 64 * 64          # about half of the rows do not fit
    out, lens, status = _device_render(gpu_ctx, segs, offs, vids, seeds, stride)
    assert gpu_ctx.get_option("last_launch_blocks") >= 2
    assert status == G.ERR_BUFFER_TOO_SMALL
    want = np.minimum(full_len, stride)
    assert np.array_equal(lens, want)
    cut = np.flatnonzero(full_len > stride)
    assert (cut < 256).any() and (cut >= 256).any()          # truncated rows in both blocks
    for u in range(n_utt):
        m = int(want[u])
        assert np.array_equal(out[u, :m].view(np.uint32), full_ref[u, :m].view(np.uint32)), u
        assert not out[u, m:].any()                           # nothing written past a row's end
    # a batch that fits reports OK again (the flag does not stick)
    out, lens, status = _device_render(gpu_ctx, segs, offs, vids, seeds, 12288)
    assert status == G.OK and np.array_equal(lens, full_len)


def test_composite_is_batch_invariant(gpu_ctx, small_machine):
    """utterance u's samples do not depend on N or on its position — here: on the block it lands in."""
    voices = W.preset_voices(8)
    gpu_ctx.set_voices(voices)
    small_machine(1)
    stride = 12288
    segs, offs, vids, seeds = _ragged(700, 8, seed=3)
    big, big_len, _ = _device_render(gpu_ctx, segs, offs, vids, seeds, stride)
    assert gpu_ctx.get_option("last_launch_blocks") >= 2
    for first, n in ((0, 40), (250, 20), (500, 200), (699, 1)):
        lo, hi = int(offs[first]), int(offs[first + n])
        osub = (offs[first:first + n + 1] - offs[first]).astype(np.uint32)
        part, part_len, _ = _device_render(gpu_ctx, segs[lo:hi], osub, vids[first:first + n], seeds[first:first + n], stride)
        assert np.array_equal(part_len, big_len[first:first + n])
        for r in range(n):
            m = int(part_len[r])
            assert np.array_equal(part[r, :m].view(np.uint32), big[first + r, :m].view(np.uint32)), (first, r)


def test_composite_pcm16_rows(gpu_ctx, small_machine):
    """i16 rows (examples/cli.rs:49 fused into the flush) through a composite launch: the conversion of the oracle's rows."""
    voices = W.single_voice()
    gpu_ctx.set_voices(voices)
    small_machine(1)
    n_utt = 333
    segs, offs, vids, seeds = _ragged(n_utt, 1, seed=21)
    stride = 12288
    pcm, lens, status = _device_render(gpu_ctx, segs, offs, vids, seeds, stride, pcm16=True)
    assert status == G.OK and gpu_ctx.get_option("last_launch_blocks") >= 2
    ref, ref_len = O.synthesize_batch(_ovoices(voices), segs, offs, vids, seeds, stride)
    assert np.array_equal(lens, ref_len)
    L = O.lib()
    conv = np.vectorize(lambda x: L.orc_pcm16(C.c_float(x)), otypes=[np.int16])
    for u in range(0, n_utt, 7):
        m = int(lens[u])
        assert np.array_equal(pcm[u, :m], conv(ref[u, :m])), u
        assert not pcm[u, m:].any()


@pytest.mark.parametrize("n_voices", [1, 8])
def test_composite_fast_mode_stays_within_the_tolerance(gpu_ctx, small_machine, n_voices):
    """Fast arithmetic: a head on the fast one-lane kernels and a rest on whatever suits its size (time-split chunks,
    the scan kernel); every row within GRAIL_FAST_TOLERANCE of the oracle, lengths identical."""
    voices = W.single_voice() if n_voices == 1 else W.preset_voices(8)
    gpu_ctx.set_voices(voices)
    small_machine(1)
    gpu_ctx.set_option("arithmetic", 1)
    n_utt = 300
    segs, offs, vids, seeds = W.make_batch(n_utt, n_voices=n_voices, length=0.25, blend_length=0.25)
    stride = W.max_samples(length=0.25)
    out, lens, status = _device_render(gpu_ctx, segs, offs, vids, seeds, stride)
    assert status == G.OK
    assert gpu_ctx.get_option("last_launch_blocks") >= 2 and gpu_ctx.get_option("last_launch_fast") == 1
    ref, ref_len = O.synthesize_batch(_ovoices(voices), segs, offs, vids, seeds, stride)
    assert np.array_equal(lens, ref_len)
    worst = 0.0
    for u in range(n_utt):
        m = int(lens[u])
        d = float(np.max(np.abs(out[u, :m].astype(np.float64) - ref[u, :m].astype(np.float64))))
        worst = max(worst, d / max(1.0, float(np.max(np.abs(ref[u, :m])))))
    print(f"composite fast vs oracle, voices={n_voices}: {worst / ULP:.1f} * 2^-23")
    assert 0.0 < worst <= G.FAST_TOLERANCE


def test_policy_follows_the_compute_unit_count_and_parity_stays(gpu_ctx, small_machine):
    """the L / family choice scales with the CU count (a CPX partition has 32), bits do not move."""
    voices = W.single_voice()
    gpu_ctx.set_voices(voices)
    n_utt = 2048
    segs, offs, vids, seeds = W.make_batch(n_utt, length=0.01, blend_length=2.0 ** -7)
    stride = W.max_samples(length=0.01)
    ref, ref_len = O.synthesize_batch(_ovoices(voices), segs, offs, vids, seeds, stride)
    assert gpu_ctx.get_option("compute_units") >= 32
    seen = {}
    for cus in (0, 64, 32, 16, 8, 2):
        small_machine(cus)
        out, lens, status = _device_render(gpu_ctx, segs, offs, vids, seeds, stride)
        assert status == G.OK
        _bit_identical(out, lens, ref, ref_len, f"cus={cus}")
        seen[cus] = (gpu_ctx.get_option("last_launch_pipelined"), gpu_ctx.get_option("last_launch_lanes"),
                     gpu_ctx.get_option("last_launch_blocks"))
    print(seen)
    assert seen[0][0] == 1                       # the whole device: pipelined workgroups (2048 <= 16 per CU)
    assert seen[64] == (1, 4, 1)                 # 64 CUs: 32 per CU, still pipelined (rounds of 16)
    assert seen[32] == (0, 4, 1)                 # 32 CUs: four lanes per utterance fill 128 SIMDs exactly
    assert seen[16] == (0, 2, 1)                 # 16 CUs: two lanes
    assert seen[8] == (0, 1, 1)                  # 8 CUs: one lane each, one round
    assert seen[2][1] == 1 and seen[2][2] == 1   # 2 CUs: four whole rounds of the one-lane kernel, nothing left over


def test_create_reports_the_device_and_refuses_nothing_on_gfx950(gpu_ctx):
    cus = gpu_ctx.get_option("compute_units")
    assert cus in (32, 64, 128, 256) or cus > 0
    assert gpu_ctx.get_option("assume_compute_units") == 0


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_composite_fuzz_over_machine_sizes_batch_sizes_and_arithmetics(gpu_ctx, small_machine, seed):
    """Random compute-unit counts (powers of two or not: every capacity is a multiple of the count), random batch sizes
    around the families' capacities, aligned or ragged, one voice or eight: exact arithmetic bit for bit, fast
    arithmetic within the tolerance, lengths always the oracle's; the cut is the one grail_plan_blocks predicts."""
    rng = np.random.default_rng(7000 + seed)
    cus = int(rng.choice([1, 2, 3, 5, 6]))
    lanes = 256 * cus
    n_utt = int(rng.choice([lanes + 1, lanes + int(rng.integers(2, 40)), lanes // 2 + int(rng.integers(1, 30)),
                            lanes + lanes // 4 + 3, 2 * lanes + int(rng.integers(1, 20)), int(rng.integers(5, 3 * lanes))]))
    n_voices = int(rng.choice([1, 8]))
    voices = W.single_voice() if n_voices == 1 else W.preset_voices(8)
    gpu_ctx.set_voices(voices)
    small_machine(cus)
    if rng.random() < 0.5:
        segs, offs, vids, seeds = _ragged(n_utt, n_voices, seed=seed)
        stride = 12288
    else:
        segs, offs, vids, seeds = W.make_batch(n_utt, n_voices=n_voices, length=0.04, blend_length=2.0 ** -5)
        stride = W.max_samples(length=0.04)
    ref, ref_len = O.synthesize_batch(_ovoices(voices), segs, offs, vids, seeds, stride)
    out, lens, status = _device_render(gpu_ctx, segs, offs, vids, seeds, stride)
    blocks = gpu_ctx.get_option("last_launch_blocks")
    assert status == G.OK
    _bit_identical(out, lens, ref, ref_len, f"cus={cus} n={n_utt} voices={n_voices} blocks={blocks}")
    gpu_ctx.set_option("arithmetic", 1)
    fast, flens, status = _device_render(gpu_ctx, segs, offs, vids, seeds, stride)
    fblocks = gpu_ctx.get_option("last_launch_blocks")
    assert status == G.OK and np.array_equal(flens, ref_len)
    worst = 0.0
    for u in range(n_utt):
        m = int(flens[u])
        if m:
            d = float(np.max(np.abs(fast[u, :m].astype(np.float64) - ref[u, :m].astype(np.float64))))
            worst = max(worst, d / max(1.0, float(np.max(np.abs(ref[u, :m])))))
    print(f"composite fuzz seed {seed}: cus={cus} n={n_utt} voices={n_voices}: {blocks} exact / {fblocks} fast launches, "
          f"fast {worst / ULP:.1f} * 2^-23")
    assert worst <= G.FAST_TOLERANCE
    assert max(blocks, fblocks) >= 2 or n_utt <= lanes


@pytest.mark.parametrize("fast", [0, 1])
def test_a_few_odd_rows_do_not_decide_the_kernels_of_all(gpu_ctx, fast):
    """20 000 ordinary utterances and a handful the lean kernel families cannot take — a segment of one sample, a
    segment of length zero, a NaN pitch, an infinite pitch, a blend length of zero: those rows go last in the slot order
    and are planned as a batch of their own (grail_batch::groups), the others keep the four-formant kernels (in fast mode: the
    fast ones).  Lengths and samples: the oracle's, odd rows included."""
    voices = W.single_voice()
    gpu_ctx.set_voices(voices)
    n_utt = 20000
    segs, offs, vids, seeds = W.make_batch(n_utt, length=0.02, blend_length=2.0 ** -6)
    odd = {17: ("length", 1.0 / 48000.0), 4000: ("length", 0.0), 9999: ("frequency", float("nan")),
           12345: ("frequency", float("inf")), 19999: ("blend_length", 0.0)}
    for u, (field, value) in odd.items():
        segs[field][offs[u] + 1] = value
    stride = W.max_samples(length=0.02)
    b = gpu_ctx.upload(segs, offs, vids, seeds)
    d_out = gpu_ctx.device_alloc(n_utt * stride * 4)
    d_len = gpu_ctx.device_alloc(n_utt * 4)
    gpu_ctx.set_option("arithmetic", fast)
    try:
        b.synthesize_async(d_out, stride, d_len)
        gpu_ctx.sync()
        name, formants, blocks = gpu_ctx.last_kernel_name(), gpu_ctx.get_option("last_launch_formants"), gpu_ctx.get_option("last_launch_blocks")
        out_len = np.zeros(n_utt, dtype=np.uint32)
        gpu_ctx.d2h(out_len, d_len, n_utt * 4)
        rows = sorted(set(list(odd) + [0, 1, 16, 18, 3999, 4001, 10000, 19998] + list(range(5000, 5016))))
        out = np.zeros((len(rows), stride), dtype=np.float32)
        for i, u in enumerate(rows):
            gpu_ctx.d2h(out[i], d_out, stride * 4, offset=u * stride * 4)
    finally:
        gpu_ctx.set_option("arithmetic", 0)
        gpu_ctx.device_free(d_out)
        gpu_ctx.device_free(d_len)
        b.free()
    assert formants == 4 and blocks >= 2, (name, formants, blocks)
    if fast:
        assert "FAST" in name, name
    sub_offs = np.zeros(len(rows) + 1, dtype=np.uint32)
    sub = []
    for i, u in enumerate(rows):
        sub.append(segs[offs[u]:offs[u + 1]])
        sub_offs[i + 1] = sub_offs[i] + offs[u + 1] - offs[u]
    with np.errstate(all="ignore"):
        ref, ref_len = O.synthesize_batch(_ovoices(voices), np.concatenate(sub), sub_offs, vids[rows], seeds[rows], stride)
    ref_len = np.minimum(ref_len, stride)
    for i, u in enumerate(rows):
        n = int(ref_len[i])
        assert out_len[u] == n, (u, int(out_len[u]), n)
        a, r = out[i, :n], ref[i, :n]
        if fast and u not in odd:
            assert float(np.abs(a.astype(np.float64) - r).max(initial=0.0)) <= G.FAST_TOLERANCE * max(1.0, float(np.abs(r).max(initial=0.0))), u
        elif not fast:
            both_nan = np.isnan(a) & np.isnan(r)
            x, y = a.view(np.uint32).copy(), r.view(np.uint32).copy()
            x[both_nan] = 0
            y[both_nan] = 0
            assert np.array_equal(x, y), (u, int(np.argmax(x != y)))


def _speech_like(n_utt, n_voices, seed=13):
    """Utterances of 4 - 24 phonemes of 5 - 20 ms (blends of 4 - 10 ms, any length; pitches of 90 - 220 Hz): lengths
    differ by a factor of ten and no two utterances have a segment boundary at the same time."""
    rng = np.random.default_rng(seed)
    counts = rng.integers(4, 25, n_utt)
    offs = np.zeros(n_utt + 1, dtype=np.uint32)
    offs[1:] = np.cumsum(counts)
    k = int(offs[-1])
    segs = np.zeros(k, dtype=G.PHONEME_DTYPE)
    segs["phoneme"] = rng.choice([G.PH_A, G.PH_E, G.PH_SILENCE, G.PH_STOP], k, p=[.4, .4, .12, .08])
    segs["phoneme"][offs[:-1]] = G.PH_SILENCE
    segs["length"] = rng.uniform(0.005, 0.02, k).astype(np.float32)
    segs["blend_length"] = rng.uniform(0.004, 0.01, k).astype(np.float32)
    segs["frequency"] = (rng.uniform(90, 220, k) / 48000.0).astype(np.float32)
    ids = np.arange(n_utt, dtype=np.uint32)
    return segs, offs, ids % np.uint32(n_voices), ids * np.uint32(17) + np.uint32(3)


@pytest.mark.parametrize("fast", [0, 1])
@pytest.mark.parametrize("n_voices", [1, 8])
def test_ragged_batches_take_wider_mappings_in_several_rounds(gpu_ctx, small_machine, n_voices, fast):
    """A machine's worth of utterances (4 CUs: 1 024) that differ in length by a factor of ten.  Laid out for one wave per
    SIMD (one lane per utterance) the launch lasts as long as its longest utterance; option "ragged_plan" weighs that
    against wider mappings in several rounds by the rows' lengths and events and takes one of them (launch_plan.cpp,
    "Ragged batches"; on the whole device: profiles/r04_ragged_plan.txt).  Exact: the oracle's bits whichever way;
    fast: within the tolerance — or, where events this dense make an exact mapping the cheaper one, the oracle's bits."""
    voices = W.single_voice() if n_voices == 1 else W.preset_voices(8)
    gpu_ctx.set_voices(voices)
    n_utt = 1024
    segs, offs, vids, seeds = _speech_like(n_utt, n_voices)
    stride = 24 * 960 + 128
    ref, ref_len = O.synthesize_batch(_ovoices(voices), segs, offs, vids, seeds, stride)
    assert ref_len.max() < stride and ref_len.max() > 6 * ref_len.min()
    gpu_ctx.set_option("arithmetic", fast)
    seen = {}
    for ragged_plan in (0, 1):
        small_machine(4, ragged_plan)
        out, lens, status = _device_render(gpu_ctx, segs, offs, vids, seeds, stride)
        seen[ragged_plan] = (gpu_ctx.get_option("last_launch_lanes"), gpu_ctx.get_option("last_launch_blocks"),
                             gpu_ctx.last_kernel_name())
        assert status == G.OK and np.array_equal(lens, ref_len)
        if gpu_ctx.get_option("last_launch_fast"):
            worst = max(float(np.abs(out[u, :lens[u]].astype(np.float64) - ref[u, :lens[u]]).max()) /
                        max(1.0, float(np.abs(ref[u, :lens[u]]).max())) for u in range(n_utt))
            assert fast and worst <= G.FAST_TOLERANCE, (ragged_plan, worst / ULP)
        else:
            # (fast arithmetic asked for, events this dense: the planner may take an exact mapping — cheaper by its model,
            # and exact bits satisfy the tolerance trivially)
            assert not fast or ragged_plan, seen
            _bit_identical(out, lens, ref, ref_len, f"ragged_plan={ragged_plan}")
    assert seen[0][0] == 1 or "SPLIT" in seen[0][2], seen          # one round: a lane (or a chunk lane) per utterance
    assert seen[1][0] in (2, 4, 8) and seen[1][1] == 1, seen       # several rounds of a wider mapping, one launch


def test_one_odd_row_does_not_cost_a_ragged_corpus_its_plan(gpu_ctx, small_machine):
    """the per-granule summary ragged_plan() needs used to be built only for batches WITHOUT row groups, so a
    single row the lean kernels cannot take (a zero-length segment) sent a whole speech-like corpus back to the one-round
    plan.  The rows of the first group now carry their own summary and are planned like the corpus without the odd row:
    same lane mapping, the odd row in a launch of its own, the oracle's bits for all."""
    voices = W.single_voice()
    gpu_ctx.set_voices(voices)
    n_utt = 1024
    segs, offs, vids, seeds = _speech_like(n_utt, 1)
    stride = 24 * 960 + 128
    small_machine(4, 1)
    _device_render(gpu_ctx, segs, offs, vids, seeds, stride)
    want_lanes = gpu_ctx.get_option("last_launch_lanes")
    assert want_lanes in (2, 4, 8)
    odd = segs.copy()
    odd["length"][offs[77] + 1] = 0.0                    # (one segment of one utterance: not "at least two samples long")
    ref, ref_len = O.synthesize_batch(_ovoices(voices), odd, offs, vids, seeds, stride)
    out, lens, status = _device_render(gpu_ctx, odd, offs, vids, seeds, stride)
    assert status == G.OK and np.array_equal(lens, ref_len)
    assert gpu_ctx.get_option("last_launch_blocks") >= 2            # the odd row apart
    assert gpu_ctx.get_option("last_launch_lanes") == want_lanes     # ... and the corpus on the plan it had without it
    _bit_identical(out, lens, ref, ref_len, "one odd row")
