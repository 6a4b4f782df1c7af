#!/usr/bin/env python3
"""bench.py — audio samples/sec of the fused Sequencer->Jitter->Synthesize HIP path.

Metric (BASELINE.json): audio samples/sec (whole node) at 48 kHz, batch = 65536
utterances x 2 s (4 segments x 0.5 s, single Voice) per GPU — BASELINE config 3 at N=1.
A "step" is one pass of the hot path over one batch whose inputs already sit in HBM; the
f32 PCM stays in HBM (25.2 GB per GPU).  Weak scaling: every rank renders its own
65536-utterance shard of the N*65536 corpus (config 5 at N=8) with no data-path
collective; the voice table is broadcast once (RCCL ncclBroadcast inside the C ABI) before
the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--utts U] [--voices V]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
ALG_BYTES_PER_SAMPLE = 4.01  # 4 B f32 written + <= 0.01 B of segment/voice input
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic.json")
RCCL_TIMEOUT_S = float(os.environ.get("GRAIL_BENCH_RCCL_TIMEOUT", "240"))


def cpu_baseline(n_cpu, voices, W):
    """The oracle (a C port of the reference's single-threaded CPU path, oracle/) timed on this
    host on the first n_cpu utterances of the same corpus.  Checker code, never the product."""
    import oracle_lib as O
    ov = [O.Voice.from_buffer_copy(bytes(v)) for v in voices]
    segs, offs, vids, seeds = W.make_batch(n_cpu, n_voices=len(voices))
    stride = W.max_samples()
    O.synthesize_batch(ov, segs[:8], offs[:3], vids[:2], seeds[:2], stride)  # warm
    t0 = time.perf_counter()
    _, out_len = O.synthesize_batch(ov, segs, offs, vids, seeds, stride)
    dt = time.perf_counter() - t0
    n = int(out_len.astype(np.uint64).sum())
    return {
        "value": n / dt, "unit": "samples/s", "cores": 1, "kind": "port",
        "sample": f"first {n_cpu} utterances of the same corpus ({n} samples, {dt:.1f} s of "
                  f"oracle/liboracle.so on 1 thread; host has {os.cpu_count()} logical cores; "
                  f"the reference itself is single-threaded)",
    }


def cpu_all_cores(voices, W):
    """Same oracle, utterances fanned out over every host core this process may use (pthreads,
    one utterance per work item).  Reported beside cpu_baseline for scale only: the reference
    has no threads, so the 1-thread figure is "the reference CPU path" (SURVEY.md §8d)."""
    import oracle_lib as O
    threads = len(os.sched_getaffinity(0))
    n_cpu = min(8192, threads * 24)
    ov = [O.Voice.from_buffer_copy(bytes(v)) for v in voices]
    segs, offs, vids, seeds = W.make_batch(n_cpu, n_voices=len(voices))
    stride = W.max_samples()
    t0 = time.perf_counter()
    _, out_len, started = O.synthesize_batch_threads(ov, segs, offs, vids, seeds, stride, threads)
    dt = time.perf_counter() - t0
    n = int(out_len.astype(np.uint64).sum())
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota = f.read().strip()          # "max 100000" or "<quota> <period>"
    except OSError:
        quota = "unknown"
    granted = started
    try:                                  # "<quota> <period>": the container's CPU-time allowance
        q, per = quota.split()
        granted = min(started, max(1, -(-int(q) // int(per))))
    except ValueError:
        pass
    return {"value": n / dt, "unit": "samples/s", "cores": granted, "kind": "port",
            "sample": f"first {n_cpu} utterances ({n} samples) in {dt:.2f} s wall on {started} "
                      f"pthreads over {threads} visible logical cores; cgroup cpu.max = '{quota}' "
                      f"grants {granted} cores' worth of CPU time, which is what `cores` reports"}


def committed_literal_ms():
    """Kernel ms of the same batch with skip_silent_formants=0, from the committed run of
    `bench.py --literal` (profiles/r01_bench_n1_literal.json); `--literal` measures it live."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_bench_n1_literal.json")) as f:
            return json.load(f)["config"]["kernel_ms_all_formants_literal"]
    except (OSError, KeyError, ValueError):
        return None


def committed_valu(workload_key, kernel_ms):
    """Secondary roofline, the binding one: VALU instructions the kernel issued per launch
    (rocprofv3 --pmc SQ_INSTS_VALU, committed in profiles/traffic.json) over this run's kernel
    time, against one instruction per 4 cycles per SIMD — the rate of the packed f32 ops that
    make up two thirds of the loop (v_pk_{mul,add,fma}_f32; profiles/r01_valu_microbench.txt)."""
    try:
        with open(TRAFFIC_FILE) as f:
            insts = json.load(f).get("valu_insts", {}).get(workload_key)
    except OSError:
        insts = None
    if not insts:
        return None
    peak = 1024 * 2.4e9 / 4.0                     # 256 CUs x 4 SIMDs, 2.4 GHz, 4 cycles per issue
    rate = insts / (kernel_ms * 1e-3)
    return {"bound": "valu-issue", "achieved": rate / 1e9, "peak": peak / 1e9,
            "unit": "G wave-instructions/s", "frac": rate / peak,
            "valu_instructions_per_launch": insts,
            "note": "SQ_INSTS_VALU from profiles/r01_pmc_sq_final_L1.txt; a lone wave per SIMD issues "
                    "at most one instruction per ~5.3 cycles (measured), i.e. frac <= 0.75"}


def committed_traffic(workload_key):
    """HBM bytes per launch measured with rocprofv3 --pmc (separate WRITE_SIZE / FETCH_SIZE
    passes, gfx950 corrections applied) for the same command; see profiles/README.md."""
    try:
        with open(TRAFFIC_FILE) as f:
            return json.load(f).get(workload_key)
    except OSError:
        return None


class stdout_to_stderr:
    """RCCL prints a version banner on fd 1 while a communicator forms; stdout must carry only
    the one JSON line, so fd 1 points at fd 2 for the duration."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)   # RCCL used C stdio: empty its buffer while fd 1 is fd 2
        os.dup2(self.saved, 1)
        os.close(self.saved)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--utts", type=int, default=65536, help="utterances per GPU")
    ap.add_argument("--voices", type=int, default=1, help="1 = single Voice, 8 = config-4 presets")
    ap.add_argument("--config", type=int, default=0, choices=[0, 2, 3, 4],
                    help="BASELINE.json config: 2 = 4096 utterances, 3 = the headline (default), "
                         "4 = 8 voice presets; shorthand for --utts / --voices")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per utterance (0 = auto)")
    ap.add_argument("--pipeline", type=int, default=1, choices=[0, 1],
                    help="0 disables the small-batch producer/consumer kernels (A/B)")
    ap.add_argument("--variant", type=int, default=0, help="kernel instantiation (experiments)")
    ap.add_argument("--pcm16", action="store_true",
                    help="i16 PCM rows (the WAV sink's conversion fused into the store), not the "
                         "headline: 2.01 algorithmic bytes per sample")
    ap.add_argument("--literal", action="store_true",
                    help="after the timed region, also time the batch with skip_silent_formants=0")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the multi-rank code path even with one rank (testing)")
    ap.add_argument("--cpu-utts", type=int, default=1536,
                    help="utterances for the CPU baseline (0 = skip); 1536 is ~12-25 s of CPU")
    args = ap.parse_args()
    if args.config == 2:
        args.utts, args.voices = 4096, 1
    elif args.config == 3:
        args.utts, args.voices = 65536, 1
    elif args.config == 4:
        args.utts, args.voices = 65536, 8

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world != 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    distributed = world > 1 or args.force_dist
    if args.force_dist and "MASTER_ADDR" not in os.environ:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")

    import __graft_entry__ as ge
    if local_rank == 0:
        ge.build()

    group = None
    if distributed:
        # Control plane (barriers, unique-id hand-off, max/sum of two scalars): files in /tmp,
        # see grail_hip/rendezvous.py for why PyTorch stays out of the bench processes.  All
        # device-side exchange — the RCCL broadcast of the voice table — is in libgrail_hip.so.
        sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
        from grail_hip.rendezvous import FileGroup
        group = FileGroup(rank, world)
        group.barrier()  # local rank 0 has finished building

    import grail_hip as G
    from grail_hip import dist as D
    from grail_hip import workload as W
    G.load()

    # test hook: GRAIL_BENCH_DEVICE pins every rank to one device (two ranks on a 1-GPU box exercise
    # the multi-process control flow; RCCL refuses duplicate GPUs, so the voice table then takes the
    # file fallback)
    device = int(os.environ.get("GRAIL_BENCH_DEVICE", local_rank))
    ctx = G.Context(device)
    n_utt = args.utts
    n_voices = max(args.voices, 1)
    stride = W.max_samples()

    # ---- voice table: rank 0 builds it; one RCCL broadcast puts it in every GPU's HBM ----
    voice_path = "local"
    voices = None
    if rank == 0:
        voices = W.single_voice() if n_voices == 1 else W.preset_voices(n_voices)
        ctx.set_voices(voices)
    abandoned_ctx = None
    if distributed:
        # The collective runs on a worker thread so that a communicator that never forms (a hung
        # ncclCommInitRank takes every rank with it) costs RCCL_TIMEOUT_S, not the whole job.
        import threading
        outcome = {}

        def native_broadcast():
            try:
                if os.environ.get("GRAIL_BENCH_FAKE_RCCL_HANG"):      # test hook
                    time.sleep(1e6)
                uid = None
                if rank == 0:        # rank 0 always publishes something: the others never wait in vain
                    try:
                        uid = G.Context.comm_unique_id()
                    except G.GrailError as e:
                        outcome["err"] = e
                        uid = b""
                uid = group.broadcast_bytes(uid)
                if not uid:
                    raise outcome.get("err") or RuntimeError("rank 0 could not create an RCCL id")
                ctx.comm_init(uid, rank, world)
                ctx.broadcast_voices(n_voices, root=0)   # ncclBroadcast over xGMI inside the C ABI
                ctx.sync()
                outcome["ok"] = True
            except Exception as e:                       # noqa: BLE001 — reported below
                outcome["err"] = e

        worker = threading.Thread(target=native_broadcast, daemon=True)
        with stdout_to_stderr():
            worker.start()
            worker.join(RCCL_TIMEOUT_S)
        hung = worker.is_alive()
        ok_everywhere = all(f[0] == 1.0 for f in
                            group.gather_doubles((1.0 if outcome.get("ok") else 0.0,)))
        if ok_everywhere:
            voices = ctx.get_voices()
            voice_path = "rccl ncclBroadcast (grail_broadcast_voices)"
        else:        # keep the job alive: hand the same bytes over through /tmp, on every rank
            why = "timed out" if hung else outcome.get("err", "failed on another rank")
            print(f"[rank {rank}] native RCCL broadcast: {why}; using the file rendezvous",
                  file=sys.stderr)
            if hung:                 # its stream may be stuck behind the collective: start afresh
                abandoned_ctx = ctx
                ctx = G.Context(device)
                if rank == 0:
                    ctx.set_voices(voices)
            blob = group.broadcast_bytes(G.voices_blob(voices) if rank == 0 else None)
            voices = G.voices_from_blob(blob)
            ctx.set_voices(voices)
            voice_path = "file rendezvous (native RCCL broadcast unavailable: %s)" % (
                "timeout" if hung else "error")

    # ---- this rank's shard of the corpus, resident in HBM ---------------------------------
    first, last, segs, offs, vids, seeds = D.shard_inputs(n_utt, rank, world, len(voices))
    batch = ctx.upload(segs, offs, vids, seeds)
    ctx.set_option("lanes_per_utterance", args.lanes)
    ctx.set_option("kernel_variant", args.variant)
    ctx.set_option("small_batch_pipeline", args.pipeline)
    d_out = ctx.device_alloc(n_utt * stride * (2 if args.pcm16 else 4))
    d_len = ctx.device_alloc(n_utt * 4)

    def step():
        if args.pcm16:
            batch.synthesize_pcm16_async(d_out, stride, d_len)
        else:
            batch.synthesize_async(d_out, stride, d_len)
        ctx.sync()
        return ctx.last_kernel_ms()   # hipEvents on the kernel's own stream

    for _ in range(args.warmup):
        step()

    def barrier():
        ctx.sync()              # hipStreamSynchronize on the stream every kernel ran on
        if distributed:
            group.barrier()

    barrier()
    t0 = time.perf_counter()
    kernel_ms = [step() for _ in range(args.steps)]
    barrier()
    elapsed = time.perf_counter() - t0

    out_len = np.zeros(n_utt, dtype=np.uint32)
    ctx.d2h(out_len, d_len, n_utt * 4)
    samples_per_step = int(out_len.astype(np.uint64).sum())
    slow = ctx.get_option("slow_division_wave_steps")

    # Outside the timed region, for the record: the same batch with every formant evaluated
    # literally ("skip_silent_formants" = 0).  Same output bits; see DESIGN.md section 4.
    literal_ms = None
    if args.literal and rank == 0 and ctx.get_option("skip_silent_formants"):
        ctx.set_option("skip_silent_formants", 0)
        literal_ms = float(np.mean([step() for _ in range(2)]))
        ctx.set_option("skip_silent_formants", 1)

    if distributed:
        stats = group.gather_doubles((elapsed, float(samples_per_step)))
        elapsed = max(e for e, _ in stats)                  # MAX over ranks
        total_samples_per_step = sum(n for _, n in stats)   # whole-job samples per step
    else:
        total_samples_per_step = float(samples_per_step)

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = total_samples_per_step * args.steps / elapsed
        k_ms = float(np.mean(kernel_ms))
        alg_bytes = samples_per_step * (ALG_BYTES_PER_SAMPLE - (2.0 if args.pcm16 else 0.0))
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        cfg = "4" if len(voices) > 1 else ("2" if n_utt == 4096 else "3")
        wl_key = f"config{cfg}_utts{n_utt}" + ("_pcm16" if args.pcm16 else "")
        line = {
            "metric": "audio samples/sec (whole node) at 48 kHz, batch=65536 utterances",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"batch={n_utt} utterances x 2 s (4 segments x 0.5 s) per GPU, "
                            f"{len(voices)} Voice preset(s), 48 kHz, {'i16' if args.pcm16 else 'f32'} PCM left in HBM "
                            f"(BASELINE config {cfg}{'; config 5 sharding' if world > 1 else ''})",
                "utterances_per_gpu": n_utt, "samples_per_utterance": int(out_len[0]),
                "samples_per_step_per_gpu": samples_per_step, "out_stride": stride,
                "lanes_per_utterance": args.lanes or "auto", "voice_table": voice_path,
                "parity": "bit-exact vs oracle (tests/test_parity_gpu.py); "
                          f"IEEE-division fallback wave-steps this run: {slow}",
                "silent_formant_skip": "on: formants with amplitude exactly 0 and zero band-pass "
                                       "state contribute exactly +0.0 and their filters are skipped "
                                       "(voices::generic() has 4 of 8 such formants; config 4's "
                                       "presets have none); output bits unchanged",
                "formants_laid_out": ctx.get_option("last_launch_formants"),
                "lanes_per_utterance_used": ctx.get_option("last_launch_lanes"),
                "pipelined": ctx.get_option("last_launch_pipelined"),
                "kernel_ms_all_formants_literal": literal_ms if literal_ms else committed_literal_ms(),
                "samples_per_s_all_formants_literal":
                    (samples_per_step / (literal_ms * 1e-3)) if literal_ms else None,
                "literal_source": "measured in this run" if literal_ms else
                                  "profiles/r01_bench_n1_literal.json (run bench.py --literal to re-measure)",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": committed_traffic(wl_key),
                "kernel": "grail::synth_kernel", "kernel_ms": k_ms,
                "algorithmic_bytes_per_launch": alg_bytes,
                "note": "HBM is the nominal bound north_star names (4.01 B/sample); the binding "
                        "limit is f32 VALU issue (~660 unfusable flops + 25 IEEE divisions per "
                        "sample, SURVEY.md §8d) — see DESIGN.md §Roofline",
            },
            "roofline_valu": committed_valu(wl_key, k_ms),
        }
        if world == 1 and args.cpu_utts > 0:
            line["cpu_baseline"] = cpu_baseline(args.cpu_utts, voices, W)
            line["speedup_vs_cpu_1thread"] = value / line["cpu_baseline"]["value"]
            line["cpu_all_cores"] = cpu_all_cores(voices, W)
        print(json.dumps(line), flush=True)

    ctx.device_free(d_out)
    ctx.device_free(d_len)
    batch.free()
    ctx.close()
    if distributed:
        group.close()
    if abandoned_ctx is not None:     # a thread is still inside RCCL: do not wait for it
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
