#!/usr/bin/env python3
"""bench.py — audio samples/sec of the fused Sequencer->Jitter->Synthesize HIP path.

Metric (BASELINE.json): audio samples/sec (whole node) at 48 kHz, batch = 65536
utterances x 2 s (4 segments x 0.5 s, single Voice) per GPU — BASELINE config 3 at N=1.
A "step" is one pass of the hot path over one batch whose inputs already sit in HBM; the
f32 PCM stays in HBM (25.2 GB per GPU).  Weak scaling: every rank renders its own
65536-utterance shard of the N*65536 corpus (config 5 at N=8) with no data-path
collective; the voice table is broadcast once (RCCL ncclBroadcast inside the C ABI) before
the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--utts U] [--voices V] [--mode exact|fast]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 and no launcher (WORLD_SIZE unset): this process starts N rank processes
itself (before it touches the GPU or loads the library), waits for them and relays rank 0's
line; any failing rank makes the whole run fail.  Rank 0 prints ONE JSON line.
"""
import argparse
import hashlib
import json
import os
import secrets
import socket
import subprocess
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
def kernel_sources():
    """Every device-side source of the library: the kernel template and its instantiation units
    (csrc/*.hip) and the headers they include (csrc/*.h), sorted by name.  (Host-only code is
    csrc/*.cpp and csrc/*.hpp.)"""
    d = os.path.join(ROOT, "grail-rs_amd", "csrc")
    return sorted(f for f in os.listdir(d) if f.endswith(".hip") or f.endswith(".h"))


def kernel_source_sha():
    """Identity of the kernels this run executes: sha256 over the kernel sources the library was
    built from (build() rebuilds whenever they change).  profiles/traffic.json entries carry the
    sha they were measured on; a counter figure from another build is reported as null."""
    h = hashlib.sha256()
    for name in kernel_sources():
        with open(os.path.join(ROOT, "grail-rs_amd", "csrc", name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


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
    n_cpu = min(2048, threads * 8)
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


def committed_counters(workload_key, kernel_symbol):
    """HBM bytes and VALU instructions per launch measured with rocprofv3 --pmc (separate
    WRITE_SIZE / FETCH_SIZE / SQ passes, gfx950 corrections applied; profiles/README.md) for the
    same command.  An entry counts only if it was measured on THIS build of the kernels (sha over
    the kernel sources) and on the kernel instantiation this run launched; otherwise None."""
    try:
        with open(TRAFFIC_FILE) as f:
            entry = json.load(f).get("entries", {}).get(workload_key)
    except (OSError, ValueError):
        return None
    if not entry or entry.get("kernel_source_sha") != kernel_source_sha():
        return None
    if kernel_symbol and entry.get("kernel") and entry["kernel"] != kernel_symbol:
        return None
    return entry


def valu_roofline(entry, kernel_ms):
    """Secondary roofline, the binding one: VALU instructions the kernel issued per launch
    (rocprofv3 --pmc SQ_INSTS_VALU) over this run's kernel time, against one instruction per 4
    cycles per SIMD — the rate of the packed f32 ops that make up most of the loop
    (v_pk_{mul,add,fma}_f32; profiles/r01_valu_microbench.txt)."""
    insts = entry.get("valu_insts") if entry else None
    if not insts:
        return None
    peak = 1024 * 2.4e9 / 4.0                     # 256 CUs x 4 SIMDs, 2.4 GHz, 4 cycles per issue
    rate = insts / (kernel_ms * 1e-3)
    return {"bound": "valu-issue", "achieved": rate / 1e9, "peak": peak / 1e9,
            "unit": "G wave-instructions/s", "frac": rate / peak,
            "valu_instructions_per_launch": insts,
            "note": "SQ_INSTS_VALU (profiles/).  The peak is the PACKED issue rate, one v_pk_*_f32 per 4 "
                    "cycles per SIMD; plain v_{add,mul,fma}_f32 issue at one per 2.8 cycles with two or more "
                    "waves per SIMD (profiles/r01_valu_microbench.txt), so this is a fraction of the packed-issue "
                    "peak, not of the SIMD's best case.  A lone wave per SIMD issues at most one instruction "
                    "per ~5 cycles whatever it is (measured), i.e. frac <= 0.8 for the headline batch"}


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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rccl_diagnose():
    """`bench.py --rccl-diagnose` (started by every rank of a run whose RCCL broadcast failed, just before
    that run exits 3): meet again over a fresh rendezvous with NCCL_DEBUG=WARN and report what RCCL says
    on stderr.  A child process, because RCCL reads NCCL_DEBUG once per process."""
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    from grail_hip.rendezvous import FileGroup
    import grail_hip as G
    group = FileGroup(rank, world)
    G.load()
    device = int(os.environ["GRAIL_BENCH_DEVICE"]) if "GRAIL_BENCH_DEVICE" in os.environ else local_rank
    tag = f"[rccl-diagnose rank {rank} device {device}]"
    try:
        ctx = G.Context(device)
        print(f"{tag} pci bus id {ctx.pci_bus_id()}", file=sys.stderr, flush=True)
        uid = G.Context.comm_unique_id() if rank == 0 else None
        uid = group.broadcast_bytes(uid)
        with stdout_to_stderr():
            ctx.comm_init(uid, rank, world)
            if rank == 0:
                from grail_hip import workload as W
                ctx.set_voices(W.single_voice())
            ctx.broadcast_voices(1, root=0)
            ctx.sync()
        print(f"{tag} second attempt succeeded: ncclCommCount={ctx.comm_info()[0]}", file=sys.stderr, flush=True)
    except Exception as e:                                  # noqa: BLE001 — the point is the message
        print(f"{tag} second attempt failed: {e}", file=sys.stderr, flush=True)
    os._exit(0)


def launch_ranks(args):
    """`--gpus N` without a launcher: start N fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set, same file rendezvous), wait, relay rank 0's single JSON line.  This parent never
    loads the library or touches the GPU, and nothing is exec'ed from a GPU-initialised process."""
    import __graft_entry__ as ge
    ge.build(load=False)
    n = args.gpus
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               GRAIL_RDZV_NONCE=secrets.token_hex(8), GRAIL_BENCH_PREBUILT="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno()))
    out0 = b""
    deadline = time.time() + float(os.environ.get("GRAIL_BENCH_LAUNCH_TIMEOUT", "3000"))
    failed = None
    try:
        out0, _ = procs[0].communicate(timeout=max(1.0, deadline - time.time()))
        for r, p in enumerate(procs):
            rc = p.wait(timeout=max(1.0, deadline - time.time()))
            if rc != 0 and failed is None:
                failed = (r, rc)
    except subprocess.TimeoutExpired:
        failed = ("timeout", -1)
    finally:
        for p in procs:                      # exact pids we started, never a pattern
            if p.poll() is None:
                p.kill()
    if failed:
        sys.stderr.write(f"bench.py: rank {failed[0]} failed (exit {failed[1]})\n")
        sys.stdout.write(out0.decode(errors="replace"))
        raise SystemExit(1)
    lines = [l for l in out0.decode().splitlines() if l.strip().startswith("{")]
    if len(lines) != 1:
        sys.stderr.write(f"bench.py: expected one JSON line from rank 0, got {len(lines)}\n")
        raise SystemExit(1)
    print(lines[0], flush=True)


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
    ap.add_argument("--mode", choices=["exact", "fast", "mid"], default="exact",
                    help="arithmetic of the timed region: exact (bit-identical to the reference, the "
                         "headline) or fast (stated tolerance, DESIGN.md §Fast mode)")
    ap.add_argument("--fast-leg", type=int, default=-1, choices=[-1, 0, 1],
                    help="with --mode exact: also time the same batch in fast mode after the timed "
                         "region and report it as `fast_mode` (default: on at N=1)")
    ap.add_argument("--other-configs", type=int, default=-1, choices=[-1, 0, 1],
                    help="after the timed region and the fast leg: BASELINE configs 4 (65 536 utterances x 8 presets) and 2 "
                         "(4 096 utterances) and the speech-like corpus (65 536), exact and fast, a few steps each, reported as `other_configs` (default: "
                         "on for the default config-3 run at N=1)")
    ap.add_argument("--corpus", default="aligned", choices=["aligned", "speech"],
                    help="aligned: the BASELINE corpus (four segments of 0.5 s per utterance, the headline); speech: the "
                         "speech-like corpus of grail_hip/workload.py (8 - 32 phonemes of 40 - 160 ms, 0.5 - 3.8 s) as the timed "
                         "batch — N=1 only, not a BASELINE config (counter passes of `other_configs.speech_like`)")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per utterance (0 = auto)")
    ap.add_argument("--pipeline", type=int, default=1, choices=[0, 1],
                    help="0 disables the small-batch producer/consumer kernels (A/B)")
    ap.add_argument("--round32", type=int, default=-1, choices=[-1, 0, 1], help="pipelined workgroups: rounds of 32 samples (A/B)")
    ap.add_argument("--pcm16", action="store_true",
                    help="i16 PCM rows (the WAV sink's conversion fused into the store), not the "
                         "headline: 2.01 algorithmic bytes per sample")
    ap.add_argument("--literal", action="store_true",
                    help="after the timed region, also time the batch with skip_silent_formants=0")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the multi-rank code path even with one rank (testing)")
    ap.add_argument("--require-rccl", action=argparse.BooleanOptionalAction, default=None,
                    help="fail unless the voice table travelled by ncclBroadcast over a communicator "
                         "of N ranks (default: on when N > 1)")
    ap.add_argument("--verify", action="store_true",
                    help="after the timed region: per-utterance digests of every rank's rows must equal "
                         "the digests rank 0 gets when it renders the same global utterances itself "
                         "(GPU-count invariance), and a re-batched subset must match (batch invariance)")
    ap.add_argument("--ramp", type=int, default=1, choices=[0, 1],
                    help="0 skips the clock-ramp launches of a small sub-batch before the warm-up "
                         "(counter passes: only the measured kernel should appear)")
    ap.add_argument("--rccl-diagnose", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-utts", type=int, default=1536,
                    help="utterances for the CPU baseline (0 = skip); 1536 is ~11 s of CPU on one "
                         "thread of the GPU box's host (samples/s does not depend on it)")
    args = ap.parse_args()
    if args.config == 2:
        args.utts, args.voices = 4096, 1
    elif args.config == 3:
        args.utts, args.voices = 65536, 1
    elif args.config == 4:
        args.utts, args.voices = 65536, 8

    if args.rccl_diagnose:
        return rccl_diagnose()
    if args.corpus == "speech" and (args.gpus > 1 or args.verify or args.pcm16):
        raise SystemExit("--corpus speech is a one-GPU f32 leg (no --gpus N, --verify, --pcm16)")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch {args.gpus} ranks "
                         f"(python bench.py --gpus {args.gpus} does it by itself)")
    distributed = world > 1 or args.force_dist
    if args.force_dist and "MASTER_ADDR" not in os.environ:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
    require_rccl = (world > 1) if args.require_rccl is None else args.require_rccl
    if distributed:
        # single-node RCCL: meet over loopback, skip the InfiniBand / interface probing.  Set here,
        # in the launcher's own process environment — the library never touches the environment.
        os.environ.setdefault("NCCL_IB_DISABLE", "1")
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import __graft_entry__ as ge
    if local_rank == 0 and not os.environ.get("GRAIL_BENCH_PREBUILT"):
        ge.build()

    group = None
    if distributed:
        # Control plane (barriers, unique-id hand-off, max/sum of two scalars): files in /tmp,
        # see grail_hip/rendezvous.py for why PyTorch stays out of the bench processes.  All
        # device-side exchange — the RCCL broadcast of the voice table — is in libgrail_hip.so.
        from grail_hip.rendezvous import FileGroup
        group = FileGroup(rank, world)
        group.barrier()  # local rank 0 has finished building

    import grail_hip as G
    from grail_hip import workload as W
    G.load()

    # test hook: GRAIL_BENCH_DEVICE pins every rank to one device (two ranks on a 1-GPU box exercise
    # the multi-process control flow; RCCL refuses duplicate GPUs, so the voice table then needs
    # --no-require-rccl and takes the file fallback)
    n_dev = G.device_count()
    if "GRAIL_BENCH_DEVICE" in os.environ:
        device = int(os.environ["GRAIL_BENCH_DEVICE"])
    else:
        device = local_rank
        if device >= n_dev:
            raise SystemExit(f"rank {rank}: --gpus {args.gpus} needs {args.gpus} GPUs, this node shows {n_dev}")
    ctx = G.Context(device)
    n_utt = args.utts
    n_voices = max(args.voices, 1)
    stride = W.max_samples()
    bus_id = ctx.pci_bus_id()
    if distributed:
        # every rank on a GPU of its own: the PCI bus ids must be distinct (the one-GPU test hook aside)
        bus_ids = [b.decode() for b in group.gather_bytes(bus_id.encode())]
        print(f"[rank {rank}] device {device} pci bus id {bus_id}", file=sys.stderr, flush=True)
        if "GRAIL_BENCH_DEVICE" not in os.environ and len(set(bus_ids)) != world:
            print(f"[rank {rank}] ranks share a GPU: pci bus ids {bus_ids}", file=sys.stderr, flush=True)
            os._exit(5)
    else:
        bus_ids = [bus_id]

    # ---- voice table: rank 0 builds it; one RCCL broadcast puts it in every GPU's HBM ----
    voice_path = "local"
    rccl_info = {"ranks": 0, "broadcast": "none (single rank)"}
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
                outcome["comm"] = ctx.comm_info()        # (ncclCommCount, ncclCommUserRank)
                outcome["ok"] = True
            except Exception as e:                       # noqa: BLE001 — reported below
                outcome["err"] = e

        worker = threading.Thread(target=native_broadcast, daemon=True)
        with stdout_to_stderr():
            worker.start()
            worker.join(RCCL_TIMEOUT_S)
        hung = worker.is_alive()
        counts = group.gather_doubles((1.0 if outcome.get("ok") else 0.0,
                                       float(outcome.get("comm", (0, 0))[0])))
        ok_everywhere = all(f[0] == 1.0 for f in counts)
        comm_ranks = int(min(f[1] for f in counts))      # what RCCL itself counted, on every rank
        if ok_everywhere:
            voices = ctx.get_voices()
            voice_path = "rccl ncclBroadcast (grail_broadcast_voices)"
            rccl_info = {"ranks": comm_ranks, "broadcast": "ncclBroadcast"}
        if require_rccl and not (ok_everywhere and comm_ranks == world):
            why = "timed out" if hung else outcome.get("err", "failed on another rank")
            print(f"[rank {rank}] --require-rccl: the RCCL broadcast over {world} ranks did not "
                  f"happen ({why}; ncclCommCount={comm_ranks})", file=sys.stderr)
            sys.stderr.flush()
            # once more, in a child per rank with NCCL_DEBUG=WARN, so that the log says why
            try:
                env = dict(os.environ, NCCL_DEBUG="WARN", GRAIL_BENCH_PREBUILT="1",
                           GRAIL_RDZV_NONCE=os.environ.get("GRAIL_RDZV_NONCE", "") + "-diagnose")
                subprocess.run([sys.executable, os.path.abspath(__file__), "--rccl-diagnose"], env=env,
                               stdout=sys.stderr.fileno(), timeout=min(90.0, RCCL_TIMEOUT_S))
            except Exception as e:                          # noqa: BLE001
                print(f"[rank {rank}] rccl diagnosis: {e}", file=sys.stderr)
            sys.stderr.flush()
            os._exit(3)              # a thread may still sit inside RCCL: do not wait for it
        if not ok_everywhere:        # keep the job alive: hand the same bytes over through /tmp, on every rank
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
            rccl_info = {"ranks": 0, "broadcast": "file-fallback"}

    # ---- this rank's shard of the corpus, resident in HBM ---------------------------------
    if args.corpus == "speech":
        first, last = 0, n_utt
        segs, offs, vids, seeds, stride = W.speech_like_batch(n_utt, np.random.default_rng(7), n_voices=len(voices))
    else:
        first, last, segs, offs, vids, seeds = W.shard_inputs(n_utt, rank, world, len(voices))
    batch = ctx.upload(segs, offs, vids, seeds)
    ctx.set_option("lanes_per_utterance", args.lanes)
    ctx.set_option("small_batch_pipeline", args.pipeline)
    if args.round32 >= 0:
        ctx.set_option("pipeline_round32", args.round32)
    # ("mid": the second tolerance tier — the reference's own filter coefficients at every sample — whatever the voices)
    ctx.set_option("arithmetic", {"exact": 0, "fast": 1, "mid": 2}[args.mode])
    d_out = ctx.device_alloc(n_utt * stride * (2 if args.pcm16 else 4))
    d_len = ctx.device_alloc(n_utt * 4)
    # first touch of the 25 GB of rows outside every measurement: a kernel that also has to fault its
    # output pages in takes twice as long once, which would sit in the profiler's per-kernel average
    ctx.memset(d_out, 0, n_utt * stride * (2 if args.pcm16 else 4))

    # clock ramp: the first kernel after idle runs at low clocks for tens of milliseconds (79 ms instead of
    # 43 for the very first launch, seen in rocprofv3's per-kernel average).  A few launches of a small
    # sub-batch — a different kernel instantiation, so it has its own row in the profiler's statistics —
    # bring the clocks up before the W warm-up steps of the real batch.  Untimed, like the warm-up.
    if args.ramp and n_utt > 8192:
        r_segs, r_offs, r_vids, r_seeds = W.make_batch(4096, n_voices=len(voices))
        ramp = ctx.upload(r_segs, r_offs, r_vids, r_seeds)
        for _ in range(4):
            ramp.synthesize_async(d_out, stride, d_len)
        ctx.sync()
        ramp.free()

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
    kernel_symbol = ctx.last_kernel_name()
    launch_info = {"formants_laid_out": ctx.get_option("last_launch_formants"),
                   "lanes_per_utterance_used": ctx.get_option("last_launch_lanes"),
                   "pipelined": ctx.get_option("last_launch_pipelined"),
                   # what the library made of the call: kernel launches it was cut into (grail_plan_blocks), the tier that ran
                   "launch_blocks": ctx.get_option("last_launch_blocks"), "compute_units": ctx.get_option("compute_units"),
                   "arithmetic_ran": ("exact", "fast (coefficients interpolated)", "fast (reference coefficients)")[
                       ctx.get_option("last_launch_fast")]}

    # ---- --verify: GPU-count invariance and batch invariance, on the device ----------------
    verify = None
    if args.verify and not args.pcm16:
        sums, _, bad = ctx.digest(d_out, stride, d_len, n_utt)
        mine = sums.tobytes() + out_len.tobytes()
        everyone = group.gather_bytes(mine) if distributed else [mine]
        if rank == 0:
            mismatches, checked = 0, 0
            for r in range(world):
                rs = np.frombuffer(everyone[r][: 8 * n_utt], dtype=np.uint64)
                rl = np.frombuffer(everyone[r][8 * n_utt:], dtype=np.uint32)
                if r == 0:
                    ref_s, ref_l = sums, out_len
                else:   # rank 0 renders rank r's utterances itself, as one batch of the same size
                    _, _, s2, o2, v2, j2 = W.shard_inputs(n_utt, r, world, len(voices))
                    b2 = ctx.upload(s2, o2, v2, j2)
                    b2.synthesize_async(d_out, stride, d_len)
                    ctx.sync()
                    ref_l = np.zeros(n_utt, dtype=np.uint32)
                    ctx.d2h(ref_l, d_len, n_utt * 4)
                    ref_s, _, _ = ctx.digest(d_out, stride, d_len, n_utt)
                    b2.free()
                mismatches += int(np.count_nonzero(rs != ref_s) + np.count_nonzero(rl != ref_l))
                checked += n_utt
            # batch invariance: every 61st utterance of rank 0's shard re-rendered as its own small
            # batch (another batch size, another position, other wave-mates; in exact mode usually another
            # lane mapping too).  Exact mode must give the same bits whatever kernel renders the subset;
            # fast mode must give the same bits within a kernel family, so the subset is pinned to the
            # family the batch took (its lane mapping, or its time-split grid)
            rebatched = None
            pin_names = ("time_split_chunks", "time_split_span_samples", "lanes_per_utterance",
                         "time_parallel_scan_max_utterances", "time_split")
            saved = {k_: ctx.get_option(k_) for k_ in pin_names}     # what is in force now: restored below, whatever happens
            sums_family = sums                                       # the digests the subset is compared with
            d_o3 = d_l3 = b3 = None
            try:
                fast_pins = {}
                if args.mode in ("fast", "mid"):
                    chunks = ctx.get_option("last_launch_chunks")
                    if chunks:
                        fast_pins = {"time_split_chunks": chunks, "time_split_span_samples": stride}
                        # the batch itself once more on the pinned grid: the digests to compare with (the timed
                        # rendering's own digests, `sums`, stay what the cross-rank comparison above used)
                        for k_, v_ in fast_pins.items():
                            ctx.set_option(k_, v_)
                        batch.synthesize_async(d_out, stride, d_len)
                        ctx.sync()
                        sums_family, _, _ = ctx.digest(d_out, stride, d_len, n_utt)
                    elif ctx.get_option("last_launch_lanes"):
                        fast_pins = {"lanes_per_utterance": ctx.get_option("last_launch_lanes")}
                    else:                                   # the scan kernel: one workgroup per utterance
                        fast_pins = {"time_parallel_scan_max_utterances": 1 << 30, "time_split": 0}
                    for k_, v_ in fast_pins.items():
                        ctx.set_option(k_, v_)
                    if not chunks:
                        # ... and the batch itself once more on the pinned family: a batch that was cut into several
                        # blocks (--utts 70000) had rows on other families than its largest block's, and a fast row's
                        # bits follow its family — the subset must be compared with a rendering on ONE family
                        batch.synthesize_async(d_out, stride, d_len)
                        ctx.sync()
                        sums_family, _, _ = ctx.digest(d_out, stride, d_len, n_utt)
                pick = np.arange(0, n_utt, 61, dtype=np.int64)
                _, _, s0, o0, v0, j0 = W.shard_inputs(n_utt, 0, world, len(voices))
                s0 = s0.reshape(n_utt, -1)[pick].reshape(-1)
                o0 = (np.arange(len(pick) + 1, dtype=np.uint64) * W.SEGMENTS_PER_UTT).astype(np.uint32)
                b3 = ctx.upload(s0, o0, v0[pick], j0[pick])
                d_o3 = ctx.device_alloc(len(pick) * stride * 4)
                d_l3 = ctx.device_alloc(len(pick) * 4)
                b3.synthesize_async(d_o3, stride, d_l3)
                ctx.sync()
                s3, _, _ = ctx.digest(d_o3, stride, d_l3, len(pick))
                rebatched = int(np.count_nonzero(s3 != sums_family[pick]))
                mismatches += rebatched
            finally:
                if d_o3 is not None:
                    ctx.device_free(d_o3)
                if d_l3 is not None:
                    ctx.device_free(d_l3)
                if b3 is not None:
                    b3.free()
                for k_, v_ in saved.items():
                    ctx.set_option(k_, v_)
            verify = {"utterances_checked": checked, "mismatches": mismatches,
                      "nonfinite_samples": int(bad.sum()), "rebatched_subset_mismatches": rebatched,
                      "method": "per-utterance bit-pattern digests (grail_batch_digest): every rank's rows "
                                "vs rank 0 rendering the same global utterances; every 61st utterance "
                                "re-rendered as a separate small batch"}
        if distributed:
            flag = group.broadcast_bytes(b"1" if (rank == 0 and verify["mismatches"]) else b"0"
                                         if rank == 0 else None)
            if flag == b"1":
                if rank == 0:
                    print(json.dumps({"verify": verify}), file=sys.stderr)
                os._exit(4)
        elif verify["mismatches"]:
            print(json.dumps({"verify": verify}), file=sys.stderr)
            raise SystemExit(4)

    # Outside the timed region, for the record: the same batch with every formant evaluated
    # literally ("skip_silent_formants" = 0).  Same output bits; see DESIGN.md section 4.
    literal_ms = None
    if args.literal and rank == 0 and ctx.get_option("skip_silent_formants"):
        ctx.set_option("skip_silent_formants", 0)
        literal_ms = float(np.mean([step() for _ in range(2)]))
        ctx.set_option("skip_silent_formants", 1)

    # Outside the timed region: the same batch in fast mode (stated tolerance), its own roofline.
    fast_leg = None
    want_fast_leg = (args.fast_leg == 1) or (args.fast_leg == -1 and world == 1)
    if args.mode == "exact" and want_fast_leg and rank == 0 and not args.pcm16:
        ctx.set_option("arithmetic", 1)
        step()
        f_ms = [step() for _ in range(args.steps)]
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        f_elapsed = time.perf_counter() - t1
        f_symbol = ctx.last_kernel_name()
        ctx.set_option("arithmetic", 0)
        fk = float(np.mean(f_ms))
        f_ach = samples_per_step * ALG_BYTES_PER_SAMPLE / (fk * 1e-3) / 1e9
        cfgk = "4" if len(voices) > 1 else ("2" if n_utt == 4096 else "3")
        f_entry = committed_counters(f"config{cfgk}_utts{n_utt}_fast", f_symbol)
        fast_leg = {
            "value": samples_per_step * args.steps / f_elapsed, "unit": "samples/s",
            "ms_per_step": f_elapsed * 1e3 / args.steps, "kernel_ms": fk, "kernel": f_symbol,
            "tolerance": G.FAST_TOLERANCE_NOTE,
            # fast arithmetic is served up to a sharpness of the voices' resonances (grail_fast_sharpness)
            "voice_sharpness": max(G.fast_sharpness(v) for v in voices), "sharpness_limit": G.FAST_SHARPNESS_LIMIT,
            "fast_kernels_ran": "FAST" in f_symbol,
            "roofline": {"bound": "hbm", "achieved": f_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": f_ach / HBM_PEAK_GBS,
                         "traffic": f_entry.get("hbm_bytes") if f_entry else None},
            "roofline_valu": valu_roofline(f_entry, fk),
        }

    # Outside the timed region, rank 0 of a one-GPU run of the headline config only: BASELINE configs 4 and 2 on a context
    # of their own (the voice table of the timed context stays what it is), exact and fast — the numbers the headline is
    # quoted beside (config 4: all eight formants live; config 2: a batch that leaves most lanes of the device idle).
    other_configs = None
    want_other = (args.other_configs == 1) or (args.other_configs == -1 and world == 1 and args.mode == "exact" and
                                                 n_utt == 65536 and n_voices == 1 and not args.pcm16 and args.lanes == 0)
    if want_other and rank == 0 and world == 1 and args.corpus == "aligned":
        other_configs = {}
        ctx2 = G.Context(device)
        # an optional extra must not cost the headline its line: a failure here (a smaller part without room for the
        # speech-like rows, an upload error) is recorded in `other_configs.error` and the run goes on
        try:
          try:
              for name, o_utts, o_voices, o_steps in (("config4", 65536, 8, 5), ("config2", 4096, 1, 10)):
                  o_table = W.single_voice() if o_voices == 1 else W.preset_voices(o_voices)
                  ctx2.set_voices(o_table)
                  o_batch = ctx2.upload(*W.make_batch(o_utts, n_voices=o_voices))
                  legs = {}
                  for o_mode in ("exact", "fast"):
                      ctx2.set_option("arithmetic", 1 if o_mode == "fast" else 0)
                      o_batch.synthesize_async(d_out, stride, d_len)      # (d_out / d_len: the timed region's rows, done with)
                      ctx2.sync()
                      o_kernel = []
                      t2 = time.perf_counter()
                      for _ in range(o_steps):
                          o_batch.synthesize_async(d_out, stride, d_len)
                          ctx2.sync()
                          o_kernel.append(ctx2.last_kernel_ms())
                      o_elapsed = time.perf_counter() - t2
                      o_len = np.zeros(o_utts, dtype=np.uint32)
                      ctx2.d2h(o_len, d_len, o_utts * 4)
                      o_samples = int(o_len.astype(np.uint64).sum())
                      o_symbol = ctx2.last_kernel_name()
                      o_k = float(np.mean(o_kernel))
                      o_ach = o_samples * ALG_BYTES_PER_SAMPLE / (o_k * 1e-3) / 1e9
                      o_entry = committed_counters(f"{name}_utts{o_utts}" + ("_fast" if o_mode == "fast" else ""), o_symbol)
                      legs[o_mode] = {
                          "value": o_samples * o_steps / o_elapsed, "unit": "samples/s", "steps": o_steps,
                          "ms_per_step": o_elapsed * 1e3 / o_steps, "kernel_ms": o_k, "kernel": o_symbol,
                          "launch_blocks": ctx2.get_option("last_launch_blocks"),
                          "roofline": {"bound": "hbm", "achieved": o_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": o_ach / HBM_PEAK_GBS, "traffic": o_entry.get("hbm_bytes") if o_entry else None,
                                       "algorithmic_bytes_per_launch": o_samples * ALG_BYTES_PER_SAMPLE},
                          "roofline_valu": valu_roofline(o_entry, o_k),
                      }
                  other_configs[name] = {
                      "workload": f"batch={o_utts} utterances x 2 s (4 segments x 0.5 s), {o_voices} Voice preset(s), 48 kHz, "
                                  f"f32 PCM left in HBM (BASELINE {name})", **legs}
                  o_batch.free()
              # ... and the headline batch's size on SPEECH-LIKE input (grail_hip/workload.py speech_like_batch: 8 - 32 phonemes of
              # 40 - 160 ms per utterance, 0.5 - 3.8 s, every lane's segment boundaries at times of its own): what the kernels do
              # when the events of a wave's lanes do not coincide, and the launch plan by the rows' lengths and events
              ctx2.set_voices(W.single_voice())
              s_segs, s_offs, s_vids, s_seeds, s_stride = W.speech_like_batch(65536, np.random.default_rng(7))
              s_batch = ctx2.upload(s_segs, s_offs, s_vids, s_seeds)
              s_out, s_len_dev = ctx2.device_alloc(65536 * s_stride * 4), ctx2.device_alloc(65536 * 4)
              try:
                  legs = {}
                  for o_mode in ("exact", "fast"):
                      ctx2.set_option("arithmetic", 1 if o_mode == "fast" else 0)
                      s_batch.synthesize_async(s_out, s_stride, s_len_dev)
                      ctx2.sync()
                      o_kernel = []
                      t2 = time.perf_counter()
                      for _ in range(3):
                          s_batch.synthesize_async(s_out, s_stride, s_len_dev)
                          ctx2.sync()
                          o_kernel.append(ctx2.last_kernel_ms())
                      o_elapsed = time.perf_counter() - t2
                      o_len = np.zeros(65536, dtype=np.uint32)
                      ctx2.d2h(o_len, s_len_dev, 65536 * 4)
                      o_samples = int(o_len.astype(np.uint64).sum())
                      o_k = float(np.mean(o_kernel))
                      o_ach = o_samples * ALG_BYTES_PER_SAMPLE / (o_k * 1e-3) / 1e9
                      s_symbol = ctx2.last_kernel_name()
                      # (counters of the same corpus as the timed batch: bench.py --corpus speech, tools/collect_counters.py)
                      s_entry = committed_counters("speech_like_utts65536" + ("_fast" if o_mode == "fast" else ""), s_symbol)
                      legs[o_mode] = {
                          "value": o_samples * 3 / o_elapsed, "unit": "samples/s", "steps": 3, "ms_per_step": o_elapsed * 1e3 / 3,
                          "kernel_ms": o_k, "kernel": s_symbol, "launch_blocks": ctx2.get_option("last_launch_blocks"),
                          "lanes_per_utterance_used": ctx2.get_option("last_launch_lanes"),
                          "arithmetic_ran": "fast" if ctx2.get_option("last_launch_fast") else "exact",
                          "roofline": {"bound": "hbm", "achieved": o_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": o_ach / HBM_PEAK_GBS, "traffic": s_entry.get("hbm_bytes") if s_entry else None,
                                       "algorithmic_bytes_per_launch": o_samples * ALG_BYTES_PER_SAMPLE},
                          "roofline_valu": valu_roofline(s_entry, o_k),
                      }
                  other_configs["speech_like"] = {
                      "workload": "batch=65536 utterances of 8 - 32 phonemes of 40 - 160 ms (0.5 - 3.8 s, 2.0 s on average: the headline "
                                  "batch's samples), blends of 30 - 80 ms, single Voice, 48 kHz, f32 PCM left in HBM (not a BASELINE "
                                  "config: the bench corpus has four aligned segments of 0.5 s)", **legs}
              finally:
                  ctx2.device_free(s_out)
                  ctx2.device_free(s_len_dev)
                  s_batch.free()
          except Exception as e:                       # noqa: BLE001 — reported in the line
              other_configs["error"] = f"{type(e).__name__}: {e}"
        finally:
            ctx2.close()

    my_elapsed = elapsed
    if distributed:
        stats = group.gather_doubles((elapsed, float(samples_per_step), float(np.mean(kernel_ms)), float(device)))
        elapsed = max(e for e, _, _, _ in stats)                  # MAX over ranks
        total_samples_per_step = sum(n for _, n, _, _ in stats)   # whole-job samples per step
    else:
        stats = [(my_elapsed, float(samples_per_step), float(np.mean(kernel_ms)), float(device))]
        total_samples_per_step = float(samples_per_step)
    # one row per rank: a straggler or a mis-pinned rank shows here
    per_rank = [{"rank": r, "device": int(dv), "pci_bus_id": bus_ids[r], "kernel_ms_mean": km,
                 "ms_per_step": e * 1e3 / args.steps, "samples_per_s": n * args.steps / e}
                for r, (e, n, km, dv) in enumerate(stats)]

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = total_samples_per_step * args.steps / elapsed
        k_ms = float(np.mean(kernel_ms))
        alg_bytes = samples_per_step * (ALG_BYTES_PER_SAMPLE - (2.0 if args.pcm16 else 0.0))
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        cfg = "4" if len(voices) > 1 else ("2" if n_utt == 4096 else "3")
        wl_key = (f"config{cfg}_utts{n_utt}" if args.corpus == "aligned" else f"speech_like_utts{n_utt}") + \
                 ("_pcm16" if args.pcm16 else "") + ("_fast" if args.mode == "fast" else "_mid" if args.mode == "mid" else "")
        entry = committed_counters(wl_key, kernel_symbol)
        line = {
            "metric": "audio samples/sec (whole node) at 48 kHz, batch=65536 utterances",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": (f"batch={n_utt} utterances x 2 s (4 segments x 0.5 s) per GPU, "
                             f"{len(voices)} Voice preset(s), 48 kHz, {'i16' if args.pcm16 else 'f32'} PCM left in HBM "
                             f"(BASELINE config {cfg}{'; config 5 sharding' if world > 1 else ''})") if args.corpus == "aligned" else
                            (f"batch={n_utt} speech-like utterances (8 - 32 phonemes of 40 - 160 ms, 0.5 - 3.8 s), "
                             f"{len(voices)} Voice preset(s), 48 kHz, f32 PCM left in HBM (NOT a BASELINE config)"),
                "arithmetic": args.mode,
                "utterances_per_gpu": n_utt, "samples_per_utterance": int(out_len[0]),
                "samples_per_step_per_gpu": samples_per_step, "out_stride": stride,
                "lanes_per_utterance": args.lanes or "auto", "voice_table": voice_path,
                "parity": ("bit-exact vs oracle (tests/test_parity_gpu.py); "
                           f"IEEE-division fallback wave-steps this run: {slow}") if args.mode == "exact"
                          else G.FAST_TOLERANCE_NOTE,
                "silent_formant_skip": "on: formants with amplitude exactly 0 and zero band-pass "
                                       "state contribute exactly +0.0 and their filters are skipped "
                                       "(voices::generic() has 4 of 8 such formants; config 4's "
                                       "presets have none); output bits unchanged",
                **launch_info,
                "kernel_ms_all_formants_literal": literal_ms,
                "samples_per_s_all_formants_literal":
                    (samples_per_step / (literal_ms * 1e-3)) if literal_ms else None,
                "all_formants_live_note": "the headline depends on the input: voices::generic() has 4 dead "
                                          "formants (skipped, same bits); with all eight live (config 4 / "
                                          "--literal) the same batch takes ~1.85x as long — see "
                                          "profiles/ for the config-4 line",
            },
            "rccl": rccl_info,
            "per_gpu_value": value / world,
            "per_rank": per_rank,
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": entry.get("hbm_bytes") if entry else None,
                "kernel": kernel_symbol, "kernel_ms": k_ms,
                "kernel_source_sha": kernel_source_sha(),
                "algorithmic_bytes_per_launch": alg_bytes,
                "note": "HBM is the nominal bound north_star names (4.01 B/sample); the binding "
                        "limit is f32 VALU issue (~660 unfusable flops + 25 IEEE divisions per "
                        "sample, SURVEY.md §8d) — see DESIGN.md §Roofline",
            },
            "roofline_valu": valu_roofline(entry, k_ms),
        }
        if verify is not None:
            line["verify"] = verify
        if fast_leg is not None:
            line["fast_mode"] = fast_leg
        if other_configs is not None:
            line["other_configs"] = other_configs
        if world == 1 and args.cpu_utts > 0:
            line["cpu_baseline"] = cpu_baseline(args.cpu_utts, voices, W)
            line["speedup_vs_cpu_1thread"] = value / line["cpu_baseline"]["value"]
            # like for like: the CPU baseline evaluates all eight formants; so does config 4 (the headline skips four dead ones)
            c4 = (other_configs or {}).get("config4", {}).get("exact")
            line["speedup_vs_cpu_1thread_all_formants"] = (c4["value"] / line["cpu_baseline"]["value"]) if c4 else None
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
