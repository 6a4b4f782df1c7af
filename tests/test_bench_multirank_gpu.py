"""bench.py's multi-rank control flow on a real GPU box.  The test boxes have one GPU, so both ranks
are pinned to GPU 0 (GRAIL_BENCH_DEVICE test hook); RCCL refuses a duplicate GPU, which exercises
(a) --require-rccl failing loudly and (b) with --no-require-rccl the all-rank file fallback for the
voice table.  The RCCL success path itself is covered single-rank in
test_parity_gpu.py::test_rccl_voice_broadcast_single_rank; N distinct GPUs only exist on the driver's
8-GPU node."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = dict(os.environ, GRAIL_BENCH_DEVICE="0", GRAIL_BENCH_RCCL_TIMEOUT="60",
           HSA_ENABLE_IPC_MODE_LEGACY="0")


def _one_line(p):
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]          # rank 0 prints ONE JSON line, nothing else
    return json.loads(lines[0])


def _check_two_ranks(d):
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 2
    per_rank = d["config"]["samples_per_step_per_gpu"]
    # whole-job value: both ranks' samples over the slowest rank's time
    assert abs(d["value"] - 2 * per_rank * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]
    assert d["rccl"]["broadcast"] in ("ncclBroadcast", "file-fallback")
    assert d["rccl"]["ranks"] == (2 if d["rccl"]["broadcast"] == "ncclBroadcast" else 0)
    assert "cpu_baseline" not in d                      # N=1 only
    # one row per rank, so that a straggler or a mis-pinned rank is visible in the driver's 8-GPU record
    assert [r["rank"] for r in d["per_rank"]] == [0, 1]
    for r in d["per_rank"]:
        assert r["device"] == 0 and r["pci_bus_id"] and r["kernel_ms_mean"] > 0 and r["ms_per_step"] > 0
        assert r["samples_per_s"] > 0
    assert abs(d["per_gpu_value"] * 2 - d["value"]) < 1e-6 * d["value"]
    assert d["ms_per_step"] >= max(r["ms_per_step"] for r in d["per_rank"]) * (1 - 1e-9)


def test_two_ranks_under_torch_distributed_run(built):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29571", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--utts", "2048", "--no-require-rccl"]
    d = _one_line(subprocess.run(cmd, cwd=ROOT, env=ENV, capture_output=True, text=True, timeout=600))
    _check_two_ranks(d)


def test_gpus_2_launches_its_own_ranks_and_verifies(built):
    """`python bench.py --gpus 2` with no launcher: the parent starts two ranks, relays one line;
    --verify proves every rank's rows equal rank 0's rendering of the same global utterances."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--utts", "2048", "--no-require-rccl", "--verify"]
    env = {k: v for k, v in ENV.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    d = _one_line(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600))
    _check_two_ranks(d)
    assert d["verify"]["mismatches"] == 0 and d["verify"]["utterances_checked"] == 2 * 2048
    assert d["verify"]["rebatched_subset_mismatches"] == 0


def test_require_rccl_fails_loudly_when_the_ranks_cannot_meet(built):
    """Default for N > 1: no silent file fallback.  Two ranks on one GPU cannot form a communicator."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--utts", "256"]
    env = {k: v for k, v in ENV.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert "require-rccl" in p.stderr
    assert "pci bus id" in p.stderr                     # every rank says which GPU it sits on
    assert "rccl-diagnose" in p.stderr                  # ... and the failed step ran once more with NCCL_DEBUG=WARN
    assert not [l for l in p.stdout.splitlines() if l.strip().startswith("{") and "n_gpus" in l]


def test_single_gpu_verify_and_fast_leg(built):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--utts", "4096",
           "--verify", "--cpu-utts", "0"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    d = _one_line(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600))
    assert d["n_gpus"] == 1 and d["verify"]["mismatches"] == 0
    assert d["verify"]["rebatched_subset_mismatches"] == 0
    assert len(d["per_rank"]) == 1 and d["per_gpu_value"] == d["value"]
    assert d["config"]["arithmetic"] == "exact"
    assert d["fast_mode"]["value"] > 0 and "FAST" in d["fast_mode"]["kernel"]
    assert d["rccl"]["ranks"] == 0


@pytest.mark.parametrize("utts", [4096, 65536])
def test_fast_mode_verify_rerenders_a_subset_within_the_same_kernel_family(built, utts):
    """--mode fast --verify: the re-batched subset must give the same bits as long as it runs in the kernel
    family the batch took (its time-split grid at 4096 utterances, one lane per utterance at 65536)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--utts", str(utts),
           "--mode", "fast", "--verify", "--cpu-utts", "0"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    d = _one_line(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600))
    assert d["config"]["arithmetic"] == "fast" and "FAST" in d["roofline"]["kernel"]
    assert ("SPLIT" in d["roofline"]["kernel"]) == (utts == 4096)
    assert d["verify"]["mismatches"] == 0 and d["verify"]["rebatched_subset_mismatches"] == 0


def test_eight_ranks_on_one_gpu_take_the_config_5_control_flow(built):
    """The driver's 8-GPU run (config 5) in miniature: eight rank processes — all pinned to the one GPU of the
    test box, so the voice table takes the file fallback — shard 8 x 512 utterances, meet at the barriers, and
    rank 0 prints one line with eight per-rank rows; --verify proves every shard equals rank 0's rendering."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
           "--utts", "512", "--no-require-rccl", "--verify"]
    env = {k: v for k, v in ENV.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    d = _one_line(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900))
    assert d["n_gpus"] == 8 and d["scaling"] == "weak"
    assert [r["rank"] for r in d["per_rank"]] == list(range(8))
    assert abs(d["per_gpu_value"] * 8 - d["value"]) < 1e-6 * d["value"]
    assert d["verify"]["mismatches"] == 0 and d["verify"]["utterances_checked"] == 8 * 512
    per_rank = d["config"]["samples_per_step_per_gpu"]
    assert abs(d["value"] - 8 * per_rank * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]
