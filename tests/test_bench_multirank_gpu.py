"""bench.py's multi-rank control flow on a real GPU box: two ranks launched the way the driver
launches them (python -m torch.distributed.run), both pinned to GPU 0 (GRAIL_BENCH_DEVICE test
hook) because the test boxes have one GPU.  RCCL refuses a duplicate GPU, so this also shows the
all-rank fallback for the voice table; the RCCL success path itself is covered single-rank in
test_parity_gpu.py::test_rccl_voice_broadcast_single_rank."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_one_json_line(built):
    env = dict(os.environ, GRAIL_BENCH_DEVICE="0", GRAIL_BENCH_RCCL_TIMEOUT="60",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29571", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--utts", "2048"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]          # rank 0 prints ONE JSON line, nothing else
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 2
    per_rank = d["config"]["samples_per_step_per_gpu"]
    # whole-job value: both ranks' samples over the slowest rank's time
    assert abs(d["value"] - 2 * per_rank * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]
    assert "rendezvous" in d["config"]["voice_table"] or "rccl" in d["config"]["voice_table"]
    assert "cpu_baseline" not in d                      # N=1 only
