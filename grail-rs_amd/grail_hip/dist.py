"""Multi-GPU plumbing shared by bench.py and the gloo tests.

The hot path shards by utterance with NO data-path collective: rank r renders utterances
[r*N/W, (r+1)*N/W) of the corpus.  The only exchange is the voice table, once, before any
synthesis: natively one ncclBroadcast inside the C ABI (grail_broadcast_voices, RCCL over
xGMI); `broadcast_voices_torch` is the same hand-off through torch.distributed, used where
RCCL cannot run (the CPU/gloo tests) and as a fallback.
"""
import numpy as np

from . import Voice, shard_range, voices_blob, voices_from_blob
from . import workload as W


def shard_inputs(utts_per_rank, rank, world, n_voices, **kw):
    """This rank's slice of the global synthetic corpus of utts_per_rank*world utterances."""
    first, last = shard_range(utts_per_rank * world, rank, world)
    return (first, last) + W.make_batch(last - first, first_utt=first, n_voices=n_voices, **kw)


def broadcast_voices_torch(voices, n_voices, dist, device="cpu", src=0):
    """Broadcast the raw grail_voice[] bytes from `src`; returns the Voice list on every rank."""
    import ctypes
    import torch
    nbytes = n_voices * ctypes.sizeof(Voice)
    if dist.get_rank() == src:
        blob = voices_blob(voices)
        assert len(blob) == nbytes
        t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    else:
        t = torch.zeros(nbytes, dtype=torch.uint8, device=device)
    dist.broadcast(t, src=src)
    return voices_from_blob(bytes(t.cpu().numpy().tobytes()))


def reduce_step_stats(elapsed_s, samples, dist, device="cpu"):
    """(max elapsed over ranks, total samples over ranks) — the bench's timing contract."""
    import torch
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    s = torch.tensor([float(samples)], dtype=torch.float64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return float(t.item()), float(s.item())


def gather_uint32(arr, dist):
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, np.asarray(arr, dtype=np.uint32).tobytes())
    return [np.frombuffer(b, dtype=np.uint32) for b in out]
