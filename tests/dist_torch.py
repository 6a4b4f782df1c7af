"""TEST plumbing (torch.distributed / gloo) for the world_size-2 CPU tests of the N>1 path.

The product never imports torch: natively the voice table travels by one ncclBroadcast inside the
C ABI (grail_broadcast_voices, RCCL over xGMI) and bench.py's control plane is the file
rendezvous.  `broadcast_voices_torch` is the same hand-off through torch.distributed where RCCL
cannot run (CPU, gloo); `reduce_step_stats` mirrors bench.py's max-over-ranks timing contract.
"""
import numpy as np

from grail_hip import Voice, voices_blob, voices_from_blob


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
