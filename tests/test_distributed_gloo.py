"""The N>1 path on CPU: two gloo ranks shard the corpus, broadcast the voice table and reduce
the timing exactly as bench.py does on RCCL.  No synthesis runs here (no GPU, no CPU fallback);
the oracle checks that rank shards are slices of the global job (batch invariance of inputs)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    import torch.distributed as dist

    import grail_hip as G
    import oracle_lib as O
    import dist_torch as D
    from grail_hip import workload as W
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_voices, per_rank = 8, 24
        voices = W.preset_voices(n_voices) if rank == 0 else None
        got = D.broadcast_voices_torch(voices, n_voices, dist)
        want = W.preset_voices(n_voices)
        assert all(bytes(a) == bytes(b) for a, b in zip(got, want)), "voice table differs"

        first, last, segs, offs, vids, seeds = W.shard_inputs(per_rank, rank, world, n_voices,
                                                              length=0.004, blend_length=0.004)
        assert (first, last) == (rank * per_rank, (rank + 1) * per_rank)
        # every rank's shard is the matching slice of the one global corpus
        gsegs, goffs, gvids, gseeds = W.make_batch(per_rank * world, n_voices=n_voices,
                                                   length=0.004, blend_length=0.004)
        assert np.array_equal(segs, gsegs[goffs[first]:goffs[last]])
        assert np.array_equal(vids, gvids[first:last]) and np.array_equal(seeds, gseeds[first:last])

        # the oracle stands in for the device here: per-rank lengths gathered == global lengths
        ov = [O.Voice.from_buffer_copy(bytes(v)) for v in got]
        lens = O.count_batch(ov, segs, offs, vids, seeds)
        all_lens = np.concatenate(D.gather_uint32(lens, dist))
        glens = O.count_batch(ov, gsegs, goffs, gvids, gseeds)
        assert np.array_equal(all_lens, glens)

        t, s = D.reduce_step_stats(0.5 + rank, int(lens.sum()), dist)
        assert t == 0.5 + (world - 1) and s == float(glens.sum())
        dist.barrier()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_gloo_shard_broadcast_reduce(built):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=500) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results


def _fg_worker(rank, world, key, q):
    sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
    from grail_hip.rendezvous import FileGroup
    try:
        g = FileGroup(rank, world, key=key, timeout=60)
        g.barrier()
        blob = g.broadcast_bytes(bytes(range(128)) if rank == 0 else None)
        assert blob == bytes(range(128))
        stats = g.gather_doubles((0.25 * (rank + 1), 1000.0 + rank))
        assert [s[0] for s in stats] == [0.25 * (r + 1) for r in range(world)]
        assert max(s[0] for s in stats) == 0.25 * world
        assert sum(s[1] for s in stats) == sum(1000.0 + r for r in range(world))
        for _ in range(20):
            g.barrier()
        g.close()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))


def test_file_rendezvous_three_ranks():
    """bench.py's torch-free control plane: barrier, byte broadcast (the RCCL unique id),
    scalar gather (max elapsed / sum samples)."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    key = f"pytest_{os.getpid()}"
    procs = [ctx.Process(target=_fg_worker, args=(r, 3, key, q)) for r in range(3)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(results) == [(0, "ok"), (1, "ok"), (2, "ok")], results
    assert not os.path.exists(os.path.join("/tmp", f"grail_rdzv_{key}"))


def _stale_worker(rank, world, key, q):
    sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
    from grail_hip.rendezvous import FileGroup
    try:
        g = FileGroup(rank, world, key=key, timeout=60)
        blob = g.broadcast_bytes(b"fresh-id" if rank == 0 else None)
        g.barrier()
        g.close()
        q.put((rank, blob.decode(), g.dir))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e), ""))


def test_file_rendezvous_ignores_leftovers_of_a_crashed_launch():
    """A crashed earlier launch with the same key left a directory full of barrier / broadcast files and
    a pointer to it (written by a process that no longer exists).  The new launch must not read any of
    it: rank 0 creates a fresh directory, the others accept only a pointer written by a live rank 0."""
    import multiprocessing as mp
    import subprocess
    key = f"pytest_stale_{os.getpid()}"
    stale = f"/tmp/grail_rdzv_{key}_stale"
    os.makedirs(stale, exist_ok=True)
    for name in ("bar1.0", "bar1.1", "bc1", "bar2.0", "bar2.1"):
        with open(os.path.join(stale, name), "wb") as f:
            f.write(b"stale-id")
    dead = subprocess.Popen([sys.executable, "-c", "pass"])
    dead.wait()
    with open(f"/tmp/grail_rdzv_{key}.ptr", "w") as f:
        f.write(f"{stale}\n{dead.pid}\n{__import__('time').time()!r}\n")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_stale_worker, args=(r, 2, key, q)) for r in (1, 0)]   # rank 1 first
    procs[0].start()
    __import__("time").sleep(0.5)              # rank 1 is already polling the stale pointer
    procs[1].start()
    results = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(30)
    assert [r[1] for r in results] == ["fresh-id", "fresh-id"], results
    assert all(r[2] != stale for r in results)
    __import__("shutil").rmtree(stale, ignore_errors=True)
