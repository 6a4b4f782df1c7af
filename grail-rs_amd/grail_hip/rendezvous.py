"""A torch-free control plane for the ranks of ONE node (the bench contract is single-node).

Why not torch.distributed: a PyTorch-ROCm wheel carries its own private HIP/HSA runtime;
loaded next to the system runtime libgrail_hip.so links, the two fight over the device
(measured on the GPU box: whichever initialises second sees "no ROCm-capable device").
The launcher (`python -m torch.distributed.run`, or bench.py's own `--gpus N` parent) only has to
spawn the ranks and export RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT; the ranks
then meet here through files under /tmp: barriers, a scalar gather, byte gathers, and the hand-off
of the 128-byte RCCL unique id.  All device-side exchange (the voice table) is RCCL, in the C ABI.

Stale files of a crashed earlier launch with the same key can never be read: rank 0 creates a
FRESH directory (mkdtemp: a random suffix) and publishes its name through one well-known pointer
file, which it first removes and which carries rank 0's pid and start time; the other ranks accept
a pointer only if that process is alive and the pointer is younger than their own launcher.
"""
import os
import shutil
import struct
import tempfile
import time


def _sanitize(key):
    return "".join(c if c.isalnum() or c in "._-" else "_" for c in key)


def _pid_alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    except PermissionError:
        return True
    return True


class FileGroup:
    def __init__(self, rank=None, world=None, key=None, timeout=600.0):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
        if key is None:
            key = "_".join([os.environ.get("MASTER_ADDR", "local"),
                            os.environ.get("MASTER_PORT", "0"),
                            os.environ.get("TORCHELASTIC_RUN_ID", "none"),
                            # a restarted attempt of an elastic agent gets a key of its own: its ranks can
                            # never join the directory of the attempt before (whose rank 0 may be hung but alive)
                            os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"),
                            os.environ.get("GRAIL_RDZV_NONCE", str(os.getppid()))])
        self.timeout = timeout
        self.seq = 0
        pointer = os.path.join("/tmp", "grail_rdzv_" + _sanitize(key) + ".ptr")
        born = time.time()
        if self.rank == 0:
            try:
                os.remove(pointer)                       # whatever an earlier launch left behind
            except FileNotFoundError:
                pass
            self.dir = tempfile.mkdtemp(prefix="grail_rdzv_" + _sanitize(key) + "_", dir="/tmp")
            tmp = f"{pointer}.tmp{os.getpid()}"
            with open(tmp, "w") as f:
                f.write(f"{self.dir}\n{os.getpid()}\n{born!r}\n")
            os.replace(tmp, pointer)
        else:
            # accept only a pointer written by a live rank 0 of THIS launch: the ranks of one launch
            # start within seconds of each other, a leftover pointer is older than that or dead
            t0 = time.perf_counter()
            while True:
                try:
                    with open(pointer) as f:
                        d, pid, stamp = f.read().split("\n")[:3]
                    if _pid_alive(int(pid)) and float(stamp) > born - 300.0 and os.path.isdir(d):
                        self.dir = d
                        break
                except (OSError, ValueError):
                    pass
                if time.perf_counter() - t0 > self.timeout:
                    raise TimeoutError(f"rendezvous timed out waiting for {pointer}")
                time.sleep(0.001)
        self.pointer = pointer

    def _path(self, name, rank=None):
        return os.path.join(self.dir, name if rank is None else f"{name}.{rank}")

    def _write(self, path, data):
        tmp = f"{path}.tmp{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(data)
        os.replace(tmp, path)          # atomic: readers never see a partial file

    def _wait(self, path):
        t0 = time.perf_counter()
        while not os.path.exists(path):
            if time.perf_counter() - t0 > self.timeout:
                raise TimeoutError(f"rendezvous timed out waiting for {path}")
            time.sleep(0.0002)
        with open(path, "rb") as f:
            return f.read()

    def barrier(self):
        self.seq += 1
        name = f"bar{self.seq}"
        self._write(self._path(name, self.rank), b"1")
        for r in range(self.world):
            self._wait(self._path(name, r))

    def broadcast_bytes(self, data, src=0):
        self.seq += 1
        path = self._path(f"bc{self.seq}")
        if self.rank == src:
            self._write(path, data)
            return data
        return self._wait(path)

    def gather_bytes(self, data):
        """Every rank contributes a byte string; every rank gets the list of all of them."""
        self.seq += 1
        name = f"gb{self.seq}"
        self._write(self._path(name, self.rank), data)
        return [self._wait(self._path(name, r)) for r in range(self.world)]

    def gather_doubles(self, values):
        """Every rank contributes a tuple of floats; every rank gets the list of all tuples."""
        out = []
        for raw in self.gather_bytes(struct.pack(f"<{len(values)}d", *values)):
            out.append(struct.unpack(f"<{len(raw) // 8}d", raw))
        return out

    def close(self):
        self.barrier()
        if self.rank == 0:
            time.sleep(0.2)            # let the others leave the last barrier's poll loop
            shutil.rmtree(self.dir, ignore_errors=True)
            try:
                os.remove(self.pointer)
            except OSError:
                pass
