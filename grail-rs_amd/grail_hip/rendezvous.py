"""A torch-free control plane for the ranks of ONE node (the bench contract is single-node).

Why not torch.distributed: a PyTorch-ROCm wheel carries its own private HIP/HSA runtime;
loaded next to the system runtime libgrail_hip.so links, the two fight over the device
(measured on the GPU box: whichever initialises second sees "no ROCm-capable device").
The launcher (`python -m torch.distributed.run`) only has to spawn the ranks and export
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT; the ranks then meet here through
files in a per-launch directory under /tmp: barriers, a scalar gather, and the hand-off of the
128-byte RCCL unique id.  All device-side exchange (the voice table) is RCCL, in the C ABI.
"""
import os
import shutil
import struct
import time


class FileGroup:
    def __init__(self, rank=None, world=None, key=None, timeout=600.0):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
        if key is None:
            key = "_".join([os.environ.get("MASTER_ADDR", "local"),
                            os.environ.get("MASTER_PORT", "0"),
                            os.environ.get("TORCHELASTIC_RUN_ID", "none"),
                            str(os.getppid())])
        self.dir = os.path.join("/tmp", "grail_rdzv_" + "".join(
            c if c.isalnum() or c in "._-" else "_" for c in key))
        os.makedirs(self.dir, exist_ok=True)
        self.timeout = timeout
        self.seq = 0

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

    def gather_doubles(self, values):
        """Every rank contributes a tuple of floats; every rank gets the list of all tuples."""
        self.seq += 1
        name = f"ga{self.seq}"
        self._write(self._path(name, self.rank), struct.pack(f"<{len(values)}d", *values))
        out = []
        for r in range(self.world):
            raw = self._wait(self._path(name, r))
            out.append(struct.unpack(f"<{len(raw) // 8}d", raw))
        return out

    def close(self):
        self.barrier()
        if self.rank == 0:
            time.sleep(0.2)            # let the others leave the last barrier's poll loop
            shutil.rmtree(self.dir, ignore_errors=True)
