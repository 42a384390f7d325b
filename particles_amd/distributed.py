"""Multi-GPU seam: independent SMC runs / SMC^2 islands sharded over one
process per GPU (the reference's ``multiSMC`` -> ``utils.distribute_work`` fan
out to loky worker processes, particles/core.py:431, utils.py:158-186).

The data path has no exchange step: each rank owns ``islands_per_rank`` whole
filters.  The only collective is the gather of the per-island log-evidences,
done with RCCL over xGMI inside libsmc_hip (``smc_comm_*``).  Host-side
rendezvous (rank discovery, barrier, distributing the RCCL unique id, timing
reduction) uses ``torch.distributed`` with the gloo backend and CPU tensors only
-- torch never touches the GPU in these processes.
"""
import ctypes
import os

import numpy as np

from . import _lib
from ._lib import DeviceArray, check, lib


def env_rank_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_islands(n_total, rank, world):
    """Contiguous block partition of ``n_total`` islands: (first, count) of
    ``rank``.  Island ids are global, so the Philox streams (and hence the
    results) do not depend on how many GPUs the job runs on."""
    base, extra = divmod(n_total, world)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


class Group:
    """Process group: gloo for the host side, RCCL (inside libsmc_hip) for the
    device-side gather.  ``device_collective=False`` keeps everything on gloo
    (CPU-only tests)."""

    def __init__(self, device_collective=True, backend_init=True):
        self.rank, self.local_rank, self.world = env_rank_world()
        self.dist = None
        self.comm = None
        self.fallback_reason = None
        if self.world > 1:
            import torch.distributed as dist
            if backend_init and not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            self.dist = dist
        if device_collective:
            try:
                self._init_rccl()
            except Exception as e:           # keep the job alive: 8 bytes/rank can go over gloo
                self.comm = None
                self.fallback_reason = "%s: %s" % (type(e).__name__, e)
            if self.world > 1:
                # all ranks agree on whether RCCL is usable
                ok = self.allreduce_min_host(1.0 if self.comm else 0.0)
                if ok < 1.0 and self.comm:
                    lib().smc_comm_destroy(self.comm)
                    self.comm = None
                    self.fallback_reason = self.fallback_reason or "RCCL unavailable on another rank"

    def _init_rccl(self):
        uid = ctypes.create_string_buffer(128)
        if self.rank == 0:
            check(lib().smc_comm_unique_id(uid))
        if self.world > 1:
            import torch
            t = torch.frombuffer(bytearray(uid.raw), dtype=torch.uint8).clone()
            self.dist.broadcast(t, src=0)
            uid = ctypes.create_string_buffer(bytes(t.numpy().tobytes()), 128)
        h = _lib.c_vp()
        check(lib().smc_comm_create(_lib.ctx().h, self.world, self.rank, uid, ctypes.byref(h)))
        self.comm = h

    # ---- host-side helpers (gloo) -------------------------------------------
    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def _allreduce_host(self, v, op):
        if self.dist is None:
            return float(v)
        import torch
        t = torch.tensor([float(v)], dtype=torch.float64)
        self.dist.all_reduce(t, op=op)
        return float(t[0])

    def allreduce_max_host(self, v):
        import torch.distributed as d
        return self._allreduce_host(v, d.ReduceOp.MAX) if self.dist else float(v)

    def allreduce_min_host(self, v):
        import torch.distributed as d
        return self._allreduce_host(v, d.ReduceOp.MIN) if self.dist else float(v)

    # ---- the collective of the path ------------------------------------------
    def gather_evidence(self, local_logLt):
        """All ranks' per-island log-evidences, concatenated in rank order.
        Every rank must contribute the same number of islands."""
        local = np.ascontiguousarray(local_logLt, dtype=np.float64)
        if self.comm is not None:
            send = DeviceArray.from_numpy(local)
            recv = DeviceArray((self.world * local.size,))
            check(lib().smc_comm_allgather_f64(self.comm, send.ptr, local.size, recv.ptr))
            return recv.get()
        if self.dist is None:
            return local.copy()
        import torch
        outs = [torch.zeros(local.size, dtype=torch.float64) for _ in range(self.world)]
        self.dist.all_gather(outs, torch.from_numpy(local.copy()))
        return np.concatenate([o.numpy() for o in outs])

    def close(self):
        if self.comm is not None:
            lib().smc_comm_destroy(self.comm)
            self.comm = None
        if self.dist is not None and self.dist.is_initialized():
            self.dist.destroy_process_group()


def log_mean_exp_host(v):
    """Combine island evidences: log of the mean of exp(v) (the SMC^2 /
    multi-run estimator of the evidence; resampling.py:291-317 on M values)."""
    v = np.asarray(v, dtype=np.float64)
    m = v.max()
    return float(m + np.log(np.mean(np.exp(v - m))))
