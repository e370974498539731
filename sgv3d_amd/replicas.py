"""Multi-GPU inference: one process per GPU, frames sharded as independent replicas.

The camera->BEV forward has no cross-frame state (BN in eval mode, one sweep), so the data path
needs NO collective: each rank runs the same HIP forward on its own frames.  What
remains is the measurement protocol the driver specifies — barrier + device synchronise on both
sides of the timed region, MAX over ranks of the elapsed time — and gathering results off the timed
path (the reference does that with ``all_gather_object``, utils/torch_dist.py:37-43).
Backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests.
"""
import os
import time

import torch


def shard_frames(items, rank, world):
    """Contiguous, balanced split of a list of frames over ranks (sizes differ by at most 1)."""
    n = len(items)
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    return items[start:start + base + (1 if rank < extra else 0)]


class ReplicaGroup:
    def __init__(self, backend=None, device=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.device = device
        self.dist = None
        # SGV3D_FORCE_DIST=1 forms the (1-rank) process group anyway: exercises the RCCL barrier / MAX-reduce of
        # the measurement protocol on a single-GPU box
        if self.world > 1 or os.environ.get("SGV3D_FORCE_DIST"):
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
            kw = {}
            if backend == "nccl" and device is not None:
                kw["device_id"] = device
            dist.init_process_group(backend, rank=self.rank, world_size=self.world, **kw)
            self.dist = dist
        self.backend = backend

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if self.dist is None:
            return float(seconds)
        dev = self.device if (self.backend == "nccl" and self.device is not None) else "cpu"
        t = torch.tensor([seconds], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def timed(self, step, steps, local_out=None):
        """Exactly ``steps`` calls of ``step`` bracketed by barrier + synchronize; MAX over ranks.
        ``local_out`` (a one-element list) receives this rank's own time: barrier | steps | device synchronise."""
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        if local_out is not None:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            local_out[0] = time.perf_counter() - t0
        self.barrier()
        return self.max_over_ranks(time.perf_counter() - t0)

    def aggregate_throughput(self, units_per_rank_per_step, steps, elapsed):
        return self.world * units_per_rank_per_step * steps / elapsed

    def all_gather_object(self, obj):
        if self.dist is None:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None
