"""Multi-GPU plumbing: one process per GPU, object points sharded over ranks.

The reduced camera system [S | g | diag] built from each rank's shard of
object points is summed over the ranks once per linearisation (SURVEY.md
8(e)); scalar packs (r'r, ||Jp||^2, ...) likewise.  The collective is
`torch.distributed.all_reduce` -- backend "nccl" is RCCL over xGMI on ROCm,
"gloo" on CPU for the world_size-2 tests.  The C core calls back into
`Comm.allreduce_ptr` with a raw device pointer (include/dbat_hip.h,
dbat_hip_allreduce_fn).
"""
from __future__ import annotations

import numpy as np


class _DevMem:
    """Expose a raw device allocation through __cuda_array_interface__."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {
            'shape': (int(count),), 'typestr': '<f8', 'data': (int(ptr), False), 'version': 2,
            'strides': None,
        }


class Comm:
    """Thin wrapper of a torch.distributed process group."""

    def __init__(self, group=None):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.n_collectives = 0
        self.bytes_reduced = 0

    def allreduce_tensor(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        self.n_collectives += 1
        self.bytes_reduced += t.numel() * t.element_size()
        return t

    def allreduce_ptr(self, ptr, count, stream):
        """Sum-all-reduce `count` doubles at device address `ptr`, ordered on
        the HIP stream `stream` (the core's stream)."""
        import torch
        t = torch.as_tensor(_DevMem(ptr, count), device='cuda')
        ext = torch.cuda.ExternalStream(stream) if stream else torch.cuda.current_stream()
        with torch.cuda.stream(ext):
            self.allreduce_tensor(t)
        return 0

    def allreduce_numpy(self, a):
        """Sum-all-reduce a host array (final gather of the sharded result)."""
        import torch
        a = np.ascontiguousarray(a, dtype=np.float64)
        t = torch.from_numpy(a.copy())
        if self.backend == 'nccl':
            t = t.cuda()
        self.allreduce_tensor(t)
        return t.cpu().numpy().reshape(a.shape)

    def barrier(self):
        self.dist.barrier(group=self.group)


def shard_ranges(s, world_size):
    """Object-point shard of every rank: [lo,hi) in the core's processing
    order (points sorted by camera signature, balanced by observation count).
    Host only -- uses dbat_hip_plan."""
    from . import _hip
    return [(_hip.plan(s, r, world_size)['pt_lo'], _hip.plan(s, r, world_size)['pt_hi'])
            for r in range(world_size)]
