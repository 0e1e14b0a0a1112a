"""Multi-GPU plumbing: one process per GPU, object points sharded over ranks.

The reduced camera system [S | g | diag] built from each rank's shard of
object points is summed over the ranks once per linearisation (SURVEY.md
8(e)); scalar packs (r'r, ||Jp||^2, ...) likewise.  On GPUs the collective
is RCCL inside libdbat_hip.so (ncclAllReduce on the handle's stream over
xGMI): `Comm.attach(handle)` only carries rank 0's unique id to the other
ranks through the torch.distributed group (any backend -- it is 128 bytes of
control plane) and calls dbat_hip_comm_init.  `Comm.allreduce_ptr` is the
test hook (dbat_hip_set_allreduce) for groups without RCCL.
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

    def __init__(self, group=None, via_host=False):
        """via_host: attach() gives a handle its sums through host memory (attach_host) instead of an RCCL
        communicator -- ranks that share a GPU, groups without RCCL."""
        import torch.distributed as dist
        self.via_host = bool(via_host)
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.n_collectives = 0
        self.bytes_reduced = 0
        self.device = None
        try:
            import torch
            if torch.cuda.is_available():
                self.device = torch.cuda.current_device()    # the rank's GPU (set_device(LOCAL_RANK) by the launcher)
        except ImportError:
            pass

    def attach(self, handle):
        """Give `handle` (this rank's shard) its RCCL communicator.  Collective."""
        from . import _hip
        if self.via_host:
            return self.attach_host(handle)
        box = [_hip.comm_unique_id() if self.rank == 0 else None]
        src = self.dist.get_global_rank(self.group, 0) if self.group is not None else 0
        self.dist.broadcast_object_list(box, src=src, group=self.group)
        handle.comm_init(box[0])

    def allreduce_tensor(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        self.n_collectives += 1
        self.bytes_reduced += t.numel() * t.element_size()
        return t

    def allreduce_ptr(self, ptr, count, stream):
        """Sum-all-reduce `count` doubles at device address `ptr`, ordered on
        the HIP stream `stream` (the core's stream)."""
        import torch
        dev = torch.device('cuda', torch.cuda.current_device() if self.device is None else self.device)
        t = torch.as_tensor(_DevMem(ptr, count), device=dev)
        if t.data_ptr() != ptr:      # a pointer of another device would have been copied, not wrapped
            raise RuntimeError('the buffer does not live on %s' % dev)
        ext = torch.cuda.ExternalStream(stream, device=dev) if stream else torch.cuda.current_stream(dev)
        with torch.cuda.stream(ext):
            self.allreduce_tensor(t)
        return 0

    def allreduce_ptr_host(self, ptr, count, stream):
        """The same sum through host memory (device -> host, all-reduce of the group's backend on the host copy,
        host -> device), ordered on `stream`: for groups without RCCL between the ranks' buffers -- several
        processes that share one GPU (the one-GPU test box), gloo."""
        import torch
        dev = torch.device('cuda', torch.cuda.current_device() if self.device is None else self.device)
        t = torch.as_tensor(_DevMem(ptr, count), device=dev)
        if t.data_ptr() != ptr:
            raise RuntimeError('the buffer does not live on %s' % dev)
        ext = torch.cuda.ExternalStream(stream, device=dev) if stream else torch.cuda.current_stream(dev)
        with torch.cuda.stream(ext):
            c = t.cpu()                      # waits for the stream's work on the buffer
            self.allreduce_tensor(c)
            t.copy_(c)
            ext.synchronize()
        return 0

    def attach_host(self, handle):
        """Give `handle` its sums over the ranks through host memory (allreduce_ptr_host) instead of an RCCL
        communicator."""
        self._hook = lambda ptr, count, stream: self.allreduce_ptr_host(ptr, count, stream)
        handle.set_allreduce(self._hook)

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
