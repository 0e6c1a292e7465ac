"""One chain split over N GPUs of a node (one process per GPU, torch.distributed / RCCL).

The genome state is tiny (68 N bytes) and replicated; every rank holds the contacts (6 GB at the
human-scale shape, 288 GB HBM per GPU) and scores the candidate CSR rows r with r % N == rank of each
move.  The partial sums are exact 64-bit integers (include/ig_detmath.h), so ONE all-reduce(SUM) of
C x 52 int64 (~2 KB) per move gives every rank bit-identical totals for any N; the zero-pixel terms,
the tail walk, the argmax and the apply are O(touched) and run redundantly on every rank.
The message is latency-bound (xGMI point-to-point, ~20 us), never bandwidth-bound.
"""
from __future__ import annotations

import numpy as np

from . import hip_lib


class _DevArray:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<i8", "data": (int(ptr), False), "version": 2}


class ShardedRunner:
    def __init__(self, ctx, rank, world, dist=None, tensor_factory=None):
        self.ctx, self.rank, self.world = ctx, rank, world
        if dist is None:
            import torch.distributed as dist
        self.dist = dist
        ctx.set_shard(rank, world)
        self._tensor_factory = tensor_factory
        self._t = None

    def _partials(self):
        if self._t is None:
            if self._tensor_factory is not None:
                self._t = self._tensor_factory()
            else:
                import torch

                self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
                ptr, n = self.ctx.partials()
                self._t = torch.as_tensor(_DevArray(ptr, n), device="cuda")
        return self._t

    def run(self, frags, cands):
        frags = np.ascontiguousarray(frags, np.int32)
        cands = np.ascontiguousarray(cands, np.int32)
        res = np.zeros(frags.size, hip_lib.MOVE_RESULT_DTYPE)
        t = None
        for i in range(frags.size):
            c = cands[i][cands[i] >= 0]
            self.ctx.step_begin(int(frags[i]), c)
            if t is None:
                t = self._partials()
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
            r, _ = self.ctx.step_finish(len(c))
            for k in res.dtype.names:
                res[k][i] = getattr(r, k)
        return res
