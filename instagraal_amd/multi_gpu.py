"""One chain split over N GPUs of a node (one process per GPU, torch.distributed / RCCL).

Two ways to split, both bit-identical to the one-GPU chain for any N:

``BatchRunner`` (used by bench.py): the candidate draws of consecutive moves do not depend on the genome, so W moves
are scored against one state ("speculative batch") and committed in order.  Every rank holds the full problem; rank r
slices and scores the slots [r * W/N, (r+1) * W/N) of each batch, the slot-major score records (exact int64 sums,
~16 KB per slot, one block of memory) are all-gathered once per batch, and every rank runs the same commit step on identical inputs.

``ShardedRunner`` (one move at a time, contact rows split):

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
        self._key = None
        self._stream = None

    def _partials(self):
        """zero-copy int64 view of the library's partial-sum buffer.  The library may move the buffer (more candidates than
        before, a new state): the view is rebuilt whenever pointer or length changed, and checked to alias the buffer."""
        if self._tensor_factory is not None:
            if self._t is None:
                self._t = self._tensor_factory()
            return self._t
        import torch

        ptr, n = self.ctx.partials()
        if self._t is None or self._key != (ptr, n):
            self._t = torch.as_tensor(_DevArray(ptr, n), device="cuda")
            assert self._t.data_ptr() == ptr, "torch copied the partial-sum buffer instead of wrapping it"
            self._key = (ptr, n)
        return self._t

    def run(self, frags, cands):
        if self._tensor_factory is None:
            import torch

            if self._stream is None:  # one stream for kernels and collectives, see BatchRunner.run
                self._stream = torch.cuda.Stream()
                self.ctx.set_stream(self._stream.cuda_stream)
            with torch.cuda.stream(self._stream):
                return self._run(frags, cands)
        return self._run(frags, cands)

    def _run(self, frags, cands):
        frags = np.ascontiguousarray(frags, np.int32)
        cands = np.ascontiguousarray(cands, np.int32)
        res = np.zeros(frags.size, hip_lib.MOVE_RESULT_DTYPE)
        for i in range(frags.size):
            c = cands[i][cands[i] >= 0]
            self.ctx.step_begin(int(frags[i]), c)
            t = self._partials()  # after step_begin: that is where the library (re)allocates
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
            r, _ = self.ctx.step_finish(len(c))
            for k in res.dtype.names:
                res[k][i] = getattr(r, k)
        return res


class _DevBytes:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


class BatchRunner:
    """W moves per batch, the slots of a batch split over the ranks (see the module docstring).

    ``tensor_factory(kind, nbytes)`` (tests) returns the uint8 tensor standing for a record buffer; by default the
    device buffers of the context are wrapped without a copy."""

    def __init__(self, ctx, rank, world, dist=None, width=None, tensor_factory=None, exchange_alone=False):
        """``exchange_alone`` (tests on a one-GPU rig): a world of ONE rank still runs its collective -- the one way to put the
        record buffers through RCCL there (two ranks may not share a device)"""
        self.ctx, self.rank, self.world = ctx, rank, world
        self._collective = world > 1 or bool(exchange_alone)
        if dist is None and self._collective:
            import torch.distributed as dist
        self.dist = dist
        self.width = int(width) if width else max(24, 8 * world)
        self.width = max(world, min(self.width, (64 // world) * world))  # equal chunks of at most 64 slots in total
        self._tensor_factory = tensor_factory
        self._bufs = None
        self._key = None
        self._stream = None
        self._into_tensor = None  # all_gather_into_tensor (RCCL) or all_gather of a list (gloo): _exchange
        self.batches = 0
        self._ema = float(self.width)  # moves a batch gets through, moving average: sets the next width (as ig_step_batch)

    def _buffers(self, cap_slots):
        """zero-copy uint8 view of the library's record buffer; re-queried at the start of every run (the library frees and
        reallocates its move buffers when a batch gets wider or the problem changes)"""
        ptr, nbytes = self.ctx.batch_records()
        key = (ptr, nbytes, cap_slots)
        if self._bufs is None or self._key != key:
            if self._tensor_factory is not None:
                t = self._tensor_factory("records", nbytes * cap_slots)
            else:
                import torch

                t = torch.as_tensor(_DevBytes(ptr, nbytes * cap_slots), device="cuda")
                assert t.data_ptr() == ptr, "torch copied the record buffer instead of wrapping it"
            self._bufs = (t, nbytes)
            self._key = key
        return self._bufs

    def _exchange(self, per):
        """ONE all-gather of the records per batch: rank r produced slots [r * per, (r + 1) * per)"""
        t, b = self._bufs
        chunk = per * b
        out = t[: self.world * chunk]
        mine = out[self.rank * chunk:(self.rank + 1) * chunk].clone()
        # which collective is decided up front from the backend, never by catching a failure: on RCCL a failed collective
        # poisons the communicator, and a fallback would hide the first real error.  gloo (the CPU / one-device test rigs) has
        # no all_gather_into_tensor; the in-process stand-ins of the tests implement what they are asked for.
        if self._into_tensor is None:
            get_backend = getattr(self.dist, "get_backend", None)
            backend = str(get_backend()).lower() if get_backend is not None else ""
            self._into_tensor = hasattr(self.dist, "all_gather_into_tensor") and backend != "gloo"
        if self._into_tensor:
            self.dist.all_gather_into_tensor(out, mine)
        else:
            parts = [out[r * chunk:(r + 1) * chunk] for r in range(self.world)]
            self.dist.all_gather(parts, mine)

    def run(self, frags, cands):
        frags = np.ascontiguousarray(frags, np.int32)
        cands = np.ascontiguousarray(cands, np.int32)
        n = frags.size
        world, rank = self.world, self.rank
        if hasattr(self.ctx, "batch_max_width"):  # work buffers are sized per slot: stay inside what fits (same on every rank)
            fit = self.ctx.batch_max_width(cands.shape[1])
            self.width = max(world, min(self.width, (fit // world) * world))
        per_max = -(-self.width // world)
        cap_slots = per_max * world  # the all-gather works on equal chunks
        if self._tensor_factory is None and self._collective:
            # kernels and collectives must be ordered on ONE stream: a dedicated torch stream (its handle is not the
            # null stream, which ig_set_stream reads as "make your own") becomes the library's stream and, inside the
            # `with`, torch's current stream, which is what ProcessGroupNCCL orders its collectives against
            import torch

            if self._stream is None:
                self._stream = torch.cuda.Stream()
                self.ctx.set_stream(self._stream.cuda_stream)
            with torch.cuda.stream(self._stream):
                return self._run(frags, cands, n, cap_slots)
        return self._run(frags, cands, n, cap_slots)

    def _run(self, frags, cands, n, cap_slots):
        world, rank = self.world, self.rank
        self.ctx.batch_upload(frags, cands, cap_slots)
        if self._collective:
            self._buffers(cap_slots)
        done = 0
        while done < n:
            # the width follows the conflict rate (identical on every rank: it only depends on the committed counts)
            w_want = max(world, min(self.width, int(1.5 * self._ema + 1.5)))
            w_now = min(w_want, n - done)
            per = -(-w_now // world)
            b, e = min(rank * per, w_now), min((rank + 1) * per, w_now)
            self.ctx.batch_score(done, w_now, b, e)
            if self._collective:
                self._exchange(per)
            got = self.ctx.batch_commit(done, w_now)
            if w_now == w_want:
                self._ema = 0.6 * self._ema + 0.4 * (min(2.0 * w_now, float(self.width)) if got >= w_now else float(got))
            done += got
            self.batches += 1
        return self.ctx.batch_results(n)
