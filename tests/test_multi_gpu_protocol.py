"""world_size-2 gloo test of the multi-GPU move protocol (instagraal_amd/multi_gpu.py): every rank
scores its share of the rows, ONE all-reduce(SUM) of exact int64 limbs, identical totals everywhere,
equal to the unsharded sums for any partition.  The device context is replaced by a CPU stand-in that
produces deterministic per-row integer contributions -- the collective plumbing is what is under test."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


class FakeShardCtx:
    """Mimics the two-phase move API; partial sums = sum over the rows this rank owns of a fixed
    pseudo-random int64 per (row, slot)."""

    N_PART = 8 * 52

    def __init__(self):
        import torch

        self.rank, self.world = 0, 1
        self.t = torch.zeros(self.N_PART, dtype=torch.int64)
        self.seen = []

    def set_shard(self, rank, world):
        self.rank, self.world = rank, world

    def step_begin(self, frag_a, cands):
        rng = np.random.RandomState(1000 + int(frag_a))
        rows = rng.randint(-2 ** 40, 2 ** 40, size=(97, self.N_PART)).astype(np.int64)
        mine = rows[np.arange(97) % self.world == self.rank].sum(axis=0)
        self.t[:] = __import__("torch").from_numpy(mine)
        self.full = rows.sum(axis=0)

    def step_finish(self, n_cands, want_scores=False):
        got = self.t.numpy().copy()
        assert np.array_equal(got, self.full)
        self.seen.append(got)

        class R:
            o = float(got[0] % 1000); dist = 0.0; mean_len = 1.0; op_sampled = int(got[1] % 24); id_f_sampled = 0
            n_contigs = 1; n_candidates = n_cands; n_slice = 0; n_evals = 0; bytes_min = 0; error = 0; pad = 0

        return R(), None


def _worker(rank, world, port, q):
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    from instagraal_amd.multi_gpu import ShardedRunner

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = FakeShardCtx()
    runner = ShardedRunner(ctx, rank, world, dist=dist, tensor_factory=lambda: ctx.t)
    frags = np.arange(5, dtype=np.int32)
    cands = np.array([[1, 2, 3, -1, -1]] * 5, np.int32)
    res = runner.run(frags, cands)
    q.put((rank, [s.tolist() for s in ctx.seen], res["op_sampled"].tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_allreduce_of_exact_limbs():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out.sort()
    assert out[0][1] == out[1][1] and out[0][2] == out[1][2]


# ---------------------------------------------------------------- slots of a speculative batch split over the ranks


class FakeBatchCtx:
    """Mimics the batch-step API (ig_batch_*): the score records of slot w of the batch starting at move0 are a
    deterministic byte pattern; the commit step checks that it sees the records of EVERY slot, whoever produced them."""

    REC_B = 112

    def __init__(self):
        self.t = {}
        self.log = []

    def tensor(self, kind, nbytes):
        import torch

        self.t[kind] = torch.zeros(nbytes, dtype=torch.uint8)
        return self.t[kind]

    def batch_records(self):
        return 0, self.REC_B

    def batch_upload(self, frags, cands, max_w):
        self.frags, self.max_w = np.asarray(frags), max_w

    @staticmethod
    def pattern(move, nbytes, salt):
        return np.random.RandomState(7919 * int(move) + salt).randint(0, 256, nbytes).astype(np.uint8)

    def batch_score(self, move0, w, b, e):
        import torch

        for kind, nb, salt in (("records", self.REC_B, 1),):
            self.t[kind].zero_()
            for slot in range(b, e):
                self.t[kind][slot * nb:(slot + 1) * nb] = torch.from_numpy(self.pattern(self.frags[move0 + slot], nb, salt))

    def batch_commit(self, move0, w):
        for kind, nb, salt in (("records", self.REC_B, 1),):
            got = self.t[kind].numpy()
            for slot in range(w):
                assert np.array_equal(got[slot * nb:(slot + 1) * nb], self.pattern(self.frags[move0 + slot], nb, salt)), (kind, move0, slot)
        n = min(w, 1 + (move0 * 7) % 5)  # a deterministic "conflict-free prefix"
        self.log.append((move0, w, n))
        return n

    def batch_results(self, n):
        from instagraal_amd import hip_lib

        return np.zeros(n, hip_lib.MOVE_RESULT_DTYPE)


def _batch_worker(rank, world, port, q):
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    from instagraal_amd.multi_gpu import BatchRunner

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = FakeBatchCtx()
    runner = BatchRunner(ctx, rank, world, dist=dist, width=7, tensor_factory=ctx.tensor)
    frags = np.arange(100, 143, dtype=np.int32)
    cands = np.tile(np.array([[1, 2, 3, -1, -1]], np.int32), (frags.size, 1))
    res = runner.run(frags, cands)
    q.put((rank, ctx.log, len(res)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_batch_slots_allgather():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_batch_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out.sort()
    assert out[0][1] == out[1][1]  # both ranks walked the same batches in lockstep
    assert sum(n for _, _, n in out[0][1]) == 43 and out[0][2] == 43
    assert all(w <= 7 for _, w, _ in out[0][1]) and any(w == 7 for _, w, _ in out[0][1])  # ragged last chunk: rank 1 owns 3 of 7 slots
