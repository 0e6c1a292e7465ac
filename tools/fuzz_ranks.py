"""Randomised comparison of the multi-GPU runners with the one-GPU path, on ONE MI355X (HIP against HIP, byte for byte).

No multi-GPU node has been available to this build, so the N > 1 protocols of instagraal_amd/multi_gpu.py are exercised the way
tests/test_hip_sampler.py::test_batch_slots_split_over_two_ranks_equal_one_gpu does it: N contexts on the one GPU, one thread per
rank, the collectives in process (an all-gather of the record blocks, an all-reduce of the int64 partial sums).  Here over seeded
random problems, world sizes 2 .. 8, batch widths, neighbours per move and pool sizes (a small slice pool forces the re-run / growth
paths on every rank):

    BatchRunner    the slots of a speculative batch split over the ranks, one all-gather per batch
    ShardedRunner  the contact rows split over the ranks, one all-reduce per move

Every rank must return the records of ``ig_step_batch`` on one context and end with the same genome.

    python tools/fuzz_ranks.py [cases] [first seed]
"""
import os
import sys
import threading
import time
import warnings

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
import torch

warnings.filterwarnings("ignore", category=RuntimeWarning)
from instagraal_amd import synth
from instagraal_amd.multi_gpu import BatchRunner, ShardedRunner
from instagraal_amd.sampler import sampler as hip_sampler

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
COLS = ["o", "dist", "op_sampled", "id_f_sampled", "mean_len", "n_contigs"]


class InProcessDist:
    class ReduceOp:
        SUM = 0

    def __init__(self, rank, world, barrier, parts):
        self.rank, self.world, self.barrier, self.parts = rank, world, barrier, parts

    def get_backend(self):
        return "in-process"

    def all_gather_into_tensor(self, out, mine):
        self.parts[("ag", self.rank)] = mine
        torch.cuda.synchronize()
        self.barrier.wait(timeout=120)
        chunk = mine.numel()
        for r in range(self.world):
            out[r * chunk:(r + 1) * chunk].copy_(self.parts[("ag", r)])
        torch.cuda.synchronize()
        self.barrier.wait(timeout=120)

    def all_reduce(self, t, op=None):
        self.parts[("ar", self.rank)] = t.clone()
        torch.cuda.synchronize()
        self.barrier.wait(timeout=120)
        tot = self.parts[("ar", 0)].clone()
        for r in range(1, self.world):
            tot += self.parts[("ar", r)]
        t.copy_(tot)
        torch.cuda.synchronize()
        self.barrier.wait(timeout=120)


def make_case(seed):
    r = np.random.RandomState(seed)
    n_frags = int(r.choice([150, 300, 700, 1500, 4000]))
    per = int(r.choice([8, 30, 100, 300]))
    per = max(8, min(per, 400000 // n_frags, n_frags))
    mean_len = int(r.choice([3, 15, 50, 200, 1000]))
    mean_len = min(mean_len, max(3, n_frags // 3))
    prob = synth.make_problem(n_frags, n_frags * per, 9000 + seed, mean_len, cis_frac=float(r.choice([0.3, 0.6, 0.8])))
    world = int(r.choice([2, 2, 3, 4, 8]))
    width = int(r.choice([world, 8, 10, 24, 40]))
    if os.environ.get("FUZZ_WORLD"):  # (narrowing a finding down)
        world = int(os.environ["FUZZ_WORLD"])
    if os.environ.get("FUZZ_WIDTH"):
        width = int(os.environ["FUZZ_WIDTH"])
    n_nb = int(r.choice([3, 5, 5, 9]))
    pool = int(r.choice([0, 0, 2000, 20000]))
    params = dict(prob.params) if r.randint(2) else synth.settled_params(prob.params)
    return prob, params, world, width, n_nb, pool, dict(n_frags=n_frags, per=per, mean_len=mean_len, world=world, width=width, neighbours=n_nb, pool=pool)


def run_world(prob, params, world, frags, cands, make_runner):
    barrier = threading.Barrier(world)
    parts = {}
    samplers = []
    for _ in range(world):
        s = hip_sampler(**prob.sampler_kwargs(), device_id=0)
        s.set_param_simu(params)
        s.eval_likelihood_init()
        samplers.append(s)
    got, errs = [None] * world, []

    def work(rk):
        try:
            torch.cuda.set_device(0)
            runner = make_runner(samplers[rk].ctx, rk, InProcessDist(rk, world, barrier, parts))
            if os.environ.get("FUZZ_SERIAL"):  # (narrowing a finding down: one rank inside the library at a time, the device drained behind every call)
                ctx = samplers[rk].ctx

                def locked(f):
                    def g(*a, **kw):
                        with SERIAL:
                            r_ = f(*a, **kw)
                            torch.cuda.synchronize()
                            return r_
                    return g
                for name in ("batch_upload", "batch_score", "batch_commit", "batch_results", "step_begin", "step_finish"):
                    setattr(ctx, name, locked(getattr(ctx, name)))
            got[rk] = runner.run(frags, cands)
        except Exception as e:
            errs.append(e)
            barrier.abort()

    th = [threading.Thread(target=work, args=(rk,)) for rk in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        for s in samplers:
            s.free_gpu()
        raise errs[0]
    out = [(got[rk][COLS].tobytes(), samplers[rk].gpu_vect_frags.copy_from_gpu().soa17().tobytes()) for rk in range(world)]
    for s in samplers:
        s.free_gpu()
    return out


SERIAL = threading.Lock()
bad = 0
t00 = time.time()
for k in range(n_cases):
    seed = seed0 + k // max(1, int(os.environ.get("FUZZ_REPEAT", "1")))  # (FUZZ_REPEAT=n: every case n times, for one-off failures)
    try:
        prob, params, world, width, n_nb, pool, desc = make_case(seed)
    except ValueError as ex:
        print("case %4d skipped: %s" % (seed, str(ex)[:80]), flush=True)
        continue
    t0 = time.time()
    try:
        os.environ.pop("IG_POOL_ENTRIES", None)
        np.random.seed(seed)
        ref = hip_sampler(**prob.sampler_kwargs(), device_id=0)
        ref.set_param_simu(params)
        ref.eval_likelihood_init()
        frags = np.resize(np.random.permutation(prob.n_frags), 160).astype(np.int32)
        cands = ref.draw_candidates(frags, n_nb)
        res = ref.ctx.step_batch(frags, cands)
        want = (res[COLS].tobytes(), ref.gpu_vect_frags.copy_from_gpu().soa17().tobytes())
        want40 = res[:40][COLS].tobytes()
        ref.free_gpu()
        if pool:
            os.environ["IG_POOL_ENTRIES"] = str(pool)
        ob = run_world(prob, params, world, frags, cands, lambda ctx, rk, d: BatchRunner(ctx, rk, world, dist=d, width=width))
        ok_b = all(o == want for o in ob)
        ok_s = True
        if k % 2 == 0:  # (one all-reduce per move: fewer moves)
            osd = run_world(prob, params, world, frags[:40], cands[:40], lambda ctx, rk, d: ShardedRunner(ctx, rk, world, dist=d))
            ok_s = all(o[0] == want40 for o in osd) and len({o[1] for o in osd}) == 1
        ok = ok_b and ok_s
        print("case %4d %s %s  (batch runner %s, sharded runner %s; %.1f s)" % (seed, "ok  " if ok else "DIFF", desc, ok_b, ok_s if k % 2 == 0 else "-",
                                                                               time.time() - t0), flush=True)
        bad += not ok
    except Exception as ex:
        bad += 1
        print("case %4d FAIL %s: %r" % (seed, desc, ex), flush=True)
    finally:
        os.environ.pop("IG_POOL_ENTRIES", None)
print("%d cases, %d bad, %.0f s" % (n_cases, bad, time.time() - t00))
sys.exit(1 if bad else 0)
