"""Randomised comparison of the nuisance run's code paths on one MI355X (no oracle: HIP against HIP, byte for byte).

For a stream of seeded random problems (bins, contacts, contig lengths, count scale), parameters (the synthetic defaults, a
settled chain's, random slopes / d_max), temperature schedules, warm-up lengths and -- now and then -- the zero-score fault
injection (ig_debug_set_zero_inject), the same run of (move, nuisance step) pairs goes

    A  through the chains, segments driven by the helper thread                  (the default)
    B  one pair per library call                                                 (ig_set_nuis_chain(0))
    C  through the chains, segments driven on the caller's thread                (IG_NUIS_ASYNC=0; every third case)
    D  with the exact pass on every step                                         (ig_set_nuis_screen(0); every fourth case)

and must return the same move records, 8-tuples, parameters, genome, generator state; the maintained exact sums must equal a
from-scratch pass and the maintained histogram one built from the final tables, in every run.

    python tools/fuzz_chains.py [cases] [first seed]
"""
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import warnings

import numpy as np

warnings.filterwarnings("ignore", category=RuntimeWarning)
from instagraal_amd import hip_lib, synth
from instagraal_amd.sampler import sampler as hip_sampler

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1


def make_case(seed):
    r = np.random.RandomState(seed)
    n_frags = int(r.choice([150, 300, 700, 1500, 4000, 9000, 20000]))
    per = int(r.choice([15, 60, 150, 400]))
    mean_len = int(r.choice([4, 15, 50, 200, 1000]))
    mean_len = min(mean_len, max(4, n_frags // 3))
    per = max(15, min(per, 1500000 // n_frags, n_frags))  # (the host builds every problem: a few seconds at most; a fifth of the pairs at most)
    prob = synth.make_problem(n_frags, n_frags * per, 1000 + seed, mean_len)
    scale = int(r.choice([1, 1, 1, 7, 40]))
    if scale > 1:
        import copy

        import scipy.sparse as sp
        prob = copy.deepcopy(prob)
        cnt = prob.coo_cnt.copy()
        cnt[::3] *= scale
        prob.coo_cnt = cnt
        M = prob.n_sub_frags
        prob.sub_csr = sp.csr_matrix((cnt, (prob.coo_row, prob.coo_col)), shape=(M, M), dtype=np.int32)
        prob.sub_csr.sort_indices()
    kind = int(r.choice([0, 1, 1, 1, 2, 3]))
    params = dict(prob.params)
    if kind == 1:
        params = synth.settled_params(prob.params)
    elif kind == 2:
        params = dict(prob.params, slope=-float(r.uniform(0.4, 1.6)), d_max=float(prob.params["d_max"]) * float(r.choice([0.5, 1.0, 30.0, 3000.0])))
    n = int(r.choice([120, 250, 500]))
    warm = int(r.choice([0, 0, 200, 900]))
    temp = int(r.randint(3))
    inject = int(r.choice([0, 0, 0, 5, 17]))
    global POOL
    POOL = int(r.choice([0, 0, 0, 3000, 30000]))  # (a small slice pool: slots re-run, the pool grown, chains ended by an overflow)
    return prob, params, scale, n, warm, temp, inject, dict(pool=POOL, n_frags=n_frags, per=per, mean_len=mean_len, scale=scale, params=kind, n=n,
                                                            warm=warm, temp=temp, inject=inject)


class cooled(hip_sampler):
    mode = 0

    def temperature(self, t, n_step):
        if self.mode == 1:
            return 0.5 + 0.25 * (t % 3)
        if self.mode == 2:
            return 3.0 / (1.0 + 0.01 * t)
        return 1.0


POOL = 0


def run(prob, params, scale, n, warm, temp, inject, seed, chain, async_, screen):
    os.environ["IG_NUIS_ASYNC"] = async_
    os.environ.pop("IG_POOL_ENTRIES", None)
    if POOL:
        os.environ["IG_POOL_ENTRIES"] = str(POOL)
    hip_lib.set_nuis_chain(chain)
    hip_lib.set_nuis_screen(screen)
    hip_lib.set_nuis_hist(2)
    hip_lib.debug_set_zero_inject(inject)
    try:
        np.random.seed(seed)
        kw = prob.sampler_kwargs()
        s = cooled(**kw, device_id=0, coo=None)
        s.mode = temp
        s.set_param_simu(params)
        s.bins = np.arange(1.0, 60.0, 1.0)
        s.eval_likelihood_init()
        frags = np.resize(np.random.permutation(prob.n_frags), n + warm)
        if warm:
            s.step_sampler_nuisance_batch(frags[:warm], 5, s.dt, 0, n + warm)
        res, tuples = s.step_sampler_nuisance_batch(frags[warm:], 5, s.dt, warm, n + warm)
        sums, _ = s.ctx.debug_globals()
        _, _, limbs = s.ctx.full_likelihood(0)
        assert [int(x) for x in sums[:5]] == [int(x) for x in limbs[:5]], "maintained sums != from-scratch pass"
        mm = s.ctx.debug_nuis_hist_check() if screen else -1  # (-1: no histogram kept: the tier's cost model left it, or screening is off)
        assert mm <= 0, "maintained histogram != one built from the final tables (%d words)" % mm
        out = (res[["o", "dist", "op_sampled", "id_f_sampled", "n_contigs"]].tobytes(),
               [tuple(float(np.ravel(x)[0]) for x in q[:7]) for q in tuples], s.gpu_vect_frags.copy_from_gpu().soa17().tobytes(),
               np.random.get_state()[1][:8].tobytes(), [float(s.param_simu[k][0]) for k in ("fact", "slope", "d_max", "v_inter")])
        st = s.ctx.debug_nuis_chain_stats()
        st["accepted"] = int(sum(q[6] for q in out[1]))
        st["zero_fallbacks"] = s.ctx.debug_zero_fallbacks()
        st["hist_kept"] = mm == 0
        s.free_gpu()
        return out, st
    finally:
        hip_lib.set_nuis_chain(1)
        hip_lib.set_nuis_screen(1)
        hip_lib.set_nuis_hist(1)
        hip_lib.debug_set_zero_inject(0)
        os.environ.pop("IG_NUIS_ASYNC", None)
        os.environ.pop("IG_POOL_ENTRIES", None)


bad = 0
t00 = time.time()
tot_pairs = tot_chain = 0
for k in range(n_cases):
    seed = seed0 + k
    try:
        prob, params, scale, n, warm, temp, inject, desc = make_case(seed)
    except ValueError as ex:  # (a shape the generator has no pairs for)
        print("case %4d skipped: %s" % (seed, str(ex)[:80]), flush=True)
        continue
    t0 = time.time()
    try:
        a, sa = run(prob, params, scale, n, warm, temp, inject, seed, 1, "1", 1)
        ta = time.time() - t0
        b, sb = run(prob, params, scale, n, warm, temp, inject, seed, 0, "1", 1)
        tb = time.time() - t0 - ta
        ok = a == b
        extra = ""
        if k % 3 == 0:
            c, _ = run(prob, params, scale, n, warm, temp, inject, seed, 1, "0", 1)
            ok = ok and c == b
            extra += " +sync"
        if k % 4 == 0:
            d, _ = run(prob, params, scale, n, warm, temp, inject, seed, 1, "1", 0)
            ok = ok and d == b
            extra += " +exact"
        tot_pairs += n + warm
        tot_chain += sa["pairs"]
        print("case %3d %s %s: %s  chains %d calls / %d pairs of %d, accepted %d, zero fallbacks %d, histogram %s  (%.2f / %.2f s chains / one pair per call; %.1f s%s)" % (
            seed, "ok  " if ok else "DIFF", desc, "", sa["calls"], sa["pairs"], n + warm, sa["accepted"], sa["zero_fallbacks"], "kept" if sa["hist_kept"] else "dropped", ta, tb,
            time.time() - t0, extra), flush=True)
        bad += not ok
    except Exception as e:  # a failed library call is a finding as well
        bad += 1
        print("case %3d FAIL %s: %r" % (seed, desc, e), flush=True)
print("%d cases, %d bad, %d of %d pairs decided in chains, %.0f s" % (n_cases, bad, tot_chain, tot_pairs, time.time() - t00))
sys.exit(1 if bad else 0)
