"""GPU parity pins added in round 4 ("cheap parity pins"):

* the HEADLINE shape (BASELINE.json configs[2]: 50 k bins / 50 M contacts) through ``step_sampler_batch`` -- the path
  ``bench.py`` times: speculative batches, two-tier scoring, the candidate draw inside the call -- against the oracle run live
  (DET mode, 16 threads: ~0.7 s per move), move records, genome and generator state;
* the oracle in LIBM mode (reference-shaped arithmetic: glibc's powf / expf / log10 composed as the CUDA source composes
  CUDA's, block-tree f64 sums: the ONE arithmetic check that does not pass through include/ig_detmath.h) run live at ``small``
  (2 x 200 moves), at cfg2 (40 moves) and on a problem with counts in the thousands: every HIP score within the north
  star's 1e-6 relative, identical winners up to the first move where the two arithmetic modes part ways, which has to be a
  near-tie (the two winners' scores within 1e-6 of each other in BOTH arithmetics);
* the decide step's zero-score rule (a score of exactly 0.0 counts as "not scored" in the reference's argmax, CL:1435-1440):
  fault injection makes every n-th move of a two-tier batch take the fallback (scored again, every column exact); records,
  genome and sums are those of the run without it, on the plain batch path and on the nuisance-on path.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL = 1e-6  # BASELINE.json north_star


def _hip(prob, coo=True):
    from instagraal_amd.sampler import sampler as hip_sampler

    s = hip_sampler(**prob.sampler_kwargs(), device_id=0, coo=(prob.coo_row, prob.coo_col, prob.coo_cnt) if coo else None)
    s.set_param_simu(prob.params)
    s.bins = np.arange(1.0, 60.0, 1.0)
    s.eval_likelihood_init()
    return s


def _oracle(prob, mode):
    from oracle.sampler_oracle import OracleSampler

    o = OracleSampler(**prob.sampler_kwargs(), mode=mode)
    o.set_param_simu(prob.params)
    o.bins = np.arange(1.0, 60.0, 1.0)
    o.eval_likelihood_init()
    return o


def test_cfg3_live_oracle_batch_path():
    """12 moves of the headline shape through the batch path against OracleSampler(DET): 6-tuples, state, generator state."""
    import os

    from instagraal_amd import synth
    from oracle import oracle_lib as ol

    ol.build()
    prob = synth.make_problem(*synth.CONFIGS["cfg3"])
    assert (prob.n_frags, prob.n_contacts) == (50_000, 50_000_000)
    s = _hip(prob)
    ol.set_threads(min(16, os.cpu_count() or 1))
    try:
        o = _oracle(prob, ol.MODE_DET)
        assert float(s.curr_likelihood_on_nz[0]) == float(o.gpu_curr_likelihood_nz[0])
        np.random.seed(11)
        frags = np.random.permutation(prob.n_frags)[:12].astype(np.int32)
        st = np.random.get_state()
        res = s.step_sampler_batch(frags, 5)
        after = np.random.get_state()
        np.random.set_state(st)
        for f, r in zip(frags, res):
            b = o.step_sampler(int(f), 5, o.dt)
            got = (float(r["o"]), float(r["dist"]), int(r["op_sampled"]), int(r["id_f_sampled"]), float(np.float32(r["mean_len"])), int(r["n_contigs"]))
            assert got == (b[0], b[1], b[2], b[3], float(b[4]), int(b[5])), (int(f), got, b)
        assert np.array_equal(np.random.get_state()[1], after[1]) and np.random.get_state()[2] == after[2], "generator state differs"
        assert np.array_equal(s.gpu_vect_frags.copy_from_gpu().soa17(), o.gpu_vect_frags.soa17())
        assert s.ctx.batch_stats()["batches"] <= 3  # the 12 moves were scored together, not one by one
    finally:
        ol.set_threads(1)
    s.free_gpu()


@pytest.mark.slow
def test_cfg3_late_live_oracle():
    """The headline's size where an assembly ENDS (synth cfg3_late: 50 k bins / 50 M contacts in 20 contigs; windows of 5 000 - 45 000
    sub-fragments: past every LDS stage, the fused commit's 4 096 and the one-column screening routine) against OracleSampler(DET):
    behind four forced mutations of long contigs, 14 moves through the batch path (window rule, two-tier scoring, draw inside the call) and
    4 more one step_sampler call at a time --
    6-tuples, genome, generator state (the reference's own GPU test ends with 15 - 45 contigs,
    /root/reference/tests/test_instagraal_gpu.py:126-340, via paste_contigs KA:3367-3693 / insert_block KA:2724-2975)."""
    import os

    from instagraal_amd import synth
    from oracle import oracle_lib as ol

    ol.build()
    prob = synth.make_problem(*synth.CONFIGS["cfg3_late"])
    assert (prob.n_frags, prob.n_contacts) == (50_000, 50_000_000)
    n_ctg = np.unique(prob.S_o_A_frags["id_c"]).size
    assert n_ctg <= 25 and int(prob.S_o_A_frags["sub_l_cont"].max()) >= 5000
    s = _hip(prob)
    ol.set_threads(min(16, os.cpu_count() or 1))
    try:
        o = _oracle(prob, ol.MODE_DET)
        assert float(s.curr_likelihood_on_nz[0]) == float(o.gpu_curr_likelihood_nz[0])
        # at the true layout 3 % of the moves change the genome: four forced mutations first (CL:2094-2151 on both sides: a flip, a pop-out /
        # pop-in, a block insert into another contig, a translocation), then their bins among the moves -- the sampler puts them back
        rng = np.random.RandomState(5)
        big = np.argsort(np.bincount(prob.S_o_A_frags["id_c"]))[::-1][:3]  # the three longest contigs
        forced = []
        for op, ca, cb in ((0, big[0], big[0]), (2, big[0], big[1]), (14, big[1], big[2]), (9, big[2], big[0])):
            a = int(rng.choice(np.nonzero(prob.S_o_A_frags["id_c"] == ca)[0][5:-5]))
            b = int(rng.choice(np.nonzero(prob.S_o_A_frags["id_c"] == cb)[0][5:-5]))
            if a == b:
                b += 1
            max_id = o.modify_gl_cuda_buffer(a, o.dt)
            o.test_copy_struct(a, b, op, max_id)
            o.modify_gl_cuda_buffer(a, o.dt)
            s.test_copy_struct(a, b, op)
            forced.append(a)
        assert np.array_equal(s.gpu_vect_frags.copy_from_gpu().soa17(), o.gpu_vect_frags.soa17())
        np.random.seed(23)
        frags = np.concatenate([forced, np.random.permutation(prob.n_frags)[:14]]).astype(np.int32)
        st = np.random.get_state()
        res = s.step_sampler_batch(frags[:14], 5)
        tuples = [(float(r["o"]), float(r["dist"]), int(r["op_sampled"]), int(r["id_f_sampled"]), float(np.float32(r["mean_len"])), int(r["n_contigs"]))
                  for r in res]
        for f in frags[14:]:
            a = s.step_sampler(int(f), 5, s.dt)
            tuples.append((float(a[0]), float(a[1]), int(a[2]), int(a[3]), float(a[4]), int(a[5])))
        after = np.random.get_state()
        np.random.set_state(st)
        n_changed = 0
        d_prev = None
        for f, got in zip(frags, tuples):
            b = o.step_sampler(int(f), 5, o.dt)
            assert got == (b[0], b[1], b[2], b[3], float(b[4]), int(b[5])), (int(f), got, b)
            n_changed += int(d_prev is not None and b[1] != d_prev)
            d_prev = b[1]
        assert np.array_equal(np.random.get_state()[1], after[1]) and np.random.get_state()[2] == after[2], "generator state differs"
        assert np.array_equal(s.gpu_vect_frags.copy_from_gpu().soa17(), o.gpu_vect_frags.soa17())
        print("cfg3_late: %d of %d compared moves changed the genome" % (n_changed, len(frags)))
        assert n_changed >= 1  # (a long window was applied and its successors scored behind it)
    finally:
        ol.set_threads(1)
    s.free_gpu()


def _libm_walk(prob, seed, n_moves):
    """HIP (the deterministic contract) and the oracle in LIBM mode, each on its own genome, fed the same candidate lists: scores
    within REL, same flags / winners / distances, until the first move where the winners differ -- a near-tie by both
    arithmetics.  -> (moves compared, worst relative score difference)"""
    from oracle import oracle_lib as ol

    s = _hip(prob, coo=False)
    o = _oracle(prob, ol.MODE_LIBM)
    nz_h, nz_o = float(s.curr_likelihood_on_nz[0]), float(o.gpu_curr_likelihood_nz[0])
    assert abs(nz_h - nz_o) <= REL * abs(nz_o)
    np.random.seed(seed)
    frags = np.random.permutation(prob.n_frags)[:n_moves]
    worst, done = 0.0, 0
    for t, f in enumerate(frags):
        cands = s.return_neighbours(int(f), 5)
        a = s.step_sampler(int(f), 5, candidates=cands)
        b = o.step_sampler(int(f), 5, o.dt, candidates=cands)
        hs, os_ = s.all_scores, o.all_scores
        scored = os_ != 0
        assert np.array_equal(scored, hs != 0), t
        rel = np.abs(hs[scored] - os_[scored]) / np.abs(os_[scored])
        worst = max(worst, float(rel.max()))
        assert rel.max() <= REL, (t, float(rel.max()))
        if (a[2], a[3]) != (b[2], b[3]):
            # the two arithmetics part ways: it has to be a near-tie in both
            ih = s.candidates.index(int(a[3])) * 24 + int(a[2])
            io = o.candidates.index(int(b[3])) * 24 + int(b[2])
            for sc in (hs, os_):
                assert abs(sc[ih] - sc[io]) <= REL * abs(sc[io]), (t, sc[ih], sc[io])
            break
        assert abs(a[0] - b[0]) <= REL * abs(b[0]), t
        assert (a[1], int(a[5])) == (b[1], int(b[5])), t
        assert np.array_equal(s.ctx.valid_insert(), o.gpu_list_valid_insert), t
        done = t + 1
    else:
        assert np.array_equal(s.gpu_vect_frags.copy_from_gpu().soa17(), o.gpu_vect_frags.soa17())
    s.free_gpu()
    return done, worst


def test_libm_live_oracle_small():
    from instagraal_amd import synth
    from oracle import oracle_lib as ol

    ol.build()
    prob = synth.make_problem(*synth.CONFIGS["small"])
    total = 0
    for seed in (21, 22, 25):
        done, worst = _libm_walk(prob, seed, 200)
        print("small, seed %d: %d moves compared with the LIBM oracle, worst relative score difference %.2e" % (seed, done, worst))
        total += done
    assert total >= 150, total  # (a near-tie ends a walk early: its two winners were checked to tie in both arithmetics)


def test_libm_live_oracle_cfg2_and_large_counts():
    import copy

    import scipy.sparse as sp

    from instagraal_amd import synth
    from oracle import oracle_lib as ol

    ol.build()
    prob = synth.make_problem(*synth.CONFIGS["cfg2"])
    done, worst = _libm_walk(prob, 23, 40)
    print("cfg2: %d moves compared with the LIBM oracle, worst relative score difference %.2e" % (done, worst))
    # counts in the hundreds and thousands (the log-factorial beyond its LDS table, the Stirling branch)
    big = copy.deepcopy(synth.make_problem(*synth.CONFIGS["small"]))
    cnt = big.coo_cnt.copy()
    cnt[::5] *= 70
    cnt[::53] *= 20
    # (the arithmetic contract clamps a term at +-2^20, include/ig_detmath.h ig_quantize -- reached by a count in the hundreds of
    # thousands against an expectation of a few: outside what the two arithmetics can be compared on)
    assert (cnt >= 1024).sum() > 100 and cnt.max() < 100_000
    big.coo_cnt = cnt
    M = big.n_sub_frags
    big.sub_csr = sp.csr_matrix((cnt, (big.coo_row, big.coo_col)), shape=(M, M), dtype=np.int32)
    big.sub_csr.sort_indices()
    done2, worst2 = _libm_walk(big, 24, 120)
    print("small with large counts: %d moves compared with the LIBM oracle, worst relative score difference %.2e" % (done2, worst2))
    assert done + done2 >= 60, (done, done2)


def test_zero_score_rule_fallback_changes_nothing():
    """every 5th move of a two-tier batch is sent through the fallback of the decide step's zero-score rule (scored again with
    every column exact): same records, genome, flags and maintained sums -- plain batches and the nuisance-on loop"""
    from instagraal_amd import hip_lib, synth

    prob = synth.make_problem(*synth.CONFIGS["small"])
    outs = []
    try:
        for inject in (0, 5):
            hip_lib.debug_set_zero_inject(inject)
            s = _hip(prob, coo=False)
            np.random.seed(31)
            frags = np.random.permutation(prob.n_frags)[:400].astype(np.int32)
            res = s.step_sampler_batch(frags[:300], 5)
            n_fb = s.ctx.debug_zero_fallbacks()
            res2, tuples = s.step_sampler_nuisance_batch(frags[300:], 5, s.dt, 0, 100)
            n_fb2 = s.ctx.debug_zero_fallbacks() - n_fb
            # ... and the reference-shaped call without all_scores (ig_step_draw: a batch of ONE in two tiers, round 5)
            # (every call is move 0 of its own batch: "every move" is the one injection period that reaches it)
            hip_lib.debug_set_zero_inject(1 if inject else 0)
            s.keep_all_scores = False
            rows3 = [tuple(s.step_sampler(int(f), 5, s.dt)) for f in np.random.permutation(prob.n_frags)[:60]]
            s.keep_all_scores = True
            hip_lib.debug_set_zero_inject(inject)
            n_fb3 = s.ctx.debug_zero_fallbacks() - n_fb - n_fb2
            sums, _ = s.ctx.debug_globals()
            _, _, limbs = s.ctx.full_likelihood(0)
            assert [int(x) for x in sums[:5]] == [int(x) for x in limbs[:5]]
            cols = ["o", "dist", "op_sampled", "id_f_sampled", "n_contigs"]
            outs.append((res[cols].tobytes(), res2[cols].tobytes(), rows3, [tuple(float(np.ravel(x)[0]) for x in q[:7]) for q in tuples],
                         s.gpu_vect_frags.copy_from_gpu().soa17().tobytes(), [int(x) for x in s.ctx.valid_insert()],
                         np.random.get_state()[1][:8].tobytes()))
            if inject:
                assert n_fb >= 20 and n_fb2 >= 5 and n_fb3 >= 5, (n_fb, n_fb2, n_fb3)
            else:
                assert n_fb == 0 and n_fb2 == 0 and n_fb3 == 0
            s.free_gpu()
    finally:
        hip_lib.debug_set_zero_inject(0)
    assert outs[0] == outs[1]
