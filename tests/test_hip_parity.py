"""GPU parity tests: the HIP path (through the C ABI) against golden vectors captured from the
reference's own sampler Python over the deterministic oracle kernels (mode 1), and against the
oracle run live.  Integer/index work bit-exact; likelihoods bit-exact as well (the arithmetic
contract of include/ig_detmath.h), which implies the 1e-6 relative bound of the north star.

What these goldens pin and what they do not: they come from the reference's own HOST Python (candidate draw, call
order, stale buffers, sort, argmax, apply, renumbering, distance, nuisance step) driving kernels that are the oracle's C
restatement of kernel_sparse_adapt.cu -- the reference's CUDA cannot run here and its tests hold no vector for this path.
Kernel semantics are therefore checked against the builder's reading of the .cu source, not against a CUDA run.  The
checks that do not pass through that reading: tests/test_hip_configs.py (P(s) on the GPU against the reference's own peval;
HIP scores within 1e-6 of the libm-mode goldens, whose arithmetic composes powf / expf / log10 as the CUDA source does)."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def make_ctx(prob, params=None):
    from instagraal_amd import hip_lib
    from instagraal_amd.sampler import problem_to_context

    return problem_to_context(prob, params=params)


@pytest.fixture(scope="module")
def tiny():
    from instagraal_amd import synth

    return synth.make_problem(*synth.CONFIGS["tiny"])


def test_terms_bit_exact(tiny, oracle_lib):
    ol = oracle_lib
    ctx = make_ctx(tiny)
    rng = np.random.default_rng(3)
    n = 200000
    s = np.exp(rng.uniform(np.log(1e-3), np.log(1e4), n)).astype(np.float32)
    s[::1000] = 0
    s[::1001] = -1
    st = (s * rng.uniform(0.5, 2.5, n)).astype(np.float32)
    ob = rng.integers(0, 40, n).astype(np.int32)
    ob[::17] = rng.integers(0, 5000, ob[::17].size)
    ol.set_mode(ol.MODE_DET)
    p = np.zeros(1, ol.PARAM_DTYPE)
    for k in p.dtype.names:
        p[k] = np.float32(tiny.params[k])
    ex, exc, term, q = ol.eval_terms(s, st, ob, p)
    gex, gexc, gterm, gq = ctx.debug_eval_terms(s, st, ob)
    assert np.array_equal(ex.view(np.uint32), gex.view(np.uint32))
    assert np.array_equal(exc.view(np.uint32), gexc.view(np.uint32))
    assert np.array_equal(term.view(np.uint64), gterm.view(np.uint64))
    assert np.array_equal(q, gq)


def test_the_clamps_reach_against_the_reference_shaped_arithmetic(tiny, oracle_lib):
    """The contract clamps a term at +-2^20 before it is quantised (include/ig_detmath.h: ig_quantize; what keeps 50 M-term sums inside two
    int64 limbs); the reference adds the unclamped double (KA:251-270, 4486).  Where the two part is a TESTED statement:
    counts from 10 to 10^7 (packed lists hold counts below 2^24) against expectations from the trans level to P(s) = 200 --
    * HIP == oracle DET bit for bit on every term, clamped or not;
    * wherever the reference-shaped term (oracle LIBM mode: powf / expf / log10 of glibc, float factorial below 15, Stirling above) lies
      inside +-2^20 the contract's term agrees with it to 1e-6 of its parts: that covers every count up to 1.5e5 whatever the
      expectation, and counts of 10^6 .. 10^7 only where the expectation is of the count's order (P(s) at a few base pairs);
    * beyond (a count of 10^6 against an expectation of a few: a term of -5e6) the contract's term IS the clamp, -2^20 exactly, and the
      reference's is not: a level-(L-1) pixel that deep is outside what the two arithmetics can be compared on."""
    ol = oracle_lib
    ctx = make_ctx(tiny)
    counts = np.array([10, 100, 1000, 10_000, 75_000, 150_000, 300_000, 1_000_000, 3_000_000, 10_000_000], np.int64)
    s_grid = np.exp(np.linspace(np.log(1e-4), np.log(2.0e3), 96)).astype(np.float32)  # kb: P(s) from 10^7 contacts per pixel down to the trans level
    s = np.tile(s_grid, counts.size).astype(np.float32)
    ob = np.repeat(counts, s_grid.size).astype(np.int32)
    st = np.zeros_like(s)
    p = np.zeros(1, ol.PARAM_DTYPE)
    for k in p.dtype.names:
        p[k] = np.float32(tiny.params[k])
    ol.set_mode(ol.MODE_DET)
    ex, exc, term_det, q_det = ol.eval_terms(s, st, ob, p)
    gex, gexc, gterm, gq = ctx.debug_eval_terms(s, st, ob)
    assert np.array_equal(term_det.view(np.uint64), gterm.view(np.uint64)) and np.array_equal(q_det, gq)
    ol.set_mode(ol.MODE_LIBM)
    try:
        _, _, term_libm, _ = ol.eval_terms(s, st, ob, p)
    finally:
        ol.set_mode(ol.MODE_DET)
    clamp = 1048576.0
    val_det = q_det.astype(np.float64) / 4294967296.0  # what joins the sums
    inside = np.abs(term_libm) < clamp * (1 - 1e-6)
    phys = s >= 0.05  # P(s) <= 4.3e3 contacts per pixel: distances a sub-fragment pair can have (below: the count's order for 10^6 .. 10^7)
    assert inside[(ob <= 150_000) & phys].all()  # every count the golden / live LIBM comparisons reach, and twice beyond
    # (1e-6 of what the term is made of -- ob log10 P, P, log10 ob! -- : near its zero crossing a term is the small difference of those)
    scale = np.abs(ob * np.log10(np.maximum(ex.astype(np.float64), 1e-300))) + ex + ob * np.log10(np.maximum(ob, 2).astype(np.float64))
    assert np.all(np.abs(val_det[inside] - term_libm[inside]) <= 1e-6 * scale[inside] + 1e-9)
    beyond = ~inside
    assert beyond.any() and (ob[beyond & phys] >= 300_000).all()
    assert np.all(np.abs(val_det[beyond]) == clamp)  # the clamp, exactly
    assert np.all(np.abs(term_libm[beyond]) >= clamp * (1 - 1e-6))
    big = ob >= 1_000_000
    print("counts >= 1e6: %d of %d terms inside the clamp (expectation of the count's order), the rest at -2^20; largest reference-shaped term %.3g"
          % (int((inside & big).sum()), int(big.sum()), float(np.abs(term_libm).max())))


def test_full_likelihood_matches_oracle(tiny, oracle_lib):
    from oracle.sampler_oracle import OracleSampler

    ol = oracle_lib
    ctx = make_ctx(tiny)
    s = OracleSampler(**tiny.sampler_kwargs(), mode=ol.MODE_DET)
    s.set_param_simu(tiny.params)
    s.eval_likelihood_init()
    hi, lo = ol.last_limbs()
    nz, z, limbs = ctx.full_likelihood()
    assert nz == float(s.gpu_curr_likelihood_nz[0])
    assert (int(limbs[0]), int(limbs[1])) == (int(hi[0]), int(lo[0]))
    d, c, st, p, ln = ctx.debug_tables()
    assert np.array_equal(d.view(np.uint32), s.vect_dist.view(np.uint32))
    assert np.array_equal(p, s.vect_pos) and np.array_equal(ln, s.vect_len) and np.array_equal(st, s.vect_s_tot)


def test_full_likelihood_rare_operands(oracle_lib):
    """the from-scratch pass on the operands its straight-line term does not cover: counts >= 256 (beyond the LDS
    log-factorial table) and >= 1024 (Stirling branch), and parameters outside the one-log domain of the contract
    (slope 0: the generic composition for every contact).  Bit-exact limbs against the oracle."""
    import copy

    import scipy.sparse as sp

    from instagraal_amd import synth
    from oracle.sampler_oracle import OracleSampler

    ol = oracle_lib
    prob = copy.deepcopy(synth.make_problem(*synth.CONFIGS["tiny"]))
    cnt = prob.coo_cnt.copy()
    cnt[::7] *= 60    # 60 .. a few hundred
    cnt[::131] *= 400  # thousands
    assert (cnt >= 256).sum() > 50 and (cnt >= 1024).sum() > 5
    prob.coo_cnt = cnt
    M = prob.n_sub_frags
    prob.sub_csr = sp.csr_matrix((cnt, (prob.coo_row, prob.coo_col)), shape=(M, M), dtype=np.int32)
    prob.sub_csr.sort_indices()
    for params in (prob.params, dict(prob.params, slope=0.0)):
        ctx = make_ctx(prob, params=params)
        s = OracleSampler(**prob.sampler_kwargs(), mode=ol.MODE_DET)
        s.set_param_simu(params)
        s.eval_likelihood_init()
        hi, lo = ol.last_limbs()
        nz, z, limbs = ctx.full_likelihood()
        assert nz == float(s.gpu_curr_likelihood_nz[0])
        assert (int(limbs[0]), int(limbs[1])) == (int(hi[0]), int(lo[0]))
        ctx.close()


# the nuisance trajectory (tiny_nuis) is replayed through the sampler class in tests/test_hip_sampler.py
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "tiny_*_mode1.npz")) if "nuis" not in p)


@pytest.mark.parametrize("case", CASES)
def test_replay_golden(case):
    """Every move of the golden trajectory: 24xC scores bit-exact, same winner, same flags, same
    return tuple; genome state identical at every checkpoint (info_frags-level parity)."""
    from instagraal_amd import synth

    g = np.load(os.path.join(GOLDEN, case + ".npz"))
    assert int(g["nuis_from"]) < 0
    prob = synth.make_problem(*synth.CONFIGS[str(g["config"])])
    ctx = make_ctx(prob)
    nz, z, _ = ctx.full_likelihood()
    assert nz == float(g["init_nz"])
    if bool(g["bomb"]):
        ctx.bomb(np.arange(prob.n_frags, dtype=np.int32))
    for t, f in enumerate(g["frag"]):
        cands = [int(c) for c in g["cands"][t] if c >= 0]
        res, sc = ctx.step(int(f), cands)
        exp = g["scores"][t][: len(cands) * 24]
        assert np.array_equal(sc, exp), (t, np.nonzero(sc != exp)[0][:8], sc[sc != exp][:4], exp[sc != exp][:4])
        r = g["ret"][t]
        assert res.op_sampled == int(r[2]) and res.id_f_sampled == int(r[3]), t
        assert res.o == r[0] and res.dist == r[1], (t, res.o, r[0], res.dist, r[1])
        assert res.mean_len == r[4] and res.n_contigs == int(r[5]), (t, res.mean_len, r[4], res.n_contigs, r[5])
        assert np.array_equal(ctx.valid_insert(), g["valid"][t]), t
        if t in g["state_every"]:
            k = list(g["state_every"]).index(t)
            assert np.array_equal(ctx.download_state(), g["states"][k]), t
    # the maintained exact sums equal a from-scratch recomputation
    nz2, z2, limbs = ctx.full_likelihood()
    res, _ = ctx.step(int(g["frag"][0]), [int(c) for c in g["cands"][0] if c >= 0])


@pytest.mark.parametrize("cut", [False, True])
@pytest.mark.parametrize("case", CASES)
def test_replay_golden_through_the_batch_path(case, cut):
    """The same golden trajectories through ``ig_step_batch`` -- the speculative batches of 24 moves, two-tier scoring (float
    screening + exact contenders), in-order commit on the device: what ``bench.py`` and ``full_em`` run -- instead of one
    ``ig_step`` per move: return tuples, stale flags after the run, genome states.  Checkpoint states are compared by cutting
    the run into calls at the checkpoints (results do not depend on where a run is cut)."""
    from instagraal_amd import hip_lib, synth

    g = np.load(os.path.join(GOLDEN, case + ".npz"))
    prob = synth.make_problem(*synth.CONFIGS[str(g["config"])])
    ctx = make_ctx(prob)
    if bool(g["bomb"]):
        ctx.bomb(np.arange(prob.n_frags, dtype=np.int32))
    frags = np.asarray(g["frag"], np.int32)
    cands = np.asarray(g["cands"], np.int32)
    # cut: one call per checkpoint interval (every stored genome state is compared); else ONE call (full-width batches), the
    # last stored state being the final one
    cuts = sorted(set(int(t) + 1 for t in g["state_every"]) | {len(frags)}) if cut else [len(frags)]
    assert int(g["state_every"][-1]) == len(frags) - 1
    b0 = ctx.batch_stats()["batches"]
    start = 0
    for end in cuts:
        res = ctx.step_batch(frags[start:end], cands[start:end])
        for t in range(start, end):
            r, q = g["ret"][t], res[t - start]
            got = (float(q["o"]), float(q["dist"]), int(q["op_sampled"]), int(q["id_f_sampled"]), float(q["mean_len"]), int(q["n_contigs"]))
            assert got == (r[0], r[1], int(r[2]), int(r[3]), r[4], int(r[5])), (t, got, list(r))
        assert np.array_equal(ctx.valid_insert(), g["valid"][end - 1]), end
        if end - 1 in g["state_every"]:
            k = list(g["state_every"]).index(end - 1)
            assert np.array_equal(ctx.download_state(), g["states"][k]), end
        start = end
    assert ctx.batch_stats()["batches"] - b0 < len(frags)  # the moves went through batches (a few contigs: many conflicts, short batches)
    sums, _ = ctx.debug_globals()
    _, _, limbs = ctx.full_likelihood()
    assert [int(x) for x in sums[:5]] == [int(x) for x in limbs[:5]]
    ctx.close()
