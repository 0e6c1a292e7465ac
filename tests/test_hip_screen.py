"""GPU tests of the two-tier scoring of the speculative batches (csrc/ig_kernels_screen.cuh): a float screening pass with a
rigorous error bound per column decides which columns can still win; only those go through the exact kernel.

* the two hardware functions the bound leans on (v_log_f32, v_exp_f32) over their WHOLE domain against the contract's
  double functions: the bound assumes 4 units, the hardware must stay below 2;
* IG_SCREEN_VERIFY=1 scores every column exactly as well and checks |screened - exact| <= bound for every column of every
  move (device-side error 7 otherwise), while the decisions are taken from the contenders only, as in production;
* the trajectories with and without screening are byte-identical (result records, genome, exact sums, stale flags)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(prob, frags, seed, env, monkeypatch, coo=False, n_neighbours=5, second_call=0):
    from instagraal_amd.sampler import sampler as hip_sampler

    for k in ("IG_SCREEN", "IG_SCREEN_VERIFY"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    s = hip_sampler(**prob.sampler_kwargs(), device_id=0, coo=(prob.coo_row, prob.coo_col, prob.coo_cnt) if coo else None)
    s.set_param_simu(prob.params)
    s.eval_likelihood_init()
    np.random.seed(seed)
    cands = s.draw_candidates(frags, n_neighbours)
    if second_call:  # a long call, then a short one: the exact kernel's grid is sized from the batches before (down to its floor)
        res = np.concatenate([s.ctx.step_batch(frags[:-second_call], cands[:-second_call]),
                              s.ctx.step_batch(frags[-second_call:], cands[-second_call:])])
    else:
        res = s.ctx.step_batch(frags, cands)
    sums, _ = s.ctx.debug_globals()
    _, _, limbs = s.ctx.full_likelihood(0)
    assert [int(x) for x in sums[:5]] == [int(x) for x in limbs[:5]]
    out = (res.tobytes(), s.gpu_vect_frags.copy_from_gpu().soa17().tobytes(), [int(x) for x in sums], [int(x) for x in s.ctx.valid_insert()])
    stats = s.ctx.debug_screen_stats()
    s.free_gpu()
    return out, stats, res


def test_hardware_log2_exp2_over_their_whole_domain():
    from instagraal_amd import synth
    from instagraal_amd.sampler import problem_to_context

    ctx = problem_to_context(synth.make_problem(*synth.CONFIGS["tiny"]))
    e_log, e_exp = ctx.debug_transcendental_error()
    print("v_log_f32: %.3f units of 2^-23 (|L| + 1); v_exp_f32: %.3f units of 2^-23 2^y" % (e_log, e_exp))
    assert 0 < e_log <= 2.0 and 0 < e_exp <= 2.0  # the bound is derived for 4
    ctx.close()


@pytest.mark.parametrize("cfg,n_moves", [("tiny", 300), ("small", 600), ("bigctg", 120), ("cfg2", 1500)])
def test_screening_bound_holds_and_changes_nothing(cfg, n_moves, monkeypatch):
    from instagraal_amd import synth

    prob = synth.make_problem(*synth.CONFIGS[cfg])
    np.random.seed(33)
    frags = np.resize(np.random.permutation(prob.n_frags), n_moves).astype(np.int32)
    exact, _, _ = _run(prob, frags, 7, {"IG_SCREEN": "0"}, monkeypatch)
    verified, stats, _ = _run(prob, frags, 7, {"IG_SCREEN_VERIFY": "1"}, monkeypatch)  # raises on a violated bound (error 7)
    screened, _, _ = _run(prob, frags, 7, {}, monkeypatch)
    assert verified == exact
    assert screened == exact
    print(cfg, "largest used fraction of a bound %.3g, largest bound %.3g, columns screened %d, scored exactly %d, terms %d / %d" % stats)
    assert 0 < stats[0] < 0.5  # the bound is rigorous, hence loose: the observed error uses a small part of it


def test_screening_with_large_counts_and_other_parameters(monkeypatch):
    """counts in the thousands (sums of ob and P in the millions per column: large bounds), and a parameter set outside the
    one-log domain (slope 0: every column's bound is void, everything goes through the exact kernel)"""
    import copy

    import scipy.sparse as sp

    from instagraal_amd import synth

    prob = copy.deepcopy(synth.make_problem(*synth.CONFIGS["small"]))
    cnt = prob.coo_cnt.copy()
    cnt[::5] *= 70
    cnt[::53] *= 500
    prob.coo_cnt = cnt
    M = prob.n_sub_frags
    prob.sub_csr = sp.csr_matrix((cnt, (prob.coo_row, prob.coo_col)), shape=(M, M), dtype=np.int32)
    prob.sub_csr.sort_indices()
    np.random.seed(5)
    frags = np.resize(np.random.permutation(prob.n_frags), 300).astype(np.int32)
    for params in (prob.params, dict(prob.params, slope=0.0), dict(prob.params, slope=-0.7, v_inter=2e-4)):
        prob.params = params
        exact, _, _ = _run(prob, frags, 9, {"IG_SCREEN": "0"}, monkeypatch)
        verified, stats, _ = _run(prob, frags, 9, {"IG_SCREEN_VERIFY": "1"}, monkeypatch)
        screened, _, _ = _run(prob, frags, 9, {}, monkeypatch)
        assert verified == exact and screened == exact
        assert stats[0] < 0.5


@pytest.mark.parametrize("cfg,n_moves", [("small", 400), ("bigctg", 160)])
def test_screening_with_a_table_longer_than_its_staged_copy(cfg, n_moves, monkeypatch):
    """a far d_max (what a nuisance chain drifts to): the P_z table is longer than the 1 024 entries the screening kernel stages.
    Staged windows hold no rank distance beyond the copy (the two-columns-per-pass loop serves them as before); unstaged ones
    (bigctg: windows of thousands of sub-fragments) read their far pairs' entries from the table itself -- neither voids a
    column's bound any more: the exact kernel keeps to the contenders"""
    from instagraal_amd import synth

    prob = synth.make_problem(*synth.CONFIGS[cfg])
    prob.params = dict(prob.params, slope=-0.6, d_max=2.5e6)
    np.random.seed(12)
    frags = np.resize(np.random.permutation(prob.n_frags), n_moves).astype(np.int32)
    exact, _, _ = _run(prob, frags, 4, {"IG_SCREEN": "0"}, monkeypatch)
    verified, stats, _ = _run(prob, frags, 4, {"IG_SCREEN_VERIFY": "1"}, monkeypatch)
    screened, stats2, _ = _run(prob, frags, 4, {}, monkeypatch)
    assert verified == exact and screened == exact
    print(cfg, "long table: used fraction %.3g, largest bound %.3g, columns screened %d, scored exactly %d, terms %d / %d" % stats2)
    assert 0 < stats[0] < 0.5
    assert stats2[3] < 0.3 * stats2[2]  # (void bounds would send every column through the exact kernel)


def test_screening_at_the_headline_shape(monkeypatch):
    from instagraal_amd import synth

    prob = synth.make_problem(*synth.CONFIGS["cfg3"])
    np.random.seed(2)
    frags = np.resize(np.random.permutation(prob.n_frags), 400).astype(np.int32)
    exact, _, _ = _run(prob, frags, 3, {"IG_SCREEN": "0"}, monkeypatch, coo=True)
    verified, stats, _ = _run(prob, frags, 3, {"IG_SCREEN_VERIFY": "1"}, monkeypatch, coo=True)
    screened, _, _ = _run(prob, frags, 3, {}, monkeypatch, coo=True)
    assert verified == exact and screened == exact
    print("cfg3: largest used fraction of a bound %.3g, largest bound %.3g, columns screened %d, scored exactly %d, terms %d / %d" % stats)
    assert stats[3] < 0.25 * stats[2]
    assert 0 < stats[0] < 0.5


def test_screening_at_the_settled_parameters(monkeypatch):
    """the headline shape under the parameters a nuisance chain settles into (synth.settled_params: slope -0.53, d_max 2.9e6 kb --
    where a default run spends 95 of its 100 cycles): the P_z table longer than its staged copy, and a ring pair's term at the
    contract's clamp (-2^20: 3e6 contacts expected per pair) -- the ring columns are ruled out by their clamped upper bound instead
    of going through the exact kernel's general loop (they were half of the exact tier's columns and nine tenths of its time)"""
    from instagraal_amd import synth

    prob = synth.make_problem(*synth.CONFIGS["cfg3"])
    prob.params = synth.settled_params(prob.params)
    np.random.seed(2)
    frags = np.resize(np.random.permutation(prob.n_frags), 400).astype(np.int32)
    exact, _, _ = _run(prob, frags, 3, {"IG_SCREEN": "0"}, monkeypatch, coo=True)
    verified, stats, _ = _run(prob, frags, 3, {"IG_SCREEN_VERIFY": "1"}, monkeypatch, coo=True)
    screened, stats2, _ = _run(prob, frags, 3, {}, monkeypatch, coo=True)
    assert verified == exact and screened == exact
    print("cfg3, settled parameters: largest used fraction of a bound %.3g, largest bound %.3g, columns screened %d, scored exactly %d, terms %d / %d" % stats2)
    assert stats2[3] < 0.08 * stats2[2]  # the ties and the current genome's columns; 10 % with the rings
    assert 0 < stats[0] < 0.5


@pytest.mark.parametrize("n_neighbours", [1, 6, 9, 16])
@pytest.mark.parametrize("cfg,n_moves", [("tiny", 160), ("small", 330)])
def test_other_candidate_counts(cfg, n_moves, n_neighbours, monkeypatch):
    """--neighborhood other than 5 (up to IG_MAX_CANDIDATES = 16): the decide wave's extra-record loads (more than 128 score
    records per move), k_contend / k_worklist sizing (a move alone may need 16 x 25 x 16 partly filled work items: the exact
    kernel's grid has a floor that follows the candidate count) and the capC = max(8, max_c) buffers.  Screened == verified ==
    exact == one move at a time, byte for byte; the run ends in a short second call (narrow batch, grid at its floor)."""
    from instagraal_amd import hip_lib, synth

    prob = synth.make_problem(*synth.CONFIGS[cfg])
    np.random.seed(41)
    frags = np.resize(np.random.permutation(prob.n_frags), n_moves).astype(np.int32)
    kw = dict(n_neighbours=n_neighbours, second_call=7)
    exact, _, res = _run(prob, frags, 7, {"IG_SCREEN": "0"}, monkeypatch, **kw)
    assert res["n_candidates"].max() <= n_neighbours and (n_neighbours <= 5 or res["n_candidates"].max() > 5)  # the wide path is taken
    verified, stats, _ = _run(prob, frags, 7, {"IG_SCREEN_VERIFY": "1"}, monkeypatch, **kw)
    screened, _, _ = _run(prob, frags, 7, {}, monkeypatch, **kw)
    assert verified == exact and screened == exact
    try:
        hip_lib.set_batch_width(1)
        single, _, _ = _run(prob, frags, 7, {}, monkeypatch, **kw)
    finally:
        hip_lib.set_batch_width(24)
    assert single == exact
    print(cfg, n_neighbours, "candidates per move: max %d, mean %.2f" % (res["n_candidates"].max(), res["n_candidates"].mean()))
