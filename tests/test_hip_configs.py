"""GPU tests at the shapes BASELINE.json names, and the checks of the HIP arithmetic that do not come from the
deterministic contract:

* cfg2 (5 k bins / 2 M contacts): 40 moves against the oracle run live (scores, winners, state bit-exact), then 2 000
  moves of properties (maintained exact sums == from scratch, independence of the batch width);
* cfg5 (200 k bins / 500 M contacts, one GPU): the same properties on ~60 moves (the oracle would need minutes per move);
* the LIBM goldens (``tiny_*_mode0.npz``: the reference's own sampler Python over kernels that compose glibc's
  powf / expf / log10 exactly as the CUDA source composes CUDA's): every HIP score within the north star's 1e-6
  relative, same winners up to the first move where the two CPU arithmetic modes themselves part ways (a near-tie);
* P(s) on the GPU against the reference's ``optim_rippe_curve_update.peval`` (captured grid);
* ``ShardedRunner`` (contact rows split over ranks, one all-reduce of exact int64 partial sums per move) on real
  device buffers: two contexts on one GPU, an in-process all-reduce, against ``ig_step``.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

REL = 1e-6  # BASELINE.json north_star: "recomputed log-likelihood must match within 1e-6 relative"


def _fresh(prob, coo=True):
    from instagraal_amd.sampler import sampler as hip_sampler

    s = hip_sampler(**prob.sampler_kwargs(), device_id=0, coo=(prob.coo_row, prob.coo_col, prob.coo_cnt) if coo else None)
    s.set_param_simu(prob.params)
    s.eval_likelihood_init()
    return s


def _valid_linear_contigs(state, n_check=200):
    from instagraal_amd import hip_lib

    f = dict(zip(hip_lib.FRAG_FIELDS, state))
    for cid in np.unique(f["id_c"])[:n_check]:
        m = np.nonzero(f["id_c"] == cid)[0]
        order = m[np.argsort(f["pos"][m])]
        assert np.array_equal(f["pos"][order], np.arange(len(m)))
        assert np.all(f["l_cont"][m] == len(m))
        if f["circ"][m[0]] == 0:
            assert f["prev"][order[0]] == -1 and f["next"][order[-1]] == -1
        assert np.array_equal(f["next"][order[:-1]], order[1:]) and np.array_equal(f["prev"][order[1:]], order[:-1])
        assert np.array_equal(f["start_bp"][order], np.concatenate([[0], np.cumsum(f["len_bp"][order])[:-1]]))


def test_cfg2_live_oracle_then_properties():
    """BASELINE.json configs[1]: synthetic 5 k fragments / 2 M contacts."""
    from instagraal_amd import hip_lib, synth
    from instagraal_amd.sampler import PARAM_NAMES
    from oracle import oracle_lib as ol
    from oracle.sampler_oracle import OracleSampler

    ol.build()
    prob = synth.make_problem(*synth.CONFIGS["cfg2"])
    assert (prob.n_frags, prob.n_contacts) == (5_000, 2_000_000)
    s = _fresh(prob)
    o = OracleSampler(**prob.sampler_kwargs(), mode=ol.MODE_DET)
    o.set_param_simu(prob.params)
    o.eval_likelihood_init()
    assert float(s.curr_likelihood_on_nz[0]) == float(o.gpu_curr_likelihood_nz[0])
    np.random.seed(3)
    frags = np.random.permutation(prob.n_frags)
    for f in frags[:40]:
        cands = s.return_neighbours(int(f), 5)
        a = s.step_sampler(int(f), 5, candidates=cands)
        b = o.step_sampler(int(f), 5, o.dt, candidates=cands)
        assert s.candidates == o.candidates
        assert np.array_equal(s.all_scores, o.all_scores)
        assert (a[0], a[1], a[2], a[3], float(a[4]), int(a[5])) == (b[0], b[1], b[2], b[3], float(b[4]), int(b[5]))
    assert np.array_equal(s.gpu_vect_frags.copy_from_gpu().soa17(), o.gpu_vect_frags.soa17())
    # the next 40 moves through the BATCH path (speculative batches, two-tier scoring, the draw inside the call) against the
    # oracle drawing with numpy and stepping one move at a time from the same generator state
    more = frags[40:80].astype(np.int32)
    st = np.random.get_state()
    res = s.step_sampler_batch(more, 5)
    after = np.random.get_state()
    np.random.set_state(st)
    for f, r in zip(more, res):
        b = o.step_sampler(int(f), 5, o.dt)
        assert (float(r["o"]), float(r["dist"]), int(r["op_sampled"]), int(r["id_f_sampled"]), float(np.float32(r["mean_len"])), int(r["n_contigs"])) == (
            b[0], b[1], b[2], b[3], float(b[4]), int(b[5]))
    assert np.array_equal(np.random.get_state()[1], after[1]) and np.random.get_state()[2] == after[2]
    assert np.array_equal(s.gpu_vect_frags.copy_from_gpu().soa17(), o.gpu_vect_frags.soa17())
    s.free_gpu()
    del o
    # 2 000 moves: maintained sums, batch-width independence, structural validity
    np.random.seed(4)
    frags = np.resize(np.random.permutation(prob.n_frags), 2000).astype(np.int32)
    outs = []
    try:
        for W in (24, 1):
            hip_lib.set_batch_width(W)
            s = _fresh(prob)
            np.random.seed(5)
            cands = s.draw_candidates(frags, 5)
            res = s.ctx.step_batch(frags, cands)
            sums, _ = s.ctx.debug_globals()
            _, _, limbs = s.ctx.full_likelihood(0)
            assert [int(x) for x in sums[:5]] == [int(x) for x in limbs[:5]], W
            if W == 24:
                # the from-scratch pass sums the tiles of trans pairs only from their count histograms: same limbs contact
                # by contact, under the model's parameters and under another set
                p1 = dict(prob.params, slope=np.float32(-1.1), fact=np.float32(prob.params["fact"] * 1.7))
                s.ctx.set_params([np.float32(p1[k]) for k in PARAM_NAMES], s.mean_kb(), 1)
                _, _, limbs1 = s.ctx.full_likelihood(1)
                hip_lib.debug_set_full_hist(0)
                try:
                    assert [int(x) for x in s.ctx.full_likelihood(0)[2]] == [int(x) for x in limbs]
                    assert [int(x) for x in s.ctx.full_likelihood(1)[2]] == [int(x) for x in limbs1]
                finally:
                    hip_lib.debug_set_full_hist(1)
            outs.append((res.tobytes(), s.gpu_vect_frags.copy_from_gpu().soa17(), [int(x) for x in s.ctx.valid_insert()],
                         s.ctx.batch_stats()))
            s.free_gpu()
    finally:
        hip_lib.set_batch_width(24)
    assert outs[0][0] == outs[1][0]
    assert np.array_equal(outs[0][1], outs[1][1]) and outs[0][2] == outs[1][2]
    assert outs[0][3]["batches"] < 400  # speculation happened: far fewer launches than moves
    _valid_linear_contigs(outs[0][1])


@pytest.mark.slow
def test_cfg5_properties():
    """BASELINE.json configs[4] on one GPU: 200 k fragments / 500 M contacts (human scale).  The oracle needs minutes per
    move here; what is checked are the size-independent properties: the incrementally maintained exact likelihood limbs
    equal a from-scratch pass over all 500 M contacts, the trajectory does not depend on the batch width, the genome
    stays a valid set of contigs."""
    from instagraal_amd import hip_lib, synth

    prob = synth.make_problem(*synth.CONFIGS["cfg5"])
    assert (prob.n_frags, prob.n_contacts) == (200_000, 500_000_000)
    np.random.seed(0)
    frags = np.resize(np.random.permutation(prob.n_frags), 60).astype(np.int32)
    outs = []
    try:
        for W in (24, 1):
            hip_lib.set_batch_width(W)
            s = _fresh(prob)
            np.random.seed(1)
            cands = s.draw_candidates(frags, 5)
            res = s.ctx.step_batch(frags, cands)
            sums, _ = s.ctx.debug_globals()
            _, _, limbs = s.ctx.full_likelihood(0)
            assert [int(x) for x in sums[:5]] == [int(x) for x in limbs[:5]], W
            outs.append((res.tobytes(), s.gpu_vect_frags.copy_from_gpu().soa17(), [int(x) for x in s.ctx.valid_insert()]))
            assert np.all(res["error"] == 0) and np.all(np.isfinite(res["o"]))
            if W == 24:  # scratch of the batches: window-sized arrays + a pool of 2 Z entries, not the worst case of a genome-wide window
                assert sum(s.ctx.scratch_bytes()) < 20e9, s.ctx.scratch_bytes()
            s.free_gpu()
            del s
    finally:
        hip_lib.set_batch_width(24)
    assert outs[0][0] == outs[1][0]
    assert np.array_equal(outs[0][1], outs[1][1]) and outs[0][2] == outs[1][2]
    _valid_linear_contigs(outs[0][1], 50)


@pytest.mark.parametrize("case", ["tiny_plain", "tiny_bomb"])
def test_libm_golden_within_1e6(case):
    """The reference-shaped arithmetic (oracle LIBM mode: float powf / expf, double log10, block-tree sums) is the
    independent statement of the numbers; the HIP path computes the deterministic contract.  On the common part of the
    two golden trajectories every one of the 24 x C HIP scores must agree with the LIBM golden to 1e-6 relative and pick
    the same winner; at the first move where the two CPU modes choose differently (a near-tie, see
    tests/test_oracle_golden.py::test_det_and_libm_modes_agree) the scores must still agree."""
    from instagraal_amd import synth
    from instagraal_amd.sampler import problem_to_context

    a = np.load(os.path.join(GOLDEN, case + "_mode0.npz"))  # LIBM
    b = np.load(os.path.join(GOLDEN, case + "_mode1.npz"))  # DET
    same = np.all(a["ret"][:, 2:4] == b["ret"][:, 2:4], axis=1)
    first = len(same) if same.all() else int(np.argmin(same))
    assert first >= 20, first
    assert np.array_equal(a["cands"][: min(first + 1, len(same))], b["cands"][: min(first + 1, len(same))])
    prob = synth.make_problem(*synth.CONFIGS[str(a["config"])])
    ctx = problem_to_context(prob)
    nz, _, _ = ctx.full_likelihood()
    assert abs(nz - float(a["init_nz"])) <= REL * abs(float(a["init_nz"]))
    if bool(a["bomb"]):
        ctx.bomb(np.arange(prob.n_frags, dtype=np.int32))
    worst = 0.0
    for t in range(min(first + 1, len(same))):
        cands = [int(c) for c in a["cands"][t] if c >= 0]
        res, sc = ctx.step(int(a["frag"][t]), cands)
        exp = a["scores"][t][: len(cands) * 24]
        scored = exp != 0
        assert np.array_equal(scored, sc != 0), t
        rel = np.abs(sc[scored] - exp[scored]) / np.abs(exp[scored])
        worst = max(worst, float(rel.max()))
        assert rel.max() <= REL, (t, rel.max())
        r = a["ret"][t]
        assert abs(res.o - r[0]) <= REL * abs(r[0]) or t == first, t
        if t < first:
            assert (res.op_sampled, res.id_f_sampled, res.n_contigs) == (int(r[2]), int(r[3]), int(r[5])), t
            assert res.dist == r[1], t
            assert np.array_equal(ctx.valid_insert(), a["valid"][t]), t
            if t in a["state_every"]:
                assert np.array_equal(ctx.download_state(), a["states"][list(a["state_every"]).index(t)]), t
    assert worst < REL
    ctx.close()


def test_nuisance_libm_golden_within_1e6():
    """the nuisance trajectory of the LIBM golden: scores, the full likelihood under the test parameters and the accepted /
    rejected flags, up to the first move where the LIBM and DET chains part ways"""
    from instagraal_amd import synth
    from instagraal_amd.sampler import sampler as hip_sampler

    a = np.load(os.path.join(GOLDEN, "tiny_nuis_mode0.npz"))
    b = np.load(os.path.join(GOLDEN, "tiny_nuis_mode1.npz"))
    same = np.all(a["ret"][:, 2:4] == b["ret"][:, 2:4], axis=1)
    first = len(same) if same.all() else int(np.argmin(same))
    nuis_from = int(a["nuis_from"])
    n_nuis_same = 0
    for k in range(len(a["nuis"])):
        if nuis_from + k < first and a["nuis"][k][6] == b["nuis"][k][6]:
            n_nuis_same += 1
        else:
            break
    assert first > nuis_from and n_nuis_same >= 5, (first, n_nuis_same)
    prob = synth.make_problem(*synth.CONFIGS[str(a["config"])])
    np.random.seed(int(a["seed"]))
    s = hip_sampler(**prob.sampler_kwargs(), device_id=0)
    s.set_param_simu(prob.params)
    s.bins = np.arange(1.0, 60.0, 1.0)
    s.eval_likelihood_init()
    frags = np.arange(0, s.n_new_frags)
    np.random.shuffle(frags)
    for t in range(nuis_from + n_nuis_same):
        r = s.step_sampler(int(a["frag"][t]), 5, s.dt)
        n = len(s.candidates) * 24
        exp = a["scores"][t][:n]
        scored = exp != 0
        assert np.array_equal(scored, s.all_scores != 0), t
        assert (np.abs(s.all_scores[scored] - exp[scored]) / np.abs(exp[scored])).max() <= REL, t
        assert (int(r[2]), int(r[3]), int(r[5])) == (int(a["ret"][t][2]), int(a["ret"][t][3]), int(a["ret"][t][5])), t
        if t >= nuis_from:
            q = s.step_nuisance_parameters(s.dt, t, len(a["frag"]))
            exp_q = a["nuis"][t - nuis_from]
            got = [float(q[0]), float(q[1]), float(q[2]), float(q[3]), float(q[4]), float(np.ravel(q[5])[0]), float(q[6])]
            assert got[6] == exp_q[6], (t, got, exp_q)  # accepted / rejected
            assert np.allclose(got[:6], exp_q[:6], rtol=REL, atol=0), (t, got, exp_q)


def test_rippe_on_gpu_against_reference_peval():
    """P(s) as the HIP kernels evaluate it (ig_rippe on gfx950) against the reference's optim_rippe_curve_update.peval
    on a grid captured from the reference (tools/gen_golden.py): no oracle in between."""
    from instagraal_amd import synth
    from instagraal_amd.sampler import problem_to_context

    g = np.load(os.path.join(GOLDEN, "host_helpers.npz"))
    s, ref = g["rippe_grid_s"], g["rippe_grid_peval"]
    names = ("kuhn", "lm", "c1", "slope", "d", "d_max", "fact", "v_inter")
    params = dict(zip(names, [float(v) for v in g["rippe_grid_params"]]))
    prob = synth.make_problem(*synth.CONFIGS["tiny"])
    ctx = problem_to_context(prob, params=params)
    ex, _, term, _ = ctx.debug_eval_terms(s, np.zeros_like(s), np.ones(s.size, np.int32))
    want = np.maximum(ref, params["v_inter"])
    assert (np.abs(ex.astype(np.float64) - want) / want).max() < REL
    # and the term built on it (ob = 1, P_z := rippe_circ(s, 0) which is d_max by quirk Q6): ob log10 P - P - log10(1!) + P_z log10 e
    pz = float(np.float32(params["d_max"])) * float(np.float32(0.43429448190325182))
    lit = np.log10(want) - want + pz
    assert (np.abs(term - lit) / (np.abs(np.log10(want)) + want + pz)).max() < REL  # relative to the size of the term's parts
    ctx.close()


def test_contact_shards_over_two_ranks_equal_one_gpu():
    """multi_gpu.ShardedRunner with world = 2 on real device buffers (two contexts on one GPU, two threads, an in-process
    all-reduce of the zero-copy int64 views): each rank slices and scores the candidate rows r with r % 2 == rank, the exact
    partial sums (and the list lengths) are summed, both ranks finish the move -- scores are not exchanged, so equal
    results mean the reduced sums were complete.  Against ig_step on one context, move by move; then the entry points
    that would silently use partial sums must refuse."""
    import threading

    import torch

    from instagraal_amd import hip_lib, synth
    from instagraal_amd.multi_gpu import ShardedRunner

    prob = synth.make_problem(*synth.CONFIGS["small"])
    np.random.seed(17)
    frags = np.random.permutation(prob.n_frags)[:60].astype(np.int32)
    ref = _fresh(prob, coo=False)
    cands = ref.draw_candidates(frags, 5)
    want = np.zeros(frags.size, hip_lib.MOVE_RESULT_DTYPE)
    for i, f in enumerate(frags):
        r, _ = ref.ctx.step(int(f), cands[i][cands[i] >= 0])
        for k in want.dtype.names:
            want[k][i] = getattr(r, k)
    want_state = ref.gpu_vect_frags.copy_from_gpu().soa17()

    world = 2
    barrier = threading.Barrier(world)
    slots = {}

    class InProcessDist:
        class ReduceOp:
            SUM = "sum"

        def __init__(self, rank):
            self.rank = rank

        def all_reduce(self, t, op=None):
            torch.cuda.synchronize()
            slots[self.rank] = t
            barrier.wait()
            total = slots[0] + slots[1]
            torch.cuda.synchronize()
            barrier.wait()
            t.copy_(total)
            torch.cuda.synchronize()
            barrier.wait()

    samplers = [_fresh(prob, coo=False) for _ in range(world)]
    got, errs = [None] * world, []

    def work(r):
        try:
            torch.cuda.set_device(0)
            runner = ShardedRunner(samplers[r].ctx, r, world, dist=InProcessDist(r))
            got[r] = runner.run(frags, cands)
            ptr, n = samplers[r].ctx.partials()
            assert runner._t.data_ptr() == ptr and runner._t.numel() == n  # a view, not a copy
        except Exception as e:  # pragma: no cover
            errs.append(e)
            barrier.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    cols = ("o", "dist", "mean_len", "op_sampled", "id_f_sampled", "n_contigs", "n_candidates", "n_slice")
    for r in range(world):
        for k in cols:
            assert np.array_equal(got[r][k], want[k]), (r, k)
        assert np.array_equal(samplers[r].gpu_vect_frags.copy_from_gpu().soa17(), want_state), r
    # a sharded handle must not run the one-GPU entry points on partial sums
    ctx = samplers[0].ctx
    for call in (lambda: ctx.step(int(frags[0]), [int(c) for c in cands[0] if c >= 0]),
                 lambda: ctx.score_move(int(frags[0]), [int(c) for c in cands[0] if c >= 0]),
                 lambda: ctx.step_batch(frags[:4], cands[:4]),
                 lambda: ctx.apply(int(frags[0]), int(cands[0][0]), 1)):
        with pytest.raises(hip_lib.HipError):
            call()
