"""The HIP path against the ORACLE where rounds 1 - 4 never compared them: on EVOLVED genomes and on random shapes.

Until round 5 every oracle-comparing test ran 12 - 160 moves from the initial state on short contigs, and the 12 000 fuzz cases were
HIP-against-HIP: a three-round-old wrong-result bug (the one-move path ignoring the slice pool's overflow flag) passed all of them.

* ``test_fuzz_cases_against_the_oracle`` -- tools/fuzz_oracle.py's generator (60 - 2 000 bins, contigs of 2 - 1 000 bins, 1 - 16
  neighbours, counts x 9 / x 60, --bomb'ed starts, small slice pools, synthetic / settled / random parameters; batches of 24, other
  widths, one step_sampler call per move): 40 seeded cases, 6-tuples, 17 x N state, stale flags, generator state.
* ``test_nuisance_run_whose_d_nuc_reaches_zero`` -- the two cases of that generator that found the round's one difference on the
  nuisance path (the generator's consumption after d_nuc rounds to 0 inside a call).
* ``test_long_trajectory_live_oracle`` -- whole cycles of the reference's loop (IG:196-262) through ``step_sampler_batch`` against
  ``OracleSampler(DET).step_sampler`` (CL:1401-1465; KA:485-607, 612-3693), state compared every 250 moves: `small` for 3 full cycles
  from the assembled and from the --bomb'ed genome, `bigctg` (windows of 3 000 - 9 000 sub-fragments) and `bigctg --bomb`.
* ``test_long_nuisance_trajectory_live_oracle`` -- the same with a nuisance step behind every move from the first cycle on
  (``step_sampler_nuisance_batch`` against ``o.step_sampler`` + ``o.step_nuisance_parameters``, CL:2961-3051), chains
  asserted to be active and steps accepted behind them.
* ``test_cfg3_oracle_moves_behind_2000_hip_moves`` -- the headline shape: 12 oracle moves BEHIND 2 000 moves of the batch path (the
  evolved genome handed to the oracle instead of being replayed there: an oracle move takes a second at this size).
"""
import importlib.util
import os

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _tool(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.slow
def test_fuzz_cases_against_the_oracle():
    fo = _tool("fuzz_oracle")
    bad, n_moves, kinds = [], 0, set()
    for seed in range(9001, 9033):
        try:
            desc, diff, t_hip, t_or = fo.live_case(seed)
        except ValueError:  # (a shape the generator has no pairs for)
            continue
        except Exception as ex:
            if "needs 1.." in str(ex):  # a move without a candidate: undefined in the reference (quirk Q13), refused here
                continue
            raise
        n_moves += desc["n"]
        kinds.add((desc["how"], bool(desc["pool"]), bool(desc["bomb"])))
        if diff:
            bad.append((desc, diff))
    print("fuzz against the oracle: %d moves, %d kinds of runs" % (n_moves, len(kinds)))
    assert not bad, bad
    assert n_moves >= 3000 and len(kinds) >= 6, (n_moves, kinds)


@pytest.mark.parametrize("seed", [5072, 5123])
def test_nuisance_run_whose_d_nuc_reaches_zero(seed):
    """tools/fuzz_oracle.py's first finding on the nuisance path (round 5): 60 bins under parameters that run away (d_max 1e30, d_nuc a
    float32 denormal) -- an accepted step rounds d_nuc to 0, sigma_d_nuc with it, and from the next step on the reference draws no normal
    for id_modif == 3 (CL:2973, 3007-3010): the one case where the stream's consumption depends on the parameters.  Until round 5 the
    host looked at sigma_d_nuc at the start of a call only and went on with the stream it had drawn up front: other candidates from the
    first such step on.  Now a run ends behind the accepted step, rewinds the generator and takes the rest one step at a time."""
    fo = _tool("fuzz_oracle")
    prob, params, desc = fo.make_case(seed)
    assert desc["how"] == "nuis"
    h = fo.run_hip(prob, params, desc)
    d_nuc = h["nuis"][:, 3]
    first = int(np.nonzero(d_nuc == 0.0)[0][0])
    assert 0 < first < desc["n"] - 20 and first % fo.CHECK_EVERY not in (0, fo.CHECK_EVERY - 1), first  # inside a call, moves behind it
    _, diff = fo.run_oracle(prob, params, desc, 0, expect=h)
    assert diff is None, diff


# (round 6: 2 cycles of `small` instead of 5, bigctg 1 000 / 1 500 instead of 2 000 / 4 000 -- the full lengths and more were run and recorded with
# tools/long_oracle.py and checked on host cores: profiles/r05a_long_oracle.txt, profiles/r06_validation.txt; the GPU suite's budget)
@pytest.mark.slow
@pytest.mark.parametrize("cfg,moves,bomb", [("small", 2000, False), ("small", 2000, True), ("bigctg", 1000, False), ("bigctg", 1500, True)])
def test_long_trajectory_live_oracle(cfg, moves, bomb):
    lo = _tool("long_oracle")
    h = lo.run_hip(cfg, moves, bomb=bomb, seed=41)
    print(h["summary"])
    diff = lo.run_oracle(cfg, moves, bomb=bomb, seed=41, expect=h)
    assert diff is None, diff
    sm = h["summary"]
    if cfg == "bigctg" and not bomb:
        assert sm["longest_contig_subfrags"] > 4096, sm  # (windows past the 32 KB LDS stage and the 4 096-sub-fragment fused commit)
    if bomb:
        assert sm["n_contigs_end"] < 0.75 * h["records"][0][5], sm  # the genome was being re-assembled on the way


@pytest.mark.slow
def test_long_nuisance_trajectory_live_oracle():
    lo = _tool("long_oracle")
    moves = 2000
    h = lo.run_hip("small", moves, bomb=False, nuis=True, seed=43, hist=2)
    sm = h["summary"]
    print(sm)
    assert sm["chain_pairs"] > 120, sm  # chains were active (pairs decided on the device: 400 of 3 000 here -- every third step is accepted on this problem)...
    acc = np.nonzero(h["nuis"][:, 6])[0]
    assert len(acc) >= 5 and acc[-1] > moves // 2, sm  # ... and steps were accepted behind them, late in the run
    diff = lo.run_oracle("small", moves, bomb=False, nuis=True, seed=43, expect=h)
    assert diff is None, diff


def test_cfg3_oracle_moves_behind_2000_hip_moves():
    lo = _tool("long_oracle")
    diff, sm = lo.behind_hip_moves("cfg3", 2000, 12, seed=11)
    print(sm)
    assert diff is None, diff
    assert sm["bins_moved_by_hip"] > 0
