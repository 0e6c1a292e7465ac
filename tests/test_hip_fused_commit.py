"""GPU test of k_decide_commit (csrc/ig_kernels_commit.cuh): the decide step and the apply step of a batch in one launch, seven waves
applying the moves behind the decide wave, and of k_decide_commit_par (round 6: the decisions in rounds of DPAR moves decided side by side
by DPAR waves, decide_rounds), which the library picks for plain chains on small windows by default (IG_FUSED_COMMIT=1); IG_FUSED_COMMIT=0 / 2
(read once per process) force the two kernels / the one-wave fused one everywhere -- also for a run's one-move launches (the record the host waits for) and
for windows of thousands of sub-fragments.  Both must leave the same words: move records (score, genome distance, statistics columns),
genome, tables behind the last move (the nuisance steps read them), generator state, maintained sums."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

_SCRIPT = r"""
import hashlib, json, sys
import numpy as np
from instagraal_amd import synth
from instagraal_amd.sampler import sampler as hip_sampler

cfg, n_plain, n_nuis = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
prob = synth.make_problem(*synth.CONFIGS[cfg])
s = hip_sampler(**prob.sampler_kwargs(), device_id=0)
s.set_param_simu(prob.params)
s.bins = np.arange(1.0, 60.0, 1.0)
s.eval_likelihood_init()
np.random.seed(21)
h = hashlib.sha256()
fr = np.resize(np.random.permutation(prob.n_frags), n_plain).astype(np.int32)
changed = 0
for lo in range(0, n_plain, 250):  # several calls: batches resumed behind pending moves, short last batches
    res = s.step_sampler_batch(fr[lo:lo + 250], 5)
    h.update(res.tobytes())
    changed += int((np.diff(res["dist"]) != 0).sum() + (np.diff(res["n_contigs"]) != 0).sum())
fr = np.resize(np.random.permutation(prob.n_frags), n_nuis)
res, tup = s.step_sampler_nuisance_batch(fr, 5, s.dt, 0, n_nuis)
h.update(res.tobytes())
h.update(repr([tuple(float(np.ravel(x)[0]) for x in q[:7]) for q in tup]).encode())
h.update(s.gpu_vect_frags.copy_from_gpu().soa17().tobytes())
h.update(np.random.get_state()[1].tobytes())
sums, _ = s.ctx.debug_globals()
_, _, limbs = s.ctx.full_likelihood(0)
assert [int(x) for x in sums[:5]] == [int(x) for x in limbs[:5]]
h.update(repr([int(x) for x in sums[:5]]).encode())
print(json.dumps({"digest": h.hexdigest(), "batches": s.ctx.batch_stats(), "changed": changed}))
"""


def _run(cfg, n_plain, n_nuis, fused):
    env = dict(os.environ, IG_FUSED_COMMIT=str(fused), PYTHONPATH=os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
    r = subprocess.run([sys.executable, "-c", _SCRIPT, cfg, str(n_plain), str(n_nuis)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.parametrize("cfg,n_plain,n_nuis", [("small", 1500, 300), ("cfg2", 2000, 300), ("bigctg", 600, 150)])
def test_fused_decide_commit_leaves_the_same_words(cfg, n_plain, n_nuis):
    two = _run(cfg, n_plain, n_nuis, 0)
    one = _run(cfg, n_plain, n_nuis, 2)
    par = _run(cfg, n_plain, n_nuis, 1)  # the default: decide_rounds (round 6: DPAR moves decided side by side) wherever a chain is eligible
    print(cfg, two, one, par)
    assert one["digest"] == two["digest"] == par["digest"]
    assert one["batches"] == two["batches"] == par["batches"] and one["batches"]["committed_in_batch"] > 0
    assert one["changed"] > 0  # moves that changed the genome went through it
