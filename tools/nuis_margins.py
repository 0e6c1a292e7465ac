"""How decisive is the Metropolis test of step_nuisance_parameters (CL:3023-3036)?  Per step of a nuisance-on run at a
synthetic shape: dL = L_test - L_move against T ln u, the margin |dL - T ln u| a screened pass would have to resolve, and
how the acceptance rate moves over the run.   python tools/nuis_margins.py [cfg3] [moves] [chunk]"""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np

from instagraal_amd import synth
from instagraal_amd.sampler import sampler as hip_sampler

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 200
prob = synth.make_problem(*synth.CONFIGS[cfg])
s = hip_sampler(**prob.sampler_kwargs(), device_id=0, coo=(prob.coo_row, prob.coo_col, prob.coo_cnt))
s.set_param_simu(prob.params)
s.bins = np.arange(1.0, 60.0, 1.0)
s.eval_likelihood_init()
np.random.seed(0)
rec = []
orig = s.ctx.nuis_step_next


def spy(temperature, u, pr, pa, mean_kb, has_next):
    r, nz, z, acc = orig(temperature, u, pr, pa, mean_kb, has_next)
    rec.append(((nz + z) - r.o, temperature * np.log(u) if u > 0 else -np.inf, acc))
    return r, nz, z, acc


s.ctx.nuis_step_next = spy
frags = np.resize(np.random.permutation(prob.n_frags), n)
for a in range(0, n, chunk):
    rec.clear()
    s.step_sampler_nuisance_batch(frags[a:a + chunk], 5, s.dt, a, n)
    d = np.array([x[0] for x in rec])
    t = np.array([x[1] for x in rec])
    acc = np.array([x[2] for x in rec])
    m = np.abs(d - t)
    q = np.percentile(m, [1, 5, 25, 50, 75])
    print("moves %5d..%5d  accept %.2f  |dL - T ln u| percentiles 1/5/25/50/75: %s   below 100: %.3f  300: %.3f  1000: %.3f  3000: %.3f; accepted with dL < 1000: %.3f" % (
        a, a + chunk, np.mean(acc == 1), " ".join("%.3g" % v for v in q), np.mean(m < 100), np.mean(m < 300), np.mean(m < 1000), np.mean(m < 3000),
        np.mean((acc == 1) & (d < 1000))), flush=True)
print("parameters now:", [float(s.param_simu[k][0]) for k in ("fact", "slope", "d_max", "v_inter")])
