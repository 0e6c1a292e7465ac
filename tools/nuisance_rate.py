"""moves/s of the reference-shaped loop WITH nuisance sampling (instagraal.py:217-262 for cycles > 4): one step_sampler and
one step_nuisance_parameters (a full pass over all contacts under test parameters) per move.  Diagnostic, not the
BASELINE metric."""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np

from instagraal_amd import synth
from instagraal_amd.sampler import sampler as hip_sampler

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
prob = synth.make_problem(*synth.CONFIGS[cfg])
s = hip_sampler(**prob.sampler_kwargs(), device_id=0, coo=(prob.coo_row, prob.coo_col, prob.coo_cnt))
s.set_param_simu(prob.params)
s.bins = np.arange(1.0, 60.0, 1.0)
s.eval_likelihood_init()
np.random.seed(0)
frags = np.random.permutation(prob.n_frags)[: n + 20]
for f in frags[:20]:
    s.step_sampler(int(f), 5, s.dt)
    s.step_nuisance_parameters(s.dt, 0, n)
t0 = time.perf_counter()
t_s = t_n = 0.0
for t, f in enumerate(frags[20:]):
    a = time.perf_counter()
    s.step_sampler(int(f), 5, s.dt)
    b = time.perf_counter()
    s.step_nuisance_parameters(s.dt, t, n)
    c = time.perf_counter()
    t_s += b - a
    t_n += c - b
dt = time.perf_counter() - t0
print("%s: %.0f moves/s with nuisance sampling (step_sampler %.0f us, step_nuisance_parameters %.0f us per move)" % (
    cfg, n / dt, 1e6 * t_s / n, 1e6 * t_n / n))
