"""Wall time of ig_full_likelihood (k_pack_tab + k_full_nz + k_full_zero + the result copy) on a synthetic configuration."""
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np

from instagraal_amd import synth
from instagraal_amd.sampler import sampler as hip_sampler

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
prob = synth.make_problem(*synth.CONFIGS[cfg])
s = hip_sampler(**prob.sampler_kwargs(), device_id=0, coo=(prob.coo_row, prob.coo_col, prob.coo_cnt))
s.set_param_simu(prob.params)
s.eval_likelihood_init()
for _ in range(5):
    s.ctx.full_likelihood(0)
t0 = time.perf_counter()
for _ in range(n):
    r = s.ctx.full_likelihood(0)
dt = (time.perf_counter() - t0) / n
Z = len(prob.coo_row)
print("%s: ig_full_likelihood %.0f us per call, %d contacts: %.2f TB/s of 12 B per contact" % (cfg, 1e6 * dt, Z, 12.0 * Z / dt / 1e12))
