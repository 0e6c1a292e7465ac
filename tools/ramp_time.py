"""how fast a fresh context reaches its sustained rate: python tools/ramp_time.py [cfg3] [moves per call] [calls]
(per call: wall time, moves/s, batches launched, pool size)"""
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np

from instagraal_amd import synth
from instagraal_amd.sampler import sampler as hip_sampler

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 120
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 12
prob = synth.make_problem(*synth.CONFIGS[cfg])
s = hip_sampler(**prob.sampler_kwargs(), device_id=0, coo=(prob.coo_row, prob.coo_col, prob.coo_cnt))
s.set_param_simu(prob.params)
s.eval_likelihood_init()
np.random.seed(0)
order = np.random.permutation(prob.n_frags).astype(np.int32)
frags = np.resize(order, k * calls)
prev = s.ctx.batch_stats()
for c in range(calls):
    f = frags[c * k: (c + 1) * k]
    t0 = time.perf_counter()
    s.step_sampler_batch(f, 5)
    dt = time.perf_counter() - t0
    st = s.ctx.batch_stats()
    print("call %2d: %7.0f us  %6.0f moves/s  %s" % (c, dt * 1e6, k / dt, {q: st[q] - prev[q] for q in st}), s.ctx.scratch_bytes()[1] >> 20, "MB pool")
    prev = st
