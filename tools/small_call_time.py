"""wall time of SHORT step_sampler_batch calls (the driver's bench call is 20 moves: one batch): python tools/small_call_time.py [cfg3] [moves per call] [calls]"""
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np

from instagraal_amd import synth
from instagraal_amd.sampler import sampler as hip_sampler

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 20
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 40
prob = synth.make_problem(*synth.CONFIGS[cfg])
s = hip_sampler(**prob.sampler_kwargs(), device_id=0, coo=(prob.coo_row, prob.coo_col, prob.coo_cnt))
s.set_param_simu(prob.params)
s.eval_likelihood_init()
np.random.seed(0)
frags = np.random.permutation(prob.n_frags).astype(np.int32)
s.step_sampler_batch(frags[:5], 5)
ts = []
b0 = s.ctx.batch_stats()["batches"]
for c in range(calls):
    f = frags[5 + c * k: 5 + (c + 1) * k]
    t0 = time.perf_counter()
    s.step_sampler_batch(f, 5)
    ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e6
print("%d calls of %d moves: median %.0f us, min %.0f us per call (%.1f us per move; %.2f batches per call)" % (
    calls, k, np.median(ts), ts.min(), np.median(ts) / k, (s.ctx.batch_stats()["batches"] - b0) / calls))
