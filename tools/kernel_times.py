"""hipEvent timings of the kernels of a batch over a bench-like run (no profiler): python tools/kernel_times.py [cfg3] [moves]"""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np

from instagraal_amd import synth
from instagraal_amd.sampler import sampler as hip_sampler

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
prob = synth.make_problem(*synth.CONFIGS[cfg])
s = hip_sampler(**prob.sampler_kwargs(), device_id=0, coo=(prob.coo_row, prob.coo_col, prob.coo_cnt))
s.set_param_simu(prob.params)
s.eval_likelihood_init()
np.random.seed(0)
frags = np.resize(np.random.permutation(prob.n_frags), n + 200).astype(np.int32)
s.step_sampler_batch(frags[:200], 5)
import time

for name in ("gather", "mutate", "slice", "screen", "score", "finalize", "commit"):
    mask = {"gather": 0, "mutate": 1, "score": 2, "finalize": 3, "commit": 7, "slice": 8, "screen": 10}[name]
    s.ctx.reset_timers(1 | ((1 << mask) << 1))
    t0 = time.perf_counter()
    res = s.step_sampler_batch(frags[200:], 5)
    dt = time.perf_counter() - t0
    ms, nl = s.ctx.kernel_time_ms(name)
    print("%-9s %8.1f us per launch, %4d launches   (run: %.0f moves/s)" % (name, 1e3 * ms, nl, n / dt))
    s.ctx.reset_timers(0)
print("screen stats", s.ctx.debug_screen_stats())
