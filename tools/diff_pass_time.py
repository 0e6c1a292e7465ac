"""average duration of the screened nuisance pass's kernel (k_full_diff_tiled, hipEvents on its stream) over a run of
nuisance steps   python tools/diff_pass_time.py [cfg3] [steps]   (IG_DEBUG_TUNING=1 IG_HIP_LIB=<tuning build> to time a variant)"""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
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
s.step_sampler_nuisance_batch(frags[:20], 5, s.dt, 0, n)
s.ctx.reset_timers(1 | ((1 << 11) << 1))
try:
    s.step_sampler_nuisance_batch(frags[20:], 5, s.dt, 0, n)
except Exception as e:  # ablated builds compute garbage: the timing is what counts
    print("   (run ended early: %s)" % str(e)[:80])
ms, k = s.ctx.kernel_time_ms("diff")
print("%s %s: k_full_diff_tiled %.1f us average over %d launches" % (os.environ.get("IG_HIP_LIB", "default build").split("/")[-1], cfg, 1e3 * ms, k))
