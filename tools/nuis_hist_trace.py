"""per-step errors and bounds of the two screened tiers against the exact pass (IG_NUIS_SCREEN_VERIFY=1, IG_NUIS_HIST_TRACE=1):
python tools/nuis_hist_trace.py [cfg] [steps] [seed]   -> the library prints one line per step to stderr"""
import os
import sys

os.environ["IG_NUIS_SCREEN_VERIFY"] = "1"
os.environ["IG_NUIS_HIST_TRACE"] = "1"
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np

from instagraal_amd import synth
from instagraal_amd.sampler import sampler as hip_sampler

cfg = sys.argv[1] if len(sys.argv) > 1 else "tiny"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 250
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 5
prob = synth.make_problem(*synth.CONFIGS[cfg])
np.random.seed(seed)
s = hip_sampler(**prob.sampler_kwargs(), device_id=0, coo=(prob.coo_row, prob.coo_col, prob.coo_cnt) if cfg == "cfg3" else None)
s.set_param_simu(prob.params)
s.bins = np.arange(1.0, 60.0, 1.0)
s.eval_likelihood_init()
frags = np.resize(np.random.permutation(prob.n_frags), n)
s.step_sampler_nuisance_batch(frags, 5, s.dt, 0, n)
print(s.ctx.debug_nuis_screen_stats())
print(s.ctx.debug_nuis_hist_stats(), "mismatch", s.ctx.debug_nuis_hist_check())
