"""How many of the <= 120 candidate genomes of a move are within a given score distance of the winner?  (Sizing of a two-tier
scoring: a screening pass with a rigorous error bound B per column needs the exact term only for columns within 2 B of the
leader.)  usage: python tools/score_gaps.py [cfg3] [n_moves] [synthetic|settled]"""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np

from instagraal_amd import synth
from instagraal_amd.sampler import sampler as hip_sampler

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
prob = synth.make_problem(*synth.CONFIGS[cfg])
s = hip_sampler(**prob.sampler_kwargs(), device_id=0, coo=(prob.coo_row, prob.coo_col, prob.coo_cnt))
which = sys.argv[3] if len(sys.argv) > 3 else "synthetic"
s.set_param_simu(prob.params if which == "synthetic" else synth.settled_params(prob.params))
s.eval_likelihood_init()
np.random.seed(0)
frags = np.random.permutation(prob.n_frags)[:n]
thr = [0.0, 1e-3, 0.01, 0.1, 0.5, 2.0, 8.0, 32.0, 128.0, 1e3, 1e4]
within = np.zeros((n, len(thr)))
ncols = np.zeros(n)
slices = np.zeros(n)
for t, f in enumerate(frags):
    s.step_sampler(int(f), 5, s.dt)
    sc = s.all_scores
    ok = sc != 0
    ncols[t] = ok.sum()
    gap = sc[ok].max() - sc[ok]
    within[t] = [(gap <= x).sum() for x in thr]
    slices[t] = s.last_result.n_slice
print(cfg, which, "moves", n, "columns scored per move %.1f" % ncols.mean(), "slice entries per move %.0f" % slices.mean())
for x, w in zip(thr, within.mean(0)):
    print("  within %8.1f of the winner: %6.2f columns per move (%.1f %%)" % (x, w, 100 * w / ncols.mean()))
