"""one move per call, the reference's loop shape: sampler.step_sampler against a one-move step_sampler_batch call: python tools/one_move_time.py [cfg3] [moves]"""
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np

from instagraal_amd import synth
from instagraal_amd.sampler import sampler as hip_sampler

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
prob = synth.make_problem(*synth.CONFIGS[cfg])
s = hip_sampler(**prob.sampler_kwargs(), device_id=0, coo=(prob.coo_row, prob.coo_col, prob.coo_cnt))
s.set_param_simu(prob.params)
s.eval_likelihood_init()
np.random.seed(0)
frags = np.random.permutation(prob.n_frags).astype(np.int32)
for f in frags[:10]:
    s.step_sampler(int(f), 5, s.dt)
def _no_scores(f):
    s.keep_all_scores = False
    try:
        return s.step_sampler(int(f), 5, s.dt)
    finally:
        s.keep_all_scores = True


for name, fn in (("step_sampler, all_scores kept", lambda f: s.step_sampler(int(f), 5, s.dt)),
                 ("step_sampler, no all_scores", _no_scores),
                 ("step_sampler_batch of one move", lambda f: s.step_sampler_batch(np.array([f], dtype=np.int32), 5))):
    ts = []
    for f in frags[10:10 + n]:
        t0 = time.perf_counter()
        fn(f)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    print("%-34s median %.0f us, p10 %.0f, p90 %.0f per move (%.0f moves/s)" % (name, np.median(ts), np.percentile(ts, 10), np.percentile(ts, 90), 1e6 / ts.mean()))
    frags = frags[n:]
print("ig_step_draw: %s, batch stats %s" % (s.ctx.debug_step_stats(), s.ctx.batch_stats()))
