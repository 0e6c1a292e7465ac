"""the from-scratch pass over all contacts (k_full_nz_tiled) on the state after a number of moves: python tools/full_pass_time.py [cfg3] [moves]
(run under rocprofv3 --kernel-trace for the kernel's own duration)"""
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np

from instagraal_amd import synth
from instagraal_amd.sampler import sampler as hip_sampler

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 600
prob = synth.make_problem(*synth.CONFIGS[cfg])
s = hip_sampler(**prob.sampler_kwargs(), device_id=0, coo=(prob.coo_row, prob.coo_col, prob.coo_cnt))
s.set_param_simu(prob.params)
s.eval_likelihood_init()
np.random.seed(0)
frags = np.resize(np.random.permutation(prob.n_frags), n).astype(np.int32)
if n:
    s.step_sampler_batch(frags, 5)
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20):
        s.ctx.full_likelihood(0)
    print("full_likelihood: %.1f us per call (host clock, all launches + sync)" % (1e6 * (time.perf_counter() - t0) / 20))
if os.environ.get("TILE_TRACE"):
    tr = s.ctx.debug_tile_trace()
    tr = tr[tr[:, 1] > 0]
    st, en, hw = tr[:, 0], tr[:, 1], tr[:, 2]
    items, nc = tr[:, 3] >> 32, tr[:, 3] & 0xffffffff
    t0 = st.min()
    dur = (en - st) / 100.0  # us
    print("workgroups %d, span %.1f us; duration mean %.1f us, p90 %.1f, max %.1f us; items per workgroup mean %.2f max %d; contacts read %d" % (
        len(tr), (en.max() - t0) / 100.0, dur.mean(), np.percentile(dur, 90), dur.max(), items.mean(), items.max(), nc.sum()))
    print("last start %.1f us; ends: p50 %.1f p90 %.1f p99 %.1f us" % ((st.max() - t0) / 100.0, *[(np.percentile(en, q) - t0) / 100.0 for q in (50, 90, 99)]))
    ev = np.concatenate([np.stack([st, np.ones(len(st))], 1), np.stack([en, -np.ones(len(st))], 1)])
    ev = ev[np.argsort(ev[:, 0])]
    conc = np.cumsum(ev[:, 1])
    tt = (ev[:, 0] - t0) / 100.0
    for a in range(0, int(tt.max()) + 1, 10):
        m = (tt >= a) & (tt < a + 10)
        if m.any():
            print("  t=%3d..%3d us: %4.0f workgroups in flight (mean)" % (a, a + 10, conc[m].mean()))
