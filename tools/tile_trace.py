"""per-workgroup trace of one from-scratch pass (ig_debug_tile_trace): items, contacts, time per workgroup   python tools/tile_trace.py [cfg3] [moves]"""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np

from instagraal_amd import synth
from instagraal_amd.sampler import sampler as hip_sampler

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
moves = int(sys.argv[2]) if len(sys.argv) > 2 else 600
prob = synth.make_problem(*synth.CONFIGS[cfg])
s = hip_sampler(**prob.sampler_kwargs(), device_id=0, coo=(prob.coo_row, prob.coo_col, prob.coo_cnt))
s.set_param_simu(prob.params)
s.eval_likelihood_init()
np.random.seed(0)
if moves:
    s.step_sampler_batch(np.random.permutation(prob.n_frags)[:moves], 5)
for rep in range(2):
    t = s.ctx.debug_tile_trace()
t = t[t[:, 1] > 0]
dur = (t[:, 1] - t[:, 0]) / 100.0  # us (100 MHz clock)
items, contacts = t[:, 3] >> 32, t[:, 3] & 0xffffffff
t0 = t[:, 0].min()
print("workgroups %d, items %d, contacts read %d; span %.1f us" % (len(t), items.sum(), contacts.sum(), (t[:, 1].max() - t0) / 100.0))
print("per workgroup: duration us min/med/max %.1f %.1f %.1f; items min/med/max %d %d %d; contacts min/med/max %d %d %d" % (
    dur.min(), np.median(dur), dur.max(), items.min(), np.median(items), items.max(), contacts.min(), np.median(contacts), contacts.max()))
print("start offsets us: med %.1f max %.1f; end offsets: min %.1f med %.1f" % (np.median(t[:, 0] - t0) / 100.0, (t[:, 0].max() - t0) / 100.0,
                                                                               (t[:, 1].min() - t0) / 100.0, np.median(t[:, 1] - t0) / 100.0))
per_item = dur / np.maximum(items, 1)
print("us per item: min/med/max %.2f %.2f %.2f; ns per contact med %.2f" % (per_item.min(), np.median(per_item), per_item.max(), 1e3 * np.median(dur / np.maximum(contacts, 1))))
xcc = (t[:, 2] >> 32) & 0xf
for x in range(8):
    m = xcc == x
    if m.any():
        print("  XCD %d: %3d workgroups, %6d items, end max %.1f us" % (x, m.sum(), items[m].sum(), (t[m, 1].max() - t0) / 100.0))

# the screened nuisance pass (csrc/ig_kernels_nuis.cuh) on the same state, under a 1 % proposal on fact
from instagraal_amd.sampler import PARAM_NAMES

p8 = [float(s.param_simu[k][0]) for k in PARAM_NAMES]
p8[6] *= 1.01
for rep in range(3):
    t, sums = s.ctx.debug_diff_trace(p8, s.mean_kb())
dur = (t[:, 1] - t[:, 0]) / 100.0
items, contacts = t[:, 3] >> 32, t[:, 3] & 0xffffffff
t0 = t[:, 0].min()
print("screened pass: workgroups %d, items %d, contacts %d; span %.1f us; output words %s" % (len(t), items.sum(), contacts.sum(), (t[:, 1].max() - t0) / 100.0, list(sums)))
print("  per workgroup us min/med/max %.1f %.1f %.1f; until staged (sum over its items) med %.1f max %.1f; contact loops med %.1f max %.1f; start offset max %.1f" % (
    dur.min(), np.median(dur), dur.max(), np.median(t[:, 4]) / 100.0, t[:, 4].max() / 100.0, np.median(t[:, 5]) / 100.0, t[:, 5].max() / 100.0, (t[:, 0].max() - t0) / 100.0))
print("  per item: staged after %.2f us, loops %.2f us (medians)" % (np.median(t[:, 4] / np.maximum(items, 1)) / 100.0, np.median(t[:, 5] / np.maximum(items, 1)) / 100.0))
# what makes a workgroup slow?  duration against its number of runs (stagings), its XCD, its place in the grid
runs = items
for k in sorted(set(runs.tolist())):
    m = runs == k
    print("  %d run(s): %3d workgroups, duration med %.1f max %.1f us, staged med %.1f, loops med %.1f" % (k, m.sum(), np.median(dur[m]), dur[m].max(), np.median(t[m, 4]) / 100.0, np.median(t[m, 5]) / 100.0))
xcc = (t[:, 2] >> 32) & 0xf
hw = t[:, 2] & 0xffffffff
cu = (hw >> 8) & 0xf
se = (hw >> 13) & 0x7
for x in range(8):
    m = xcc == x
    if m.any():
        print("  XCD %d: %3d workgroups, duration med %.1f max %.1f" % (x, m.sum(), np.median(dur[m]), dur[m].max()))
order = np.argsort(-dur)[:12]
print("  slowest:", [(int(i), round(float(dur[i]), 1), int(runs[i]), int(xcc[i]), int(contacts[i])) for i in order])
q = np.arange(len(dur))
for lo in range(0, len(dur), 64):
    print("  workgroups %3d..%3d: duration med %.1f" % (lo, min(lo + 63, len(dur) - 1), np.median(dur[lo:lo + 64])))
