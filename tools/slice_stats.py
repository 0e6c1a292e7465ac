"""distribution of the slice sizes of a trajectory (diagnostic)"""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
from instagraal_amd import hip_lib, synth
from instagraal_amd.sampler import sampler as hip_sampler
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
prob = synth.make_problem(*synth.CONFIGS[cfg])
s = hip_sampler(**prob.sampler_kwargs(), device_id=0, coo=(prob.coo_row, prob.coo_col, prob.coo_cnt))
s.set_param_simu(prob.params); s.eval_likelihood_init()
np.random.seed(0)
n = 2200
order = np.arange(prob.n_frags); np.random.shuffle(order)
frags = np.resize(order, n).astype(np.int32)
cands = s.draw_candidates(frags, 5)
res = s.ctx.step_batch(frags, cands)
for name in ("n_slice", "n_evals", "n_candidates", "op_sampled"):
    v = res[name].astype(np.float64)
    print(name, "mean %.0f  p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (v.mean(), *np.percentile(v, [10, 50, 90, 99]), v.max()))
st = s.gpu_vect_frags.copy_from_gpu()
L = np.asarray(st.l_cont); ids = np.asarray(st.id_c)
u, cnt = np.unique(ids, return_counts=True)
print("contigs", len(u), "frags/contig mean %.1f max %d" % (cnt.mean(), cnt.max()))
print("identity winners: %.3f" % np.mean(res["op_sampled"] == 24) if False else "", "changed ops histogram", np.bincount(res["op_sampled"].astype(int), minlength=24))
print(s.ctx.batch_stats())
