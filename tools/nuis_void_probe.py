"""why steps of a nuisance run fall out of the screened tiers on an evolved genome: python tools/nuis_void_probe.py [cfg3] [plain moves] [steps]
(run with IG_NUIS_HIST_TRACE=1 for the void flags of every such step)"""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np

from instagraal_amd import synth
from instagraal_amd.sampler import sampler as hip_sampler

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n_plain = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
n_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
prob = synth.make_problem(*synth.CONFIGS[cfg])
s = hip_sampler(**prob.sampler_kwargs(), device_id=0, coo=(prob.coo_row, prob.coo_col, prob.coo_cnt))
s.set_param_simu(prob.params)
s.bins = np.arange(1.0, 60.0, 1.0)
s.eval_likelihood_init()
np.random.seed(0)
done = 0
while done < n_plain:
    k = min(prob.n_frags, n_plain - done)
    s.step_sampler_batch(np.random.permutation(prob.n_frags)[:k].astype(np.int32), 5)
    done += k
st = s.gpu_vect_frags.copy_from_gpu()
ids, first = np.unique(st.id_c, return_index=True)
print("after %d plain moves: %d contigs, %d of them rings (%s bins)" % (done, len(ids), int(st.circ[first].sum()), st.l_cont[first][st.circ[first] != 0].tolist()), flush=True)
for r in range(4):
    s.step_sampler_nuisance_batch(np.random.permutation(prob.n_frags)[:n_steps], 5, s.dt, 0, n_steps)
    st = s.gpu_vect_frags.copy_from_gpu()
    ids, first = np.unique(st.id_c, return_index=True)
    print("run %d: rings now %d; histogram tier %s" % (r, int(st.circ[first].sum()), s.ctx.debug_nuis_hist_stats()), flush=True)
