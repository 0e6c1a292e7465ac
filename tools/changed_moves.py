"""fraction of the moves of a bench-like trajectory whose winner changes the genome (what the decide step's fast path can count on)   python tools/changed_moves.py"""
import sys, numpy as np
sys.path.insert(0,'/root/repo')
from instagraal_amd import synth
from instagraal_amd.sampler import sampler as hip_sampler
for cfg in ("cfg3","cfg2"):
    prob = synth.make_problem(*synth.CONFIGS[cfg])
    s = hip_sampler(**prob.sampler_kwargs(), device_id=0, coo=(prob.coo_row, prob.coo_col, prob.coo_cnt))
    s.set_param_simu(prob.params); s.eval_likelihood_init()
    np.random.seed(0)
    order = np.random.permutation(prob.n_frags)
    for rep in range(3):
        fr = np.resize(np.roll(order, -rep*4000), 4000).astype(np.int32)
        res = s.step_sampler_batch(fr, 5)
        d = np.concatenate([[True], (np.diff(res["dist"]) != 0) | (np.diff(res["n_contigs"]) != 0)])
        print(cfg, "moves %d..%d: changed genome %.3f; ops histogram top: %s; batches %s" % (rep*4000, rep*4000+4000, d.mean(),
              np.bincount(res["op_sampled"], minlength=24).tolist(), s.ctx.batch_stats()))
    s.free_gpu()
