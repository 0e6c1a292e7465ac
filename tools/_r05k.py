import sys, os, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tools')
os.environ["IG_WIDE_LISTS"]="1"
from instagraal_amd import synth, hip_lib
from instagraal_amd.sampler import sampler as S
src=open('tools/fuzz_batches.py').read()
ns={'__file__':os.path.abspath('tools/fuzz_batches.py')}
sys.argv=['x','0','1']
exec(compile(src[:src.index('def run(')],'fb','exec'),ns)
prob, params, n, width, wide, inject, n_nb, desc = ns['make_case'](240)
print(desc)
out={}
for win in (0, 31):
    hip_lib.set_window(win)
    np.random.seed(240)
    s = S(**prob.sampler_kwargs(), device_id=0); s.set_param_simu(params); s.eval_likelihood_init()
    frags = np.resize(np.random.permutation(prob.n_frags), n).astype(np.int32)
    cands = s.draw_candidates(frags, n_nb)
    res = s.ctx.step_batch(frags, cands)
    sums,_ = s.ctx.debug_globals(); _,_,limbs = s.ctx.full_likelihood(0)
    print(win, "sums ok:", [int(x) for x in sums[:5]] == [int(x) for x in limbs[:5]])
    out[win] = (res.copy(), s.gpu_vect_frags.copy_from_gpu().soa17())
    s.free_gpu()
a,b = out[0][0], out[31][0]
for i in range(len(a)):
    ra=(a["o"][i],a["op_sampled"][i],a["id_f_sampled"][i],a["n_contigs"][i],a["dist"][i]); rb=(b["o"][i],b["op_sampled"][i],b["id_f_sampled"][i],b["n_contigs"][i],b["dist"][i])
    if ra!=rb:
        print("first difference at move", i, "frag", frags[i], "cands", cands[i], "\n old", ra, "\n win", rb); break
else: print("no difference in", len(a), "records; states equal:", np.array_equal(out[0][1], out[31][1]))
