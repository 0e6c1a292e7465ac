import numpy as np, sys
sys.path.insert(0,'.')
from instagraal_amd import synth, hip_lib
from instagraal_amd.sampler import problem_to_context
g=np.load("tests/golden/tiny_plain_mode1.npz")
prob=synth.make_problem(*synth.CONFIGS["tiny"])
ctx=problem_to_context(prob)
for t in range(4):
    f=int(g["frag"][t]); cands=[int(c) for c in g["cands"][t] if c>=0]
    res,sc=ctx.step(f,cands)
    sums,ints=ctx.debug_globals()
    nz,z,limbs=ctx.full_likelihood()
    print(t,"A",f,"cands",cands,"op",res.op_sampled,"B",res.id_f_sampled,"ch",ints[2:], "maintained",sums[:2],"fresh",limbs[:2], "diff_q", (int(sums[0])-int(limbs[0]))*2**32+int(sums[1])-int(limbs[1]), "z ok", np.array_equal(sums[2:],limbs[2:]), sums[2:], limbs[2:])
    st=ctx.download_state()
    k=list(g["state_every"]).index(t) if t in g["state_every"] else None
