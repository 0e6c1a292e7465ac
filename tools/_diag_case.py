import os, sys, warnings
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
warnings.filterwarnings("ignore", category=RuntimeWarning)
seed, nmv = int(sys.argv[1]), int(sys.argv[2])
sys.argv = [sys.argv[0], "0"]
src = open(os.path.join(os.path.dirname(__file__), "fuzz_batches.py")).read().split("\nbad = 0\n")[0]
exec(src)
from oracle import oracle_lib as ol
from oracle.sampler_oracle import OracleSampler
ol.build()
ol.set_threads(16)
prob, params, n, width, wide, inject, n_nb, desc = make_case(seed)
print(desc, params, flush=True)


def tup(r):
    return (float(r["o"]), float(r["dist"]), int(r["op_sampled"]), int(r["id_f_sampled"]), float(np.float32(r["mean_len"])), int(r["n_contigs"]))


np.random.seed(seed)
s = hip_sampler(**prob.sampler_kwargs(), device_id=0)
s.set_param_simu(params)
s.eval_likelihood_init()
frags = np.resize(np.random.permutation(prob.n_frags), n).astype(np.int32)[:nmv]
st0 = np.random.get_state()
resA = s.step_sampler_batch(frags, n_nb)
candsA = np.array(s.last_candidates)
s.free_gpu()
np.random.set_state(st0)
s = hip_sampler(**prob.sampler_kwargs(), device_id=0)
s.set_param_simu(params)
s.eval_likelihood_init()
o = OracleSampler(**prob.sampler_kwargs(), mode=ol.MODE_DET)
o.set_param_simu(params)
o.eval_likelihood_init()
for i, f in enumerate(frags):
    stm = np.random.get_state()
    b = s.step_sampler(int(f), n_nb)
    cb = list(s.candidates)
    sc = np.array(s.all_scores).reshape(len(cb), -1)
    np.random.set_state(stm)
    q = o.step_sampler(int(f), n_nb, o.dt)
    tb = (float(b[0]), float(b[1]), int(b[2]), int(b[3]), float(b[4]), int(b[5]))
    tq = (float(q[0]), float(q[1]), int(q[2]), int(q[3]), float(q[4]), int(q[5]))
    ta = tup(resA[i])
    flag = "" if ta == tb == tq else "   <<<<<< A==B %s, B==oracle %s, A==oracle %s" % (ta == tb, tb == tq, ta == tq)
    print(i, int(f), "cands", cb, "batch cands", [int(x) for x in candsA[i] if x >= 0], "\n   A", ta, "\n   B", tb, "\n   O", tq, flag, flush=True)
    if flag:
        print("   B's scores (candidates x 24):")
        for c, row in zip(cb, sc):
            print("     ", c, " ".join("%.3f" % v for v in row))
        so = np.array(o.all_scores).reshape(len(cb), -1)
        print("   oracle's scores:")
        for c, row in zip(cb, so):
            print("     ", c, " ".join("%.3f" % v for v in row))
        print("   columns that differ (candidate, column, B, oracle):", [(cb[a], b_, float(sc[a, b_]), float(so[a, b_])) for a in range(sc.shape[0]) for b_ in range(sc.shape[1]) if sc[a, b_] != so[a, b_]][:40])
        break
