"""plain and nuisance-on rates under the synthetic parameters (d_max ~ 450 kb: a P_z table of 250 entries) and under parameters as a
settled chain has them (d_max 3e6 kb: the table longer than the 1 024 entries the kernels stage, rank distances beyond it read from
the table in memory / the formula):   python tools/long_table_probe.py [bigctg] [plain moves] [nuisance moves]"""
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np

from instagraal_amd import synth
from instagraal_amd.sampler import sampler as hip_sampler

cfg = sys.argv[1] if len(sys.argv) > 1 else "bigctg"
n_plain = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
n_nuis = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
prob = synth.make_problem(*synth.CONFIGS[cfg])
amp = prob.params["c1"] * prob.params["fact"]
sets = {"synthetic": prob.params, "settled-like": dict(prob.params, slope=-0.53, d_max=3.0e6, v_inter=float(amp * 3.0e6 ** -0.53))}
for name, params in sets.items():
    s = hip_sampler(**prob.sampler_kwargs(), device_id=0, coo=(prob.coo_row, prob.coo_col, prob.coo_cnt))
    s.set_param_simu(params)
    s.bins = np.arange(1.0, 60.0, 1.0)
    s.eval_likelihood_init()
    np.random.seed(0)
    fr = np.resize(np.random.permutation(prob.n_frags), n_plain + 200).astype(np.int32)
    s.step_sampler_batch(fr[:200], 5)
    t0 = time.perf_counter()
    s.step_sampler_batch(fr[200:], 5)
    dt = time.perf_counter() - t0
    print("%s, %s: plain %.0f moves/s" % (cfg, name, n_plain / dt), flush=True)
    fr = np.resize(np.random.permutation(prob.n_frags), n_nuis + 100)
    s.step_sampler_nuisance_batch(fr[:100], 5, s.dt, 0, n_nuis)
    t0 = time.perf_counter()
    res, tup = s.step_sampler_nuisance_batch(fr[100:], 5, s.dt, 0, n_nuis)
    dt = time.perf_counter() - t0
    print("%s, %s: nuisance on %.0f moves/s, accept %.2f\n    screened pass %s\n    histogram tier %s" % (
        cfg, name, n_nuis / dt, np.mean([q[6] for q in tup]), s.ctx.debug_nuis_screen_stats(), s.ctx.debug_nuis_hist_stats()), flush=True)
    print("    parameters now:", {k: float(s.param_simu[k][0]) for k in ("slope", "d_max", "v_inter", "fact")}, flush=True)
    s.free_gpu()
