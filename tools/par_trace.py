#!/usr/bin/env python3
"""Where k_decide_commit_par's time goes (a build with -DIG_PAR_TRACE, loaded through IG_DEBUG_TUNING=1 IG_HIP_LIB=...): per launch, in
microseconds -- set-up (first loads + state), the rounds, the epilogue, the commit waves' end, rounds and moves per launch.
usage: IG_DEBUG_TUNING=1 IG_HIP_LIB=variants/libig_partrace.so python tools/par_trace.py [config] [moves]"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from instagraal_amd import synth  # noqa: E402
from instagraal_amd.sampler import sampler as hip_sampler  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
prob = synth.make_problem(*synth.CONFIGS[cfg])
s = hip_sampler(**prob.sampler_kwargs(), device_id=0, coo=(prob.coo_row, prob.coo_col, prob.coo_cnt))
s.set_param_simu(prob.params)
s.eval_likelihood_init()
np.random.seed(0)
fr = np.resize(np.random.permutation(prob.n_frags), 640 + n).astype(np.int32)
s.step_sampler_batch(fr[:640], 5)
s.ctx.debug_dbg(clear=True)
res = s.step_sampler_batch(fr[640:], 5)
d = s.ctx.debug_dbg().astype(np.float64)
L = max(d[0], 1.0)
if os.environ.get("IG_PAR_TRACE_MODE") == "2":
    print("%s: %d launches, %.1f rounds, %.1f moves per launch; commit waves (us per launch): prologue done at %.1f, waiting for decisions %.1f, "
          "from the last decision to their end %.1f; decide wave 0 done at %.1f, commit waves at %.1f" % (
              cfg, int(d[0]), d[1] / L, d[5] / L, d[2] / L / 100, d[3] / L / 100, d[4] / L / 100, d[7] / L / 100, d[6] / L / 100))
elif os.environ.get("IG_PAR_TRACE_MODE") == "3":
    R = max(d[1], 1.0)
    print("%s: %d launches, %.1f rounds, %.1f moves per launch; decide wave 0, us per ROUND: decision %.2f, waiting at the first barrier %.2f, "
          "scan + finalize %.2f, waiting at the second barrier %.2f; done at %.1f" % (cfg, int(d[0]), d[1] / L, d[5] / L, d[2] / R / 100, d[3] / R / 100,
                                                                                    d[4] / R / 100, d[6] / R / 100, d[7] / L / 100))
else:
    print("%s: %d launches, %.1f rounds and %.1f moves per launch; per launch (us): set-up %.1f, rounds %.1f (%.2f per round), epilogue %.1f; "
          "first decide wave done at %.1f, commit waves done at %.1f" % (cfg, int(d[0]), d[1] / L, d[5] / L, d[2] / L / 100, d[3] / L / 100,
                                                                      d[3] / max(d[1], 1) / 100, d[4] / L / 100, d[7] / L / 100, d[6] / L / 100))
changed = float(np.mean(np.diff(np.concatenate([[res["dist"][0]], res["dist"]])) != 0))
print("moves that changed the genome distance: %.1f %%" % (100 * changed))
