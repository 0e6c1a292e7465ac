import sys, os, json, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
from instagraal_amd import hip_lib, synth
from instagraal_amd.sampler import sampler as hip_sampler
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
prob = synth.make_problem(*synth.CONFIGS[cfg])
s = hip_sampler(**prob.sampler_kwargs(), device_id=0, coo=(prob.coo_row, prob.coo_col, prob.coo_cnt))
s.set_param_simu(prob.params); s.eval_likelihood_init()
np.random.seed(0)
n = 1200
order = np.arange(prob.n_frags); np.random.shuffle(order)
frags = np.resize(order, n).astype(np.int32)
cands = s.draw_candidates(frags, 5)
s.ctx.step_batch(frags[:200], cands[:200])
for W in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,8,16").split(",")]:
    hip_lib.set_batch_width(W)
    s.ctx.reset_timers(1)
    b0 = s.ctx.batch_stats()
    t0 = time.perf_counter()
    res = s.ctx.step_batch(frags[200:], cands[200:])
    dt = time.perf_counter() - t0
    b1 = s.ctx.batch_stats()
    out = {"W": W, "us_per_move_with_timers": 1e6 * dt / (n - 200), "batches": b1["batches"] - b0["batches"],
           "tails": b1["one_move_tails"] - b0["one_move_tails"]}
    for k in ["gather", "mutate", "slice", "score", "finalize", "argmax", "delta", "apply", "post", "commit"]:
        ms, cnt = s.ctx.kernel_time_ms(k)
        out[k] = (round(ms * 1e3, 1), cnt)
    print(json.dumps(out))
    s.ctx.reset_timers(0)
