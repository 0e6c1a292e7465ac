"""What ONE step_sampler_batch call costs beyond its moves: calls of 120 .. 7 680 moves at cfg3, the median of a few each, and the line
through them (seconds = intercept + moves / rate).  The driver's bench line times 480 moves in one call (--steps 20): the intercept is a
tenth of that.

    python tools/call_overhead.py [CFG] [--reps R]
"""
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np

from instagraal_amd import synth
from instagraal_amd.sampler import sampler as hip_sampler


def main(argv):
    cfg = argv[0] if argv and not argv[0].startswith("-") else "cfg3"
    reps = int(argv[argv.index("--reps") + 1]) if "--reps" in argv else 5
    prob = synth.make_problem(*synth.CONFIGS[cfg])
    np.random.seed(3)
    s = hip_sampler(**prob.sampler_kwargs(), device_id=0, coo=(prob.coo_row, prob.coo_col, prob.coo_cnt))
    s.set_param_simu(prob.params)
    s.eval_likelihood_init()
    N = prob.n_frags
    s.step_sampler_batch(np.random.permutation(N)[:960].astype(np.int32), 5)  # warm
    sizes = [120, 240, 480, 960, 1920, 3840, 7680]
    rows = []
    for n in sizes:
        ts, bs = [], []
        for _ in range(reps):
            frags = np.random.permutation(N)[:n].astype(np.int32)
            b0 = s.ctx.batch_stats()["batches"]
            t0 = time.perf_counter()
            s.step_sampler_batch(frags, 5)
            ts.append(time.perf_counter() - t0)
            bs.append(s.ctx.batch_stats()["batches"] - b0)
        t = float(np.median(ts))
        rows.append((n, t, float(np.median(bs))))
        print("%5d moves per call: %8.3f ms  (%.1f k moves/s, %.1f launch chains, %.1f moves per chain)" % (n, t * 1e3, n / t / 1e3, rows[-1][2], n / rows[-1][2]),
              flush=True)
    x = np.array([r[0] for r in rows], float)
    y = np.array([r[1] for r in rows], float)
    A = np.vstack([np.ones_like(x), x]).T
    (a, b), *_ = np.linalg.lstsq(A, y, rcond=None)
    print("seconds = %.0f us + moves / %.1f k per s" % (a * 1e6, 1.0 / b / 1e3))
    s.free_gpu()
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
