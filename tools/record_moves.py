"""Record a trajectory of the HIP batch path for tools/commit_sim.py: per move the focal bin, its candidate
list and the winner (partner, operator) -- everything else the simulation needs (the contigs a move READS: those of the focal bin and
of its candidates; the contigs it WRITES; whether it changed the genome) follows from replaying these on the genome, which needs no
GPU (tools/commit_sim.py does it with the oracle's operators).

    python tools/record_moves.py CFG MOVES OUT.npz [--warm MOVES] [--params settled] [--seed S]

--warm: that many moves first, unrecorded (an evolved genome; the state the recording starts from is stored with it).
"""
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np

from instagraal_amd import synth
from instagraal_amd.sampler import sampler as hip_sampler


def main(argv):
    warm, params, seed = 0, "synthetic", 3
    pos = []
    it = iter(argv)
    for a in it:
        if a == "--warm":
            warm = int(next(it))
        elif a == "--params":
            params = next(it)
        elif a == "--seed":
            seed = int(next(it))
        else:
            pos.append(a)
    cfg, n_moves, out = pos[0], int(pos[1]), pos[2]
    prob = synth.make_problem(*synth.CONFIGS[cfg])
    np.random.seed(seed)
    s = hip_sampler(**prob.sampler_kwargs(), device_id=0, coo=(prob.coo_row, prob.coo_col, prob.coo_cnt))
    s.set_param_simu(prob.params if params == "synthetic" else synth.settled_params(prob.params))
    s.eval_likelihood_init()
    N = prob.n_frags
    order = np.arange(N)

    def cycle_moves(n):
        parts = []
        while sum(len(p) for p in parts) < n:
            np.random.shuffle(order)
            parts.append(order.copy())
        return np.concatenate(parts)[:n].astype(np.int32)

    if warm:
        t0 = time.time()
        s.step_sampler_batch(cycle_moves(warm), 5)
        print("%d moves of warm-up in %.1f s" % (warm, time.time() - t0), flush=True)
    state0 = s.gpu_vect_frags.copy_from_gpu().soa17()
    frags = cycle_moves(n_moves)
    t0 = time.time()
    res = s.step_sampler_batch(frags, 5)
    dt = time.time() - t0
    st = s.ctx.batch_stats()
    print("%s %s: %d moves recorded in %.2f s (%.1f k moves/s), %s" % (cfg, params, n_moves, dt, n_moves / dt / 1e3, st), flush=True)
    np.savez_compressed(out, cfg=cfg, params=params, warm=warm, seed=seed, state0=state0, frags=frags, cands=np.asarray(s.last_candidates, np.int32),
                        op=res["op_sampled"].astype(np.int8), idf=res["id_f_sampled"].astype(np.int32), n_contigs=res["n_contigs"].astype(np.int32),
                        moves_per_s=n_moves / dt, batches=st["batches"])
    s.free_gpu()
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
