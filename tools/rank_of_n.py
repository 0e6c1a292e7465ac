"""What ONE rank of an N-GPU BatchRunner does per batch, measured on the one GPU there is.

multi_gpu.BatchRunner splits the SLOTS of a speculative batch over the ranks: a rank gathers every slot's windows, builds /
slices / screens / scores its own W / N slots, the slot-major score records are all-gathered, and every rank runs the same
decide + apply step.  Here rank 0's share is run for N = 1, 2, 4, 8 on one MI355X with the other ranks' records REPLAYED: per
batch the whole batch is scored once (untimed) and its record buffer kept; then the timed pass scores rank 0's slots only, the
other slots' records are copied into place on the device (what the all-gather would deliver; the collective itself -- ~20-30 us
on xGMI -- is NOT in the figure) and the batch is committed.  The trajectory is the one-GPU trajectory for every N (identical
records, identical decisions), so the per-batch times compare like for like: time(N) is an N-GPU run's period minus its
collective, time(N) - (time(1) - time(N)) / (N - 1) ... its serial part.

    python tools/rank_of_n.py [cfg3] [batches] [width]
"""
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
import torch

from instagraal_amd import hip_lib, synth
from instagraal_amd.multi_gpu import _DevBytes
from instagraal_amd.sampler import sampler as hip_sampler

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n_batches = int(sys.argv[2]) if len(sys.argv) > 2 else 60
W = int(sys.argv[3]) if len(sys.argv) > 3 else 24
prob = synth.make_problem(*synth.CONFIGS[cfg])
n_moves = (n_batches + 8) * W


def run(world):
    s = hip_sampler(**prob.sampler_kwargs(), device_id=0, coo=(prob.coo_row, prob.coo_col, prob.coo_cnt))
    s.set_param_simu(prob.params)
    s.eval_likelihood_init()
    np.random.seed(0)
    frags = np.resize(np.random.permutation(prob.n_frags), n_moves).astype(np.int32)
    cands = s.draw_candidates(frags, 5)
    stream = torch.cuda.Stream()
    s.ctx.set_stream(stream.cuda_stream)
    times, moves = [], 0
    with torch.cuda.stream(stream):
        s.ctx.batch_upload(frags, cands, W)
        ptr, nbytes = s.ctx.batch_records()
        rec = torch.as_tensor(_DevBytes(ptr, nbytes * W), device="cuda")
        assert rec.data_ptr() == ptr
        done, k = 0, 0
        while done < n_moves and k < n_batches + 8:
            w_now = min(W, n_moves - done)
            per = -(-w_now // world)
            b, e = 0, min(per, w_now)  # rank 0's slots
            if world > 1:  # the whole batch once, its records kept: what the other ranks would send
                s.ctx.batch_score(done, w_now, 0, w_now)
                keep = rec.clone()
            stream.synchronize()
            t0 = time.perf_counter()
            s.ctx.batch_score(done, w_now, b, e)
            if world > 1:
                rec[e * nbytes:w_now * nbytes].copy_(keep[e * nbytes:w_now * nbytes])
            got = s.ctx.batch_commit(done, w_now)
            stream.synchronize()
            dt = time.perf_counter() - t0
            if k >= 8:
                times.append(dt)
                moves += got
            done += got
            k += 1
    res = s.ctx.batch_results(done)
    state = s.gpu_vect_frags.copy_from_gpu().soa17().tobytes()
    s.free_gpu()
    return np.array(times) * 1e6, moves, res[["o", "op_sampled", "id_f_sampled", "n_contigs"]].tobytes(), state


base = None
print("%s, batches of %d slots, rank 0 of N with the other ranks' records replayed (collective not included):" % (cfg, W))
t1 = None
for world in (1, 2, 4, 8):
    t, moves, recs, state = run(world)
    if base is None:
        base, t1 = (recs, state), t.mean()
    assert (recs, state) == base, "the replayed run left the one-GPU trajectory"
    serial = (t.mean() - t1 / world) / (1.0 - 1.0 / world) if world > 1 else float("nan")
    print("  N = %d: %6.1f us per batch (median %6.1f, p10 %6.1f, p90 %6.1f), %4.1f moves per batch -> %5.2f x one GPU before the collective%s" % (
        world, t.mean(), np.median(t), np.percentile(t, 10), np.percentile(t, 90), moves / len(t), t1 / t.mean(),
        "" if world == 1 else "; serial part (Amdahl fit) %.0f us, with a 25 us all-gather: %.2f x" % (serial, t1 / (t.mean() + 25.0))))
