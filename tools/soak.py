"""Soak: many cycles over all bins at a synthetic shape (contigs merge, windows grow, pools and window buffers are regrown),
then cycles with nuisance sampling; after every cycle the maintained exact likelihood is compared with a from-scratch pass
and the genome checked for structural validity.   python tools/soak.py [cfg3] [cycles] [nuisance_cycles] [moves_per_cycle]"""
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np

from instagraal_amd import hip_lib, synth
from instagraal_amd.sampler import sampler as hip_sampler

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ncycles = int(sys.argv[3]) if len(sys.argv) > 3 else 1
per = int(sys.argv[4]) if len(sys.argv) > 4 else 0
prob = synth.make_problem(*synth.CONFIGS[cfg])
s = hip_sampler(**prob.sampler_kwargs(), device_id=0, coo=(prob.coo_row, prob.coo_col, prob.coo_cnt))
s.set_param_simu(prob.params)
s.bins = np.arange(1.0, 60.0, 1.0)
s.eval_likelihood_init()
np.random.seed(0)
n = per if per > 0 else prob.n_frags


def check(tag, dt, moves):
    sums, ints = s.ctx.debug_globals()
    _, _, limbs = s.ctx.full_likelihood(0)
    ok = [int(x) for x in sums[:5]] == [int(x) for x in limbs[:5]]
    st = s.gpu_vect_frags.copy_from_gpu()
    ids, cnt = np.unique(st.id_c, return_counts=True)
    valid = bool(np.all(st.l_cont == cnt[np.searchsorted(ids, st.id_c)]))
    print("%-12s %7d moves in %6.2f s = %7.0f moves/s; contigs %6d (longest %5d bins); maintained == from scratch: %s; lengths consistent: %s; scratch %.1f GB" % (
        tag, moves, dt, moves / dt, len(ids), cnt.max(), ok, valid, sum(s.ctx.scratch_bytes()) / 1e9), flush=True)
    assert ok and valid


for c in range(cycles):
    frags = np.random.permutation(prob.n_frags)[:n]
    t0 = time.perf_counter()
    res = s.step_sampler_batch(frags, 5)
    check("cycle %d" % c, time.perf_counter() - t0, n)
for c in range(ncycles):
    frags = np.random.permutation(prob.n_frags)[:n]
    t0 = time.perf_counter()
    w0 = s.ctx.debug_nuis_wait()
    res, tup = s.step_sampler_nuisance_batch(frags, 5, s.dt, c * n, ncycles * n)
    check("nuisance %d" % c, time.perf_counter() - t0, n)
    print("             host per move: " + ", ".join("%s %.0f us" % (k, 1e6 * v / n) for k, v in s.nuis_profile.items()) +
          "; of the library call %.0f us waiting for the device" % (1e6 * (s.ctx.debug_nuis_wait() - w0) / n))
    print("             screened pass %s" % (s.ctx.debug_nuis_screen_stats(),))
    print("             histogram tier %s" % (s.ctx.debug_nuis_hist_stats(),))
    print("             batches %s" % (s.ctx.batch_stats(),))
    print("             accepted %.2f of the nuisance steps; parameters %s" % (np.mean([q[6] for q in tup]), [float(s.param_simu[k][0]) for k in ("fact", "slope", "d_max", "v_inter")]))
print("soak ok")
