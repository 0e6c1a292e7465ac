"""Long trajectories of the HIP path against the ORACLE (CPU restatement of the reference, DET arithmetic) on an EVOLVED genome.

The oracle-vs-HIP evidence of rounds 1 - 4 was 12 - 160 moves per shape, from the initial state, on short contigs.  Here a
whole run of the reference's loop (instagraal.py:196-262: per cycle one shuffle of the bins, one step_sampler per bin -- and with `nuis`
one step_nuisance_parameters behind every move) goes

    HIP     through sampler.step_sampler_batch / sampler.step_sampler_nuisance_batch, CHUNK moves per call
    ORACLE  through OracleSampler.step_sampler (+ step_nuisance_parameters), one call per move        (CL:1401-1465, 2961-3051)

from the same seed; compared: the 6-tuple of every move, the 7 scalars of every nuisance step, the 17 x N genome and the stale insert
flags behind every chunk, numpy's generator state at the end.  Rings, windows past 1 024 and 4 096 sub-fragments, pool growth and
windowed same-contig winners all come up on the way (`summary` in the result says how often).

    python tools/long_oracle.py CFG MOVES [--bomb] [--nuis] [--seed S]                      both, live (a GPU box)
    python tools/long_oracle.py CFG MOVES [...] --record FILE.npz                          HIP only          (a GPU box)
    python tools/long_oracle.py --check FILE.npz                                           the oracle against a recording (no GPU)
"""
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np

warnings.filterwarnings("ignore", category=RuntimeWarning)

CHUNK = 250
N_NB = 5


def _problem(cfg):
    from instagraal_amd import synth

    return synth.make_problem(*synth.CONFIGS[cfg])


def _cycles(n_frags, n_moves):
    """the bins of the run in the reference's order: the SAME array shuffled once more per cycle (IG:207, 213)"""
    frags = np.arange(0, n_frags)
    out = []
    while sum(len(x) for x in out) < n_moves:
        np.random.shuffle(frags)
        out.append(frags.copy())
    return np.concatenate(out)[:n_moves].astype(np.int32)


def _row(r):
    return (float(r["o"]), float(r["dist"]), int(r["op_sampled"]), int(r["id_f_sampled"]), float(np.float32(r["mean_len"])), int(r["n_contigs"]))


def _nuis_row(q):
    return tuple(float(np.ravel(x)[0]) for x in q[:7])


def run_hip(cfg, n_moves, bomb=False, nuis=False, seed=1, params=None, hist=None):
    """-> dict: records (n, 6), nuis (n, 7) or empty, states / flags per chunk, generator state, summary"""
    from instagraal_amd import hip_lib
    from instagraal_amd.sampler import sampler as hip_sampler

    prob = _problem(cfg)
    if hist is not None:
        hip_lib.set_nuis_hist(hist)
    try:
        np.random.seed(seed)
        s = hip_sampler(**prob.sampler_kwargs(), device_id=0)
        s.set_param_simu(prob.params if params is None else params)
        s.eval_likelihood_init()
        if bomb:
            s.bomb_the_genome()
        # (the draws of the cycles' shuffles interleave with the moves' draws in the reference: one shuffle at the head of each cycle)
        N = prob.n_frags
        frags_all = np.arange(0, N)
        rec, nrec, states, flags = [], [], [], []
        longest, rings, done = 0, 0, 0
        t0 = time.time()
        while done < n_moves:
            np.random.shuffle(frags_all)
            cyc = frags_all[: min(N, n_moves - done)].astype(np.int32)
            for i0 in range(0, len(cyc), CHUNK):
                part = cyc[i0:i0 + CHUNK]
                if nuis:
                    res, tuples = s.step_sampler_nuisance_batch(part, N_NB, s.dt, done + i0, n_moves)
                    nrec += [_nuis_row(q) for q in tuples]
                else:
                    res = s.step_sampler_batch(part, N_NB)
                rec += [_row(r) for r in res]
                g = s.gpu_vect_frags.copy_from_gpu()
                states.append(g.soa17())
                flags.append(np.array(s.ctx.valid_insert(), np.int32))
                longest = max(longest, int(g.sub_l_cont.max()))
                rings += int((g.circ != 0).any())
            done += len(cyc)
        st = np.random.get_state()
        stats = dict(s.ctx.batch_stats(), pool_retries=s.ctx.debug_pool_retries())
        summary = dict(cfg=cfg, moves=n_moves, bomb=int(bomb), nuis=int(nuis), seed=seed, seconds_hip=round(time.time() - t0, 2),
                       longest_contig_subfrags=longest, chunks_with_a_ring=rings, n_contigs_end=int(rec[-1][5]),
                       batches=int(stats["batches"]), one_move_tails=int(stats["one_move_tails"]), pool_retries=int(stats["pool_retries"]))
        if nuis:
            cs = s.ctx.debug_nuis_chain_stats()
            summary.update(chain_pairs=cs["pairs"], chain_calls=cs["calls"], accepted=int(sum(q[6] for q in nrec)))
        s.free_gpu()
        return dict(records=np.array(rec, np.float64).reshape(-1, 6), nuis=np.array(nrec, np.float64).reshape(-1, 7),
                    states=np.array(states, np.int32), flags=np.array(flags, np.int32), rng_key=np.array(st[1], np.uint32), rng_pos=int(st[2]),
                    rng_gauss=(int(st[3]), float(st[4])), summary=summary)
    finally:
        if hist is not None:
            hip_lib.set_nuis_hist(1)


def run_oracle(cfg, n_moves, bomb=False, nuis=False, seed=1, params=None, expect=None, threads=0, progress=False):
    """the same run through the oracle, compared with `expect` (run_hip's result) as it goes -> first difference or None"""
    from oracle import oracle_lib as ol
    from oracle.sampler_oracle import OracleSampler

    ol.build()
    prob = _problem(cfg)
    ol.set_threads(threads if threads else min(16, os.cpu_count() or 1))
    try:
        np.random.seed(seed)
        o = OracleSampler(**prob.sampler_kwargs(), mode=ol.MODE_DET)
        o.set_param_simu(prob.params if params is None else params)
        o.eval_likelihood_init()
        if bomb:
            o.bomb_the_genome()
        N = prob.n_frags
        frags_all = np.arange(0, N)
        done, k = 0, 0
        t0 = time.time()
        while done < n_moves:
            np.random.shuffle(frags_all)
            cyc = frags_all[: min(N, n_moves - done)].astype(np.int32)
            for i, f in enumerate(cyc):
                t = done + i
                cands = [c for c in o.return_neighbours(int(f), N_NB) if c != int(f)]  # (quirk Q13: sampler._clean)
                b = o.step_sampler(int(f), N_NB, o.dt, candidates=cands)
                row = (float(b[0]), float(b[1]), int(b[2]), int(b[3]), float(np.float32(b[4])), int(b[5]))
                if tuple(expect["records"][t]) != row:
                    return "move %d (bin %d): HIP %r / oracle %r" % (t, int(f), tuple(expect["records"][t]), row)
                if nuis:
                    q = _nuis_row(o.step_nuisance_parameters(o.dt, t, n_moves))
                    if tuple(expect["nuis"][t]) != q:
                        return "nuisance step %d: HIP %r / oracle %r" % (t, tuple(expect["nuis"][t]), q)
                if (i + 1) % CHUNK == 0 or i + 1 == len(cyc):
                    if not np.array_equal(expect["states"][k], o.gpu_vect_frags.soa17()):
                        bad = np.argwhere(expect["states"][k] != o.gpu_vect_frags.soa17())
                        return "genome after move %d: %d entries differ, first (field %d, bin %d)" % (t, len(bad), bad[0][0], bad[0][1])
                    if not np.array_equal(expect["flags"][k], np.array(o.gpu_list_valid_insert, np.int32)):
                        return "stale insert flags after move %d" % t
                    k += 1
                    if progress:
                        print("  ... %d moves agree (%.0f s, %d contigs)" % (t + 1, time.time() - t0, int(o.n_contigs)), flush=True)
            done += len(cyc)
        st = np.random.get_state()
        if not (np.array_equal(expect["rng_key"], np.array(st[1], np.uint32)) and int(expect["rng_pos"]) == int(st[2])):
            return "generator state differs at the end of the run"
        return None
    finally:
        ol.set_threads(1)


def behind_hip_moves(cfg, n_hip, n_oracle, seed=11, threads=0):
    """`n_oracle` moves against the oracle BEHIND `n_hip` moves of the HIP path: the evolved genome is handed to the oracle (state, contig
    ids as ig_download_state renumbers them -- CL:2715-2881 --, stale insert flags) instead of being replayed there.  For shapes where an
    oracle move takes a second (the headline shape).  -> (first difference or None, summary)"""
    from instagraal_amd.sampler import sampler as hip_sampler
    from oracle import oracle_lib as ol
    from oracle.oracle_lib import FRAG_FIELDS
    from oracle.sampler_oracle import OracleSampler

    ol.build()
    prob = _problem(cfg)
    np.random.seed(seed)
    s = hip_sampler(**prob.sampler_kwargs(), device_id=0, coo=(prob.coo_row, prob.coo_col, prob.coo_cnt))
    s.set_param_simu(prob.params)
    s.eval_likelihood_init()
    frags = _cycles(prob.n_frags, n_hip + n_oracle)
    t0 = time.time()
    res0 = s.step_sampler_batch(frags[:n_hip], N_NB)
    t_hip = time.time() - t0
    g = s.gpu_vect_frags.copy_from_gpu()
    changed = int(np.count_nonzero((g.l_cont != prob.S_o_A_frags["l_cont"]) | (g.pos != prob.S_o_A_frags["pos"]) | (g.ori != 1)))
    ol.set_threads(threads if threads else min(16, os.cpu_count() or 1))
    try:
        o = OracleSampler(**prob.sampler_kwargs(), mode=ol.MODE_DET)
        o.set_param_simu(prob.params)
        for k in FRAG_FIELDS:
            getattr(o.gpu_vect_frags, k)[:] = getattr(g, k)
        o.gpu_id_contigs[:] = g.id_c
        o.gpu_list_valid_insert[:] = np.array(s.ctx.valid_insert(), np.int32)
        st = np.random.get_state()
        res = s.step_sampler_batch(frags[n_hip:], N_NB)
        after = np.random.get_state()
        np.random.set_state(st)
        diff = None
        for t, (f, r) in enumerate(zip(frags[n_hip:], res)):
            b = o.step_sampler(int(f), N_NB, o.dt)
            row = (float(b[0]), float(b[1]), int(b[2]), int(b[3]), float(np.float32(b[4])), int(b[5]))
            if _row(r) != row:
                diff = "move %d behind %d (bin %d): HIP %r / oracle %r" % (t, n_hip, int(f), _row(r), row)
                break
        if diff is None:
            if not (np.array_equal(np.random.get_state()[1], after[1]) and np.random.get_state()[2] == after[2]):
                diff = "generator state differs"
            elif not np.array_equal(s.gpu_vect_frags.copy_from_gpu().soa17(), o.gpu_vect_frags.soa17()):
                diff = "genome differs behind the oracle's moves"
            elif not np.array_equal(np.array(s.ctx.valid_insert(), np.int32), np.array(o.gpu_list_valid_insert, np.int32)):
                diff = "stale insert flags differ"
    finally:
        ol.set_threads(1)
    summary = dict(cfg=cfg, hip_moves=n_hip, oracle_moves=n_oracle, seconds_hip=round(t_hip, 2), bins_moved_by_hip=changed,
                   n_contigs=int(res0[-1]["n_contigs"]), longest_contig_subfrags=int(g.sub_l_cont.max()))
    s.free_gpu()
    return diff, summary


def main(argv):
    record = check = None
    bomb = nuis = False
    seed = 1
    pos = []
    it = iter(argv)
    for a in it:
        if a == "--record":
            record = next(it)
        elif a == "--check":
            check = next(it)
        elif a == "--bomb":
            bomb = True
        elif a == "--nuis":
            nuis = True
        elif a == "--seed":
            seed = int(next(it))
        else:
            pos.append(a)
    t0 = time.time()
    if check:
        z = np.load(check, allow_pickle=False)
        exp = dict(records=z["records"], nuis=z["nuis_rows"], states=z["states"], flags=z["flags"], rng_key=z["rng_key"], rng_pos=int(z["rng_pos"]))
        cfg, n_moves, bomb, nuis, seed = str(z["cfg"]), int(z["moves"]), bool(z["bomb"]), bool(z["nuis_on"]), int(z["seed"])
        print("recorded:", str(z["summary"]))
        diff = run_oracle(cfg, n_moves, bomb, nuis, seed, expect=exp, progress=True)
        print("%s %d moves%s%s against the oracle: %s (%.0f s)" % (cfg, n_moves, " --bomb" if bomb else "", " + nuisance steps" if nuis else "",
                                                                   "DIFF " + diff if diff else "identical", time.time() - t0))
        return 1 if diff else 0
    cfg, n_moves = pos[0], int(pos[1])
    hist = 2 if nuis else None  # (the histogram tier whatever its cost model says: chains need it)
    h = run_hip(cfg, n_moves, bomb, nuis, seed, hist=hist)
    print("HIP:", h["summary"], flush=True)
    if record:
        np.savez_compressed(record, records=h["records"], nuis_rows=h["nuis"], states=h["states"], flags=h["flags"], rng_key=h["rng_key"],
                            rng_pos=np.int64(h["rng_pos"]), cfg=cfg, moves=n_moves, bomb=int(bomb), nuis_on=int(nuis), seed=seed,
                            summary=repr(h["summary"]))
        print("recorded in %s (%.1f MB)" % (record, os.path.getsize(record) / 1e6))
        return 0
    diff = run_oracle(cfg, n_moves, bomb, nuis, seed, expect=h, progress=True)
    print("%s %d moves against the oracle: %s (%.0f s)" % (cfg, n_moves, "DIFF " + diff if diff else "identical", time.time() - t0))
    return 1 if diff else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
