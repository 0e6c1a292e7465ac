"""Randomised comparison of the HIP path with the ORACLE (the CPU restatement of the reference, oracle/; DET arithmetic).

tools/fuzz_batches.py / fuzz_chains.py / fuzz_ranks.py compare the library with itself -- they catch a path that disagrees with its
siblings, not a mistake the paths share.  Here every case is one run of moves through

    HIP     sampler.step_sampler_batch (the window of scored slots, two-tier scoring, the candidate draw inside the call: the path bench.py
            times; some cases at another window width, without the window, with a small slice pool, one reference-shaped step_sampler call
            per move, or with a nuisance step behind every move: step_sampler_nuisance_batch)
    ORACLE  OracleSampler(DET).step_sampler, one move at a time, on numpy's generator (CL:1401-1465, 3103-3141; KA:485-607, 612-3693)

from the same seeded state and the two must agree on: the 6-tuple of every move (score, genome distance, winner, partner, mean contig
length, contig count), the 17 x N genome state and the stale insert flags at every checkpoint, and numpy's generator state at the end.

The generator follows fuzz_batches.py's (bins, contacts per bin, contig lengths 2 .. 1 000 bins, cis share, counts x 9 / x 60,
1 .. 16 neighbours, synthetic / settled / random parameters) plus --bomb'ed starts and small slice pools, sized so that the oracle -- a
full-N, full-Z algorithm like the reference -- finishes a case in seconds.

The two halves need different machines' strengths (one MI355X for seconds, host cores for minutes), so they can be run apart:

    python tools/fuzz_oracle.py CASES [FIRST_SEED]                      both, live (a GPU box)
    python tools/fuzz_oracle.py CASES [FIRST_SEED] --record FILE.npz    HIP only: what it returned, per case      (a GPU box)
    python tools/fuzz_oracle.py --check FILE.npz [--jobs J]             the oracle against a recording            (no GPU needed)
    ... --how nuis | batch | width | nowindow | one                      every case in that mode (with --check: as recorded)
    ... --long K                                                         K times the moves (and the oracle's budget) per case
    ... --big                                                            5 000 - 20 000 bins, up to 1.5 M contacts, mostly small pools (record
                                                                         on a GPU box, check on host cores: an oracle move takes 0.1 - 0.3 s)
"""
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np

warnings.filterwarnings("ignore", category=RuntimeWarning)

CHECK_EVERY = 50


def _digest(a):
    import hashlib

    return np.frombuffer(hashlib.blake2b(np.ascontiguousarray(a).tobytes(), digest_size=8).digest(), np.uint64)[0]


def _state_equal(expect, k, arr):
    """checkpoint k of a run: the full genome where the recording holds it (live runs; the last checkpoint of a file), else its digest"""
    if "states" in expect and k < len(expect["states"]):
        return np.array_equal(expect["states"][k], arr)
    if k == len(expect["state_digests"]) - 1 and "state_last" in expect:
        return np.array_equal(expect["state_last"], arr)
    return expect["state_digests"][k] == _digest(arr)


FORCE_HOW = None  # (--how: every case of a run in one mode)
BIG = 0   # (--big: problems of 5 000 - 20 000 bins; the oracle half is meant for --check on host cores)
LONG = 1  # (--long K: K times the moves per case, and K times the oracle's budget -- runaway parameters take hundreds of accepted steps)


def make_case(seed):
    """-> (problem, parameters, description dict) -- deterministic in the seed, no GPU touched"""
    from instagraal_amd import synth

    r = np.random.RandomState(seed)
    n_frags = int(r.choice([60, 150, 300, 700, 1200, 2000]))
    per = int(r.choice([8, 30, 100, 400]))
    per = max(8, min(per, 600000 // n_frags, n_frags))
    if BIG:  # (--big: 5 000 - 20 000 bins, up to 1.5 M contacts -- lists that overflow small pools and the exact kernel's first grid)
        n_frags = int(r.choice([5000, 10000, 20000]))
        per = max(15, min(int(r.choice([30, 75, 150])), 1500000 // n_frags))
    mean_len = int(r.choice([2, 4, 15, 50, 200, 1000]))
    mean_len = min(mean_len, max(2, n_frags // 3))
    prob = synth.make_problem(n_frags, n_frags * per, 7000 + seed, mean_len, cis_frac=float(r.choice([0.3, 0.6, 0.8, 0.95])))
    scale = int(r.choice([1, 1, 1, 9, 60]))
    if scale > 1:
        import copy

        import scipy.sparse as sp
        prob = copy.deepcopy(prob)
        cnt = prob.coo_cnt.copy()
        cnt[::3] *= scale
        prob.coo_cnt = cnt
        M = prob.n_sub_frags
        prob.sub_csr = sp.csr_matrix((cnt, (prob.coo_row, prob.coo_col)), shape=(M, M), dtype=np.int32)
        prob.sub_csr.sort_indices()
    kind = int(r.choice([0, 0, 1, 2, 2]))
    params = dict(prob.params)
    if kind == 1:
        params = synth.settled_params(prob.params)
    elif kind == 2:
        slope = -float(r.uniform(0.3, 2.2))
        kuhn, lm = float(prob.params["kuhn"]), float(prob.params["lm"])
        params = dict(prob.params, slope=slope, c1=float(np.float32((0.53 * np.power(lm / kuhn, slope)) * np.power(kuhn, -3))),
                      d_max=float(prob.params["d_max"]) * float(r.choice([0.3, 1.0, 30.0, 3000.0])),
                      v_inter=float(prob.params["v_inter"]) * float(r.choice([0.1, 1.0, 10.0])))
    n_nb = int(r.choice([1, 3, 5, 5, 9, 16]))
    bomb = int(r.randint(4) == 0)
    pool = int(r.choice([0, 0, 0, 3000, 30000])) if not BIG else int(r.choice([0, 3000, 3000, 30000]))
    # the default window / another width / no window (round 4's batches) / one step_sampler call per move / a nuisance step behind every
    # move (step_sampler_nuisance_batch: chains, screened tiers -- against o.step_sampler + o.step_nuisance_parameters, CL:2961-3051)
    how = str(r.choice(["batch", "batch", "width", "nowindow", "one", "nuis"]))
    if FORCE_HOW:
        how = FORCE_HOW
    width = int(r.choice([2, 3, 7, 16, 31, 40, 64])) if how in ("width", "nowindow") else 0
    # the oracle's cost per move: (1 + C) passes over all contacts, C x 24 columns over the slice (about 2 Z / contigs entries, all of
    # them late in an assembly), ~250 rewrites of the genome per candidate; moves so that a case stays within ~10 core-seconds
    Z, C = prob.n_contacts, min(n_nb, 16)
    n_ctg = max(1, n_frags // mean_len) if not bomb else n_frags
    per_move = 8e-9 * Z * (1 + C) + 45e-9 * 21 * C * min(Z, 3.0 * Z / n_ctg) + 1.5e-6 * n_frags * C + 3e-3
    if how == "nuis":
        per_move += 8e-9 * Z
    n = int(min(int(r.choice([100, 250, 500])) * LONG, max(40, 10.0 * LONG / per_move)))
    # the nuisance runs' own knobs: the histogram tier always / where its cost model says / never, chains on / off, the moves per call
    # (drawn behind everything else: the shapes of the seeds of earlier rounds stay what they were)
    hist = int(r.choice([2, 2, 1, 0]))
    chain = int(r.randint(4) != 0)
    chunk = int(r.choice([50, 50, 17, 1000])) if how == "nuis" else CHECK_EVERY
    desc = dict(seed=int(seed), n_frags=n_frags, per=per, mean_len=mean_len, scale=scale, params=kind, n=n, neighbours=n_nb, bomb=bomb,
                pool=pool, how=how, width=width)
    nwidth = int(r.choice([0, 0, 16, 24]))  # (a fixed width of a run's batches: both halves scored W slots at a time -- more overflows per batch)
    if how == "nuis":
        desc.update(hist=hist, chain=chain, chunk=chunk)
        if BIG:
            desc.update(nwidth=nwidth)
    return prob, params, desc


def _frags(prob, n):
    return np.resize(np.random.permutation(prob.n_frags), n).astype(np.int32)


def run_hip(prob, params, desc):
    """-> dict(records (n, 6), states [17 x N at every CHECK_EVERY-th move and at the end], flags [...], rng (key, pos))"""
    from instagraal_amd import hip_lib
    from instagraal_amd.sampler import sampler as hip_sampler

    os.environ.pop("IG_POOL_ENTRIES", None)
    if desc["pool"]:
        os.environ["IG_POOL_ENTRIES"] = str(desc["pool"])
    try:
        hip_lib.set_window(0 if desc["how"] == "nowindow" else (desc["width"] if desc["how"] == "width" else 48))
    except AttributeError:  # (an older build of the library under IG_DEBUG_TUNING=1 IG_HIP_LIB=...: no window rule)
        pass
    hip_lib.set_batch_width(min(desc["width"], 64) if desc["how"] == "nowindow" else 24)
    every = desc.get("chunk", CHECK_EVERY)
    if desc["how"] == "nuis":
        hip_lib.set_nuis_hist(desc.get("hist", 2))  # (2: the histogram tier whatever its cost model says -- chains need it)
        hip_lib.set_nuis_chain(desc.get("chain", 1))
        hip_lib.set_nuis_width(desc.get("nwidth", 0))
    try:
        np.random.seed(desc["seed"])
        # (one call per move: every other case the way ``simulation`` constructs the sampler -- no all_scores, the batch of one in two tiers)
        s = hip_sampler(**prob.sampler_kwargs(), device_id=0, keep_all_scores=bool(desc["seed"] & 1))
        s.set_param_simu(params)
        s.eval_likelihood_init()
        if desc["bomb"]:
            s.bomb_the_genome()
        frags = _frags(prob, desc["n"])
        rec, states, flags, nrec = [], [], [], []
        for i0 in range(0, len(frags), every):
            part = frags[i0:i0 + every]
            if desc["how"] == "nuis":
                res, tuples = s.step_sampler_nuisance_batch(part, desc["neighbours"], s.dt, i0, len(frags))
                rec += [(float(q["o"]), float(q["dist"]), int(q["op_sampled"]), int(q["id_f_sampled"]), float(np.float32(q["mean_len"])),
                         int(q["n_contigs"])) for q in res]
                nrec += [tuple(float(np.ravel(x)[0]) for x in q[:7]) for q in tuples]
            elif desc["how"] == "one":
                for f in part:
                    o, dist, op, idf, ml, nc = s.step_sampler(int(f), desc["neighbours"])
                    rec.append((float(o), float(dist), int(op), int(idf), float(ml), int(nc)))
            else:
                res = s.step_sampler_batch(part, desc["neighbours"])
                rec += [(float(q["o"]), float(q["dist"]), int(q["op_sampled"]), int(q["id_f_sampled"]), float(np.float32(q["mean_len"])),
                         int(q["n_contigs"])) for q in res]
            states.append(s.gpu_vect_frags.copy_from_gpu().soa17())
            flags.append(np.array(s.ctx.valid_insert(), np.int32))
        st = np.random.get_state()
        stats = dict(s.ctx.batch_stats(), pool_retries=s.ctx.debug_pool_retries())
        if desc["how"] == "nuis":
            stats["chain_pairs"] = s.ctx.debug_nuis_chain_stats()["pairs"]
        s.free_gpu()
        return dict(records=np.array(rec, np.float64).reshape(-1, 6), states=np.array(states, np.int32), flags=np.array(flags, np.int32),
                    nuis=np.array(nrec, np.float64).reshape(-1, 7), rng_key=np.array(st[1], np.uint32), rng_pos=int(st[2]), stats=stats)
    finally:
        os.environ.pop("IG_POOL_ENTRIES", None)
        hip_lib.set_batch_width(24)
        try:
            hip_lib.set_window(48)
        except AttributeError:
            pass
        hip_lib.set_nuis_hist(1)
        hip_lib.set_nuis_chain(1)
        hip_lib.set_nuis_width(0)


def run_oracle(prob, params, desc, threads=0, expect=None):
    """the same run through the oracle; with `expect` (run_hip's result) it stops at the first move that differs.
    -> (result dict like run_hip's, first difference or None)"""
    from oracle import oracle_lib as ol
    from oracle.sampler_oracle import OracleSampler

    ol.build()
    ol.set_threads(threads if threads else min(16, os.cpu_count() or 1))
    try:
        np.random.seed(desc["seed"])
        o = OracleSampler(**prob.sampler_kwargs(), mode=ol.MODE_DET)
        o.set_param_simu(params)
        o.eval_likelihood_init()
        if desc["bomb"]:
            o.bomb_the_genome()
        frags = _frags(prob, desc["n"])
        rec, states, flags = [], [], []
        diff = None
        for t, f in enumerate(frags):
            cands = [c for c in o.return_neighbours(int(f), desc["neighbours"]) if c != int(f)]  # (quirk Q13: sampler._clean)
            b = o.step_sampler(int(f), desc["neighbours"], o.dt, candidates=cands)
            row = (float(b[0]), float(b[1]), int(b[2]), int(b[3]), float(np.float32(b[4])), int(b[5]))
            rec.append(row)
            if expect is not None and diff is None and tuple(expect["records"][t]) != row:
                diff = "move %d (bin %d): HIP %r / oracle %r" % (t, int(f), tuple(expect["records"][t]), row)
                break
            if desc["how"] == "nuis":
                q = tuple(float(np.ravel(x)[0]) for x in o.step_nuisance_parameters(o.dt, t, len(frags))[:7])
                if expect is not None and diff is None and tuple(expect["nuis"][t]) != q:
                    diff = "nuisance step %d: HIP %r / oracle %r" % (t, tuple(expect["nuis"][t]), q)
                    break
            if (t + 1) % desc.get("chunk", CHECK_EVERY) == 0 or t + 1 == len(frags):
                k = len(states)
                states.append(o.gpu_vect_frags.soa17())
                flags.append(np.array(o.gpu_list_valid_insert, np.int32))
                if expect is not None and diff is None:
                    if not _state_equal(expect, k, states[-1]):
                        diff = "genome after move %d differs" % t
                        break
                    if not np.array_equal(expect["flags"][k], flags[-1]):
                        diff = "stale insert flags after move %d: HIP %r / oracle %r" % (t, expect["flags"][k].tolist(), flags[-1].tolist())
                        break
        st = np.random.get_state()
        out = dict(records=np.array(rec, np.float64).reshape(-1, 6), states=np.array(states, np.int32), flags=np.array(flags, np.int32),
                   rng_key=np.array(st[1], np.uint32), rng_pos=int(st[2]), n_contigs_end=int(o.n_contigs))
        if expect is not None and diff is None:
            if not (np.array_equal(expect["rng_key"], out["rng_key"]) and int(expect["rng_pos"]) == out["rng_pos"]):
                diff = "generator state differs at the end of the run"
        return out, diff
    finally:
        ol.set_threads(1)


def live_case(seed, threads=0):
    """one case on this machine (GPU + host cores) -> (description, first difference or None, seconds HIP, seconds oracle)"""
    prob, params, desc = make_case(seed)
    t0 = time.time()
    h = run_hip(prob, params, desc)
    t1 = time.time()
    _, diff = run_oracle(prob, params, desc, threads, expect=h)
    desc = dict(desc, batches=h["stats"]["batches"], pool_retries=h["stats"]["pool_retries"])
    return desc, diff, t1 - t0, time.time() - t1


def _check_one(args):
    global FORCE_HOW, LONG, BIG
    path, seed, threads = args
    z = np.load(path, allow_pickle=False)
    FORCE_HOW = (str(z["how"]) or None) if "how" in z.files else None
    LONG = int(z["long"]) if "long" in z.files else 1
    BIG = int(z["big"]) if "big" in z.files else 0
    pre = "c%d_" % seed
    exp = dict(records=z[pre + "records"], state_digests=z[pre + "state_digests"], state_last=z[pre + "state_last"], flags=z[pre + "flags"],
               nuis=z[pre + "nuis"] if pre + "nuis" in z.files else np.zeros((0, 7)), rng_key=z[pre + "rng_key"], rng_pos=int(z[pre + "rng_pos"]))
    prob, params, desc = make_case(seed)
    assert desc["n"] == len(exp["records"]), "the recording was made by another generator"
    t0 = time.time()
    out, diff = run_oracle(prob, params, desc, threads, expect=exp)
    return seed, desc, diff, time.time() - t0, out.get("n_contigs_end")


def main(argv):
    record = check = None
    jobs = 1
    pos = []
    it = iter(argv)
    for a in it:
        if a == "--record":
            record = next(it)
        elif a == "--check":
            check = next(it)
        elif a == "--jobs":
            jobs = int(next(it))
        elif a == "--how":
            global FORCE_HOW
            FORCE_HOW = next(it)
        elif a == "--long":
            global LONG
            LONG = int(next(it))
        elif a == "--big":
            global BIG
            BIG = 1
        else:
            pos.append(a)
    t00 = time.time()
    bad = 0
    if check:
        z = np.load(check, allow_pickle=False)
        seeds = [int(x) for x in z["seeds"]]
        threads = max(1, (os.cpu_count() or 1) // jobs)
        work = [(check, sd, threads) for sd in seeds]
        if jobs > 1:
            import multiprocessing as mp

            results = mp.get_context("spawn").Pool(jobs).imap_unordered(_check_one, work)
        else:
            results = map(_check_one, work)
        n_moves = 0
        for seed, desc, diff, dt, nc in results:
            n_moves += desc["n"]
            print("case %5d %s %s: contigs at the end %s  (oracle %.1f s)%s" % (seed, "DIFF" if diff else "ok  ", desc, nc, dt, "  " + diff if diff else ""),
                  flush=True)
            bad += diff is not None
        print("%d recorded cases (%d moves) against the oracle: %d differ, %.0f s" % (len(seeds), n_moves, bad, time.time() - t00))
        return 1 if bad else 0
    n_cases = int(pos[0]) if pos else 20
    seed0 = int(pos[1]) if len(pos) > 1 else 1
    store = {}
    done = []
    for seed in range(seed0, seed0 + n_cases):
        try:
            prob, params, desc = make_case(seed)
        except ValueError as ex:  # (a shape the generator has no pairs for)
            print("case %5d skipped: %s" % (seed, str(ex)[:80]), flush=True)
            continue
        try:
            t0 = time.time()
            h = run_hip(prob, params, desc)
            t1 = time.time()
            if record:
                pre = "c%d_" % seed
                for k in ("records", "flags", "rng_key", "nuis"):
                    store[pre + k] = h[k]
                store[pre + "state_digests"] = np.array([_digest(x) for x in h["states"]], np.uint64)  # (64 MiB come back from a GPU box)
                store[pre + "state_last"] = h["states"][-1]
                store[pre + "rng_pos"] = np.int64(h["rng_pos"])
                done.append(seed)
                print("case %5d recorded %s: batches %d, pool retries %d  (%.1f s)" % (seed, desc, h["stats"]["batches"], h["stats"]["pool_retries"],
                                                                                      t1 - t0), flush=True)
                continue
            _, diff = run_oracle(prob, params, desc, 0, expect=h)
            print("case %5d %s %s: batches %d, pool retries %d  (HIP %.1f s, oracle %.1f s)%s" % (
                seed, "DIFF" if diff else "ok  ", desc, h["stats"]["batches"], h["stats"]["pool_retries"], t1 - t0, time.time() - t1,
                "  " + diff if diff else ""), flush=True)
            bad += diff is not None
        except Exception as ex:  # a failed library call is a finding as well
            if "needs 1.." in str(ex):  # the uniform fallback draw returned the focal bin alone: undefined in the reference (quirk Q13)
                print("case %5d skipped (a move without a candidate: quirk Q13) %s" % (seed, desc), flush=True)
                continue
            bad += 1
            print("case %5d FAIL %s: %r" % (seed, desc, ex), flush=True)
    if record:
        store["seeds"] = np.array(done, np.int64)
        store["how"] = np.array(FORCE_HOW or "")
        store["long"] = np.int64(LONG)
        store["big"] = np.int64(BIG)
        np.savez_compressed(record, **store)
        print("%d cases recorded in %s (%.1f MB), %d failed, %.0f s" % (len(done), record, os.path.getsize(record) / 1e6, bad, time.time() - t00))
    else:
        print("%d cases against the oracle, %d bad, %.0f s" % (n_cases, bad, time.time() - t00))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
