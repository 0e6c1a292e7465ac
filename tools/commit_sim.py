"""What a commit rule other than "stop at the first conflict" would buy -- measured on recorded trajectories,
without a GPU.

Input: tools/record_moves.py's recordings of the HIP batch path (per move: focal bin, candidate list, winner).  The genome is replayed
with the oracle's operators (oracle/ig_oracle_ops.c through OracleSampler.test_copy_struct: CL:2094-2151), which gives, per move in
sequential order,

    R   the contigs it READS   (those of the focal bin and of its <= 5 candidates: everything its scores depend on)
    W   the contigs it WRITES  (ids before and after, of every bin whose state changed) -- empty for the ~90 % of the moves whose
        winner leaves the genome alone

A slot scored when `t0` moves were committed is STALE at its turn iff a move in [t0, its turn) wrote a contig of its R.  Rules:

    (i)   today's: batches of W slots scored against one state, committed in order up to the first stale slot, the rest re-scored at
          the head of the next batch (width 1.5 x a moving average of the moves the last batches got through);
    (ii)  out of order (VERDICT's proposal): a batch of W; slot j commits if no contig of its R was written -- or MAY be written: R of an
          unresolved predecessor -- earlier in the batch; blocked slots go to the head of the next batch;
    (iii) in order over a WINDOW of scored slots: slots stay scored across launches; a launch re-scores the stale front slot, every
          slot known to be stale, and fills the window up with fresh slots; decisions strictly in order up to the first stale slot
          (all the live scalars, the stale insert flags and the zero-score rule stay exactly what they are today).

For each: slots scored per committed move, moves per launch chain, and moves/s under the cost model of a chain RE-MEASURED in round 5 on
today's code (bench.py at IG_BATCH_W = 24 / 32 / 48 / 64: 45.4 / 44.8 / 42.2 / 38.4 k moves/s, i.e. 521 / 900 / 1 172 us per batch of 24 / 48 / 64
slots; profiles/r05b_cfg3_W24_kernel_stats.csv, r05b_cfg3_W48_kernel_stats.csv: every kernel of a batch grows 1.4 - 1.8 x from 24 to 48
slots): a + b * slots scored + c * moves decided with a = 142 us, b = 15 us per slot scored, c = 1 us per decision.  (Round 4's DESIGN
took 360 of a batch's 521 us for latency chains that "do not grow with W": they do.)  The model reproduces the recorded run -- rule (i) at
W = 24 gives the recording's own number of batches and its rate to 1 %.

    python tools/commit_sim.py RECORDING.npz [...] [--out TABLE.txt]
"""
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np

A_US, B_US, C_US = 142.0, 15.0, 1.0


def replay(path, cache=True):
    """-> (R: list of frozensets, W: list of frozensets (empty: the move left the genome alone)) in sequential order"""
    side = path[:-4] + "_sets.npz"
    if cache and os.path.exists(side):
        z = np.load(side, allow_pickle=True)
        return list(z["R"]), list(z["W"])
    from oracle import oracle_lib as ol
    from oracle.oracle_lib import FRAG_FIELDS, FragStruct
    from oracle.sampler_oracle import N_INSERT_BLOCKS, N_TMP, LIST_SIZE, OracleSampler

    ol.build()
    z = np.load(path, allow_pickle=False)
    state0, frags, cands, op, idf = z["state0"], z["frags"], z["cands"], z["op"], z["idf"]
    N = state0.shape[1]
    o = OracleSampler.__new__(OracleSampler)  # (only what test_copy_struct touches: no contacts, no distributions)
    o.lib = ol.lib()
    o.n_new_frags = np.int32(N)
    o.gpu_vect_frags = FragStruct(N, {k: state0[i] for i, k in enumerate(FRAG_FIELDS)})
    o.gpu_id_contigs = np.array(state0[2], np.int32)
    o.collector_gpu_vect_frags = [FragStruct(N) for _ in range(N_TMP)]
    o.pop_gpu_vect_frags = FragStruct(N)
    o.pop_gpu_id_contigs = o.gpu_id_contigs.copy()
    o.trans1_gpu_vect_frags = FragStruct(N)
    o.trans1_gpu_id_contigs = o.gpu_id_contigs.copy()
    o.trans2_gpu_vect_frags = FragStruct(N)
    o.trans2_gpu_id_contigs = o.gpu_id_contigs.copy()
    o.gpu_list_valid_insert = np.zeros(N_INSERT_BLOCKS * 2, np.int32)
    o.gpu_list_bounds = np.array(LIST_SIZE[:N_INSERT_BLOCKS], np.int32)
    o.gpu_list_f_upstream = np.zeros(N_INSERT_BLOCKS, np.int32)
    o.gpu_list_f_downstream = np.zeros(N_INSERT_BLOCKS, np.int32)
    g = o.gpu_vect_frags
    key = ("pos", "id_c", "ori", "l_cont", "circ")
    R, W = [], []
    t0 = time.time()
    for t in range(len(frags)):
        a, b, m = int(frags[t]), int(idf[t]), int(op[t])
        cs = [int(c) for c in cands[t] if c >= 0]
        R.append(frozenset([int(g.id_c[a])] + [int(g.id_c[c]) for c in cs]))
        ca, cb = int(g.id_c[a]), int(g.id_c[b])
        members = np.nonzero((g.id_c == ca) | (g.id_c == cb))[0]
        before = [getattr(g, k)[members].copy() for k in key]
        o.test_copy_struct(a, b, m, np.int32(g.id_c.max()))
        changed = any(not np.array_equal(x, getattr(g, k)[members]) for x, k in zip(before, key))
        if changed:
            W.append(frozenset([ca, cb] + [int(x) for x in np.unique(g.id_c[members])]))
        else:
            W.append(frozenset())
        if (t + 1) % 5000 == 0:
            print("  replayed %d moves (%.0f s), %d contigs, %.1f %% changed the genome" % (
                t + 1, time.time() - t0, len(np.unique(g.id_c)), 100.0 * sum(1 for w in W if w) / len(W)), file=sys.stderr, flush=True)
    if cache:
        np.savez_compressed(side, R=np.array(R, dtype=object), W=np.array(W, dtype=object))
    return R, W


class Writes:
    """contig id -> the times it was written (ascending); stale(R, t0, t1): was a contig of R written by a move in [t0, t1)?"""

    def __init__(self, W):
        self.when = {}
        for t, w in enumerate(W):
            for c in w:
                self.when.setdefault(c, []).append(t)

    def stale(self, R, t0, t1):
        import bisect

        for c in R:
            ts = self.when.get(c)
            if ts:
                i = bisect.bisect_left(ts, t0)
                if i < len(ts) and ts[i] < t1:
                    return True
        return False


def rate(chains, slots, moves):
    us = A_US * chains + B_US * slots + C_US * moves
    return moves / us * 1e6


def rule_today(R, W, Wmax, adaptive=True):
    wr, n = Writes(W), len(R)
    done = chains = slots = 0
    ema = float(Wmax)
    while done < n:
        w_now = min(max(2, min(Wmax, int(1.5 * ema + 1.5))) if adaptive else Wmax, n - done)
        k = 0
        while k < w_now and not wr.stale(R[done + k], done, done + k):
            k += 1
        k = max(k, 1)
        chains += 1
        slots += w_now
        if w_now == (max(2, min(Wmax, int(1.5 * ema + 1.5))) if adaptive else Wmax):
            ema = 0.6 * ema + 0.4 * (min(2.0 * w_now, float(Wmax)) if k >= w_now else float(k))
        done += k
    return chains, slots, n


def rule_out_of_order(R, W, Wmax):
    """(ii): returns also the share of the slots of a batch that commit"""
    n = len(R)
    queue = list(range(min(Wmax, n)))
    nxt = len(queue)
    chains = slots = committed_total = 0
    while queue:
        chains += 1
        slots += len(queue)
        written, maybe, blocked = set(), set(), []
        for m in queue:  # in sequential order; every slot of the batch was scored against the state at the batch's start
            if (R[m] & written) or (R[m] & maybe):
                blocked.append(m)
                maybe |= R[m]
            else:
                written |= W[m]
                committed_total += 1
        if len(blocked) == len(queue):  # (cannot happen: the first slot of a batch is never blocked)
            raise RuntimeError("no progress")
        fresh = list(range(nxt, min(n, nxt + Wmax - len(blocked))))
        nxt += len(fresh)
        queue = blocked + fresh
    return chains, slots, n


def rule_window(R, W, Wmax):
    """(iii) in order over a window of scored slots"""
    wr, n = Writes(W), len(R)
    scored_at = {}
    done = chains = slots = 0
    while done < n:
        # a launch: re-score every slot of the window that is stale NOW (the front one is, unless this is the first launch), fill up
        hi = min(n, done + Wmax)
        k_scored = 0
        for m in range(done, hi):
            if m not in scored_at or wr.stale(R[m], scored_at[m], done):
                scored_at[m] = done
                k_scored += 1
        chains += 1
        slots += k_scored
        m = done
        while m < hi and not wr.stale(R[m], scored_at[m], m):
            scored_at.pop(m, None)
            m += 1
        if m == done:
            raise RuntimeError("no progress")
        done = m
    return chains, slots, n


def main(argv):
    out = None
    paths = []
    it = iter(argv)
    for a in it:
        if a == "--out":
            out = next(it)
        else:
            paths.append(a)
    lines = ["# tools/commit_sim.py: commit rules replayed on recorded trajectories of the HIP batch path (tools/record_moves.py, one MI355X);",
             "# cost of a launch chain: %.0f us + %.1f us per slot scored + %.1f us per move decided (re-measured in round 5: bench.py at" % (A_US, B_US, C_US),
             "# IG_BATCH_W = 24 / 48 / 64 takes 521 / 900 / 1 172 us per batch -- every kernel of a batch grows with its slots, profiles/r05b_*)",
             "# rule (i) today: in-order, stop at the first stale slot, batch re-scored behind it; (ii) out of order: a slot commits unless a",
             "# contig it reads was written or may be written (read set of a blocked predecessor) earlier in the batch; (iii) in order over a",
             "# window of scored slots: stale slots re-scored, the window filled up, decisions strictly in order"]
    for p in paths:
        z = np.load(p, allow_pickle=False)
        R, W = replay(p)
        n = len(R)
        ch = sum(1 for w in W if w)
        lines.append("")
        lines.append("%s (%s, %s parameters, %d moves behind %d): %.1f %% of the moves change the genome; recorded at %.1f k moves/s in %d batches" % (
            os.path.basename(p), str(z["cfg"]), str(z["params"]), n, int(z["warm"]), 100.0 * ch / n, float(z["moves_per_s"]) / 1e3, int(z["batches"])))
        lines.append("  rule   W    launches  slots scored / move  moves / launch  share of scored slots that commit  modelled k moves/s")
        for name, fn in (("(i)", rule_today), ("(ii)", rule_out_of_order), ("(iii)", rule_window)):
            for Wm in (24, 48, 96, 192):
                c, s, m = fn(R, W, Wm)
                lines.append("  %-5s %4d  %8d  %19.3f  %14.1f  %33.3f  %18.1f" % (name, Wm, c, s / m, m / c, m / s, rate(c, s, m) / 1e3))
        print("\n".join(lines[-14:]), flush=True)
    if out:
        open(out, "w").write("\n".join(lines) + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
