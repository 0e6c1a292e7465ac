#!/usr/bin/env python3
"""MCMC moves/s of the instaGRAAL scoring path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the path over one BATCH of moves: ``--moves-per-step`` (default 128; until round 5 it was 24, the
speculative batch width of rounds 1 - 4 -- since the window rule a launch chain covers ~35 moves and the driver's
``--steps 20 --warmup 5`` timed 480 moves in 10 ms, a quarter of BASELINE.md section 3's protocol of >= 2 000 moves behind
>= 200) consecutive ``step_sampler`` calls (CL:1401-1465), each one the candidate
draw (return_neighbours, CL:3103-3141, on numpy's generator stream), one focal bin, <= 5 partner bins, up to 5 x 24
candidate genomes scored, argmax applied.  ``value`` is MOVES per second (BASELINE.json's metric) = K x moves-per-step /
elapsed; ``ms_per_step`` is per batch; ``config.moves_timed`` says how many moves the timed region held.  (Rounds 1 and
early 2 counted one move as a step, so the driver's ``--steps 20`` timed ONE launch of 20 moves, 0.7 ms: a sample of
the call overhead, not of the path.)  Workload at N=1: BASELINE.json configs[2] (synthetic 50 k bins / 50 M contacts, the configuration the metric is
quoted on).  The problem (contacts, genome state, jump distributions) is resident before the timed region; the timed
region covers the K x moves-per-step consecutive moves END TO END, issued as ONE ``step_sampler_batch`` call (the way
``full_em`` issues a cycle): draw of the candidate lists (host thread, ahead of the launches), their
H2D, every kernel, the D2H of the result records.  ``ig_step_batch_draw`` scores the moves W at a time against one
state and commits them in order on the device ("speculative batches": results identical to K single calls,
tests/test_hip_sampler.py), so one launch of the dominant kernel covers several moves.

N > 1: launched by ``torch.distributed.run`` (the driver's way), or -- when WORLD_SIZE is not set -- this script starts
the N workers itself (child processes, before anything touches the GPU) and relays rank 0's line.  One process per GPU,
RCCL.  ``--runner batch`` (default): every rank holds the full problem, the slots of each speculative batch are split
over the ranks, ONE all-gather of the slot-major score records (exact int64 sums) per batch, identical commit on every
rank.  ``--runner sharded`` (BASELINE config 4's wording): contact rows split over the ranks, one all-reduce of the
exact partial sums per move.  Both are bit-identical to one GPU for any N ("strong" scaling: one chain, split N ways).

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def prob_name(prob):
    return "%dk bins / %.0fM contacts" % (prob.n_frags // 1000, prob.n_contacts / 1e6)


def cpu_baseline(prob, frags, cands, budget_s=45.0, max_moves=72, hip_res=None, params=None):
    """The oracle (a CPU port of the reference ALGORITHM: full-N genome rewrites, full-Z slice scans) timed on a bounded
    sample of the same workload -- the first moves of the same trajectory -- on one thread and on all host cores (OpenMP
    over the contact-length loops: slice scans, full likelihood).  hip_res: the HIP path's records of the SAME moves from the same
    initial state (the first moves of the run): what the oracle computes while it is timed is compared with them, move by move.
    The oracle is the checker and the baseline here, never the thing measured as the product."""
    from oracle import oracle_lib as ol
    from oracle.sampler_oracle import OracleSampler

    ol.build()
    t0 = time.time()
    ol.set_threads(1)
    s = OracleSampler(**prob.sampler_kwargs(), mode=ol.MODE_DET)
    s.set_param_simu(prob.params if params is None else params)
    s.eval_likelihood_init()
    log("[cpu_baseline] oracle set-up %.1fs" % (time.time() - t0))
    ncores = os.cpu_count() or 1
    rates, n_done, spent = {}, 0, {}
    n_checked = n_same = 0
    # one thread (a few moves: ~1.3 s each at cfg3) and 16 threads (the figure worth quoting: OpenMP over the contact-length loops and
    # the slice kernels' blocks; >= 60 moves, SURVEY 8(d) asks for a sample that is not a handful).  No all-cores leg: the rest of a
    # move is serial in the reference's algorithm, 256 threads read slower than one (round 4: 0.49 against 0.81 moves/s)
    par = min(16, ncores)
    plan = sorted({1, par})
    quota = {t: (6 if t == 1 else max_moves) for t in plan}
    share = {t: budget_s * (0.2 if (t == 1 and par > 1) else 0.8) for t in plan}
    n_avail = min(len(frags), len(cands))
    done_by = {}
    for threads in plan:
        ol.set_threads(threads)
        n = 0
        t0 = time.time()
        while n < quota[threads] and n_done < n_avail:
            c = [int(x) for x in cands[n_done] if x >= 0]
            b = s.step_sampler(int(frags[n_done]), len(c), s.dt, candidates=c)
            if hip_res is not None and n_done < len(hip_res):
                r = hip_res[n_done]
                same = (float(r["o"]), float(r["dist"]), int(r["op_sampled"]), int(r["id_f_sampled"]), int(r["n_contigs"])) == (
                    float(b[0]), float(b[1]), int(b[2]), int(b[3]), int(b[5]))
                n_checked += 1
                n_same += int(same)
                if not same:
                    log("[cpu_baseline] move %d DIFFERS: HIP %r, oracle %r" % (n_done, r, b))
            n += 1
            n_done += 1
            if time.time() - t0 > share[threads] and n >= 2:
                break
        spent[threads] = time.time() - t0
        rates[threads] = n / spent[threads]
        done_by[threads] = n
    ol.set_threads(1)
    best = max(rates, key=lambda k: rates[k])
    return dict(value=rates[best], unit="moves/s", cores=best, kind="port", value_1_thread=rates[1],
                checked_against_hip={"moves": n_checked, "identical": n_same,
                                     "what": "o, dist, op_sampled, id_f_sampled, n_contigs of the HIP batch path's records of the same moves"},
                value_by_threads={str(k): v for k, v in rates.items()}, host_cores=ncores,
                sample="the first %d moves of the same seeded trajectory on %s, oracle DET mode: " % (n_done, prob_name(prob)) +
                       ", ".join("%d moves in %.1f s on %d thread%s" % (done_by[k], spent[k], k, "s" if k > 1 else "") for k in plan))


def nuisance_rate(s, prob, n_moves, n_neighbours, settle=0):
    """moves/s of the reference's loop for cycles > 4 (IG:241-252): one step_sampler + one step_nuisance_parameters
    (the likelihood of all contacts under test parameters, CL:2961-3051) per move -- 95 of the default 100 cycles.
    settle: that many (move, step) pairs first, untimed: the proposals of the first few hundred steps are accepted by the
    thousands of log-likelihood units (35 % of them); a run spends its 4.75 M steps in the regime behind that."""
    s.bins = np.arange(1.0, 60.0, 1.0)
    frags = np.resize(np.random.permutation(prob.n_frags), n_moves + 10 + settle)
    run = getattr(s, "step_sampler_nuisance_batch", None)
    if run is not None:
        run(frags[:10 + settle], n_neighbours, s.dt, 0, n_moves)
        b0, c0 = s.ctx.batch_stats(), s.ctx.debug_nuis_chain_stats()
        t0 = time.perf_counter()
        out = run(frags[10 + settle:], n_neighbours, s.dt, 0, n_moves)
        dt = time.perf_counter() - t0
        b1, c1 = s.ctx.batch_stats(), s.ctx.debug_nuis_chain_stats()
        extra = {"batches_per_1000_moves": 1000.0 * (b1["batches"] - b0["batches"]) / n_moves,
                 "pairs_decided_in_chains_pct": 100.0 * (c1["pairs"] - c0["pairs"]) / n_moves,
                 "chain_calls": c1["calls"] - c0["calls"], "chain_segments": c1["segments"] - c0["segments"],
                 "chain_ends": {k: c1["ends"][k] - c0["ends"][k] for k in c1["ends"]}}
        return n_moves / dt, float(np.mean([q[6] for q in out[1]])), "step_sampler_nuisance_batch", extra
    for t, f in enumerate(frags[:10]):
        s.step_sampler(int(f), n_neighbours, s.dt)
        s.step_nuisance_parameters(s.dt, t, n_moves)
    acc = 0
    t0 = time.perf_counter()
    for t, f in enumerate(frags[10:]):
        s.step_sampler(int(f), n_neighbours, s.dt)
        acc += s.step_nuisance_parameters(s.dt, t, n_moves)[6]
    dt = time.perf_counter() - t0
    return n_moves / dt, acc / float(n_moves), "step_sampler + step_nuisance_parameters per move", {}


def secondary_shape(make_sampler, prob, n_neighbours, bomb, n_warm, n_moves, what):
    """moves/s of the batch path on another genome of the headline's size -- where an assembly starts (`--bomb`: every bin its own contig,
    IG:206) or ends (cfg3_late: ~20 long contigs; the reference merges contigs, KA:3367-3693 / 2724-2975, and its own GPU test ends with
    15 - 45 of them, tests/test_instagraal_gpu.py:126-340): secondary fields of the line, same call as the headline's timed region"""
    import torch

    t0 = time.time()
    s2 = make_sampler(prob)
    if bomb:
        s2.bomb_the_genome()
    up_s = time.time() - t0
    fr = np.resize(np.random.permutation(prob.n_frags), n_warm + n_moves).astype(np.int32)
    if n_warm:
        s2.step_sampler_batch(fr[:n_warm], n_neighbours)
    b0 = s2.ctx.batch_stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = s2.step_sampler_batch(fr[n_warm:], n_neighbours)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    b1 = s2.ctx.batch_stats()
    sums, _ = s2.ctx.debug_globals()
    _, _, limbs = s2.ctx.full_likelihood(0)
    nb = max(b1["batches"] - b0["batches"], 1)
    out = {"what": what, "moves_per_s": n_moves / dt, "moves": n_moves, "moves_warmup": n_warm, "launch_chains": nb, "moves_per_launch_chain": n_moves / nb,
           "us_per_launch_chain": 1e6 * dt / nb, "n_contigs_first_last": [int(res["n_contigs"][0]), int(res["n_contigs"][-1])],
           "moves_changing_the_genome_distance_pct": 100.0 * float(np.mean(np.diff(np.concatenate([[res["dist"][0]], res["dist"]])) != 0)),
           "bytes_min_per_move": float(res["bytes_min"].mean()), "slice_contacts_per_move": float(res["n_slice"].mean()),
           "maintained_likelihood_exact": bool(int(sums[0]) == int(limbs[0]) and int(sums[1]) == int(limbs[1])), "setup_s": up_s}
    s2.free_gpu()
    return out


def spawn_workers(a):
    """--gpus N without a launcher: start N ranks as children of this process (which never touches the GPU)."""
    import socket

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("[bench] starting %d ranks: %s" % (a.gpus, " ".join(cmd)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.run(cmd, env=env)
    sys.exit(p.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed steps of --moves-per-step moves")
    ap.add_argument("--warmup", type=int, default=5, help="untimed steps before them")
    ap.add_argument("--moves-per-step", type=int, default=128,
                    help="moves (step_sampler calls) per step: 128, so that the driver's --steps 20 --warmup 5 times 2 560 moves behind 640 "
                         "(BASELINE.md section 3: >= 2 000 timed after >= 200)")
    ap.add_argument("--config", default="cfg3", help="synthetic shape: cfg2 | cfg3 | cfg5 | small | tiny")
    ap.add_argument("--neighbours", type=int, default=5)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--runner", default="batch", choices=("batch", "sharded", "replicas"),
                    help="N > 1: what is split over the ranks (batch / sharded: ONE chain, bit-identical to one GPU, 'strong'); replicas: "
                         "N independent chains, one per GPU, no data-path collective ('weak': the aggregate BASELINE's >= 6x asks for)")
    ap.add_argument("--params", default="synthetic", choices=("synthetic", "settled"),
                    help="P(s) parameters of the timed moves: the synthetic set of BASELINE.md section 3 (slope -1.5: the headline), or the set a "
                         "nuisance chain settles into on this data (slope -0.53, d_max 2.9e6 kb: synth.settled_params)")
    ap.add_argument("--settled-batches", type=int, default=40, help="batches of the default line's config.settled_parameters sample (0: skip)")
    ap.add_argument("--timer-sampling", type=int, default=0, help="hipEvent pairs around every N-th launch of the scoring kernels (0: every launch of a short run, every 4th of a long one)")
    ap.add_argument("--reference-loop-moves", type=int, default=200, help="step_sampler calls of config.reference_loop, one per move (0: skip)")
    ap.add_argument("--late-moves", type=int, default=1024, help="timed moves of config.late_assembly (synth cfg3_late: the headline's size in ~20 contigs) "
                                                                 "and of config.bombed_start (the headline's genome after bomb_the_genome); 0: skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=45.0, help="seconds of the oracle's timed sample (a fifth on one thread, the rest on 16)")
    ap.add_argument("--nuisance-moves", type=int, default=150, help="moves of the nuisance-on loop timed after the run (0: skip)")
    ap.add_argument("--nuisance-settle", type=int, default=2400, help="untimed (move, step) pairs in front of a second, settled measurement (0: skip)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_workers(a)  # does not return

    import torch

    from instagraal_amd import synth
    from instagraal_amd.sampler import sampler as hip_sampler

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE %d" % (a.gpus, world))
    # IG_BENCH_ONE_DEVICE=1 (test rigs with a single GPU): every rank uses cuda:0 and the collectives go through gloo --
    # exercises the multi-process protocol end to end, not a performance configuration
    one_device = os.environ.get("IG_BENCH_ONE_DEVICE") == "1"
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if one_device:
        local_rank = 0
    elif local_rank >= n_dev:
        raise SystemExit("bench.py: rank %d has no device (%d visible, --gpus %d): refusing to share a GPU between ranks" % (rank, n_dev, world))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            # one process per GPU: no two ranks on the same device
            ids = [None] * world
            dist.all_gather_object(ids, (os.uname().nodename, torch.cuda.current_device()))
            if len(set(ids)) != world:
                raise SystemExit("bench.py: ranks share a device: %r" % (ids,))

    t0 = time.time()
    prob = synth.make_problem(*synth.CONFIGS[a.config])
    log("[rank %d] problem %s generated in %.1fs" % (rank, prob_name(prob), time.time() - t0))
    t0 = time.time()
    kw = prob.sampler_kwargs()
    s = hip_sampler(**kw, device_id=local_rank, coo=(prob.coo_row, prob.coo_col, prob.coo_cnt))
    s.set_param_simu(prob.params if a.params == "synthetic" else synth.settled_params(prob.params))
    s.eval_likelihood_init()
    log("[rank %d] uploaded + initial likelihood %.6f in %.1fs" % (rank, float(s.curr_likelihood_on_nz[0]), time.time() - t0))

    # trajectory: a shuffled cycle prefix; the candidate lists are drawn INSIDE the timed region, as step_sampler does
    replicas = world > 1 and a.runner == "replicas"
    np.random.seed(a.seed + (rank if replicas else 0))  # replicas: every rank its own chain
    mps = max(1, a.moves_per_step)
    n_warm, n_moves = a.warmup * mps, a.steps * mps
    n_total = n_warm + n_moves
    order = np.arange(prob.n_frags)
    np.random.shuffle(order)
    frags = np.resize(order, n_total).astype(np.int32)

    runner = None
    if world > 1 and not replicas:
        from instagraal_amd.multi_gpu import BatchRunner, ShardedRunner

        runner = (BatchRunner if a.runner == "batch" else ShardedRunner)(s.ctx, rank, world, dist=dist)

    def run(fr):
        """K complete step_sampler calls: draw + score + apply"""
        if runner is None:
            res = s.step_sampler_batch(fr, a.neighbours)
            return res, s.last_candidates
        cands = s.draw_candidates(fr, a.neighbours)  # every rank draws the same lists from the same generator state
        return runner.run(fr, cands), cands

    first_res = first_cands = None  # the run's first moves (from the initial state): the cpu_baseline leg replays them on the oracle
    if n_warm:
        first_res, first_cands = run(frags[:n_warm])
        first_res, first_cands = first_res[:96].copy(), np.array(first_cands[:96])
    # hipEvent pairs around the two scoring kernels (k_screen, k_score_list) on the library's stream, every 4th launch: an event
    # record between two kernels of a stream costs ~6 us of idle queue, four of them per batch were 4 % of the timed region
    # (a short call -- the driver's 480 moves are ~15 launches of the dominant kernel, of 9 to 48 slots each since round 5's window rule --
    # times every launch: four samples of launches that different read 0.093 - 0.114 for the same kernel; --timer-sampling N fixes it)
    t_every = a.timer_sampling if a.timer_sampling > 0 else (1 if n_moves <= 1200 else 4)
    s.ctx.set_timer_sampling(t_every)
    s.ctx.reset_timers(1 | (((1 << 2) | (1 << 10)) << 1))
    batches_before = s.ctx.batch_stats()["batches"] if (world == 1 or replicas) else 0
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res, cands = run(frags[n_warm:])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if one_device else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    score_ms, n_launch = s.ctx.kernel_time_ms("score")
    screen_ms, n_screen = s.ctx.kernel_time_ms("screen")
    s.ctx.reset_timers(0)
    s.ctx.set_timer_sampling(1)
    n_timed = int(n_screen if (n_screen and screen_ms >= score_ms) else n_launch)
    screened = s.ctx.debug_screen_stats() if (world == 1 or replicas) else None
    # the dominant kernel: the screening pass when the batches are scored in two tiers, else the exact kernel
    dom_name, dom_ms, dom_key = ("k_screen", screen_ms, "k_screen") if (n_screen and screen_ms >= score_ms) else ("k_score_list", score_ms, "k_score_list")
    if n_screen and screen_ms >= score_ms:
        n_launch = n_screen
    bstats = s.ctx.batch_stats() if (world == 1 or replicas) else None
    # launches of the dominant kernel in the timed region: one per scored batch (the timed ones are a sample of them)
    n_launch = (bstats["batches"] - batches_before) if (bstats is not None and bstats["batches"] > batches_before) else t_every * int(n_launch)

    # the draw alone, for the record (it ran on a host thread next to the launches above)
    st = np.random.get_state()
    t0 = time.perf_counter()
    s.draw_candidates(frags[n_warm:], a.neighbours)
    draw_us = 1e6 * (time.perf_counter() - t0) / n_moves
    np.random.set_state(st)

    # self-check: the incrementally maintained exact likelihood equals a from-scratch recomputation
    exact_ok = None
    if world == 1 or a.runner in ("batch", "replicas"):
        sums, _ = s.ctx.debug_globals()
        _, _, limbs = s.ctx.full_likelihood(0)
        exact_ok = bool(int(sums[0]) == int(limbs[0]) and int(sums[1]) == int(limbs[1]))

    # the reference's own loop shape (instagraal.py:221-228): ONE step_sampler call per bin, the result read before the next -- what a
    # caller that keeps that loop gets (sampler.step_sampler -> ig_step_draw), behind the timed moves on the same genome
    ref_loop = None
    if rank == 0 and world == 1 and a.reference_loop_moves > 0:
        try:
            fr = np.random.permutation(prob.n_frags)[: 2 * a.reference_loop_moves]
            ref_loop = {"what": "one sampler.step_sampler call per move, its 6-tuple read before the next call (median of %d calls each)" % a.reference_loop_moves}
            keep0 = s.keep_all_scores
            for key, keep in (("us_per_call", False), ("us_per_call_with_all_scores", True)):
                s.keep_all_scores = keep
                ts = []
                for f in fr[:a.reference_loop_moves] if not keep else fr[a.reference_loop_moves:]:
                    t0 = time.perf_counter()
                    s.step_sampler(int(f), a.neighbours, s.dt)
                    ts.append(time.perf_counter() - t0)
                ref_loop[key] = 1e6 * float(np.median(ts))
                ref_loop[key.replace("us_per_call", "moves_per_s")] = len(ts) / float(np.sum(ts))
            s.keep_all_scores = keep0
        except Exception as e:  # a diagnostic next to the headline, never instead of it
            ref_loop = {"us_per_call": None, "error": repr(e)}

    nuis = None
    if rank == 0 and world == 1 and a.nuisance_moves > 0:
        try:
            rate, acc, how, extra = nuisance_rate(s, prob, a.nuisance_moves, a.neighbours)
            nuis = {"moves_per_s": rate, "accept_rate": acc, "moves": a.nuisance_moves, "loop": how, **extra,
                    "regime": "the first %d (move, step) pairs behind the timed moves: large proposals, decisive tests" % a.nuisance_moves}
            if a.nuisance_settle > 0:
                n_set = 16 * a.nuisance_moves  # (2 400 pairs: a 600-pair sample read 15.5 - 17.3 k from run to run)
                rate2, acc2, _, extra2 = nuisance_rate(s, prob, n_set, a.neighbours, settle=a.nuisance_settle)
                nuis["settled"] = {"moves_per_s": rate2, "accept_rate": acc2, "moves": n_set, "after_steps": a.nuisance_moves + 10 + a.nuisance_settle, **extra2,
                                   "regime": "behind %d more (move, step) pairs: where a run spends its 95 cycles" % a.nuisance_settle}
            st = getattr(s.ctx, "debug_nuis_screen_stats", None)
            if st is not None:
                nuis["screened_pass"] = st()
            st = getattr(s.ctx, "debug_nuis_hist_stats", None)
            if st is not None:  # its first tier: the Metropolis test from a histogram of the cis contacts' distances
                nuis["screened_pass"]["histogram_tier"] = st()
        except Exception as e:  # a diagnostic next to the headline, never instead of it
            nuis = {"moves_per_s": None, "error": repr(e)}

    # the plain batch under the parameters a nuisance chain settles into (95 of the default 100 cycles run there): the same call,
    # the model's parameters replaced (maintained sums recomputed), on the genome as the measurements above left it
    settled = None
    if rank == 0 and world == 1 and a.settled_batches > 0 and a.params == "synthetic":
        try:
            sp = synth.settled_params(prob.params)
            s.set_param_simu(sp)
            nb = a.settled_batches
            fr = np.resize(np.random.permutation(prob.n_frags), (nb + 8) * mps).astype(np.int32)
            s.step_sampler_batch(fr[:8 * mps], a.neighbours)
            sc0 = s.ctx.debug_screen_stats()
            b0 = s.ctx.batch_stats()
            s.ctx.set_timer_sampling(4)
            s.ctx.reset_timers(1 | (((1 << 2) | (1 << 10)) << 1))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rs = s.step_sampler_batch(fr[8 * mps:], a.neighbours)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ex_ms, _ = s.ctx.kernel_time_ms("score")
            sc_ms, _ = s.ctx.kernel_time_ms("screen")
            s.ctx.reset_timers(0)
            s.ctx.set_timer_sampling(1)
            sc1 = s.ctx.debug_screen_stats()
            b1 = s.ctx.batch_stats()
            sums, _ = s.ctx.debug_globals()
            _, _, limbs = s.ctx.full_likelihood(0)
            nbat = max(b1["batches"] - b0["batches"], 1)
            settled = {"moves_per_s": nb * mps / dt, "k_screen_ms": sc_ms, "exact_ms": ex_ms,
                       "exact_columns_pct": 100.0 * (sc1[3] - sc0[3]) / max(sc1[2] - sc0[2], 1),
                       "exact_terms_pct": 100.0 * (sc1[5] - sc0[5]) / max(sc1[4] - sc0[4], 1),
                       "moves": nb * mps, "batches": nbat, "moves_per_batch": nb * mps / nbat, "ms_per_batch": 1e3 * dt / nbat,
                       "maintained_likelihood_exact": bool(int(sums[0]) == int(limbs[0]) and int(sums[1]) == int(limbs[1])),
                       "parameters": {k: float(sp[k]) for k in ("slope", "fact", "d_max", "v_inter")}}
            del rs
        except Exception as e:  # a diagnostic next to the headline, never instead of it
            settled = {"moves_per_s": None, "error": repr(e)}

    # the two ends of an assembly at the headline's size (VERDICT r5 missing 2): secondary fields, never the headline
    late = bombed = None
    if rank == 0 and world == 1 and a.late_moves > 0 and a.config == "cfg3":
        def mk(pr):
            s2 = hip_sampler(**pr.sampler_kwargs(), device_id=local_rank, coo=(pr.coo_row, pr.coo_col, pr.coo_cnt))
            s2.set_param_simu(pr.params)
            s2.eval_likelihood_init()
            return s2
        try:
            bombed = secondary_shape(mk, prob, a.neighbours, True, 256, 4 * a.late_moves,
                                     "cfg3 after bomb_the_genome (IG:206): 50 k singleton contigs, where a --bomb run starts")
        except Exception as e:  # a diagnostic next to the headline, never instead of it
            bombed = {"moves_per_s": None, "error": repr(e)}
        try:
            t0 = time.time()
            prob_late = synth.make_problem(*synth.CONFIGS["cfg3_late"])
            log("[late] problem cfg3_late generated in %.1fs" % (time.time() - t0))
            late = secondary_shape(mk, prob_late, a.neighbours, False, 128, a.late_moves,
                                   "synth cfg3_late: %s in %d contigs (longest %d bins): where an assembly ends" % (
                                       prob_name(prob_late), int(np.unique(prob_late.S_o_A_frags["id_c"]).size),
                                       int(np.bincount(prob_late.S_o_A_frags["id_c"]).max())))
            del prob_late
        except Exception as e:
            late = {"moves_per_s": None, "error": repr(e)}

    if rank == 0:
        # algorithmic bytes of one launch of the dominant kernel = sum of the per-move B_min of the moves it scored
        # (committed moves only: a slot that had to be re-scored is work, not algorithmic traffic)
        n_launch = max(int(n_launch), 1)
        split = world if (world > 1 and not replicas) else 1  # N > 1, one chain: a rank scores 1/N of the slots (or of the contact rows) of a launch
        bytes_min = float(res["bytes_min"].sum()) / n_launch / split
        n_evals = float(res["n_evals"].sum()) / n_launch / split
        achieved = bytes_min / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # the reference algorithm's streaming model B_ref: (1 + C) 12 Z + C (12 S + 24 20 M + 50 136 N) + 2 20 M
        C = float(res["n_candidates"].mean())
        S = float(res["n_slice"].mean()) / max(C, 1.0)
        b_ref = (1 + C) * 12.0 * prob.n_contacts + C * (12.0 * S + 24 * 20.0 * prob.n_sub_frags + 50 * 136.0 * prob.n_frags) + 40.0 * prob.n_sub_frags
        # HBM bytes per launch of the dominant kernel: rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE,
        # separate passes) of THIS workload, committed under profiles/ -- replayed from the file named below, not measured
        # in this run (the counters need the profiler around the process)
        traffic = valu_busy = traffic_src = batch_traffic = prof_mpl = None
        traffic_scale = 1.0
        try:
            import glob

            pmc = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_%s_pmc_traffic.json" % a.config)))
            if pmc and world == 1:
                allk = json.load(open(pmc[-1]))
                prof = allk[dom_key]
                # REPLAYED, not measured here: per launch of the profiled run, scaled to this run's moves per launch chain where the
                # summary says how many moves its launches covered (a launch's traffic follows the slots it scores)
                prof_mpl = allk.get("moves_per_launch")
                traffic_scale = (n_moves / max(int(n_launch), 1)) / float(prof_mpl) if prof_mpl else 1.0
                traffic = float(prof["traffic_bytes_per_launch"]) * traffic_scale
                valu_busy = float(prof["VALUBusy_pct"]) / 100.0  # fraction of cycles the VALUs issue
                traffic_src = "profiles/" + os.path.basename(pmc[-1])
                # the whole batch: every kernel of it that was counted (the slice lists' round trip -- written by k_slice, read
                # by the scoring kernels -- is in neither B_min nor the dominant kernel's figure)
                # every kernel of a batch that was counted (k_predict runs twice per batch; k_full_nz_tiled and k_tail are not part of one)
                per_k = {k: float(v["traffic_bytes_per_launch"]) * traffic_scale * float(v.get("launches_per_chain", 1.0)) for k, v in allk.items()
                         if isinstance(v, dict) and "traffic_bytes_per_launch" in v and k not in ("k_full_nz_tiled", "k_tail", "k_commit_batch")}
                batch_traffic = {"bytes_per_batch": sum(per_k.values()), "by_kernel": per_k}
        except Exception:
            traffic = valu_busy = traffic_src = batch_traffic = None
        # instruction-issue ceiling of the dominant kernel (tools/microbench/ubench.hip on this part, profiles/r03_microbench.txt):
        # plain v_fma_f32 sustains 115 lane-ops per CU per clock of the 128 a SIMD-32 x 4 CU can issue (v_pk_fma_f32: the same
        # FLOPs, half the instructions), v_log_f32 / v_exp_f32 a quarter of that.  A screened term = 24 full-rate + 2
        # quarter-rate instructions = 32 issue slots; an exact f64 term ~75 instructions at half rate or less.
        slots_per_term = 32.0 if dom_name == "k_screen" else 150.0
        valu_peak = 256 * 128 * 2.4e9
        out = {
            "metric": "MCMC moves/s (accepted+rejected) at fixed n_frags x nnz",
            "value": (world if replicas else 1) * n_moves / elapsed,  # replicas: every rank ran its own n_moves
            "unit": "moves/s",
            "n_gpus": dist.get_world_size() if dist is not None else 1,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True,
            "scaling": "weak" if replicas else "strong",
            "vs_baseline": None,
            # BASELINE.md section 3's protocol read literally -- "wall-clock around step_sampler", ONE call per move, the reference's loop
            # (IG:221-228) unchanged but for the import: what a drop-in-by-import caller gets; `value` needs the caller to hand a cycle's
            # bins to step_sampler_batch, results identical
            "value_unchanged_caller": None if not ref_loop else ref_loop.get("moves_per_s"),
            "value_unchanged_caller_unit": "moves/s, one sampler.step_sampler call per move (median %s us per call)" % (
                "%.0f" % ref_loop["us_per_call"] if ref_loop and ref_loop.get("us_per_call") else "?"),
            "dtype": "f32 screening tier with a rigorous bound (every column) + f64 terms (f32 inputs) for the contenders / exact i64 fixed-point sums",
            "data": "synthetic",
            "config": {"workload": "synthetic Hi-C %s (%d sub-frags), level 4, %d neighbours, nuisance sampling off%s" % (
                prob_name(prob), prob.n_sub_frags, a.neighbours, "" if a.params == "synthetic" else ", P(s) parameters of a settled chain"), "name": a.config, "seed": a.seed,
                "step": "one batch of %d moves (step_sampler calls)" % mps, "moves_per_step": mps, "moves_timed": n_moves,
                "moves_warmup": n_warm,
                "parallelism": "1 GPU" if world == 1 else ("replicas only: %d independent chains, one per GPU, no data-path collective" % world) if replicas else (
                    "batch slots split over %d ranks, all-gather of score records" % world if a.runner == "batch" else
                    "contact rows split over %d ranks, all-reduce of exact partial sums per move" % world),
                "timed_region": "candidate draw (host thread) + H2D + kernels + D2H of the result records",
                "draw_us_per_move_alone": draw_us,
                "candidate_genomes_per_s": float(res["n_candidates"].sum()) * 24 / elapsed,  # 24 mutations per candidate: built and screened
                "columns_scored_exactly_per_s": None if not screened else screened[3] / max(screened[2], 1) * float(res["n_candidates"].sum()) * 24 / elapsed,
                "term_evals_per_move": float(res["n_evals"].mean()), "moves_per_launch": n_moves / n_launch,
                "batches": bstats, "maintained_likelihood_exact": exact_ok,
                "reference_loop": ref_loop,
                "nuisance_on": nuis,
                "settled_parameters": settled,
                "late_assembly": late,
                "bombed_start": bombed,
                "parameters": a.params,
                "nuisance_on_moves_per_s": None if nuis is None else nuis.get("moves_per_s"),
                "reference_equivalent_GBps": b_ref * (n_moves / elapsed) / 1e9,
                "B_ref_bytes_per_move": b_ref},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": traffic, "traffic_source": traffic_src, "valu_busy_profiled": valu_busy, "kernel": dom_name,
                         "traffic_replayed": None if traffic is None else {
                             "replayed": True, "profiled_moves_per_launch": prof_mpl, "this_run_moves_per_launch": n_moves / n_launch,
                             "scale_applied": traffic_scale,
                             "note": "HBM bytes come from separate rocprofv3 --pmc passes of this workload (the file in traffic_source), per launch "
                                     "of that run, scaled by this run's moves per launch chain / the profiled run's; not counted in this process"},
                         "traffic_over_algorithmic": None if not traffic else traffic / max(bytes_min, 1.0),
                         "batch_traffic": batch_traffic,
                         "batch": {"bytes": bytes_min, "ms": 1e3 * elapsed / n_launch, "frac": bytes_min / (elapsed / n_launch) / 8e12,
                                   "traffic_over_algorithmic": None if not batch_traffic else batch_traffic["bytes_per_batch"] / max(bytes_min, 1.0),
                                   "note": "the WHOLE batch (every kernel, launch gaps, the host's turnaround): algorithmic bytes of the moves of "
                                           "a launch / the wall time per scored batch -- `frac` above is the dominant kernel alone"},
                         "valu": {"bound": "valu issue", "achieved": (n_evals / (dom_ms * 1e-3)) * slots_per_term if dom_ms > 0 else 0.0,
                                  "peak": valu_peak, "unit": "lane-ops/s", "issue_slots_per_term": slots_per_term,
                                  "frac": ((n_evals / (dom_ms * 1e-3)) * slots_per_term / valu_peak) if dom_ms > 0 else 0.0,
                                  "measured_sustained_fma_lane_ops_per_s": 70.6e12,
                                  "note": "the dominant kernel works on an L2-resident set: its ceiling is instruction issue (and the LDS "
                                          "gathers behind it), not HBM; peak = 256 CUs x 128 lanes x 2.4 GHz (SIMD-32, one wave64 "
                                          "instruction per 2 clocks), microbenchmark in profiles/r03_microbench.txt"},
                         "avg_launch_ms": dom_ms, "launches": int(n_launch), "launches_timed": n_timed,
                         "algorithmic_bytes_per_launch": bytes_min,
                         "term_evals_per_launch": n_evals,
                         "term_evals_per_s": (n_evals / (dom_ms * 1e-3)) if dom_ms > 0 else 0.0,
                         "exact_kernel_avg_launch_ms": score_ms, "screen_kernel_avg_launch_ms": screen_ms,
                         "columns_screened_vs_scored_exactly": None if not screened else [screened[2], screened[3]],
                         "note": "B_min = sum_c[12 S_c + 20 m_c U + 8 U] + 68 n_touched per move (SURVEY 8(d)), summed over the moves "
                                 "of a launch.  Batches are scored in two tiers: every (contact, column) term through the float "
                                 "screening kernel (the dominant one: VALU-issue bound on an L2-resident working set, not HBM bound), "
                                 "the exact f64 kernel only for the columns that can still win: DESIGN.md section 4.3-4.4"},
        }
        mismatch = None
        if not a.no_cpu_baseline and world == 1:
            try:
                if first_res is None:
                    first_res, first_cands = res[:96], np.array(cands[:96])
                out["cpu_baseline"] = cpu_baseline(prob, frags, first_cands, a.cpu_budget, hip_res=first_res,
                                                   params=None if a.params == "synthetic" else synth.settled_params(prob.params))
                cb = out["cpu_baseline"].get("checked_against_hip")
                if cb and cb["identical"] != cb["moves"]:
                    mismatch = "bench.py: the oracle disagrees with the HIP path on %d of %d moves" % (cb["moves"] - cb["identical"], cb["moves"])
                    out["error"] = mismatch
            except Exception as e:  # the baseline is a report, never the product path
                out["cpu_baseline"] = {"value": None, "unit": "moves/s", "cores": 1, "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)
        if mismatch:  # (the line is out -- with the mismatch in it -- before the run ends in an error)
            raise SystemExit(mismatch)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
