#!/usr/bin/env python3
"""MCMC moves/s of the instaGRAAL scoring path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W          one GPU
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one move = one ``step_sampler`` call (CL:1401-1465): one focal bin, <= 5 partner bins,
up to 5 x 24 candidate genomes scored, argmax applied.  Workload at N=1: BASELINE.json configs[2]
(synthetic 50 k bins / 50 M contacts, the configuration the metric is quoted on).  Inputs are
resident in HBM before the timed region; the timed region covers K consecutive moves including the
H2D of the pre-drawn candidate lists and the D2H of the K result records.  ``ig_step_batch`` scores the
moves W at a time against one state and commits them in order on the device ("speculative batches":
the results are identical to K single calls, tests/test_hip_sampler.py), so one launch of the
dominant kernel covers several moves.

Multi-GPU (N > 1): every rank holds the full problem; the slots of each speculative batch are split
over the ranks (rank r slices and scores W/N of the W moves), the slot-major score records (exact
int64 sums, ~15 KB per slot) are all-gathered over RCCL once per batch, and every rank runs the same
commit step on identical inputs.  Results are bit-identical for any N ("strong" scaling: the same
chain, split N ways); the commit step is the serial fraction.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(prob, frags, cands, budget_s=20.0, max_moves=8):
    """The oracle (a CPU port of the reference algorithm: full-N genome rewrites, full-Z slice scans)
    timed on a bounded sample of the same workload: the first moves of the same trajectory."""
    from oracle import oracle_lib as ol
    from oracle.sampler_oracle import OracleSampler

    ol.build()
    cores = 1
    t0 = time.time()
    s = OracleSampler(**prob.sampler_kwargs(), mode=ol.MODE_DET)
    s.set_param_simu(prob.params)
    s.eval_likelihood_init()
    log("[cpu_baseline] oracle set-up %.1fs" % (time.time() - t0))
    n = 0
    t0 = time.time()
    while n < min(max_moves, len(frags)):
        c = [int(x) for x in cands[n] if x >= 0]
        s.step_sampler(int(frags[n]), len(c), s.dt, candidates=c)
        n += 1
        if time.time() - t0 > budget_s:
            break
    dt = time.time() - t0
    return dict(value=n / dt, unit="moves/s", cores=cores, kind="port",
                sample="first %d moves of the same seeded trajectory on %s, oracle DET mode, %.1f s" % (n, prob_name(prob), dt))


def prob_name(prob):
    return "%dk bins / %.0fM contacts" % (prob.n_frags // 1000, prob.n_contacts / 1e6)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--config", default="cfg3", help="synthetic shape: cfg2 | cfg3 | cfg5 | small | tiny")
    ap.add_argument("--neighbours", type=int, default=5)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    a = ap.parse_args()

    import torch

    from instagraal_amd import hip_lib, synth
    from instagraal_amd.sampler import sampler as hip_sampler

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        log("WARNING: --gpus %d but WORLD_SIZE %d" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # IG_BENCH_ONE_DEVICE=1 (test rigs with a single GPU): every rank uses cuda:0 and the collectives go through gloo --
    # exercises the multi-process protocol end to end, not a performance configuration
    one_device = os.environ.get("IG_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    t0 = time.time()
    prob = synth.make_problem(*synth.CONFIGS[a.config])
    log("[rank %d] problem %s generated in %.1fs" % (rank, prob_name(prob), time.time() - t0))
    t0 = time.time()
    kw = prob.sampler_kwargs()
    s = hip_sampler(**kw, device_id=local_rank, coo=(prob.coo_row, prob.coo_col, prob.coo_cnt))
    s.set_param_simu(prob.params)
    s.eval_likelihood_init()
    log("[rank %d] uploaded + initial likelihood %.6f in %.1fs" % (rank, float(s.curr_likelihood_on_nz[0]), time.time() - t0))

    # trajectory: one shuffled cycle prefix, candidates pre-drawn with the reference's RNG consumption
    np.random.seed(a.seed)
    n_total = a.warmup + a.steps
    order = np.arange(prob.n_frags)
    np.random.shuffle(order)
    frags = np.resize(order, n_total).astype(np.int32)
    t0 = time.time()
    cands = s.draw_candidates(frags, a.neighbours)
    t_draw = time.time() - t0
    log("[rank %d] %d candidate lists drawn in %.2fs (%.1f us each, host numpy RNG)" % (rank, n_total, t_draw, 1e6 * t_draw / n_total))

    if world > 1:
        from instagraal_amd.multi_gpu import BatchRunner

        runner = BatchRunner(s.ctx, rank, world, dist=dist)
        run = runner.run
    else:
        run = s.ctx.step_batch

    if a.warmup:
        run(frags[: a.warmup], cands[: a.warmup])
    s.ctx.reset_timers(1 | ((1 << 2) << 1))  # hipEvent pairs around k_score only
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = run(frags[a.warmup:], cands[a.warmup:])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    score_ms, n_launch = s.ctx.kernel_time_ms("score")
    s.ctx.reset_timers(0)
    bstats = s.ctx.batch_stats() if world == 1 else None

    # self-check: the incrementally maintained exact likelihood equals a from-scratch recomputation
    sums, _ = s.ctx.debug_globals()
    _, _, limbs = s.ctx.full_likelihood(0)
    exact_ok = bool(int(sums[0]) == int(limbs[0]) and int(sums[1]) == int(limbs[1]))

    if rank == 0:
        # algorithmic bytes of one launch of the dominant kernel = sum of the per-move B_min of the moves it scored
        # (committed moves only: a slot that had to be re-scored is work, not algorithmic traffic)
        n_launch = max(int(n_launch), 1)
        bytes_min = float(res["bytes_min"].sum()) / n_launch / world  # N > 1: a rank scores 1/N of the slots of a launch
        n_evals = float(res["n_evals"].sum()) / n_launch / world
        achieved = bytes_min / (score_ms * 1e-3) / 1e9 if score_ms > 0 else 0.0
        # HBM bytes per launch of the dominant kernel from rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction +
        # WRITE_SIZE, separate passes; profiles/*_pmc_traffic.json): a committed measurement of THIS workload, or null
        traffic = valu_busy = None
        try:
            import glob

            pmc = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_%s_pmc_traffic.json" % a.config)))
            if pmc and world == 1:
                prof = json.load(open(pmc[-1]))["k_score_list"]
                traffic = float(prof["traffic_bytes_per_launch"])
                valu_busy = float(prof["VALUBusy_pct"]) / 100.0  # the bound that applies: fraction of cycles the VALUs issue
        except Exception:
            traffic = valu_busy = None
        out = {
            "metric": "MCMC moves/s (accepted+rejected) at fixed n_frags x nnz",
            "value": a.steps / elapsed,
            "unit": "moves/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64 terms (f32 inputs) / exact i64 fixed-point sums",
            "data": "synthetic",
            "config": {"workload": "synthetic Hi-C %s (%d sub-frags), level 4, %d neighbours, nuisance sampling off" % (
                prob_name(prob), prob.n_sub_frags, a.neighbours), "name": a.config, "seed": a.seed,
                "parallelism": "1 GPU" if world == 1 else "batch slots split over %d ranks, all-gather of score records" % world,
                "candidates_scored_per_s": float(res["n_candidates"].sum()) * 24 / elapsed,
                "term_evals_per_move": float(res["n_evals"].mean()), "moves_per_launch": a.steps / n_launch,
                "batches": bstats, "maintained_likelihood_exact": exact_ok},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": traffic, "valu_busy_profiled": valu_busy, "kernel": "k_score_list", "avg_launch_ms": score_ms, "launches": int(n_launch),
                         "algorithmic_bytes_per_launch": bytes_min,
                         "term_evals_per_launch": n_evals,
                         "term_evals_per_s": (n_evals / (score_ms * 1e-3)) if score_ms > 0 else 0.0,
                         "note": "B_min = sum_c[12 S_c + 20 m_c U + 8 U] + 68 n_touched per move (SURVEY 8(d)), summed over the moves "
                                 "of a launch; the kernel is bound by VALU issue (VALUBusy 85 %, profiles/) of the exact f64 term "
                                 "arithmetic on an L2-resident working set, not by HBM: DESIGN.md section 4.3"},
        }
        if not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(prob, frags, cands, a.cpu_budget)
            except Exception as e:  # the baseline is a report, never the product path
                out["cpu_baseline"] = {"value": None, "unit": "moves/s", "cores": 1, "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
