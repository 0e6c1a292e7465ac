"""moves/s of the loop WITH nuisance sampling (instagraal.py:217-262 for cycles > 4): one step_sampler and one
step_nuisance_parameters (a full pass over all contacts under test parameters) per move, through
sampler.step_sampler_nuisance_batch (the two in flight together) and through the one-call-at-a-time methods; with a
breakdown of where the host waits.  Diagnostic, not the BASELINE metric.   python tools/nuisance_rate.py [cfg3] [moves]"""
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np

from instagraal_amd import synth
from instagraal_amd.sampler import sampler as hip_sampler

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
prob = synth.make_problem(*synth.CONFIGS[cfg])
s = hip_sampler(**prob.sampler_kwargs(), device_id=0, coo=(prob.coo_row, prob.coo_col, prob.coo_cnt))
s.set_param_simu(prob.params)
s.bins = np.arange(1.0, 60.0, 1.0)
s.eval_likelihood_init()
np.random.seed(0)
frags = np.random.permutation(prob.n_frags)[: 2 * n + 40]
s.step_sampler_nuisance_batch(frags[:20], 5, s.dt, 0, n)
b0 = s.ctx.batch_stats()
w0 = s.ctx.debug_nuis_wait()
t0 = time.perf_counter()
res, tup = s.step_sampler_nuisance_batch(frags[20:20 + n], 5, s.dt, 0, n)
dt = time.perf_counter() - t0
print("%s: %.0f moves/s through step_sampler_nuisance_batch (accept rate %.2f)" % (cfg, n / dt, np.mean([q[6] for q in tup])))
b1 = s.ctx.batch_stats()
print("   of the library call: %.0f us per move waiting for the device" % (1e6 * (s.ctx.debug_nuis_wait() - w0) / n))
print("   batches scored: %d for %d moves; one-move tails %d" % (b1["batches"] - b0["batches"], n, b1["one_move_tails"] - b0["one_move_tails"]))
print("   screened pass:", s.ctx.debug_nuis_screen_stats())
print("   its histogram tier:", s.ctx.debug_nuis_hist_stats())
if hasattr(s, "nuis_profile"):
    tot = sum(s.nuis_profile.values())
    print("   host time per move: " + ", ".join("%s %.0f us" % (k, 1e6 * v / n) for k, v in s.nuis_profile.items()) + " (sum %.0f us)" % (1e6 * tot / n))
s.nuis_step_trace = []
if os.environ.get("NUIS_LONG"):  # the rate as the chain settles: chunks of 600 moves
    CH = int(os.environ.get("NUIS_CHUNK", "600"))  # moves per call
    for k in range(int(os.environ["NUIS_LONG"])):
        fr = np.random.permutation(prob.n_frags)[:CH]
        w0 = s.ctx.debug_nuis_wait()
        clk = [(n_, getattr(time, n_)) for n_ in ("CLOCK_MONOTONIC", "CLOCK_BOOTTIME", "CLOCK_REALTIME", "CLOCK_MONOTONIC_RAW") if hasattr(time, n_)]
        ns0 = [time.clock_gettime_ns(c_) for _, c_ in clk]
        t0 = time.perf_counter()
        res, tup = s.step_sampler_nuisance_batch(fr, 5, s.dt, 0, CH)
        dt = time.perf_counter() - t0
        # (for tools/rocprof_stats.py: the chunk's window on whichever clock the tracer stamps its launches with)
        print("CHUNK_WINDOW_NS %d " % k + " ".join("%d,%d" % (a_, time.clock_gettime_ns(c_)) for a_, (_, c_) in zip(ns0, clk)))
        tr = np.array(s.nuis_step_trace)
        s.nuis_step_trace.clear()
        if tr.size == 0:
            tr = np.zeros((1, 2))
        rej, acc = tr[tr[:, 1] == 0, 0] * 1e6, tr[tr[:, 1] == 1, 0] * 1e6
        if len(rej) == 0:
            rej = np.zeros(1)
        print("            chains so far:", s.ctx.debug_nuis_chain_stats(), "; plain pairs this chunk: %d" % len(tr))
        print("            per PLAIN step: rejected n=%d median %.0f us, p25 %.0f, p75 %.0f, p95 %.0f, mean %.0f; accepted n=%d median %.0f us mean %.0f" % (
            len(rej), np.median(rej), np.percentile(rej, 25), np.percentile(rej, 75), np.percentile(rej, 95), rej.mean(), len(acc),
            np.median(acc) if len(acc) else 0, acc.mean() if len(acc) else 0))
        edges = [0, 80, 120, 200, 300, 450, 700, 1e9]
        hh = np.histogram(tr[:, 0] * 1e6, edges)[0]
        tot = [float((tr[(tr[:, 0] * 1e6 >= a) & (tr[:, 0] * 1e6 < b), 0] * 1e6).sum()) / CH for a, b in zip(edges[:-1], edges[1:])]
        print("            steps by duration (us) " + ", ".join("<%g: %d (%.0f us/step)" % (b, n_, t_) for b, n_, t_ in zip(edges[1:], hh, tot)))
        print("   chunk %d: %.0f moves/s, accept %.2f, device wait %.0f us/move, host %s, %s" % (
            k, CH / dt, np.mean([q[6] for q in tup]), 1e6 * (s.ctx.debug_nuis_wait() - w0) / CH,
            ", ".join("%s %.0f" % (a, 1e6 * v / CH) for a, v in s.nuis_profile.items()), s.ctx.debug_nuis_screen_stats()), flush=True)
        print("            histogram tier:", s.ctx.debug_nuis_hist_stats(), flush=True)
        print("            parameters now:", {k: float(s.param_simu[k][0]) for k in s.param_simu.dtype.names}, flush=True)
        ss = s.ctx.debug_screen_stats()
        print("            two-tier scoring so far: columns screened %d, scored exactly %d (%.1f %%); terms %.3g / %.3g (%.1f %%)" % (
            ss[2], ss[3], 100.0 * ss[3] / max(ss[2], 1), ss[4], ss[5], 100.0 * ss[5] / max(ss[4], 1)), flush=True)
if os.environ.get("NUIS_ONLY"):
    sys.exit(0)
t_s = t_n = 0.0
rest = frags[20 + n:20 + n + min(n, 100)]
for t, f in enumerate(rest):
    a = time.perf_counter()
    s.step_sampler(int(f), 5, s.dt)
    b = time.perf_counter()
    s.step_nuisance_parameters(s.dt, t, n)
    c = time.perf_counter()
    t_s += b - a
    t_n += c - b
print("%s: %.0f moves/s one call at a time (step_sampler %.0f us, step_nuisance_parameters %.0f us per move)" % (
    cfg, len(rest) / (t_s + t_n), 1e6 * t_s / len(rest), 1e6 * t_n / len(rest)))
