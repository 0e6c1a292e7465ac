"""What bounds k_screen: the kernel without some of its parts, timed on the REAL trajectory.  The library launches a probe
instance of the screening kernel behind the real launch of every batch (IG_SCREEN_PROBE=mask: bit 0 no P_z gather, bit 1 the
partner's record made up from the row's -- one LDS gather per entry and column instead of two --, bit 2 no v_log / v_exp), on the
same lists and columns, its sums into scratch words: the results are the real kernel's, so every mask sees the same moves
(ablated BUILDS do not: other winners, other contigs, other lists).     python tools/screen_probe.py [cfg3] [moves]"""
import json
import os
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
    import numpy as np

    from instagraal_amd import synth
    from instagraal_amd.sampler import sampler as hip_sampler

    cfg, n = sys.argv[2], int(sys.argv[3])
    prob = synth.make_problem(*synth.CONFIGS[cfg])
    s = hip_sampler(**prob.sampler_kwargs(), device_id=0, coo=(prob.coo_row, prob.coo_col, prob.coo_cnt))
    s.set_param_simu(prob.params)
    s.eval_likelihood_init()
    np.random.seed(0)
    frags = np.resize(np.random.permutation(prob.n_frags), n + 200).astype(np.int32)
    cands = s.draw_candidates(frags, 5)
    s.ctx.step_batch(frags[:200], cands[:200])
    s.ctx.reset_timers(1)
    s.ctx.set_timer_sampling(1) if hasattr(s.ctx, "set_timer_sampling") else None
    res = s.ctx.step_batch(frags[200:], cands[200:])
    out = {k: s.ctx.kernel_time_ms(k) for k in ("screen", "probe")}
    print(json.dumps({"mask": int(os.environ.get("IG_SCREEN_PROBE", "-1")), "screen_us": round(out["screen"][0] * 1e3, 1), "probe_us": round(out["probe"][0] * 1e3, 1),
                      "launches": out["probe"][1], "checksum": int(res["op_sampled"].astype(np.int64).sum())}))
    sys.exit(0)

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = sys.argv[2] if len(sys.argv) > 2 else "1200"
names = {0: "everything (the kernel as it is, without the tail walks)", 1: "no P_z gather", 2: "one record gather per entry and column instead of two",
         3: "no P_z gather, one record gather", 4: "no v_log / v_exp", 5: "no P_z gather, no v_log / v_exp", 6: "one record gather, no v_log / v_exp",
         7: "none of the three", 8: "no entries at all: set-up, staging, reduction, publication",
         16: "dispatch and the first round of loads only", 32: "... and the second round, the staging, its barrier",
         512: "everything requested and staged; no barrier, no pass", 1024: "the kernel without the pairs that take the one-column routine",
         1056: "up to the staging's barrier, without those pairs", 2048: "the kernel without the pairs with a ring on a window",
         4096: "the kernel without a candidate's last, odd column (and other one-column cases without a ring)",
         1032: "no entries, and without the pairs that take the one-column routine", 1544: "... up to the staging only (no barrier, no reduction, no publication)",
         8192: "the kernel without the loads of its entries inside the loop", 8199: "... and without P_z gather, record gather, v_log / v_exp",
         96: "... without the P_z table", 160: "... without the columns", 288: "... without the first entries", 480: "... without any of the three"}
masks = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else list(range(9)) + [16, 32]
for mask in masks:
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", cfg, n], env=dict(os.environ, IG_SCREEN_PROBE=str(mask)), capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print("mask", mask, "failed:", r.stderr[-500:])
        continue
    d = json.loads(line[-1])
    print("%-62s %6.1f us   (real launch next to it: %6.1f us, %d launches, moves checksum %d)" % (names[mask], d["probe_us"], d["screen_us"], d["launches"], d["checksum"]))
