#!/usr/bin/env python3
"""Assemble profiles/<tag>_cfg3_pmc_traffic.json from the per-pass counter files tools/profile_round.sh leaves in gpurun_out/
(<tag>_pmc_pass1..6.json: FETCH_SIZE | WRITE_SIZE | instruction counts | VALUBusy ... | cycles | TCC hits/misses).
FETCH_SIZE / WRITE_SIZE are KB per dispatch; FETCH_SIZE x 2 = the gfx950 correction of MI355X_MICROARCH.md.
usage: python tools/make_pmc_summary.py <tag> [outdir]"""
import glob
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
tag = sys.argv[1]
cfg = os.environ.get("CONFIG", "cfg3")  # the shape tools/profile_round.sh was run with
if os.environ.get("PARAMS", "synthetic") == "settled":
    cfg += "settled"
outdir = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles")
KERNELS = {"k_screen": ("_Z13k_screen_tail", "_Z8k_screen"), "k_score_list": "_Z12k_score_list", "k_slice": "_Z7k_slice", "k_tail": "_Z6k_tail",
           "k_full_nz_tiled": "_Z15k_full_nz_tiled", "k_decide_batch": ("_Z14k_decide_batch", "_Z15k_decide_commit", "_Z19k_decide_commit_par"), "k_mutate": "_Z8k_mutate",
           "k_gather": "_Z8k_gather", "k_delta": "_Z7k_delta", "k_offsets": "_Z9k_offsets", "k_contend": "_Z9k_contend", "k_worklist": "_Z10k_worklist",
           "k_records": "_Z9k_records", "k_predict": "_Z9k_predict", "k_commit_batch": "_Z14k_commit_batch"}
merged = {}
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "%s_%s_pmc_pass*.json" % (tag, cfg)))):
    d = json.load(open(f))
    for name, ctrs in d.items():
        for short, prefix in KERNELS.items():
            if name.startswith(prefix if isinstance(prefix, tuple) else (prefix,)):
                e = merged.setdefault(short, {})
                for c, v in ctrs.items():
                    e[c] = v["avg"]
                    e.setdefault("launches", v["launches"])
# the bench line of the first counter pass (tools/profile_round.sh keeps it): how many moves a launch chain of the PROFILED run covered --
# bench.py scales the replayed per-launch traffic to its own run's moves per launch with it
mpl = None
try:
    line = json.load(open(os.path.join(ROOT, "gpurun_out", "%s_%s_pmc_bench_line.json" % (tag, cfg))))
    mpl = float(line["config"]["moves_per_launch"])
except Exception:
    pass
out = {"command": "tools/profile_round.sh %s: rocprofv3 --pmc <one group per pass> -- python3 bench.py --no-cpu-baseline --nuisance-moves 0 "
                  "--reference-loop-moves 0 --late-moves 0 --config %s%s --settled-batches 0 --steps 8 --warmup 2 (8 steps of 128 moves; one MI355X; "
                  "%s moves per launch chain)" % (tag, cfg.replace("settled", ""), " --params settled" if cfg.endswith("settled") else "",
                                                   "%.1f" % mpl if mpl else "?"),
       "moves_per_launch": mpl,
       "note": "FETCH_SIZE / WRITE_SIZE are KB per dispatch (separate passes). traffic_bytes_per_launch = 2 x FETCH_SIZE (gfx950 "
               "correction of MI355X_MICROARCH.md, calibrated in round 1 on k_full_nz) + WRITE_SIZE.  Instruction counts are per "
               "dispatch and per counter instance as rocprofv3 reports them."}
for short, e in merged.items():
    r = {"launches": e.get("launches")}
    if "FETCH_SIZE" in e:
        r["FETCH_SIZE_KB_avg"] = e["FETCH_SIZE"]
    if "WRITE_SIZE" in e:
        r["WRITE_SIZE_KB_avg"] = e["WRITE_SIZE"]
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        r["traffic_bytes_per_launch"] = 1024.0 * (2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"])
    if "TCC_HIT_sum" in e:
        r["L2_hit_rate"] = e["TCC_HIT_sum"] / max(e["TCC_HIT_sum"] + e["TCC_MISS_sum"], 1.0)
    for k_in, k_out in (("VALUBusy", "VALUBusy_pct"), ("VALUUtilization", "VALUUtilization_pct"), ("LdsUtil", "LdsUtil_pct"),
                        ("LdsBankConflict", "LdsBankConflict_pct"), ("SQ_INSTS_VALU", "SQ_INSTS_VALU_per_dispatch"),
                        ("SQ_INSTS_SALU", "SQ_INSTS_SALU_per_dispatch"), ("SQ_INSTS_LDS", "SQ_INSTS_LDS_per_dispatch"),
                        ("SQ_WAVES", "SQ_WAVES_per_dispatch"), ("GRBM_GUI_ACTIVE", "GRBM_GUI_ACTIVE_cycles")):
        if k_in in e:
            r[k_out] = e[k_in]
    out[short] = r
path = os.path.join(outdir, "%s_%s_pmc_traffic.json" % (tag, cfg))
json.dump(out, open(path, "w"), indent=1)
print("wrote", path, {k: round(v.get("traffic_bytes_per_launch", 0) / 1e6, 1) for k, v in out.items() if isinstance(v, dict)})
