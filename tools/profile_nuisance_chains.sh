#!/bin/bash
# kernel trace of the nuisance-on loop with chains (settled chunks): per-kernel statistics + a timeline excerpt.
# usage: bash tools/profile_nuisance_chains.sh <tag>   (through gpurun; copy gpurun_out/<tag>_nuis_* into profiles/)
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_nuis
NUIS_LONG=5 NUIS_ONLY=1 timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_nuis -o nuis -- python3 $R/tools/nuisance_rate.py cfg3 600 > $R/gpurun_out/${TAG}_nuis_prof.log 2>&1
DB=$(find /tmp/prof_nuis -name "*.db" | head -1)
python3 $R/tools/rocprof_stats.py $DB $R/gpurun_out/${TAG}_nuis_settled_kernel_stats.csv \
  "rocprofv3 --kernel-trace -- NUIS_LONG=5 NUIS_ONLY=1 python3 tools/nuisance_rate.py cfg3 600 (3 620 (move, step) pairs, chains on; one MI355X); aggregated by tools/rocprof_stats.py"
python3 $R/tools/rocprof_timeline.py $DB 120 0.9 k_chain_hist_eval > $R/gpurun_out/${TAG}_nuis_settled_timeline.txt 2>&1
# the same for the LAST chunk alone (the settled 600 pairs the tool times), cut out by the window the tool prints
WIN=$(grep CHUNK_WINDOW_NS $R/gpurun_out/${TAG}_nuis_prof.log | tail -1 | cut -d' ' -f3-)
python3 $R/tools/rocprof_stats.py $DB $R/gpurun_out/${TAG}_nuis_lastchunk_kernel_stats.csv \
  "rocprofv3 --kernel-trace -- NUIS_LONG=5 NUIS_ONLY=1 python3 tools/nuisance_rate.py cfg3 600: the launches of the last chunk of 600 (move, step) pairs only; aggregated by tools/rocprof_stats.py" "$WIN" > $R/gpurun_out/${TAG}_nuis_lastchunk_top.txt 2>&1
