#!/bin/bash
# kernel trace of the nuisance-on loop once the chain has settled (2 400 pairs of warm-up inside the traced run):
# bash tools/profile_nuisance_settled.sh <tag>   (through gpurun; copy gpurun_out/<tag>_* into profiles/)
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export NUIS_ONLY=1 NUIS_LONG=5
rm -rf /tmp/prof_nus
timeout 500 rocprofv3 --kernel-trace -d /tmp/prof_nus -o run -- python3 $R/tools/nuisance_rate.py cfg3 600 > /tmp/nus.log 2>&1
grep "chunk" /tmp/nus.log | cut -c1-120
DB=$(find /tmp/prof_nus -name "*.db" | head -1)
python3 $R/tools/rocprof_timeline.py $DB 70 0.85 k_gather > $R/gpurun_out/${TAG}_nuis_settled_timeline.txt
python3 $R/tools/rocprof_stats.py $DB $R/gpurun_out/${TAG}_nuis_settled_kernel_stats.csv \
  "rocprofv3 --kernel-trace -- NUIS_LONG=5 python3 tools/nuisance_rate.py cfg3 600 (3 620 moves + nuisance steps, one MI355X); aggregated by tools/rocprof_stats.py"
