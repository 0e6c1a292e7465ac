#!/bin/bash
# kernel trace of the nuisance-on loop: bash tools/profile_nuisance.sh <tag>   (through gpurun; copy gpurun_out/<tag>_* into profiles/)
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export NUIS_ONLY=1
rm -rf /tmp/prof_nu
timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_nu -o run -- python3 $R/tools/nuisance_rate.py cfg3 600 > /tmp/nu.log 2>&1
python3 $R/tools/rocprof_stats.py $(find /tmp/prof_nu -name "*.db" | head -1) $R/gpurun_out/${TAG}_nuis_kernel_stats.csv \
  "rocprofv3 --kernel-trace -- python3 tools/nuisance_rate.py cfg3 600 (620 moves + nuisance steps, one MI355X); aggregated by tools/rocprof_stats.py"
grep -v "Warning\|ratio =" /tmp/nu.log | tail -4
python3 $R/tools/rocprof_timeline.py $(find /tmp/prof_nu -name "*.db" | head -1) 90 > $R/gpurun_out/${TAG}_nuis_timeline.txt
