#!/bin/bash
# Profiles of one round on the GPU box: kernel trace stats + PMC passes (separate runs), summaries into gpurun_out/.
# usage: bash tools/profile_round.sh <tag>      (run through gpurun; copy gpurun_out/<tag>_* into profiles/)
TAG=${1:-rXX}
CFG=${CONFIG:-cfg3}   # CONFIG=cfg5 bash tools/profile_round.sh <tag>: the human-scale shape
PAR=${PARAMS:-synthetic}  # PARAMS=settled: the P(s) parameters a nuisance chain settles into (bench.py --params settled); files are named <cfg>settled
NAME=$CFG; [ "$PAR" = "settled" ] && NAME=${CFG}settled
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_kt -o run -- python3 $R/bench.py --config $CFG --params $PAR --settled-batches 0 --no-cpu-baseline --nuisance-moves 0 --reference-loop-moves 0 --late-moves 0 --steps 20 --warmup 5 > /tmp/kt.log 2>&1
python3 $R/tools/rocprof_stats.py $(find /tmp/prof_kt -name "*.db" | head -1) $R/gpurun_out/${TAG}_${NAME}_kernel_stats.csv \
  "rocprofv3 --kernel-trace -- python3 bench.py --config $CFG --params $PAR --settled-batches 0 --no-cpu-baseline --nuisance-moves 0 --reference-loop-moves 0 --late-moves 0 --steps 20 --warmup 5 (2 560 timed moves; one MI355X); aggregated by tools/rocprof_stats.py"
tail -1 /tmp/kt.log | cut -c1-400
[ -n "$SKIP_PMC" ] && exit 0
i=0
for CT in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "VALUBusy VALUUtilization LdsUtil LdsBankConflict" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  [ -n "$TRAFFIC_ONLY" ] && [ $i -gt 2 ] && [ $i -ne 4 ] && continue   # FETCH_SIZE, WRITE_SIZE, VALUBusy only
  rm -rf /tmp/pmc$i
  timeout 600 rocprofv3 --pmc $CT -d /tmp/pmc$i -o run -- python3 $R/bench.py --config $CFG --params $PAR --settled-batches 0 --no-cpu-baseline --nuisance-moves 0 --reference-loop-moves 0 --late-moves 0 --steps 8 --warmup 2 > /tmp/pmc$i.log 2>&1
  [ $i -eq 1 ] && grep '^{' /tmp/pmc1.log | tail -1 > $R/gpurun_out/${TAG}_${NAME}_pmc_bench_line.json
  python3 $R/tools/rocprof_pmc.py $(find /tmp/pmc$i -name "*.db" | head -1) $R/gpurun_out/${TAG}_${NAME}_pmc_pass$i.json | grep -i "score_list\|k_screen\|k_slice" | cut -c1-300
done
