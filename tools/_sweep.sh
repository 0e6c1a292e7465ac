cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_kt -o run -- python3 $R/bench.py --no-cpu-baseline --nuisance-moves 0 --steps 1000 --warmup 100 > /tmp/kt.log 2>&1
python3 $R/tools/rocprof_timeline.py $(find /tmp/prof_kt -name "*.db" | head -1) 36 > $R/gpurun_out/r02k_batch_timeline.txt
