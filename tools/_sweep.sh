for v in "IG_SLICE_SHARED_PLANES=1" "IG_SLICE_SHARED_PLANES=2" "IG_SLICE_SHARED_PLANES=3" "IG_SLICE_SHARED_PLANES=5" "IG_SLICE_SHARED_PLANES=2 IG_SLICE_RB=96" "IG_SLICE_SHARED_PLANES=4 IG_SLICE_RB=64"; do
echo "== $v"
env $v python bench.py --no-cpu-baseline --nuisance-moves 0 2>&1 | tail -1 | cut -c77-100
env $v python tools/kernel_times.py cfg3 1500 2>&1 | grep "slice"
done
