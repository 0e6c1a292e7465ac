python -m pytest tests/test_hip_parity.py tests/test_hip_configs.py tests/test_hip_sampler.py -x -q -m gpu -k "not cfg5" 2>&1 | tail -2
for v in "A=1" "IG_FULL_GRID=256"; do
echo "== $v"
env $v NUIS_ONLY=1 python tools/nuisance_rate.py cfg3 600 2>&1 | grep "moves/s\|host time"
done
IG_FULL_GRID=256 bash tools/profile_nuisance.sh r02k > /dev/null 2>&1; head -60 gpurun_out/r02k_nuis_timeline.txt
