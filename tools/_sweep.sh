for i in 1 2; do
python bench.py --config cfg2 --no-cpu-baseline --nuisance-moves 0 2>&1 | tail -1 | cut -c1-120
IG_HIP_LIB=$GRAFT_REPO_ROOT/tune_nolb.so python bench.py --config cfg2 --no-cpu-baseline --nuisance-moves 0 2>&1 | tail -1 | cut -c1-120
done
