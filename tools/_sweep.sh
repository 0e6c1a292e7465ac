for v in "IG_NUIS_W=1" "IG_NUIS_W=2" "IG_NUIS_W=3" "IG_NUIS_W=4" "IG_NUIS_W=6" "IG_NUIS_W=8"; do
echo "== $v"
env $v NUIS_ONLY=1 python tools/nuisance_rate.py cfg3 600 2>&1 | grep "moves/s\|batches"
done
