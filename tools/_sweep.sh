T=r02m
bash tools/profile_round.sh $T > gpurun_out/${T}_profile_round.log 2>&1
bash tools/profile_nuisance.sh $T > gpurun_out/${T}_profile_nuis.log 2>&1
python tools/nuisance_rate.py cfg3 600 2>&1 | grep -v "Warning\|ratio =\|amdgpu.ids" > gpurun_out/${T}_nuisance_rate.txt
python bench.py 2>&1 | tail -1 > gpurun_out/${T}_bench_cfg3.json
python bench.py --config cfg2 2>&1 | tail -1 > gpurun_out/${T}_bench_cfg2.json
python bench.py --config bigctg 2>&1 | tail -1 > gpurun_out/${T}_bench_bigctg.json
python bench.py --config cfg5 --steps 1000 --warmup 100 --cpu-budget 5 2>&1 | tail -1 > gpurun_out/${T}_bench_cfg5.json
