set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_dp_rccl_gpu.py tests/test_dp_gpu.py -q --tb=short -x 2>&1 | tail -5 > gpurun_out/r05_p_pytest_dp.txt
cat gpurun_out/r05_p_pytest_dp.txt
for w in cfg2 cfg3; do
timeout 600 python bench.py --workload $w --no-cpu-baseline --steps 20 --dp-one-rank-probe 2>&1 | tail -1 > gpurun_out/r05_p_dp_one_rank_$w.txt
cut -c1-3500 gpurun_out/r05_p_dp_one_rank_$w.txt
timeout 600 python bench.py --workload $w --no-cpu-baseline --no-secondary --windows 0 --steps 20 2>&1 | tail -1 | cut -c1-200
done
