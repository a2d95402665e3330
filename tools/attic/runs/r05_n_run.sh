set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_dp_rccl_gpu.py tests/test_dp_gpu.py -q --tb=short -x 2>&1 | tail -5 > gpurun_out/r05_n_pytest_dp.txt
cat gpurun_out/r05_n_pytest_dp.txt
for w in cfg2 cfg3; do
timeout 600 python bench.py --workload $w --no-cpu-baseline --steps 20 --dp-one-rank-probe 2>&1 | tail -1 > gpurun_out/r05_n_dp_one_rank_$w.txt
cut -c1-3000 gpurun_out/r05_n_dp_one_rank_$w.txt
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pj_dp -- python $R/bench.py --workload cfg3 --no-cpu-baseline --steps 10 --dp-one-rank-probe > /tmp/pj_dp.log 2>&1
cd $R
python tools/prof_breakdown.py /tmp/pj_dp 40 gpurun_out/r05_n_dp_cfg3_step_sequence.txt > gpurun_out/r05_n_dp_cfg3_one_rank_breakdown.txt 2>&1
head -45 gpurun_out/r05_n_dp_cfg3_one_rank_breakdown.txt
TAG=r05 bash tools/prof_round.sh > gpurun_out/r05_n_prof_round.log 2>&1
tail -30 gpurun_out/r05_n_prof_round.log
head -30 gpurun_out/r05_one_step_breakdown.txt
