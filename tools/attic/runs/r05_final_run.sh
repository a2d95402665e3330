set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout 3000 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -25 > gpurun_out/r05_final_pytest_gpu.txt
cat gpurun_out/r05_final_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time timeout 1500 python bench.py ) > gpurun_out/r05_final_bench_default.txt 2>&1
tail -5 gpurun_out/r05_final_bench_default.txt | cut -c1-1500
TAG=r05 bash tools/prof_round.sh > gpurun_out/r05_final_prof_round.log 2>&1
tail -25 gpurun_out/r05_final_prof_round.log
