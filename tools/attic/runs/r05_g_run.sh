set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_decode_gpu.py -q --tb=short -x -k "persistent" -s 2>&1 | tail -40 > gpurun_out/r05_g_pytest_mega.txt
cat gpurun_out/r05_g_pytest_mega.txt
timeout 900 python -m pytest tests/test_decode_gpu.py -q --tb=short 2>&1 | tail -15 > gpurun_out/r05_g_pytest_decode.txt
cat gpurun_out/r05_g_pytest_decode.txt
timeout 900 python -m pytest tests/test_dp_rccl_gpu.py tests/test_dp_gpu.py -q --tb=short 2>&1 | tail -15 > gpurun_out/r05_g_pytest_dp.txt
cat gpurun_out/r05_g_pytest_dp.txt
timeout 900 python -m pytest "tests/test_full_size_gpu.py::test_cfg2_loss_and_gradients_match_oracle_through_the_captured_step" -q -s --tb=line 2>&1 | grep -E "worst error|passed|failed|Error" > gpurun_out/r05_g_pytest_fs.txt
cat gpurun_out/r05_g_pytest_fs.txt
timeout 300 python bench_decode.py 2>&1 | tail -3 > gpurun_out/r05_g_bench_decode.txt
cat gpurun_out/r05_g_bench_decode.txt
one() { env $1 timeout 300 python bench.py --no-cpu-baseline --no-secondary --windows 4 --steps 30 $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', d['ms_per_step'], d['config']['median_window_ms_per_step'], d['roofline'].get('avg_us_per_launch'))"; }
{ for i in 1 2; do one "MTN_STEP_HEAD=0" ""; one "MTN_STEP_HEAD=1" ""; done; } > gpurun_out/r05_g_head_ab.txt 2>&1
grep -v "^+" gpurun_out/r05_g_head_ab.txt
