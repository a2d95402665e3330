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
