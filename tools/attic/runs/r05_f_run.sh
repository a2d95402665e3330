set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_dp_rccl_gpu.py tests/test_dp_gpu.py -q --tb=short 2>&1 | tail -70 > gpurun_out/r05_f_pytest_dp.txt
cat gpurun_out/r05_f_pytest_dp.txt
timeout 900 python -m pytest "tests/test_full_size_gpu.py::test_cfg2_loss_and_gradients_match_oracle_through_the_captured_step" -q -s --tb=line 2>&1 | tail -30 > gpurun_out/r05_f_pytest_fs.txt
cat gpurun_out/r05_f_pytest_fs.txt
