set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest "tests/test_dp_rccl_gpu.py::test_one_rank_rccl_chain_matches_plain_schedule_bitwise" -x -q 2>&1 | tail -60 > gpurun_out/r05_e_pytest_dp.txt
cat gpurun_out/r05_e_pytest_dp.txt
timeout 900 python -m pytest "tests/test_full_size_gpu.py::test_cfg2_loss_and_gradients_match_oracle_through_the_captured_step" -q -s 2>&1 | tail -80 > gpurun_out/r05_e_pytest_fs.txt
cat gpurun_out/r05_e_pytest_fs.txt
