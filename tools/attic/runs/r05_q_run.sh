set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_counted_waits_gpu.py tests/test_decode_gpu.py tests/test_model_gpu.py -q --tb=short -x 2>&1 | tail -8 > gpurun_out/r05_q_pytest.txt
cat gpurun_out/r05_q_pytest.txt
