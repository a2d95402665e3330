set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_decode_gpu.py -q --tb=short -x -k "other_widths" -s 2>&1 | tail -25 > gpurun_out/r05_zw_pytest_decode.txt
cat gpurun_out/r05_zw_pytest_decode.txt
