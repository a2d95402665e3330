set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout 300 ./tools/l2_written_probe.bin > gpurun_out/r05_l2_written_probe_b.txt 2>&1
cat gpurun_out/r05_l2_written_probe_b.txt
timeout 900 python -m pytest tests/test_dp_gpu.py tests/test_dp_rccl_gpu.py -x -q 2>&1 | tail -5
