set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_decode_gpu.py -q --tb=short -x -k "persistent" -s 2>&1 | tail -25 > gpurun_out/r05_h_pytest_mega.txt
cat gpurun_out/r05_h_pytest_mega.txt
timeout 300 python tools/decode_timeline.py > gpurun_out/r05_h_decode_timeline.txt 2>&1
cat gpurun_out/r05_h_decode_timeline.txt
timeout 300 python bench_decode.py --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r05_h_bench_decode.txt
cut -c1-900 gpurun_out/r05_h_bench_decode.txt
MTN_DECODE_MEGA=0 timeout 300 python bench_decode.py --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r05_h_bench_decode_off.txt
cut -c1-900 gpurun_out/r05_h_bench_decode_off.txt
timeout 900 python -m pytest "tests/test_full_size_gpu.py::test_cfg2_loss_and_gradients_match_oracle_through_the_captured_step" -q -s --tb=line 2>&1 | grep -E "worst error|passed|failed|Error" > gpurun_out/r05_h_pytest_fs.txt
cat gpurun_out/r05_h_pytest_fs.txt
