set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_decode_gpu.py -q --tb=short -x 2>&1 | tail -12 > gpurun_out/r05_l_pytest_decode.txt
cat gpurun_out/r05_l_pytest_decode.txt
timeout 300 python tools/decode_timeline.py > gpurun_out/r05_l_decode_timeline.txt 2>&1
tail -12 gpurun_out/r05_l_decode_timeline.txt
timeout 300 python bench_decode.py --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r05_l_bench_decode.txt
cut -c1-700 gpurun_out/r05_l_bench_decode.txt
timeout 3000 python -m pytest tests -m gpu -q --tb=short -x --deselect tests/test_decode_gpu.py 2>&1 | tail -25 > gpurun_out/r05_l_pytest_gpu.txt
cat gpurun_out/r05_l_pytest_gpu.txt
