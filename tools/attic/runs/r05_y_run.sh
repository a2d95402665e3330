set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_decode_gpu.py -q --tb=short -x -s 2>&1 | tail -14 > gpurun_out/r05_y_pytest_decode.txt
cat gpurun_out/r05_y_pytest_decode.txt
timeout 300 python tools/decode_timeline.py > gpurun_out/r05_y_decode_timeline.txt 2>&1
tail -34 gpurun_out/r05_y_decode_timeline.txt
timeout 300 python bench_decode.py --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r05_y_bench_decode.txt
cut -c1-700 gpurun_out/r05_y_bench_decode.txt
