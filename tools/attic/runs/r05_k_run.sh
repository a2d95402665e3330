set -x
R=$GRAFT_REPO_ROOT
cd $R
for NW in 16 8; do
export MTN_DEC_NW=$NW
timeout 600 python -m pytest tests/test_decode_gpu.py -q --tb=short -x -k "persistent" -s 2>&1 | tail -12 > gpurun_out/r05_k_pytest_mega_nw$NW.txt
cat gpurun_out/r05_k_pytest_mega_nw$NW.txt
timeout 300 python tools/decode_timeline.py > gpurun_out/r05_k_decode_timeline_nw$NW.txt 2>&1
tail -30 gpurun_out/r05_k_decode_timeline_nw$NW.txt
timeout 300 python bench_decode.py --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r05_k_bench_decode_nw$NW.txt
cut -c1-600 gpurun_out/r05_k_bench_decode_nw$NW.txt
done
unset MTN_DEC_NW
MTN_DECODE_MEGA=0 timeout 300 python bench_decode.py --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r05_k_bench_decode_launch.txt
cut -c1-600 gpurun_out/r05_k_bench_decode_launch.txt
