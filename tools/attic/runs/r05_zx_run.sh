set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout 300 python bench_decode.py --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r05_zx_bench_decode.txt
cut -c1-1100 gpurun_out/r05_zx_bench_decode.txt
