set -x
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15
timeout 900 python bench.py > gpurun_out/r04_i_bench.json 2> gpurun_out/r04_i_bench.err; tail -c 600 gpurun_out/r04_i_bench.json; tail -5 gpurun_out/r04_i_bench.err
