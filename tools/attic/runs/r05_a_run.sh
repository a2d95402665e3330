set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout 300 ./tools/l2_written_probe.bin > gpurun_out/r05_l2_written_probe.txt 2>&1
cat gpurun_out/r05_l2_written_probe.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_a_bench.json 2> gpurun_out/r05_a_bench.err
cut -c1-400 gpurun_out/r05_a_bench.json
tail -3 gpurun_out/r05_a_bench.err
