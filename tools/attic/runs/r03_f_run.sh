set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 5 120 ./tools/seam_bench.bin > gpurun_out/r03_seam_bench.txt 2>&1
cat gpurun_out/r03_seam_bench.txt
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d /tmp/dp_stats -- python $R/tools/dp_rccl_probe.py --steps 10 > $R/gpurun_out/r03_f_dp_probe_rocprof.txt 2>&1
python $R/tools/prof_summary.py /tmp/dp_stats $R/gpurun_out/r03_f_dp_rccl_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/dp_rccl_probe.py --steps 10 (one-rank RCCL group, MTN_FORCE_DIST=1, cfg2 batch 32)" | head -40
grep -i nccl $R/gpurun_out/r03_f_dp_rccl_kernel_stats.csv | head
