set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 5 300 python tools/nt_gemm_probe.py > gpurun_out/r03_nt_gemm_probe.txt 2>&1
cat gpurun_out/r03_nt_gemm_probe.txt
cd /tmp && export TMPDIR=/tmp
for v in "X=1" "MTN_KEEP_WT=1"; do
  rm -rf /tmp/pl_stats
  env $v timeout -k 5 300 rocprofv3 --kernel-trace --stats -d /tmp/pl_stats -- python $R/bench.py --no-cpu-baseline --no-secondary --no-record --windows 0 --steps 10 > /tmp/pl_b.log 2>&1
  (cd $R && python tools/prof_breakdown.py /tmp/pl_stats 60 /tmp/seq_$v.txt > gpurun_out/r03_l_breakdown_$v.txt)
done
head -40 $R/gpurun_out/r03_l_breakdown_X=1.txt
head -40 $R/gpurun_out/r03_l_breakdown_MTN_KEEP_WT=1.txt
