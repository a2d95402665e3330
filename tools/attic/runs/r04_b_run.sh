# round 4, run b: where the LayerNorm-epilogue step spends its time (rocprofv3 kernel trace, both settings, same box)
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 1; do
  if [ $v = 0 ]; then export MTN_LN_EPI=0; else unset MTN_LN_EPI; fi
  rm -rf /tmp/pj_$v
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pj_$v -- python $R/bench.py --no-cpu-baseline --no-secondary --windows 0 --steps 10 > /tmp/pj_b$v.log 2>&1
  tail -1 /tmp/pj_b$v.log | cut -c1-200
  (cd $R && python tools/prof_breakdown.py /tmp/pj_$v 60 gpurun_out/r04_b_epi${v}_step_sequence.txt > gpurun_out/r04_b_epi${v}_one_step_breakdown.txt)
done
