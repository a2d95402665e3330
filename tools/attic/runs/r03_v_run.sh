set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pl_stats
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d /tmp/pl_stats -- python $R/bench.py --no-cpu-baseline --no-secondary --no-record --windows 0 --steps 10 > /tmp/pl_b.log 2>&1
(cd $R && python tools/prof_breakdown.py /tmp/pl_stats 60 gpurun_out/r03_v_step_sequence.txt > gpurun_out/r03_v_breakdown.txt)
head -40 $R/gpurun_out/r03_v_breakdown.txt
