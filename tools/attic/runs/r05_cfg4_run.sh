set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pj_cfg4 -- python $R/bench.py --workload cfg4 --no-cpu-baseline --no-secondary --windows 0 --steps 10 > /tmp/pj_c4.log 2>&1
tail -1 /tmp/pj_c4.log | cut -c1-300
cd $R
python tools/prof_breakdown.py /tmp/pj_cfg4 45 gpurun_out/r05_cfg4_step_sequence.txt > gpurun_out/r05_cfg4_one_step_breakdown.txt
head -48 gpurun_out/r05_cfg4_one_step_breakdown.txt
