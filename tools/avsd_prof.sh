cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pj_avsd -- python $R/bench.py --workload avsd32 --no-cpu-baseline --no-secondary --windows 0 --steps 10 > /tmp/pj_a.log 2>&1
tail -c 600 /tmp/pj_a.log
cd $R
python tools/prof_breakdown.py /tmp/pj_avsd 45 gpurun_out/r06_avsd32_step_sequence.txt > gpurun_out/r06_avsd32_one_step_breakdown.txt
cat gpurun_out/r06_avsd32_one_step_breakdown.txt
