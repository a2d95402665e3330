set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0 MTN_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29577
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pj_dp -- python $R/bench.py --workload cfg3 --no-cpu-baseline --no-secondary --windows 0 --steps 10 > /tmp/pj_dp.log 2>&1
tail -1 /tmp/pj_dp.log | cut -c1-200
cd $R
python tools/prof_breakdown_dp.py /tmp/pj_dp 40 gpurun_out/r04_x_dp_cfg3_step_sequence.txt > gpurun_out/r04_x_dp_cfg3_one_step_breakdown.txt
head -70 gpurun_out/r04_x_dp_cfg3_one_step_breakdown.txt
