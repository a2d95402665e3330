set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ln_epilogue_gpu.py -q -x 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pj_1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pj_1 -- python $R/bench.py --no-cpu-baseline --no-secondary --windows 0 --steps 10 > /tmp/pj_b1.log 2>&1
(cd $R && python tools/prof_breakdown.py /tmp/pj_1 70 gpurun_out/r04_r_step_sequence.txt > gpurun_out/r04_r_one_step_breakdown.txt; grep "embed\|fold\|step wall" gpurun_out/r04_r_one_step_breakdown.txt)
