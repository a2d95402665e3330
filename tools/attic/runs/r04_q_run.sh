set -x
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_full_size_gpu.py tests/test_dp_gpu.py tests/test_batch_assembly.py -q -x 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 30 --windows 2 2>&1 | tail -1 | cut -c1-160
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pj_1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pj_1 -- python $R/bench.py --no-cpu-baseline --no-secondary --windows 0 --steps 10 > /tmp/pj_b1.log 2>&1
(cd $R && python tools/prof_breakdown.py /tmp/pj_1 70 gpurun_out/r04_q_step_sequence.txt > gpurun_out/r04_q_one_step_breakdown.txt; grep "embed\|fold\|step wall" gpurun_out/r04_q_one_step_breakdown.txt)
