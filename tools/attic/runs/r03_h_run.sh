set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 5 300 python tools/two_chain_probe.py > gpurun_out/r03_two_chain_probe.txt 2>&1
tail -4 gpurun_out/r03_two_chain_probe.txt
timeout -k 5 120 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "attention_fwd_bwd or gemm" 2>&1 | tail -4
TAG=r03_b timeout -k 5 900 bash tools/prof_round.sh > gpurun_out/r03_b_prof.log 2>&1
tail -22 gpurun_out/r03_b_prof.log
cd /tmp && export TMPDIR=/tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d /tmp/c3_stats -- python $R/bench.py --workload cfg3 --gpus 1 --no-cpu-baseline --no-secondary --windows 0 --steps 10 > /tmp/c3.log 2>&1
python $R/tools/prof_breakdown.py /tmp/c3_stats 60 $R/gpurun_out/r03_b_cfg3_step_sequence.txt > $R/gpurun_out/r03_b_cfg3_one_step_breakdown.txt
head -24 $R/gpurun_out/r03_b_cfg3_one_step_breakdown.txt
