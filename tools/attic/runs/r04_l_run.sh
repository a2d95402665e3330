set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_fused_gpu.py -q -x -k "four_launch" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_ln_epilogue_gpu.py tests/test_full_size_gpu.py -q -x 2>&1 | tail -3
for v in 0 1 0 1; do
  export MTN_FH_TWO_SIZE=$v
  echo "== MTN_FH_TWO_SIZE=$v (batch 64)"
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --windows 2 --steps 30 --batch-per-gpu 64 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('window_ms_per_step'))"
done
unset MTN_FH_TWO_SIZE
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pj_1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pj_1 -- python $R/bench.py --no-cpu-baseline --no-secondary --windows 0 --steps 10 --batch-per-gpu 64 > /tmp/pj_b1.log 2>&1
(cd $R && python tools/prof_breakdown.py /tmp/pj_1 60 gpurun_out/r04_l_b64_step_sequence.txt > gpurun_out/r04_l_b64_one_step_breakdown.txt)
