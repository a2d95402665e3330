set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ln_epilogue_gpu.py -x -q -s 2>&1 | grep "linearity\|passed\|failed\|Error\|assert" | tail -20
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -6
for v in 0 1 0 1; do
  if [ $v = 0 ]; then export MTN_LN_LIN=0; else unset MTN_LN_LIN; fi
  echo "== MTN_LN_LIN=$v"
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --windows 2 --steps 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('window_ms_per_step'))"
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
unset MTN_LN_LIN
rm -rf /tmp/pj_1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pj_1 -- python $R/bench.py --no-cpu-baseline --no-secondary --windows 0 --steps 10 > /tmp/pj_b1.log 2>&1
(cd $R && python tools/prof_breakdown.py /tmp/pj_1 60 gpurun_out/r04_j_lin_step_sequence.txt > gpurun_out/r04_j_lin_one_step_breakdown.txt)
