# round 4, run a: LayerNorm backward in the dLN-out GEMM's epilogue — tests, then A/B on the cfg2 step (same box)
set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ln_epilogue_gpu.py -x -q -s 2>&1 | tail -25
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_kernels_gpu.py -x -q 2>&1 | tail -8
for v in 0 1 0 1; do
  if [ $v = 0 ]; then export MTN_LN_EPI=0; else unset MTN_LN_EPI; fi
  echo "== MTN_LN_EPI=$v"
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --windows 2 --steps 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('window_ms_per_step'))"
done
