set -x
R=$GRAFT_REPO_ROOT
cd $R
ab() { for i in 1 2; do
  for cfg in "MTN_TT_AUX=0 MTN_STEP_HEAD=0" "MTN_TT_AUX=1 MTN_STEP_HEAD=1"; do
    echo "== $cfg $1"; env $cfg timeout 300 python bench.py --no-cpu-baseline --no-secondary --windows 4 --steps 30 $1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['median_window_ms_per_step'], d['config']['window_ms_per_step'], d['roofline'].get('avg_us_per_launch'), d['roofline']['all_gemm_kernels']['launches_per_step'])"
  done; done; }
ab "" > gpurun_out/r05_c_tail_head_ab.txt 2>&1
ab "--workload cfg3" >> gpurun_out/r05_c_tail_head_ab.txt 2>&1
cat gpurun_out/r05_c_tail_head_ab.txt
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r05_c_pytest.txt
cat gpurun_out/r05_c_pytest.txt
