set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r03_p_ab.txt
rm -f $O
timeout -k 5 900 python -m pytest tests/test_full_size_gpu.py tests/test_model_gpu.py tests/test_dp_gpu.py -x -q 2>&1 | tail -5 | tee -a $O
one() {  # $1 = env assignment, $2.. = bench flags
  v=$1; shift
  echo "== $v  $*" >> $O
  env $v timeout -k 5 200 python bench.py --no-cpu-baseline --no-secondary --no-record --windows 1 --steps 30 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; k=r['kernels'].get('gemm_tt_dma128_table_kernel', {}); print('samples/s', d['value'], 'step ms', d['ms_per_step'], '| table launch us', k.get('avg_us'))" >> $O 2>&1
}
for v in "X=1" "MTN_STEP_BRANCHES=0" "X=1" "MTN_STEP_BRANCHES=0"; do one $v; done
for v in "X=1" "MTN_STEP_BRANCHES=0"; do one $v --batch-per-gpu 64; done
cat $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pl_stats
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d /tmp/pl_stats -- python $R/bench.py --no-cpu-baseline --no-secondary --no-record --windows 0 --steps 10 > /tmp/pl_b.log 2>&1
(cd $R && python tools/prof_breakdown.py /tmp/pl_stats 60 gpurun_out/r03_p_step_sequence.txt > gpurun_out/r03_p_breakdown.txt)
head -3 $R/gpurun_out/r03_p_breakdown.txt; tail -8 $R/gpurun_out/r03_p_step_sequence.txt
