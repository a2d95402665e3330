set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r03_ac_nw8.txt
rm -f $O
timeout -k 5 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" 2>&1 | tail -3 | tee -a $O

one() {  # $1 = env assignment, $2.. = bench flags
  v=$1; shift
  echo "== $v  $*" >> $O
  env $v timeout -k 5 200 python bench.py --no-cpu-baseline --no-secondary --no-record --windows 1 --steps 30 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels'].get('gemm_dma_kernel<64,64>', {}); print('samples/s', d['value'], 'step ms', d['ms_per_step'], '| <64,64> launches', k.get('launches_per_step'), 'avg us', k.get('avg_us'))" >> $O 2>&1
}
for v in "X=1" "MTN_GEMM_128X_NW8=1" "X=1" "MTN_GEMM_128X_NW8=1"; do one $v; done
for v in "X=1" "MTN_GEMM_128X_NW8=1" "X=1" "MTN_GEMM_128X_NW8=1"; do one $v --batch-per-gpu 64; done
cat $O
