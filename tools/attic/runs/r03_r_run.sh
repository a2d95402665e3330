set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r03_r_ab.txt
rm -f $O
timeout -k 5 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "k512" 2>&1 | tail -5 | tee -a $O
one() {  # $1 = env assignment, $2.. = bench flags
  v=$1; shift
  echo "== $v  $*" >> $O
  env $v timeout -k 5 200 python bench.py --no-cpu-baseline --no-secondary --no-record --windows 1 --steps 30 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; k=r['kernels'].get('gemm_k512_kernel', {}); print('samples/s', d['value'], 'step ms', d['ms_per_step'], '| k512 launches', k.get('launches_per_step'), 'avg us', k.get('avg_us'), 'TFLOP/s', k.get('achieved_TFLOPs'))" >> $O 2>&1
}
for v in "X=1" "MTN_K512_PERSIST=0" "X=1" "MTN_K512_PERSIST=0"; do one $v; done
for v in "X=1" "MTN_K512_PERSIST=0"; do one $v --batch-per-gpu 64; done
cat $O
timeout -k 5 900 python -m pytest tests/test_full_size_gpu.py tests/test_model_gpu.py tests/test_decode_gpu.py -x -q 2>&1 | tail -4 | tee -a $O
