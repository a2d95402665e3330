set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r04_s_nw16_ab.txt
rm -f $O
echo "# gemm_dma_kernel<64,64,512> on sixteen waves (4 x 4 grid of 16 x 16 wave tiles) against eight: MTN_GEMM_NW16 = 1 plain launches, 2 launches with a LayerNorm epilogue, 3 both" >> $O
MTN_GEMM_NW16=3 timeout -k 5 900 python -m pytest tests/test_ln_epilogue_gpu.py tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -4 | tee -a $O
one() {  # $1 = env assignment, $2.. = bench flags
  v=$1; shift
  echo "== $v  $*" >> $O
  env $v timeout -k 5 200 python bench.py --no-cpu-baseline --no-secondary --no-record --windows 1 --steps 30 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('samples/s', d['value'], 'step ms', d['ms_per_step'])
for k,v in r['kernels'].items():
    if 'gemm_dma_kernel<64,64>' in k and 'half' not in k: print('   ', k, 'launches', v.get('launches_per_step'), 'avg us', v.get('avg_us'), 'total us', v.get('total_us_per_step'))
" >> $O 2>&1
}
for v in "X=1" "MTN_GEMM_NW16=1" "MTN_GEMM_NW16=2" "MTN_GEMM_NW16=3" "X=1" "MTN_GEMM_NW16=1" "MTN_GEMM_NW16=2" "MTN_GEMM_NW16=3"; do one $v; done
for v in "X=1" "MTN_GEMM_NW16=3" "X=1" "MTN_GEMM_NW16=3"; do one $v --batch-per-gpu 64; done
cat $O
