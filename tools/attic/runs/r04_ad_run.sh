set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r04_ad_kv_projection_kernels.txt
rm -f $O
echo "# the two hoisted K|V projection launches (K = 512, 3 024 tiles of 128 x 128): gemm_k512_kernel (default) against gemm_dma128x_kernel (four stages, sixteen waves; forced with MTN_GEMM_K512_MIN_TILES=0 MTN_GEMM_128X_MIN_TILES=192) and the general LDS-DMA kernels (MTN_GEMM_K512_MIN_TILES=0)" >> $O
one() {
  v="$1"; shift
  echo "== $v  $*" >> $O
  env $v timeout -k 5 200 python bench.py --no-cpu-baseline --no-secondary --no-record --windows 1 --steps 30 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('samples/s', d['value'], 'step ms', d['ms_per_step'])
for k,v in r['kernels'].items():
    if v.get('launches_per_step',0) <= 3 and v.get('avg_us',0) > 30: print('   ', k, 'launches', v.get('launches_per_step'), 'avg us', v.get('avg_us'), 'TFLOP/s', v.get('achieved_TFLOPs'))
" >> $O 2>&1
}
for v in "X=1" "MTN_GEMM_K512_MIN_TILES=0 MTN_GEMM_128X_MIN_TILES=192" "MTN_GEMM_K512_MIN_TILES=0" "X=1" "MTN_GEMM_K512_MIN_TILES=0 MTN_GEMM_128X_MIN_TILES=192"; do one "$v"; done
cat $O
