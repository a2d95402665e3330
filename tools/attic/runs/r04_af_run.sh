set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r04_af_k512_wide_ab.txt
rm -f $O
echo "# K|V projection launches (K = 512): 128 x 256 tiles, two column blocks per wave (gemm_k512w_kernel, default) against 128 x 128 (MTN_K512_WIDE=0)" >> $O
timeout -k 5 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -k "k512 or golden or gradient" 2>&1 | tail -3 | tee -a $O
for w in 1 0; do echo "== MTN_K512_WIDE=$w (tools/k512_probe.py)" >> $O; MTN_K512_WIDE=$w timeout 120 python tools/k512_probe.py 2>&1 | grep "tile per workgroup\|persistent" >> $O; done
one() {
  v="$1"; shift
  echo "== $v  $*" >> $O
  env $v timeout -k 5 200 python bench.py --no-cpu-baseline --no-secondary --no-record --windows 1 --steps 30 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('samples/s', d['value'], 'step ms', d['ms_per_step'])
for k,v in r['kernels'].items():
    if 'k512' in k: print('   ', k, 'launches', v.get('launches_per_step'), 'avg us', v.get('avg_us'), 'TFLOP/s', v.get('achieved_TFLOPs'))
" >> $O 2>&1
}
for v in "X=1" "MTN_K512_WIDE=0" "X=1" "MTN_K512_WIDE=0"; do one "$v"; done
for v in "X=1" "MTN_K512_WIDE=0" "MTN_K512_PERSIST=0" ; do one "$v" --batch-per-gpu 64; done
cat $O
