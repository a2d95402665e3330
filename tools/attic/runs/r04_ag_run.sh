set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r04_ag_gemm_dma_ablation.txt
true
echo "# second set: NO_STORE = no epilogue stores; NO_LOAD_NO_COMPUTE; NOTHING = neither DMA, nor MFMAs, nor stores (prologue + epilogue loads + launch)" >> $O
one() {
  v="$1"; shift
  echo "== $v" >> $O
  env $v timeout -k 5 200 python bench.py --no-cpu-baseline --no-secondary --no-record --windows 1 --steps 20 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('step ms', d['ms_per_step'])
for k,v in r['kernels'].items():
    if 'gemm_dma' in k: print('   %-44s launches %3d avg us %6.2f total %7.1f' % (k, v.get('launches_per_step'), v.get('avg_us'), v.get('total_us_per_step')))
" >> $O 2>&1
}
one "X=1"
for v in NO_STORE NO_LOAD_NO_COMPUTE NOTHING; do one "MTN_HIP_LIB=tools/libmtn_hip_gd_$v.so"; done
cat $O
