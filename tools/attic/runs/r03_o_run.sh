set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r03_o_ab.txt
rm -f $O
one() {  # $1 = env assignment, $2.. = bench flags
  v=$1; shift
  echo "== $v  $*" >> $O
  env $v timeout -k 5 200 python bench.py --no-cpu-baseline --no-secondary --no-record --windows 1 --steps 30 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; k=r['kernels'].get('gemm_tt_dma128_table_kernel', {}); print('samples/s', d['value'], 'step ms', d['ms_per_step'], '| table launch us', k.get('avg_us'))" >> $O 2>&1
}
for v in "X=1" "MTN_GEMM_C64H=1" "X=1" "MTN_GEMM_C64H=1"; do one $v; done
for v in "X=1" "MTN_GEMM_C64H=1" "X=1" "MTN_GEMM_C64H=1"; do one $v --batch-per-gpu 64; done
cat $O
