set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r03_z_ab.txt
rm -f $O
one() {  # $1 = env assignment, $2.. = bench flags
  v=$1; shift
  echo "== $v  $*" >> $O
  env $v timeout -k 5 200 python bench.py --no-cpu-baseline --no-secondary --no-record --windows 1 --steps 30 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; k=r['kernels'].get('gemm_dma128x_kernel (128x128, four stages)', {}); print('samples/s', d['value'], 'step ms', d['ms_per_step'], '| 128x launches', k.get('launches_per_step'), 'avg us', k.get('avg_us'))" >> $O 2>&1
}
for v in "X=1" "MTN_GEMM_128X_MIN_TILES=0" "X=1" "MTN_GEMM_128X_MIN_TILES=0"; do one $v; done
for v in "X=1" "MTN_GEMM_128X_MIN_TILES=0"; do one $v --batch-per-gpu 64; done
cat $O
for b in fh_bench fh_bench_xl; do echo "== tools/$b.bin" >> gpurun_out/r03_fh_xrows.txt; timeout -k 5 120 tools/$b.bin 2>&1 | grep -A8 "^g0 \|^g4 \|^g1 " >> gpurun_out/r03_fh_xrows.txt; done
cat gpurun_out/r03_fh_xrows.txt | cut -c1-180
