set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r03_n_ab.txt
rm -f $O
timeout -k 5 600 python -m pytest tests/test_kernels_gpu.py tests/test_full_size_gpu.py -x -q 2>&1 | tail -5 | tee -a $O
timeout -k 5 300 python tools/nt_gemm_probe.py > gpurun_out/r03_nt_gemm_probe2.txt 2>&1
one() {  # $1 = env assignment, $2.. = bench flags
  v=$1; shift
  echo "== $v  $*" >> $O
  env $v timeout -k 5 200 python bench.py --no-cpu-baseline --no-secondary --no-record --windows 1 --steps 30 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; k=r['kernels'].get('gemm_tt_dma128_table_kernel', {}); print('samples/s', d['value'], 'step ms', d['ms_per_step'], '| table launch us', k.get('avg_us'))" >> $O 2>&1
}
for v in "X=1" "MTN_HIP_LIB=$R/tools/libmtn_hip_nopipe.so" "X=1" "MTN_HIP_LIB=$R/tools/libmtn_hip_nopipe.so"; do one $v; done
for v in "X=1" "MTN_HIP_LIB=$R/tools/libmtn_hip_nopipe.so" "X=1" "MTN_HIP_LIB=$R/tools/libmtn_hip_nopipe.so"; do one $v --batch-per-gpu 64; done
cat $O
timeout -k 5 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee -a $O
