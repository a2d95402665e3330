set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r03_ab_epi.txt
rm -f $O
timeout -k 5 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" 2>&1 | tail -3 | tee -a $O
one() {  # $1 = env assignment, $2.. = bench flags
  v=$1; shift
  echo "== $v  $*" >> $O
  env $v timeout -k 5 200 python bench.py --no-cpu-baseline --no-secondary --no-record --windows 1 --steps 30 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('samples/s', d['value'], 'step ms', d['ms_per_step'])" >> $O 2>&1
}
for v in "X=1" "MTN_HIP_LIB=$R/tools/libmtn_hip_prevgemm.so" "X=1" "MTN_HIP_LIB=$R/tools/libmtn_hip_prevgemm.so"; do one $v; done
for v in "X=1" "MTN_HIP_LIB=$R/tools/libmtn_hip_prevgemm.so"; do one $v --batch-per-gpu 64; done
cat $O
