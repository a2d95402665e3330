set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r04_am_fh_counted_wait_ab.txt
rm -f $O
echo "# fused forward: behind the LayerNorm a counted wait (the 8 NP weight loads may still fly) + raw barrier instead of vmcnt(0) + __syncthreads, weight fragments permuted step by step inside the projection loop — against the previous library" >> $O
timeout -k 5 2000 python -m pytest tests/test_fused_gpu.py tests/test_model_gpu.py tests/test_full_size_gpu.py tests/test_decode_gpu.py tests/test_random_sweeps_gpu.py -x -q 2>&1 | tail -3 | tee -a $O
one() {
  v="$1"; shift
  echo "== $v  $*" >> $O
  env $v timeout -k 5 200 python bench.py --no-cpu-baseline --no-secondary --no-record --windows 1 --steps 30 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('samples/s', d['value'], 'step ms', d['ms_per_step'])
" >> $O 2>&1
}
for v in "X=1" "MTN_HIP_LIB=tools/libmtn_hip_prev.so" "X=1" "MTN_HIP_LIB=tools/libmtn_hip_prev.so" "X=1" "MTN_HIP_LIB=tools/libmtn_hip_prev.so"; do one "$v"; done
for v in "X=1" "MTN_HIP_LIB=tools/libmtn_hip_prev.so" "X=1" "MTN_HIP_LIB=tools/libmtn_hip_prev.so"; do one "$v" --batch-per-gpu 64; done
for v in "X=1" "MTN_HIP_LIB=tools/libmtn_hip_prev.so"; do one "$v" --workload cfg4; done
cat $O
