set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r04_ak_seed_scalar_fb_order_ab.txt
rm -f $O
echo "# (1) dropout seed as a scalar load (its consumer no longer waits for vmcnt(0) = every load / LDS-DMA in flight: all kernels with dropout), (2) fused backward: first-stage operands issued first, counted landing wait, dO stage under the late loads — against the previous library (tools/libmtn_hip_prev.so)" >> $O
timeout -k 5 2000 python -m pytest tests/test_dropout_stream_gpu.py tests/test_kernels_gpu.py tests/test_fused_gpu.py tests/test_model_gpu.py tests/test_full_size_gpu.py tests/test_ln_epilogue_gpu.py tests/test_decode_gpu.py -x -q 2>&1 | tail -3 | tee -a $O
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
