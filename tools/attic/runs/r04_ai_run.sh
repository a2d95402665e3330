set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r04_ai_fh_interleave_lead_ab.txt
rm -f $O
echo "# fused forward kernel, interleaved weight issue: blocks 0 and 1 ahead of the first LayerNorm row group (tools/libmtn_hip_fh_lead.so, -DFH_INTL_LEAD) against block 0 only (default)" >> $O
MTN_HIP_LIB=tools/libmtn_hip_fh_lead.so timeout -k 5 900 python -m pytest tests/test_fused_gpu.py -x -q 2>&1 | tail -2 | tee -a $O
one() {
  v="$1"; shift
  echo "== $v  $*" >> $O
  env $v timeout -k 5 200 python bench.py --no-cpu-baseline --no-secondary --no-record --windows 1 --steps 30 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('samples/s', d['value'], 'step ms', d['ms_per_step'])
" >> $O 2>&1
}
for v in "X=1" "MTN_HIP_LIB=tools/libmtn_hip_fh_lead.so" "X=1" "MTN_HIP_LIB=tools/libmtn_hip_fh_lead.so" "X=1" "MTN_HIP_LIB=tools/libmtn_hip_fh_lead.so"; do one "$v"; done
for v in "X=1" "MTN_HIP_LIB=tools/libmtn_hip_fh_lead.so" "X=1" "MTN_HIP_LIB=tools/libmtn_hip_fh_lead.so"; do one "$v" --batch-per-gpu 64; done
for v in "X=1" "MTN_HIP_LIB=tools/libmtn_hip_fh_lead.so"; do one "$v" --workload cfg4; done
cat $O
