set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r04_aq_fh_row_bands_ab.txt
rm -f $O
echo "# fused forward: an XCD group's row blocks as a CONTIGUOUS band (the row bands of the producing GEMM's 2-D tile map: part of the x rows then sits in the reader's own L2) — tools/libmtn_hip_fh_bands.so, -DFH_XCD_ROW_BANDS — against blocks g mod sg (default)" >> $O
MTN_HIP_LIB=tools/libmtn_hip_fh_bands.so timeout -k 5 900 python -m pytest tests/test_fused_gpu.py -x -q 2>&1 | tail -2 | tee -a $O
one() {
  v="$1"; shift
  echo "== $v  $*" >> $O
  env $v timeout -k 5 200 python bench.py --no-cpu-baseline --no-secondary --no-record --windows 1 --steps 30 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('samples/s', d['value'], 'step ms', d['ms_per_step'])
" >> $O 2>&1
}
for v in "X=1" "MTN_HIP_LIB=tools/libmtn_hip_fh_bands.so" "X=1" "MTN_HIP_LIB=tools/libmtn_hip_fh_bands.so" "X=1" "MTN_HIP_LIB=tools/libmtn_hip_fh_bands.so"; do one "$v"; done
for v in "X=1" "MTN_HIP_LIB=tools/libmtn_hip_fh_bands.so"; do one "$v" --batch-per-gpu 64; done
cat $O
