set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r04_w_fused_weights_nt_ab.txt
rm -f $O
echo "# fused forward kernel: the weight slice (192 KB per workgroup, read once by it, re-read by the other row blocks' workgroups of its XCD) as non-temporal loads (tools/libmtn_hip_fh_wnt.so, -DFH_W_NT) against plain loads" >> $O
one() {  # $1 = env assignment, $2.. = bench flags
  v=$1; shift
  echo "== $v  $*" >> $O
  env $v timeout -k 5 200 python bench.py --no-cpu-baseline --no-secondary --no-record --windows 1 --steps 30 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('samples/s', d['value'], 'step ms', d['ms_per_step'])
" >> $O 2>&1
}
for v in "X=1" "MTN_HIP_LIB=tools/libmtn_hip_fh_wnt.so" "X=1" "MTN_HIP_LIB=tools/libmtn_hip_fh_wnt.so"; do one $v; done
for v in "X=1" "MTN_HIP_LIB=tools/libmtn_hip_fh_wnt.so"; do one $v --batch-per-gpu 64; done
cat $O
