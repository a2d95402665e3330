set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r04_aa_prefetch_ab.txt
rm -f $O
echo "# operand prefetch (mtn_prefetch_next: a group's last GEMM launch brings the next group's weights into the L2s of the XCDs that will read them) against MTN_PREFETCH=0" >> $O
timeout -k 5 1500 python -m pytest tests/test_fused_gpu.py tests/test_model_gpu.py tests/test_kernels_gpu.py -x -q 2>&1 | tail -4 | tee -a $O
one() {  # $1 = env assignment, $2.. = bench flags
  v=$1; shift
  echo "== $v  $*" >> $O
  env $v timeout -k 5 200 python bench.py --no-cpu-baseline --no-secondary --no-record --windows 1 --steps 30 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('samples/s', d['value'], 'step ms', d['ms_per_step'])
" >> $O 2>&1
}
for v in "X=1" "MTN_PREFETCH=0" "X=1" "MTN_PREFETCH=0" "X=1" "MTN_PREFETCH=0"; do one $v; done
for v in "X=1" "MTN_PREFETCH=0" "X=1" "MTN_PREFETCH=0"; do one $v --batch-per-gpu 64; done
cat $O
