set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r04_v_unit_walk_ab.txt
rm -f $O
echo "# fused forward / backward kernels: launches of more units than CUs (batch 64) with ONE workgroup per CU walking the units (default) against one workgroup per unit, i.e. two rounds (MTN_FH_WALK=0)" >> $O
timeout -k 5 1500 python -m pytest tests/test_fused_gpu.py tests/test_model_gpu.py tests/test_full_size_gpu.py tests/test_ln_epilogue_gpu.py -x -q 2>&1 | tail -4 | tee -a $O
one() {  # $1 = env assignment, $2.. = bench flags
  v=$1; shift
  echo "== $v  $*" >> $O
  env $v timeout -k 5 200 python bench.py --no-cpu-baseline --no-secondary --no-record --windows 1 --steps 30 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('samples/s', d['value'], 'step ms', d['ms_per_step'])
" >> $O 2>&1
}
for v in "X=1" "MTN_FH_WALK=0" "X=1" "MTN_FH_WALK=0"; do one $v --batch-per-gpu 64; done
for v in "X=1" "MTN_FH_WALK=0" "X=1" "MTN_FH_WALK=0"; do one $v; done
for v in "X=1" "MTN_FH_WALK=0"; do one $v --batch-per-gpu 48; done
for v in "X=1" "MTN_FH_WALK=0"; do one $v --batch-per-gpu 96; done
cat $O
