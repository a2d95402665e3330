set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r03_af_lnbwd8.txt
rm -f $O
timeout -k 5 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -k "layernorm or ln or model or golden" 2>&1 | tail -3 | tee -a $O
one() {  # $1 = env assignment, $2.. = bench flags
  v=$1; shift
  echo "== $v  $*" >> $O
  env $v timeout -k 5 200 python bench.py --no-cpu-baseline --no-secondary --no-record --windows 1 --steps 30 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('samples/s', d['value'], 'step ms', d['ms_per_step'])" >> $O 2>&1
}
for v in "X=1" "MTN_LN_BWD_4W=1" "X=1" "MTN_LN_BWD_4W=1"; do one $v; done
for v in "X=1" "MTN_LN_BWD_4W=1" "X=1" "MTN_LN_BWD_4W=1"; do one $v --batch-per-gpu 64; done
cat $O
