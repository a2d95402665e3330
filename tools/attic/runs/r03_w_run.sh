set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r03_w_ab.txt
rm -f $O
timeout -k 5 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" 2>&1 | tail -3 | tee -a $O
timeout -k 5 300 python tools/nt_gemm_probe.py > gpurun_out/r03_nt_gemm_probe4.txt 2>&1
grep -A6 "^dmem\|^generator" gpurun_out/r03_nt_gemm_probe4.txt | cut -c1-260 >> $O
one() {  # $1 = env assignment, $2.. = bench flags
  v=$1; shift
  echo "== $v  $*" >> $O
  env $v timeout -k 5 200 python bench.py --no-cpu-baseline --no-secondary --no-record --windows 1 --steps 30 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('samples/s', d['value'], 'step ms', d['ms_per_step'])" >> $O 2>&1
}
for v in "X=1" "MTN_GEMM_128X_MIN_TILES=0" "X=1" "MTN_GEMM_128X_MIN_TILES=0"; do one $v; done
for v in "X=1" "MTN_GEMM_128X_MIN_TILES=0" "X=1" "MTN_GEMM_128X_MIN_TILES=0"; do one $v --batch-per-gpu 64; done
cat $O
