set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r04_ap_ln_lin_again.txt
rm -f $O
echo "# LayerNorm forward by linearity (MTN_LN_LIN=1: bf16 rows by LDS-DMA, statistics from the producer) re-measured on the interleaved fused forward kernel (the x rows are half of a stream-bound launch's bytes now)" >> $O
one() {
  v="$1"; shift
  echo "== $v  $*" >> $O
  env $v timeout -k 5 200 python bench.py --no-cpu-baseline --no-secondary --no-record --windows 1 --steps 30 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('samples/s', d['value'], 'step ms', d['ms_per_step'])
" >> $O 2>&1
}
for v in "X=1" "MTN_LN_LIN=1" "X=1" "MTN_LN_LIN=1" "X=1" "MTN_LN_LIN=1"; do one "$v"; done
for v in "X=1" "MTN_LN_LIN=1"; do one "$v" --batch-per-gpu 64; done
cat $O
