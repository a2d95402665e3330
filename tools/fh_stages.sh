# cumulative cost of the stages of fused_head_fwd_kernel (csrc/fused.hip): the bench step under rocprofv3 with the kernel
# leaving after stage MTN_FH_STOP = 1 (loads + LayerNorm), 2 (+ projections), 3 (+ stores), 0 (everything)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for st in 1 2 3 0; do
  MTN_FH_STOP=$st timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ps$st -- python $R/bench.py --no-cpu-baseline --steps 6 > /tmp/bs$st.log 2>&1
  echo "== stop $st"; python $R/tools/prof_breakdown.py /tmp/ps$st 60 | grep -E "fused_head|step wall"
done
