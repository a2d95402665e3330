# cumulative cost of the stages of fused_head_bwd_kernel (csrc/fused_bwd.hip): the bench step under rocprofv3 with the kernel
# leaving after stage MTN_FB_STOP = 1 (loads landed), 2 (+ dO image), 3 (+ attention math, no stores), 0 (everything)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for st in 1 2 3 0; do
  MTN_FB_STOP=$st timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pb$st -- python $R/bench.py --no-cpu-baseline --no-secondary --windows 0 --steps 6 > /tmp/bb$st.log 2>&1
  echo "== stop $st"; python $R/tools/prof_breakdown.py /tmp/pb$st 60 | grep -E "fused_head_bwd|step wall"
done
