# LayerNorm-epilogue ablation (timing only; the ablated libraries compute wrong gradients): the step under rocprofv3 with each
# piece of the consume epilogue switched off, per-launch averages of the GEMM launches that carry it.
#   (on the build box)  bash tools/lne_ablation.sh build      (on the GPU box)  bash tools/lne_ablation.sh run
cd "$(dirname "$0")/.."
VARS="nocolpart:-DLNE_ABL_NO_COLPART nolp:-DLNE_ABL_NO_LP late:-DLNE_ABL_LATE nopart:-DLNE_ABL_NO_PART nothing:-DLNE_ABL_NO_COLPART,-DLNE_ABL_NO_LP,-DLNE_ABL_NO_PART"
if [ "$1" = build ]; then
  for v in $VARS; do n=${v%%:*}; f=$(echo ${v#*:} | tr , ' '); bash tools/build_variant.sh lne_$n $f; done
  exit 0
fi
export TMPDIR=/tmp
R=$PWD
summ() {
python - "$1" <<'PY'
import sqlite3, glob, sys
from collections import defaultdict
db = sorted(glob.glob(sys.argv[1] + '/**/*_results.db', recursive=True))[-1]
rows = sqlite3.connect(db).execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
adam = [i for i, r in enumerate(rows) if 'adam_chunks_kernel' in r[0]]
cands = [rows[adam[i] + 1:adam[i + 1] + 1] for i in range(max(0, len(adam) - 5), len(adam) - 1)]
step = min(cands, key=lambda st: st[-1][2] - st[0][1])
agg = defaultdict(list)
for r in step:
    n = r[0].split('(')[0]
    if 'true>' in n and 'gemm' in n or 'fused_head_bwd' in n or 'ln_bwd_small' in n or 'ln_fold' in n:
        agg[(n[-40:], r[3] // max(1, r[4]))].append((r[2] - r[1]) / 1e3)
print(f"  step {(step[-1][2] - step[0][1]) / 1e3:.1f} us, {len(step)} kernels")
for k, v in sorted(agg.items()):
    print(f"  {k[0]:42s} wgs={k[1]:5d} n={len(v):3d} avg={sum(v) / len(v):7.2f}  total={sum(v):7.1f}")
PY
}
for n in epi0 full nocolpart nolp nopart nothing; do
  unset MTN_HIP_LIB MTN_LN_EPI
  [ $n = epi0 ] && export MTN_LN_EPI=0
  [ $n != epi0 ] && [ $n != full ] && export MTN_HIP_LIB=$R/tools/libmtn_hip_lne_$n.so
  echo "== $n"
  rm -rf /tmp/abl_$n
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/abl_$n -- python $R/bench.py --no-cpu-baseline --no-secondary --windows 0 --steps 10 > /tmp/abl_$n.log 2>&1)
  summ /tmp/abl_$n
done
