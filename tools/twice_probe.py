"""Upper bound of what operand prefetch could give: the library built with -DMTN_DBG_TWICE issues every fused forward / fused backward /
LDS-DMA GEMM launch twice; the second copy finds ALL of its operands (weights AND activations) in the L2s the first one pulled them
into.  From a rocprofv3 kernel trace of the bench step: duration of the first against the second copy of every pair.
    python tools/twice_probe.py <rocprof dir>"""
import sqlite3, glob, sys
from collections import defaultdict
db = sorted(glob.glob(sys.argv[1] + '/**/*_results.db', recursive=True))[-1]
c = sqlite3.connect(db)
rows = c.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
adam = [i for i, r in enumerate(rows) if 'adam_chunks_kernel' in r[0]]
step = rows[adam[-2] + 1: adam[-1] + 1]
agg = defaultdict(lambda: [0, 0.0, 0.0])
i = 0
while i + 1 < len(step):
    a, b = step[i], step[i + 1]
    if a[0] == b[0] and a[3] == b[3] and any(k in a[0] for k in ("fused_head_fwd", "fused_head_bwd", "gemm_dma_kernel")):
        k = (a[0].split('(')[0][-52:], a[3] // max(1, a[4]))
        agg[k][0] += 1; agg[k][1] += (a[2] - a[1]) / 1e3; agg[k][2] += (b[2] - b[1]) / 1e3
        i += 2
    else:
        i += 1
t1 = sum(v[1] for v in agg.values()); t2 = sum(v[2] for v in agg.values())
print(f"paired launches {sum(v[0] for v in agg.values())}: first copies {t1:.1f} us, second copies (everything in L2) {t2:.1f} us per step -> {t1 - t2:.1f} us ({100 * (t1 - t2) / t1:.1f} %) is memory latency beyond L2")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[0]:54s} wgs={k[1]:5d} n={v[0]:3d}  first {v[1] / v[0]:7.2f} us  second {v[2] / v[0]:7.2f} us  ({100 * (v[1] - v[2]) / v[1]:5.1f} %)")
