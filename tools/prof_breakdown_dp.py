"""One step of the data-parallel schedule out of a rocprofv3 kernel trace: steps are cut at noam_tick_kernel (the optimiser kernels
run once per slice there, so prof_breakdown.py's optimiser-to-optimiser rule does not apply); besides the per-kernel table it lists
the idle gaps between consecutive kernels (segment boundaries, eager collectives) and what runs concurrently."""
import sqlite3, glob, sys
from collections import defaultdict
db = sorted(glob.glob(sys.argv[1] + '/**/*_results.db', recursive=True))[-1]
c = sqlite3.connect(db)
rows = c.execute("select name, start, end, grid_x, workgroup_x, queue_id from kernels order by start").fetchall()
ticks = [i for i, r in enumerate(rows) if 'noam_tick_kernel' in r[0]]
cands = [rows[ticks[i]:ticks[i + 1]] for i in range(max(0, len(ticks) - 6), len(ticks) - 1)]
step = min(cands, key=lambda st: st[-1][2] - st[0][1])
t0, t1 = step[0][1], step[-1][2]
iv = sorted((r[1], r[2]) for r in step)
busy = 0; cs, ce = iv[0]; gaps = []
for a, b in iv[1:]:
    if a > ce:
        busy += ce - cs; gaps.append((a - ce, ce)); cs, ce = a, b
    else:
        ce = max(ce, b)
busy += ce - cs
print(f"step (tick to tick) {(cands[-1][-1][2] - cands[-1][0][1]) / 1e3:.1f} us last, {(t1 - t0) / 1e3:.1f} us shortest; kernels {len(step)}, chip busy {busy / 1e3:.1f} us, "
      f"idle {(t1 - t0 - busy) / 1e3:.1f} us in {len(gaps)} gaps, sum of kernel durations {sum(r[2] - r[1] for r in step) / 1e3:.1f} us")
qs = defaultdict(float)
for r in step:
    qs[r[5]] += (r[2] - r[1]) / 1e3
print("per queue: " + ", ".join(f"q{k}: {v:.0f} us" for k, v in sorted(qs.items())))
print("largest gaps (us, at offset):")
for g, at in sorted(gaps, reverse=True)[:14]:
    prev = max((r for r in step if r[2] <= at + 1), key=lambda r: r[2])
    nxt = min((r for r in step if r[1] >= at + g - 1), key=lambda r: r[1])
    print(f"   {g / 1e3:7.1f} at {(at - t0) / 1e3:8.1f}   after {prev[0].split('(')[0][-40:]}   before {nxt[0].split('(')[0][-40:]}")
agg = defaultdict(list)
for r in step:
    n = r[0].split('(')[0][-44:]
    agg[(n, r[3] // max(1, r[4]))].append((r[2] - r[1]) / 1e3)
tot = sum(sum(v) for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{k[0]:46s} wgs={k[1]:6d} n={len(v):4d} avg={sum(v) / len(v):8.2f} us  total={sum(v):8.1f} ({100 * sum(v) / tot:4.1f}%)")
if len(sys.argv) > 3:
    with open(sys.argv[3], "w") as f:
        for i, r in enumerate(step):
            f.write(f"{i:3d} {(r[1] - t0) / 1e3:8.1f} {(r[2] - r[1]) / 1e3:7.2f} us q{r[5]} wgs={r[3] // max(1, r[4]):5d}  {r[0].split('(')[0][-60:]}\n")
