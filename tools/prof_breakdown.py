import sqlite3, glob, sys
from collections import defaultdict
db = sorted(glob.glob(sys.argv[1] + '/**/*_results.db', recursive=True))[-1]
c = sqlite3.connect(db)
rows = c.execute("select name, start, end, grid_x, workgroup_x, queue_id from kernels order by start").fetchall()
adam = [i for i,r in enumerate(rows) if 'adam_kernel' in r[0] or 'adam_chunks_kernel' in r[0]]
if len(adam) < 3:
    # round 5: the rest of the optimiser rides in the table launch (no adam_chunks launch behind it): a step ends with the table launch
    adam = [i for i,r in enumerate(rows) if 'gemm_tt_dma128_table_kernel' in r[0]]
# optimiser-to-optimiser intervals that are whole train steps (bench.py's per-kernel census replays only the GEMM launches, and
# those passes end in the table launch too); the shortest one is a replayed graph step (the eager census pass is longer)
cands = [rows[adam[i]+1:adam[i+1]+1] for i in range(len(adam)-1)]
cands = [c for c in cands if any('fused_head_fwd' in r[0] for r in c) and any('fused_head_bwd' in r[0] for r in c)][-6:]
step = min(cands, key=lambda st: st[-1][2] - st[0][1])
t0, t1 = step[0][1], step[-1][2]
iv = sorted((r[1], r[2]) for r in step)
busy = 0; cs, ce = iv[0]
for a,b in iv[1:]:
    if a > ce: busy += ce-cs; cs, ce = a,b
    else: ce = max(ce,b)
busy += ce-cs
print(f"step wall {(t1-t0)/1e3:.1f} us, kernels {len(step)}, busy {busy/1e3:.1f} us, sum {sum(r[2]-r[1] for r in step)/1e3:.1f} us")
agg = defaultdict(list)
for r in step:
    n = r[0].split('(')[0][-44:]
    agg[(n, r[3]//max(1,r[4]))].append((r[2]-r[1])/1e3)
tot = sum(sum(v) for v in agg.values())
for k,v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[2]) if len(sys.argv)>2 else 30]:
    print(f"{k[0]:46s} wgs={k[1]:6d} n={len(v):4d} avg={sum(v)/len(v):8.2f} us  total={sum(v):8.1f} ({100*sum(v)/tot:4.1f}%)")
if len(sys.argv) > 3:      # the step's launches in order: index, start offset, duration, workgroups, name
    with open(sys.argv[3], "w") as f:
        for i, r in enumerate(step):
            f.write(f"{i:3d} {(r[1]-t0)/1e3:8.1f} {(r[2]-r[1])/1e3:7.2f} us wgs={r[3]//max(1,r[4]):5d}  {r[0].split('(')[0][-60:]}\n")
