"""Summarise a rocprofv3 results .db (kernel-trace --stats) into a small CSV for profiles/."""
import sqlite3, sys, glob, os
db = sys.argv[1]
if os.path.isdir(db):
    db = sorted(glob.glob(os.path.join(db, "**", "*_results.db"), recursive=True))[-1]
out, header = sys.argv[2], sys.argv[3:]
c = sqlite3.connect(db)
rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc limit 45").fetchall()
tot = c.execute("select sum(total_duration), sum(total_calls) from top_kernels").fetchone()
with open(out, "w") as f:
    for h in header:
        f.write("# " + h + "\n")
    f.write(f"# all kernels: {tot[1]} dispatches, {tot[0]:.0f} us total GPU kernel time\n")
    f.write("name,total_calls,total_duration_us,average_us,percentage\n")
    for r in rows:
        n = r[0] if len(r[0]) < 120 else r[0][:117] + "..."
        f.write(f"\"{n}\",{r[1]},{r[2]:.1f},{r[3]:.2f},{r[4]:.2f}\n")
print(open(out).read()[:1500])
