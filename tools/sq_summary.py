"""Summarise one rocprofv3 PMC pass of SQ counters per kernel (all dispatches of the run):
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d <dir> -- <cmd>
wait_* as fractions of SQ_WAVE_CYCLES, conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.   usage: sq_summary.py <dir> [header line]"""
import csv, glob, os, re, sys
from collections import defaultdict
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*_counter_collection.csv"), recursive=True))[-1]
agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
for r in csv.DictReader(open(f)):
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:46]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k].add(r["Dispatch_Id"])
if len(sys.argv) > 2:
    print("# " + sys.argv[2])
print("# wait_* as fractions of SQ_WAVE_CYCLES, conflict as a fraction of SQ_LDS_IDX_ACTIVE; all dispatches of the run")
for k, c in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"]):
    wc, act = max(c["SQ_WAVE_CYCLES"], 1.0), c["SQ_LDS_IDX_ACTIVE"]
    print(f"{k:46s} n={len(cnt[k]):5d} wave_cyc={wc/1e6:9.1f}M wait_any={c['SQ_WAIT_ANY']/wc:.2f} wait_inst={c['SQ_WAIT_INST_ANY']/wc:.2f} "
          f"lds_act={act/1e6:8.1f}M conflict={(c['SQ_LDS_BANK_CONFLICT']/act if act else 0):.2f} mfma_busy={c['SQ_VALU_MFMA_BUSY_CYCLES']/1e6:8.1f}M")
