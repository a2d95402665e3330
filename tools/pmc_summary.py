"""Summarise two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE runs: the TCC block has 4 slots,
FETCH_SIZE takes 3 and WRITE_SIZE 2) into profiles/<name>_pmc_traffic.json: HBM-side bytes per launch for every kernel of
one train step, with the gfx950 corrections of MI355X_MICROARCH.md §HBM:
  * counters are in KiB per dispatch (request counts x 64 B);
  * FETCH_SIZE reports exactly half of a wide (16 B/lane) coalesced read on gfx950 -> doubled;
  * WRITE_SIZE is uncalibrated -> calibrated here on adam_kernel, whose traffic is known exactly
    (reads p,g,m,v = 16 B/param; writes p,m,v + bf16 copy = 14 B/param), and the same check is reported for FETCH_SIZE.
    Round 5: the fused step has no stand-alone optimiser launch any more (it rides in the table launch), so the passes are
    collected with `bench.py --pmc-calibration`, which appends ONE mtn_adam_step over 2^24 scratch elements to the run.
usage: pmc_summary.py <fetch_dir> <write_dir> <out.json> <n_elements_of_the_calibration_launch>"""
import csv, glob, json, os, re, sys
from collections import defaultdict


def load(d, counter):
    f = sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True))[-1]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows


def short(name):
    name = re.sub(r"\(.*", "", name).replace("void ", "").replace("unsigned short", "bf16")
    return name[-70:]


def calibration(rows):
    r = [r for r in rows if "adam_kernel" in r["Kernel_Name"] or "adam_chunks_kernel" in r["Kernel_Name"]][-1]
    return float(r["Counter_Value"]) * 1024.0


def one_step(rows):
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"] or "adam_chunks_kernel" in r["Kernel_Name"]]
    if len(adam) < 3:                  # a step ends with the table launch (tools/prof_breakdown.py)
        adam = [i for i, r in enumerate(rows) if "gemm_tt_dma128_table_kernel" in r["Kernel_Name"]]
    cands = [rows[adam[i] + 1: adam[i + 1] + 1] for i in range(len(adam) - 1)]
    # whole train steps only (bench.py's per-kernel census replays just the GEMM launches, and ends in the table launch too)
    cands = [c for c in cands if any("fused_head_fwd" in r["Kernel_Name"] for r in c) and any("fused_head_bwd" in r["Kernel_Name"] for r in c)]
    return min(cands, key=len)          # a replayed graph step (bench.py's eager census pass has extra kernels)


def per_kernel(rows):
    agg = defaultdict(lambda: [0, 0.0])
    for r in rows:
        k = (short(r["Kernel_Name"]), int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"]) * 1024.0
    return agg


fetch_dir, write_dir, out, n_params = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
rows_f, rows_w = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
F, W = per_kernel(one_step(rows_f)), per_kernel(one_step(rows_w))
fetch_corr = 2.0                                                 # guide: wide coalesced reads are tallied at half
write_cal = (14.0 * n_params) / calibration(rows_w)               # known bytes / raw counter
fetch_check = (16.0 * n_params) / (fetch_corr * calibration(rows_f))
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --steps 2 --warmup 1",
       "corrections": {"unit": "counter x 1024 B", "fetch_x": fetch_corr, "write_x_calibrated_on_adam_kernel": round(write_cal, 3),
                       "fetch_known_over_corrected_on_adam_kernel": round(fetch_check, 3)},
       "kernels": {}, "step": {}}
tot_f = tot_w = 0.0
for k in sorted(set(F) | set(W), key=lambda k: -(F.get(k, [0, 0])[1] + W.get(k, [0, 0])[1])):
    nf, bf = F.get(k, [0, 0.0]); nw, bw = W.get(k, [0, 0.0])
    n = max(nf, nw)
    fb, wb = fetch_corr * bf, write_cal * bw
    tot_f += fb; tot_w += wb
    res["kernels"][f"{k[0]} wgs={k[1]}"] = {"launches_per_step": n, "fetch_bytes_per_launch": round(fb / max(1, nf)),
                                            "write_bytes_per_launch": round(wb / max(1, nw)),
                                            "hbm_bytes_per_launch": round(fb / max(1, nf) + wb / max(1, nw))}
res["step"] = {"fetch_GB": round(tot_f / 1e9, 3), "write_GB": round(tot_w / 1e9, 3), "total_GB": round((tot_f + tot_w) / 1e9, 3)}
# per kernel family (all grid sizes): what bench.py's roofline.traffic reads
fam = defaultdict(lambda: [0, 0.0])
for name, v in res["kernels"].items():
    f = name.split(" wgs=")[0]
    fam[f][0] += v["launches_per_step"]; fam[f][1] += v["hbm_bytes_per_launch"] * v["launches_per_step"]
res["families"] = {f: {"launches_per_step": n, "hbm_bytes_per_launch": round(b / n)} for f, (n, b) in sorted(fam.items(), key=lambda kv: -kv[1][1])}
# what the passes were collected on: bench.py's roofline.traffic_source compares it with the sources it runs on
import hashlib
import subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = hashlib.sha256()
for f in sorted(glob.glob(os.path.join(root, "mtn_amd", "csrc", "*"))):
    h.update(os.path.basename(f).encode())
    h.update(open(f, "rb").read())
try:
    head = subprocess.check_output(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
except Exception:
    head = os.environ.get("MTN_GIT_HEAD")          # (the GPU box has no .git: tools/prof_round.sh passes it in)
res["collected_on"] = {"csrc_sha16": h.hexdigest()[:16], "git_head": head}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["corrections"]), json.dumps(res["step"]))
for f, v in list(res["families"].items())[:16]:
    print(f"{f:72s} n={v['launches_per_step']:4d} {v['hbm_bytes_per_launch']/1e6:9.3f} MB/launch")
