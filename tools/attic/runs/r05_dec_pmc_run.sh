set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pj_decf -- python $R/tools/decode_probe.py > /tmp/pj_decf.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pj_decw -- python $R/tools/decode_probe.py > /tmp/pj_decw.log 2>&1
cd $R
python - <<'PY' > gpurun_out/r05_decode_pmc_traffic.txt 2>&1
import csv, glob, os
def load(d, name):
    f = sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True))[-1]
    return [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == name and "decode_step_kernel" in r["Kernel_Name"]]
F, W = load("/tmp/pj_decf", "FETCH_SIZE"), load("/tmp/pj_decw", "WRITE_SIZE")
import statistics as st
for g in (64 + 32 + 128 - 32, None):
    pass
by = {}
for r in F:
    by.setdefault(int(r["Grid_Size"]) // int(r["Workgroup_Size"]), []).append(float(r["Counter_Value"]) * 1024 * 2)
print("decode_step_kernel, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python tools/decode_probe.py; KiB counters x 1024, FETCH x 2 (gfx950 correction)")
for wgs, v in sorted(by.items()):
    print(f"workgroups {wgs:4d}: launches {len(v):4d}  fetched per launch: median {st.median(v)/1e6:8.2f} MB  min {min(v)/1e6:8.2f}  max {max(v)/1e6:8.2f}")
byw = {}
for r in W:
    byw.setdefault(int(r["Grid_Size"]) // int(r["Workgroup_Size"]), []).append(float(r["Counter_Value"]) * 1024)
for wgs, v in sorted(byw.items()):
    print(f"workgroups {wgs:4d}: launches {len(v):4d}  written per launch: median {st.median(v)/1e6:8.2f} MB")
PY
cat gpurun_out/r05_decode_pmc_traffic.txt
