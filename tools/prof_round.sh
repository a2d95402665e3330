set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pj_stats -- python $R/bench.py --no-cpu-baseline --no-secondary --windows 0 --steps 10 > /tmp/pj_b.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pj_fetch -- python $R/bench.py --no-cpu-baseline --no-secondary --windows 0 --steps 3 --warmup 1 > /tmp/pj_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pj_write -- python $R/bench.py --no-cpu-baseline --no-secondary --windows 0 --steps 3 --warmup 1 > /tmp/pj_w.log 2>&1
# batch 64 on one GPU (cfg3's per-GPU batch) and the decode path: kernel traces only
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pj_cfg3 -- python $R/bench.py --workload cfg3 --no-cpu-baseline --no-secondary --windows 0 --steps 10 > /tmp/pj_c3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pj_dec -- python $R/tools/decode_probe.py > /tmp/pj_dec.log 2>&1
tail -2 /tmp/pj_c3.log /tmp/pj_dec.log
cd $R
NREST=$(python - <<'PY'
import torch, sys
sys.path.insert(0, ".")
from mtn_amd import make_model
from mtn_amd.synthetic import CONFIGS
cfg = CONFIGS["cfg2"]
m = make_model(cfg["vocab"], cfg["vocab"], N=cfg["N"], d_model=cfg["d_model"], d_ff=cfg["d_ff"], h=cfg["h"], dropout=0.1, ft_sizes=cfg["ft_sizes"], diff_encoder=True, auto_encoder_ft="query", compute_dtype=torch.bfloat16).cuda()
m.prepare()
print(int(m.rest_tables(frozenset(t[0] for t in m._fusable))[0][1].sum()))
PY
)
echo NREST=$NREST
python tools/prof_summary.py /tmp/pj_stats gpurun_out/${TAG:-r04_z}_kernel_stats.csv 2>&1 | tail -3
python tools/prof_breakdown.py /tmp/pj_stats 60 gpurun_out/${TAG:-r04_z}_step_sequence.txt > gpurun_out/${TAG:-r04_z}_one_step_breakdown.txt
python tools/pmc_summary.py /tmp/pj_fetch /tmp/pj_write gpurun_out/${TAG:-r04_z}_pmc_traffic.json $NREST 2>&1 | tail -20
python tools/prof_breakdown.py /tmp/pj_cfg3 60 gpurun_out/${TAG:-r04_z}_cfg3_step_sequence.txt > gpurun_out/${TAG:-r04_z}_cfg3_one_step_breakdown.txt
python tools/prof_summary.py /tmp/pj_dec gpurun_out/${TAG:-r04_z}_decode_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/decode_probe.py (beam 4 x 20 steps: 4 dialogues one at a time + greedy + 8 dialogues side by side, cfg2 model)" 2>&1 | tail -3
