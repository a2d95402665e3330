set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pj_stats -- python $R/bench.py --no-cpu-baseline --no-secondary --windows 0 --steps 10 > /tmp/pj_b.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pj_fetch -- python $R/bench.py --no-cpu-baseline --no-secondary --windows 0 --steps 3 --warmup 1 --pmc-calibration > /tmp/pj_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pj_write -- python $R/bench.py --no-cpu-baseline --no-secondary --windows 0 --steps 3 --warmup 1 --pmc-calibration > /tmp/pj_w.log 2>&1
# batch 64 on one GPU (cfg3's per-GPU batch) and the decode path: kernel traces only
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pj_cfg3 -- python $R/bench.py --workload cfg3 --no-cpu-baseline --no-secondary --windows 0 --steps 10 > /tmp/pj_c3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pj_dec -- python $R/tools/decode_probe.py > /tmp/pj_dec.log 2>&1
tail -2 /tmp/pj_c3.log /tmp/pj_dec.log
cd $R
NREST=16777216      # the calibration launch of bench.py --pmc-calibration
echo NREST=$NREST
python tools/prof_summary.py /tmp/pj_stats gpurun_out/${TAG:-r05}_kernel_stats.csv 2>&1 | tail -3
python tools/prof_breakdown.py /tmp/pj_stats 60 gpurun_out/${TAG:-r05}_step_sequence.txt > gpurun_out/${TAG:-r05}_one_step_breakdown.txt
python tools/pmc_summary.py /tmp/pj_fetch /tmp/pj_write gpurun_out/${TAG:-r05}_pmc_traffic.json $NREST 2>&1 | tail -20
python tools/prof_breakdown.py /tmp/pj_cfg3 60 gpurun_out/${TAG:-r05}_cfg3_step_sequence.txt > gpurun_out/${TAG:-r05}_cfg3_one_step_breakdown.txt
python tools/prof_summary.py /tmp/pj_dec gpurun_out/${TAG:-r05}_decode_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/decode_probe.py (beam 4 x 20 steps: 4 dialogues one at a time + greedy + 8 dialogues side by side, cfg2 model)" 2>&1 | tail -3
