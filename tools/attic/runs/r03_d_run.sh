set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 5 200 python -m pytest tests/test_fused_gpu.py -m gpu -q -k "directly or layer_gradients" 2>&1 | tail -8 > gpurun_out/r03_d_pytest.txt
tail -4 gpurun_out/r03_d_pytest.txt
timeout -k 5 200 python tools/overlap_cu_mask_probe.py > gpurun_out/r03_overlap_cu_mask.txt 2>&1
tail -7 gpurun_out/r03_overlap_cu_mask.txt
timeout -k 5 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_d_bench.json 2> gpurun_out/r03_d_bench.err
tail -1 gpurun_out/r03_d_bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print(d['config']['secondary']['dp_schedule_one_rank']); r=d['roofline']; print(r['frac'], r['peak_measured'], r['frac_of_measured_peak'], r['avg_us_per_launch'])"
B="python bench.py --no-cpu-baseline --no-secondary --windows 2 --steps 20"
rm -f gpurun_out/r03_d_ab.txt
for v in "X=1" "MTN_GEMM_NTB_MIN_TILES=150" "X=1" "MTN_GEMM_NTB_MIN_TILES=150"; do
  echo "== cfg3 $v" >> gpurun_out/r03_d_ab.txt
  env $v timeout -k 5 150 $B --workload cfg3 --gpus 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['window_ms_per_step'])" >> gpurun_out/r03_d_ab.txt 2>&1
done
echo "== cfg4" >> gpurun_out/r03_d_ab.txt
timeout -k 5 150 $B --workload cfg4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['window_ms_per_step'])" >> gpurun_out/r03_d_ab.txt 2>&1
cat gpurun_out/r03_d_ab.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d /tmp/c4_stats -- python $R/bench.py --workload cfg4 --no-cpu-baseline --no-secondary --windows 0 --steps 10 > /tmp/c4.log 2>&1
python $R/tools/prof_breakdown.py /tmp/c4_stats 60 $R/gpurun_out/r03_d_cfg4_step_sequence.txt > $R/gpurun_out/r03_d_cfg4_one_step_breakdown.txt
head -30 $R/gpurun_out/r03_d_cfg4_one_step_breakdown.txt
timeout -k 5 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/cal_f -- python $R/tools/pmc_gemm_calib.py > /tmp/cal_f.log 2>&1
timeout -k 5 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/cal_w -- python $R/tools/pmc_gemm_calib.py > /tmp/cal_w.log 2>&1
python $R/tools/pmc_gemm_calib.py --summarise /tmp/cal_f /tmp/cal_w > $R/gpurun_out/r03_pmc_gemm_calibration.txt 2>&1
cat $R/gpurun_out/r03_pmc_gemm_calibration.txt
