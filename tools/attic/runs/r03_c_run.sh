set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 5 420 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r03_c_pytest.txt
tail -8 gpurun_out/r03_c_pytest.txt
timeout -k 5 240 python tools/dp_rccl_probe.py > gpurun_out/r03_c_dp_probe.txt 2>&1
tail -12 gpurun_out/r03_c_dp_probe.txt
timeout -k 5 200 python tools/overlap_cu_mask_probe.py > gpurun_out/r03_c_overlap_probe.txt 2>&1
tail -8 gpurun_out/r03_c_overlap_probe.txt
B="python bench.py --no-cpu-baseline --no-secondary --windows 2 --steps 20"
for v in "" "MTN_FH_KSPLIT=0" "MTN_FUSED=0"; do
  echo "== cfg4 $v" >> gpurun_out/r03_c_ab.txt
  env $v timeout -k 5 150 $B --workload cfg4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['window_ms_per_step'])" >> gpurun_out/r03_c_ab.txt 2>&1
done
for v in "" "MTN_GEMM_XCD2D=0" "" "MTN_GEMM_XCD2D=0"; do
  echo "== cfg2 $v" >> gpurun_out/r03_c_ab.txt
  env $v timeout -k 5 150 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['window_ms_per_step'])" >> gpurun_out/r03_c_ab.txt 2>&1
done
cat gpurun_out/r03_c_ab.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 5 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/cal_f -- python $R/tools/pmc_gemm_calib.py > /tmp/cal_f.log 2>&1
timeout -k 5 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/cal_w -- python $R/tools/pmc_gemm_calib.py > /tmp/cal_w.log 2>&1
tail -2 /tmp/cal_f.log
python $R/tools/pmc_gemm_calib.py --summarise /tmp/cal_f /tmp/cal_w > $R/gpurun_out/r03_pmc_gemm_calibration.txt 2>&1
cat $R/gpurun_out/r03_pmc_gemm_calibration.txt
