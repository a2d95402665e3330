set -x
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_fused_gpu.py tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_random_sweeps_gpu.py -m gpu -q 2>&1 | tail -30 > gpurun_out/r03_b_pytest.txt
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/dp_stats -- python $R/tools/dp_rccl_probe.py > $R/gpurun_out/r03_b_dp_probe.txt 2>&1
python $R/tools/prof_summary.py /tmp/dp_stats $R/gpurun_out/r03_b_dp_rccl_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/dp_rccl_probe.py (one-rank RCCL group, MTN_FORCE_DIST=1, cfg2 batch 32)" | tail -3
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/cal_f -- python $R/tools/pmc_gemm_calib.py > /tmp/cal_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/cal_w -- python $R/tools/pmc_gemm_calib.py > /tmp/cal_w.log 2>&1
tail -3 /tmp/cal_f.log
python $R/tools/pmc_gemm_calib.py --summarise /tmp/cal_f /tmp/cal_w > $R/gpurun_out/r03_pmc_gemm_calibration.txt 2>&1
cat $R/gpurun_out/r03_pmc_gemm_calibration.txt
cat $R/gpurun_out/r03_b_dp_probe.txt | tail -15
tail -6 $R/gpurun_out/r03_b_pytest.txt
