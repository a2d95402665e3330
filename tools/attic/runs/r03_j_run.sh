set -x
R=$GRAFT_REPO_ROOT
cd $R
rm -f gpurun_out/r03_tt_ablation.txt
for v in "0" "2" "10" "18" "26" "8" "0"; do
  echo "== MTN_TT_ABLATE=$v (2 = no contraction, 4 = no optimiser epilogue, 8 = no transposed copy, 16 = no bf16 copy)" >> gpurun_out/r03_tt_ablation.txt
  MTN_HIP_LIB=$R/tools/libmtn_hip_ablate.so MTN_TT_ABLATE=$v timeout -k 5 150 python bench.py --no-cpu-baseline --no-secondary --windows 1 --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; k=r['kernels']['gemm_tt_dma128_table_kernel']; print('step ms', d['ms_per_step'], '| table launch us', k['avg_us'], '| peak_measured GB/s', r.get('peak_measured'))" >> gpurun_out/r03_tt_ablation.txt 2>&1
done
cat gpurun_out/r03_tt_ablation.txt
