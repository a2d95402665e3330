set -x
R=$GRAFT_REPO_ROOT
cd $R
B="python bench.py --no-cpu-baseline --no-secondary --windows 2 --steps 20 --workload cfg3 --gpus 1"
rm -f gpurun_out/r03_g_ab.txt
for v in "X=1" "MTN_GEMM_DMA_MAX_TILES=900" "MTN_GEMM_DMA_MAX_TILES=1300" "MTN_GEMM_DMA_MAX_TILES=2100" "X=1" "MTN_GEMM_DMA_MAX_TILES=1300" "MTN_GEMM_TILE=64" "MTN_GEMM_TILE=32"; do
  echo "== cfg3 $v" >> gpurun_out/r03_g_ab.txt
  env $v timeout -k 5 150 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['window_ms_per_step'])" >> gpurun_out/r03_g_ab.txt 2>&1
done
cat gpurun_out/r03_g_ab.txt
