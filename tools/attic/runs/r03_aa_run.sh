set -x
R=$GRAFT_REPO_ROOT
cd $R
rm -f gpurun_out/r03_fh_order.txt
for b in fh_bench_wf fh_bench; do echo "== tools/$b.bin $( [ $b = fh_bench_wf ] && echo '(every wave: weights, then LayerNorm)' || echo '(waves 0-3 weights first, waves 4-7 LayerNorm first)')" >> gpurun_out/r03_fh_order.txt; timeout -k 5 120 tools/$b.bin 2>&1 | grep -A9 "^g[0-9] " >> gpurun_out/r03_fh_order.txt; done
grep -A4 "^==\|^g[0-9] " gpurun_out/r03_fh_order.txt | grep "^==\|^g[0-9]\|median" | cut -c1-170
timeout -k 5 600 python -m pytest tests/test_fused_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -3
