set -x
R=$GRAFT_REPO_ROOT
cd $R
rm -f gpurun_out/r03_fh_lines.txt
for b in fh_bench_w64 fh_bench; do echo "== tools/$b.bin $( [ $b = fh_bench_w64 ] && echo '(weight loads: 16 rows x 64 B per instruction)' || echo '(weight loads: 8 rows x 128 B per instruction)')" >> gpurun_out/r03_fh_lines.txt; timeout -k 5 120 tools/$b.bin 2>&1 | grep -A9 "^g[0-9] " >> gpurun_out/r03_fh_lines.txt; done
grep "^==\|^g[0-9]\|median\|loads issued" gpurun_out/r03_fh_lines.txt | cut -c1-170
timeout -k 5 600 python -m pytest tests/test_fused_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -3
