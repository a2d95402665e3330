# Round 6 closing validation on the GPU box (one gpurun call): full GPU suite, smoke, default bench (terse line + full record), the profile set.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 | tail -6 > gpurun_out/r06_gpu_suite.txt; cat gpurun_out/r06_gpu_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 > gpurun_out/r06_smoke.txt; cat gpurun_out/r06_smoke.txt
timeout 900 python bench.py --full-record gpurun_out/r06_bench_full.json > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench.err; wc -c gpurun_out/r06_bench_line.json; head -c 600 gpurun_out/r06_bench_line.json; echo
TAG=r06 bash tools/prof_round.sh > gpurun_out/r06_prof_round.log 2>&1; tail -25 gpurun_out/r06_prof_round.log
timeout 300 python bench_decode.py > gpurun_out/r06_bench_decode.json 2>/dev/null; cat gpurun_out/r06_bench_decode.json
