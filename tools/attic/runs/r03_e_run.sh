set -x
R=$GRAFT_REPO_ROOT
cd $R
B="python bench.py --no-cpu-baseline --no-secondary --windows 2 --steps 20"
rm -f gpurun_out/r03_e_ab.txt
for v in "X=1" "MTN_FB_RING_MIN=100000" "MTN_FUSED=0" "X=1" "MTN_FB_RING_MIN=100000"; do
  echo "== cfg4 $v" >> gpurun_out/r03_e_ab.txt
  env $v timeout -k 5 150 $B --workload cfg4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['window_ms_per_step'])" >> gpurun_out/r03_e_ab.txt 2>&1
done
cat gpurun_out/r03_e_ab.txt
timeout -k 5 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_e_bench.json 2> gpurun_out/r03_e_bench.err
tail -3 gpurun_out/r03_e_bench.err
tail -1 gpurun_out/r03_e_bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print(d['config']['secondary']['dp_schedule_one_rank']); r=d['roofline']; print(r['frac'], r['peak_measured'], r['frac_of_measured_peak'], r['avg_us_per_launch'])"
