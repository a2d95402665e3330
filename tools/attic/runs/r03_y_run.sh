set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 5 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r03_f_pytest.txt
cat gpurun_out/r03_f_pytest.txt
TAG=r03_f timeout -k 5 1200 bash tools/prof_round.sh > gpurun_out/r03_f_prof_round.log 2>&1
tail -5 gpurun_out/r03_f_prof_round.log
cd $R
cp gpurun_out/r03_f_pmc_traffic.json profiles/r03_f_pmc_traffic.json 2>/dev/null
timeout -k 5 600 python bench.py > gpurun_out/r03_f_bench.log 2>&1
tail -1 gpurun_out/r03_f_bench.log > gpurun_out/r03_f_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r03_f_bench.json')); r=d['roofline']
print(d['value'], d['ms_per_step'], {k:r.get(k) for k in ['achieved','frac','traffic','peak_measured','frac_of_measured_peak','avg_us_per_launch','algorithmic_bytes_per_launch']})
print(d['cpu_baseline']); print({k:(v if not isinstance(v,dict) else {kk:v[kk] for kk in list(v)[:4]}) for k,v in d['config'].get('secondary',{}).items()})
"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
