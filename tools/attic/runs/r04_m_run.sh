set -x
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -12
timeout 900 python bench.py --no-cpu-baseline --steps 20 > gpurun_out/r04_m_bench.json 2> gpurun_out/r04_m_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_m_bench.json'))
print(d['value'], d['ms_per_step'])
s=d['config']['secondary']
print('b64', s['batch64_one_gpu']['samples_per_s'], s['batch64_one_gpu']['ms_per_step'])
print('dp', s['dp_schedule_one_rank'])
print('decode', s['decode']['beam'], s['decode']['beam_batched'])
PY
tail -3 gpurun_out/r04_m_bench.err
MTN_DP_LP_GATHER=0 timeout 600 python bench.py --no-cpu-baseline --steps 20 --dp-one-rank-probe 2>&1 | tail -2
