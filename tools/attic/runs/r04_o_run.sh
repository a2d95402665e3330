set -x
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6
MTN_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['value'], d['n_gpus'], d['config']['secondary'].get('exchange'), d['config']['secondary'].get('one_rank_no_exchange'))"
