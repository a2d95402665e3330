set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r03_final_workloads.txt
rm -f $O
for w in cfg2 cfg3 cfg4; do
  echo "== --workload $w" >> $O
  timeout -k 5 300 python bench.py --no-cpu-baseline --no-secondary --no-record --windows 2 --steps 30 --workload $w 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('samples/s', d['value'], 'step ms', d['ms_per_step'], d['config'].get('workload','')[:80])" >> $O 2>&1
done
cat $O
