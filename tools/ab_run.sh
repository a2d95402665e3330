#!/bin/bash
# Same-box A/B of the train step under environment variants, inside ONE gpurun call (box-to-box spread on the pool is +-4 %):
#   gpurun -- 'bash tools/ab_run.sh OUT.txt [--workload cfgN ...] -- "VAR=1" "VAR=0 OTHER=2" ...'
# Every variant is run twice, interleaved; one line per run: variant, ms/step of the timed window, median of five windows, average
# duration of the dominant kernel.  (The rounds' one-off scripts that produced profiles/r03_* and r04_* were under tools/attic/runs/ until round 6 — tools/attic/INDEX.md lists them:
# they name switches of their own round's library.)
OUT=$1; shift
ARGS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ARGS+=("$1"); shift; done
shift
cd "${GRAFT_REPO_ROOT:-.}"
for i in 1 2; do
  for v in "$@"; do
    env $v timeout 300 python bench.py --no-cpu-baseline --no-secondary --windows 4 --steps 30 "${ARGS[@]}" 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['config']['median_window_ms_per_step'], d['roofline'].get('avg_us_per_launch'))"
  done
done | tee "gpurun_out/$OUT"
