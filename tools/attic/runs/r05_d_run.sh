set -x
R=$GRAFT_REPO_ROOT
cd $R
one() { env $1 timeout 300 python bench.py --no-cpu-baseline --no-secondary --windows 4 --steps 30 $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', d['ms_per_step'], d['config']['median_window_ms_per_step'], d['roofline'].get('avg_us_per_launch'))"; }
{
for i in 1 2; do
  one "MTN_TT_AUX=0 MTN_STEP_HEAD=0" ""
  one "MTN_TT_AUX=0 MTN_STEP_HEAD=1" ""
  one "MTN_TT_AUX=1 MTN_STEP_HEAD=0" ""
  one "MTN_TT_AUX=1 MTN_STEP_HEAD=0 MTN_TT_AUX_GAP=0" ""
  one "MTN_TT_AUX=1 MTN_STEP_HEAD=0 MTN_TT_AUX_GAP=100000" ""
  one "MTN_TT_AUX=1 MTN_STEP_HEAD=1" ""
done
one "MTN_TT_AUX=0 MTN_STEP_HEAD=0" "--workload cfg3"
one "MTN_TT_AUX=1 MTN_STEP_HEAD=1" "--workload cfg3"
one "MTN_TT_AUX=0 MTN_STEP_HEAD=0" "--workload cfg3"
one "MTN_TT_AUX=1 MTN_STEP_HEAD=1" "--workload cfg3"
} > gpurun_out/r05_d_tail_head_ab.txt 2>&1
grep -v "^+" gpurun_out/r05_d_tail_head_ab.txt
cd /tmp && export TMPDIR=/tmp
MTN_TT_AUX=1 MTN_STEP_HEAD=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pj_new -- python $R/bench.py --no-cpu-baseline --no-secondary --windows 0 --steps 10 > /tmp/pj_new.log 2>&1
MTN_TT_AUX=0 MTN_STEP_HEAD=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pj_old -- python $R/bench.py --no-cpu-baseline --no-secondary --windows 0 --steps 10 > /tmp/pj_old.log 2>&1
cd $R
python tools/prof_breakdown.py /tmp/pj_new 60 gpurun_out/r05_d_new_step_sequence.txt > gpurun_out/r05_d_new_one_step_breakdown.txt
python tools/prof_breakdown.py /tmp/pj_old 60 gpurun_out/r05_d_old_step_sequence.txt > gpurun_out/r05_d_old_one_step_breakdown.txt
head -3 gpurun_out/r05_d_new_one_step_breakdown.txt gpurun_out/r05_d_old_one_step_breakdown.txt
head -8 gpurun_out/r05_d_new_step_sequence.txt; tail -8 gpurun_out/r05_d_new_step_sequence.txt
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -25 > gpurun_out/r05_d_pytest.txt
cat gpurun_out/r05_d_pytest.txt
