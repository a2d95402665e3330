set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for w in cfg2 cfg3; do
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pj_dp_$w -- python $R/bench.py --workload $w --no-cpu-baseline --steps 10 --dp-one-rank-probe > /tmp/pj_dp_$w.log 2>&1
tail -1 /tmp/pj_dp_$w.log | cut -c1-300
python $R/tools/prof_breakdown_dp.py /tmp/pj_dp_$w 50 $R/gpurun_out/r05_o_dp_${w}_step_sequence.txt > $R/gpurun_out/r05_o_dp_${w}_one_rank_breakdown.txt 2>&1
head -70 $R/gpurun_out/r05_o_dp_${w}_one_rank_breakdown.txt
done
