set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pj_on -- python $R/bench.py --no-cpu-baseline --no-secondary --no-record --windows 0 --steps 8 > /tmp/pj_on.log 2>&1
MTN_PREFETCH=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pj_off -- python $R/bench.py --no-cpu-baseline --no-secondary --no-record --windows 0 --steps 8 > /tmp/pj_off.log 2>&1
cd $R
python tools/prof_breakdown.py /tmp/pj_on 30 > gpurun_out/r04_ab_on.txt
python tools/prof_breakdown.py /tmp/pj_off 30 > gpurun_out/r04_ab_off.txt
paste -d'\n' gpurun_out/r04_ab_on.txt gpurun_out/r04_ab_off.txt | head -60
