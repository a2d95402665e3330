set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MTN_HIP_LIB=$R/tools/libmtn_hip_twice.so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pj_tw -- python $R/bench.py --no-cpu-baseline --no-secondary --no-record --windows 0 --steps 6 > /tmp/pj_tw.log 2>&1
tail -1 /tmp/pj_tw.log | cut -c1-150
cd $R
python tools/twice_probe.py /tmp/pj_tw > gpurun_out/r04_y_twice_probe.txt
cat gpurun_out/r04_y_twice_probe.txt
