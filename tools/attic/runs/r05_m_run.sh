set -x
R=$GRAFT_REPO_ROOT
cd $R
( time timeout 1500 python bench.py ) > gpurun_out/r05_m_bench_default.txt 2>&1
tail -5 gpurun_out/r05_m_bench_default.txt | cut -c1-6000
TAG=r05 bash tools/prof_round.sh > gpurun_out/r05_m_prof_round.log 2>&1
tail -40 gpurun_out/r05_m_prof_round.log
cd $R
timeout 300 python tools/decode_host_profile.py > gpurun_out/r05_m_decode_host_profile.txt 2>&1
head -50 gpurun_out/r05_m_decode_host_profile.txt
