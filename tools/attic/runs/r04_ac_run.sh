set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_full_size_gpu.py tests/test_dp_rccl_gpu.py -x -q 2>&1 | tail -3
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pj_a -- python $R/bench.py --no-cpu-baseline --no-secondary --no-record --windows 0 --steps 8 > /tmp/pj_a.log 2>&1
cd $R
python tools/prof_breakdown.py /tmp/pj_a 40 | grep "step wall\|adam_chunks\|ln_bwd_finalize\|embed_bwd"
