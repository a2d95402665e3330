set -x
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_model_gpu.py tests/test_dp_gpu.py tests/test_dp_rccl_gpu.py tests/test_full_size_gpu.py -q 2>&1 | tail -8
timeout 600 python bench.py --no-cpu-baseline --steps 20 --dp-one-rank-probe 2>&1 | tail -1
MTN_DP_LP_GATHER=0 timeout 600 python bench.py --no-cpu-baseline --steps 20 --dp-one-rank-probe 2>&1 | tail -1
timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 30 --windows 2 2>&1 | tail -1 | cut -c1-200
