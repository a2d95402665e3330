set -x
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^\.\|^$" | tail -60
MTN_FUZZ_N=80 timeout 900 python -m pytest tests/test_fused_gpu.py -q -k random_shapes 2>&1 | tail -30
