set -x
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_dropout_stream_gpu.py -q -x 2>&1 | grep -v "^$" | tail -40
timeout 600 python -m pytest tests/test_fused_gpu.py -q -s -k "fp64_oracle" 2>&1 | grep "worst\|passed\|failed\|Assertion" | tail
timeout 600 python -m pytest tests/test_decode_gpu.py tests/test_full_size_gpu.py -q -x 2>&1 | tail -5
MTN_FUZZ_N=80 timeout 900 python -m pytest tests/test_fused_gpu.py -q -k random_shapes 2>&1 | tail -4
./tools/seam_bench.bin
