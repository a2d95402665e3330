# Builds tools/libmtn_hip_ablate.so: the library with -DMTN_TT_ABLATION (the parameter-gradient + optimiser table launch can skip
# its contraction, MTN_TT_ABLATE=2, or its optimiser epilogue, MTN_TT_ABLATE=4).  Timing only — the results are wrong by design.
#   bash tools/tt_ablation.sh && MTN_HIP_LIB=tools/libmtn_hip_ablate.so MTN_TT_ABLATE=2 python bench.py --no-cpu-baseline --no-secondary
set -e
cd "$(dirname "$0")/.."
OBJ=/tmp/mtn_ablate_obj; mkdir -p $OBJ
for f in gemm layernorm attention fused fused_bwd elementwise sublayer losshead assemble select gemm_k512; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DMTN_TT_ABLATION -c mtn_amd/csrc/$f.hip -o $OBJ/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/libmtn_hip_ablate.so $OBJ/*.o
ls -la tools/libmtn_hip_ablate.so
