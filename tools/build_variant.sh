# Builds tools/libmtn_hip_<name>.so = the library with extra compile flags on ONE source (SRC=gemm by default; the other objects
# are taken from mtn_amd/build): same-box A/B of a compile-time choice through MTN_HIP_LIB.
#   SRC=fused_bwd bash tools/build_variant.sh fbtl -DFB_TIMELINE && MTN_HIP_LIB=tools/libmtn_hip_fbtl.so python tools/fb_timeline.py
#   (the ablation switches of rounds 3-5 — MTN_TTB_NOPIPE, GP_ABLATE_*, LNE_ABL_*, ... — left the sources in round 6: tools/attic/INDEX.md)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
python -m mtn_amd.build > /dev/null
SRC=${SRC:-gemm}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c mtn_amd/csrc/$SRC.hip -o /tmp/gemm_$name.o 2>/dev/null
objs=$(ls mtn_amd/build/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/libmtn_hip_$name.so /tmp/gemm_$name.o $objs
ls -la tools/libmtn_hip_$name.so
