"""Build libmtn_hip.so (gfx950) in-tree with hipcc.  `python -m mtn_amd.build`."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libmtn_hip.so")
SOURCES = ["gemm.hip", "layernorm.hip", "attention.hip", "fused.hip", "fused_bwd.hip", "elementwise.hip", "sublayer.hip", "losshead.hip", "assemble.hip", "select.hip", "gemm_k512.hip", "decode.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
# The fused attention kernels wait with COUNTED s_waitcnt vmcnt(N) (N = the loads the compiler emits behind the LDS-DMA today).  The same
# library with full waits instead is the reference tests/test_counted_waits_gpu.py compares the shipped one with, bit for bit.
SAFE_LIB = os.path.join(HERE, "libmtn_hip_safewaits.so")
SAFE_SOURCES = ["fused.hip", "fused_bwd.hip"]


def _headers_mtime() -> float:
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "mtn_hip.h")]
    return max(os.path.getmtime(h) for h in hs if os.path.exists(h))


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "mtn_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = True) -> str:
    """One object per source, compiled in parallel (only the sources newer than their object, or everything when a header
    changed), then one link."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libmtn_hip.so")
    os.makedirs(OBJ, exist_ok=True)
    ht = _headers_mtime()
    todo = []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s[:-4] + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), ht):
            todo.append((src, obj))

    def cc(job):
        cmd = [hipcc] + FLAGS + ["-c", job[0], "-o", job[1]]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max(1, min(len(todo), os.cpu_count() or 1))) as ex:
        list(ex.map(cc, todo))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp"] + [os.path.join(OBJ, s[:-4] + ".o") for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


def build_safe_waits(verbose: bool = True) -> str:
    """libmtn_hip_safewaits.so: SAFE_SOURCES recompiled with -DMTN_SAFE_WAITS, every other object shared with the shipped library."""
    build(verbose=verbose)
    srcs = [os.path.join(CSRC, s) for s in SAFE_SOURCES]
    if os.path.exists(SAFE_LIB) and os.path.getmtime(SAFE_LIB) >= max(os.path.getmtime(LIB), max(os.path.getmtime(s) for s in srcs)):
        return SAFE_LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

    def cc(s):
        obj = os.path.join(OBJ, s[:-4] + "_safewaits.o")
        cmd = [hipcc] + FLAGS + ["-DMTN_SAFE_WAITS", "-c", os.path.join(CSRC, s), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(len(SAFE_SOURCES)) as ex:
        safe = list(ex.map(cc, SAFE_SOURCES))
    objs = safe + [os.path.join(OBJ, s[:-4] + ".o") for s in SOURCES if s not in SAFE_SOURCES]
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SAFE_LIB + ".tmp"] + objs)
    os.replace(SAFE_LIB + ".tmp", SAFE_LIB)
    return SAFE_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
    print(build_safe_waits())
