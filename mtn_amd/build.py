"""Build libmtn_hip.so (gfx950) in-tree with hipcc.  `python -m mtn_amd.build`."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmtn_hip.so")
SOURCES = ["gemm.hip", "layernorm.hip", "attention.hip", "elementwise.hip", "sublayer.hip", "losshead.hip", "assemble.hip"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "mtn_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libmtn_hip.so")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", LIB + ".tmp"]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
