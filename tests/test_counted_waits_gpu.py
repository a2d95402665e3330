"""The fused attention kernels wait for their LDS-DMA images with COUNTED waits (`s_waitcnt vmcnt(N)`, N = the number of younger loads
the compiler emits today: csrc/fused.hip, csrc/fused_bwd.hip).  A compiler that merged, hoisted or dropped one of those loads would
let LDS be read before the image has landed.  mtn_amd/libmtn_hip_safewaits.so is the same library with full waits at those sites
(build.py build_safe_waits); the whole captured train step must come out bit-identical on both — loss, every gradient, every weight
after the fused Adam — at the benchmark's launch shapes (B = 32 and the ragged batch 64: one and two rounds of units, wide and byte masks)."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import hashlib, json, sys, torch
sys.path.insert(0, %(root)r)
from mtn_amd import make_model
from mtn_amd.synthetic import CONFIGS, synthetic_batch
from mtn_amd.train_step import TrainStep
cfg = dict(CONFIGS["cfg2"])
dev = torch.device("cuda:0")
out = {}
for B, ragged in ((32, False), (64, True)):
    torch.manual_seed(7)
    model = make_model(cfg["vocab"], cfg["vocab"], N=cfg["N"], d_model=cfg["d_model"], d_ff=cfg["d_ff"], h=cfg["h"], dropout=0.1,
                       ft_sizes=cfg["ft_sizes"], diff_encoder=True, auto_encoder_ft="query", compute_dtype=torch.bfloat16).to(dev)
    batch = synthetic_batch(cfg["vocab"], B, cfg["Q"], cfg["H"], cfg["C"], cfg["T"], cfg["frames"], cfg["ft_sizes"], device=dev, seed=3, ragged=ragged)
    step = TrainStep(model, batch, cfg["vocab"], pad=1, warmup=4000)
    losses = [float(step()) for _ in range(3)]
    torch.cuda.synchronize()
    h = hashlib.sha256()
    h.update(model._flat.detach().cpu().numpy().tobytes())
    out[f"B{B}{'_ragged' if ragged else ''}"] = {"losses": losses, "weights_sha256": h.hexdigest()}
print("RESULT " + json.dumps(out))
'''


def _run(lib_path):
    env = dict(os.environ)
    if lib_path:
        env["MTN_HIP_LIB"] = lib_path
    else:
        env.pop("MTN_HIP_LIB", None)
    p = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
    assert lines, (p.stdout[-2000:], p.stderr[-2000:])
    return json.loads(lines[-1][7:])


@pytest.mark.gpu
def test_counted_waits_equal_full_waits_bit_for_bit():
    safe = os.path.join(ROOT, "mtn_amd", "libmtn_hip_safewaits.so")
    from mtn_amd import build
    if not os.path.exists(safe) or os.path.getmtime(safe) < os.path.getmtime(build.LIB):
        build.build_safe_waits(verbose=False)      # test infrastructure, built here (not by __graft_entry__.build())
    shipped, full = _run(None), _run(safe)
    assert shipped == full, (shipped, full)
    for v in shipped.values():
        assert all(l == l and abs(l) < 1e4 for l in v["losses"])
