import numpy as np
import torch


def relmax(a, b):
    """max|a-b| / max(1e-6, max|b|)  — the 'relative-to-max' metric the bf16 tolerance is stated in."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / max(1e-6, float(b.abs().max())))


def absmax(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


TOL = {torch.float32: 1e-3, torch.bfloat16: 1e-2}     # BASELINE.json north_star: 1e-3 fp32 / 1e-2 bf16
DTYPES = [torch.float32, torch.bfloat16]


def lp_round(t, dtype):
    """Round a float tensor to the compute dtype (and back), to build a reference that sees the same inputs."""
    return t.to(dtype).float()
