"""Calibration of the gfx950 FETCH_SIZE / WRITE_SIZE counters on the GEMM kernels' OWN access patterns (MI355X_MICROARCH.md §HBM:
"other access widths are uncalibrated: calibrate on a known byte count in your own access pattern before trusting an absolute").
tools/pmc_summary.py doubles FETCH_SIZE for every kernel because a wide coalesced stream (adam_kernel) is tallied at half; the
LDS-DMA operand loads of gemm_dma_kernel (16 B per lane, chunks XOR-swizzled inside a 512-byte row segment) and its fp32 residual
reads (16 B per lane, 64 contiguous bytes per quad) are different patterns.  Every launch below reads each operand byte ONCE from
HBM by construction (one column of tiles: N = one tile; the row operand is far beyond the Infinity Cache), so known bytes /
raw counter gives the factor for that pattern.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/cal_f -- python tools/pmc_gemm_calib.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/cal_w -- python tools/pmc_gemm_calib.py
    python tools/pmc_gemm_calib.py --summarise /tmp/cal_f /tmp/cal_w > profiles/r03_pmc_gemm_calibration.txt
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the cases must run on gemm_dma_kernel (the kernel whose traffic is in question), whatever their tile count
os.environ["MTN_GEMM_DMA_MAX_TILES"] = "100000000"
os.environ["MTN_GEMM_K512_MIN_TILES"] = "0"
os.environ["MTN_GEMM_NO_HALF"] = "1"

# name -> (M, N, K, residual fp32?, out fp32?)       A [M,K] bf16 is read once; B [N,K] is tiny; C [M,N]
CASES = {
    "dma_A_only_bf16out": (1 << 20, 64, 512, False, False),
    "dma_A_resid_f32out": (1 << 20, 64, 512, True, True),
    "dma_A_k2048_bf16out": (1 << 18, 64, 2048, False, False),
    "dma32_A_only_bf16out": (1 << 20, 32, 512, False, False),
    "dma_A_and_B_square": (8192, 8192, 512, False, False),      # 128 x 128 tiles of 64 x 64: every panel shared by 128 workgroups
}


def known_bytes(M, N, K, resid, f32):
    rd = M * K * 2 + N * K * 2 + (M * N * 4 if resid else 0)
    wr = M * N * (4 if f32 else 2)
    return rd, wr


def run():
    import torch
    from mtn_amd import lib as L, ops
    dev = torch.device("cuda:0")
    lib = L.load()
    order = []
    for name, (M, N, K, resid, f32) in CASES.items():
        A = torch.randn(M, K, device=dev).to(torch.bfloat16)
        B = torch.randn(N, K, device=dev).to(torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
        res = torch.randn(M, N, device=dev) if resid else None
        p = L.GemmProblem()
        p.A, p.B, p.lda, p.ldb, p.M, p.N, p.K, p.a_trans, p.b_trans, p.gate_scale = A.data_ptr(), B.data_ptr(), K, K, M, N, K, 0, 0, 1.0
        if f32:
            p.out_f32 = out.data_ptr()
        else:
            p.out_lp = out.data_ptr()
        p.ldc = N
        if resid:
            p.residual, p.ldr = res.data_ptr(), N
        # flush caches between cases: a 1 GiB fill
        junk = torch.empty(1 << 28, device=dev, dtype=torch.float32)
        junk.fill_(1.0)
        torch.cuda.synchronize()
        lib.mtn_census_begin()
        ops.gemm(L.MTN_BF16, [p])
        torch.cuda.synchronize()
        n = lib.mtn_census_end()
        import ctypes as C
        info = L.CensusLaunch()
        L.check(lib.mtn_census_info(0, C.byref(info)))
        order.append((name, lib.mtn_census_variant_name(info.variant).decode(), info.workgroups))
        del junk
    json.dump(order, open("/tmp/pmc_gemm_calib_order.json", "w"))
    print(order)


def summarise(fdir, wdir):
    def load(d, counter):
        f = sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True))[-1]
        rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and "gemm" in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        return rows
    F, W = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
    order = json.load(open("/tmp/pmc_gemm_calib_order.json"))
    print("# FETCH_SIZE / WRITE_SIZE (raw counter x 1024 B) against known bytes, every operand byte read once by construction")
    print("# tools/pmc_summary.py applies fetch x2.0 (calibrated on adam_kernel's wide coalesced stream) to EVERY kernel")
    print(f"{'case':26s} {'kernel':38s} {'wgs':>7s} {'known rd MB':>12s} {'raw fetch MB':>13s} {'known/raw':>10s} {'known wr MB':>12s} {'raw write MB':>13s} {'known/raw':>10s}")
    for i, (name, variant, wgs) in enumerate(order):
        M, N, K, resid, f32 = CASES[name]
        rd, wr = known_bytes(M, N, K, resid, f32)
        rf, rw = float(F[i]["Counter_Value"]) * 1024, float(W[i]["Counter_Value"]) * 1024
        print(f"{name:26s} {variant:38s} {wgs:7d} {rd / 1e6:12.1f} {rf / 1e6:13.1f} {rd / rf:10.3f} {wr / 1e6:12.1f} {rw / 1e6:13.1f} {wr / rw:10.3f}")


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2], sys.argv[3])
    else:
        run()
