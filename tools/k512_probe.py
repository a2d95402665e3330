"""The memories' K|V projection launch (K = 512, N = 1024, 16 problems of the cfg2 step) on both forms of csrc/gemm_k512.hip:
microseconds per launch (40 launches between one event pair, after a 1 GiB fill so that the first is cold), cold single launch.
Ablation libraries (tools/build_variant.sh, SRC=gemm_k512): MTN_HIP_LIB=tools/libmtn_hip_k_NO_MFMA.so etc.

    python tools/k512_probe.py [--batch 32]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    from mtn_amd import lib as L
    lib = L.load()
    dev = torch.device("cuda:0")
    B = args.batch
    rows = [B * 128, B * 40, B * 20, B * 32, B * 32]                 # history, caption, query, two feature streams
    Ms = (rows * 4)[:16]
    probs = (L.GemmProblem * len(Ms))()
    keep = []
    for i, M in enumerate(Ms):
        a = (torch.randn(M, 512, device=dev) * 0.5).to(torch.bfloat16)
        w = (torch.randn(1024, 512, device=dev) * 0.05).to(torch.bfloat16)
        b = torch.randn(1024, device=dev)
        out = torch.zeros(M * 1024 + 8192, device=dev, dtype=torch.bfloat16)       # (+ room for the timing build's stamps)
        p = probs[i]
        p.A, p.B, p.lda, p.ldb, p.M, p.N, p.K = a.data_ptr(), w.data_ptr(), 512, 512, M, 1024, 512
        p.bias, p.gate_scale, p.out_lp, p.ldc = b.data_ptr(), 1.0, out.data_ptr(), 1024
        keep += [a, w, b, out]
    flops = sum(2.0 * M * 1024 * 512 for M in Ms)
    junk = torch.empty(1 << 28, device=dev)
    st = L.stream_ptr()
    for form in ("persistent", "tile per workgroup"):
        os.environ["MTN_K512_PERSIST"] = "1" if form == "persistent" else "0"
        L.reload_env()
        for _ in range(3):
            L.check(lib.mtn_gemm(L.MTN_BF16, len(Ms), probs, st))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cold = []
        for _ in range(5):
            junk.fill_(1.0)
            e0.record(); L.check(lib.mtn_gemm(L.MTN_BF16, len(Ms), probs, st)); e1.record()
            torch.cuda.synchronize()
            cold.append(e0.elapsed_time(e1) * 1e3)
        e0.record()
        for _ in range(40):
            L.check(lib.mtn_gemm(L.MTN_BF16, len(Ms), probs, st))
        e1.record()
        torch.cuda.synchronize()
        warm = e0.elapsed_time(e1) * 1e3 / 40
        print(f"{form:20s}: back to back {warm:7.2f} us = {flops / warm / 1e6:6.1f} TFLOP/s | after a 1 GiB fill (incl. event pair) {min(cold):7.2f} us   [{len(Ms)} problems, {sum(Ms)} rows, {flops / 1e9:.1f} GFLOP]")
    os.environ.pop("MTN_K512_PERSIST", None)
    L.reload_env()
    if os.environ.get("MTN_HIP_LIB", "").endswith("STAMPS.so"):
        os.environ["MTN_K512_PERSIST"] = "1"
        L.reload_env()
        keep[3].zero_()
        L.check(lib.mtn_gemm(L.MTN_BF16, len(Ms), probs, st))
        torch.cuda.synchronize()
        raw = keep[3][Ms[0] * 1024:].view(torch.int64).cpu()
        for name, base in (("wave 0 (brings the x images)", 0), ("wave 4 (stores the tiles)", 512)):
            t = [int(v) for v in raw[base:base + 500] if int(v) != 0]
            if len(t) < 4:
                continue
            t0 = t[0]
            us = lambda c: (c - t0) / 100.0             # s_memrealtime-style 100 MHz counter
            print(f"{name}: first unit found +0.00, W + two images landed +{us(t[1]):.2f} us")
            k = 2
            n = 0
            while k + 4 < len(t):
                a, b, c, d, e = (us(x) for x in t[k:k + 5])
                print(f"   unit {n:2d}: image wait over {a:7.2f} | barrier A {b:7.2f} | MFMAs issued {c:7.2f} | staged {d:7.2f} | barrier B {e:7.2f}")
                k += 5; n += 1


if __name__ == "__main__":
    main()
