"""dX = dY W with W as it lies (b_trans = 1, [K][N] tiles + transposing LDS reads) against dX = dY (W^T)^T through a transposed
copy (b_trans = 0), on the train step's backward shapes.  Each case: 200 launches back to back between one pair of events
(operands stay in L2 / Infinity Cache: a relative figure), per tile choice.

    python tools/nt_gemm_probe.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

# (label, [(M, N, K)] grouped in one launch)
CASES = [
    ("dxn = dqkv Wqkv, 3 members  [640 x 512 x 1536] x3", [(640, 512, 1536)] * 3),
    ("dxn = dq Wq, 1 member       [640 x 512 x 512]", [(640, 512, 512)]),
    ("dh  = dyl W2                [640 x 2048 x 512]", [(640, 2048, 512)]),
    ("dxn = dh W1                 [640 x 512 x 2048]", [(640, 512, 2048)]),
    ("dmem = dkv Wkv              [8064 x 512 x 1024]", [(8064, 512, 1024)]),
    ("dmem, 3 members             [2688 x 512 x 1024] x3", [(2688, 512, 1024)] * 3),
    ("dmem at 64 samples          [16128 x 512 x 1024]", [(16128, 512, 1024)]),
    ("generator logits            [1920 x 3000 x 512]", [(1920, 3000, 512)]),
]


def main():
    from mtn_amd import lib as L, ops
    dev = torch.device("cuda:0")
    lib = L.load()
    import ctypes as C

    def build(shapes, bt):
        keep, probs = [], []
        for (M, N, K) in shapes:
            A = torch.randn(M, K, device=dev).to(torch.bfloat16)
            W = torch.randn(K, N, device=dev).to(torch.bfloat16)            # the weight as the forward pass keeps it: [out K][in N]
            B = W if bt else W.t().contiguous()
            out = torch.empty(M, N, device=dev)
            p = L.GemmProblem()
            p.A, p.B, p.lda, p.ldb, p.M, p.N, p.K, p.a_trans, p.b_trans, p.gate_scale = A.data_ptr(), B.data_ptr(), K, (N if bt else K), M, N, K, 0, bt, 1.0
            p.out_f32, p.ldc = out.data_ptr(), N
            probs.append(p)
            keep += [A, W, B, out]
        return probs, keep

    def timed(probs, reps=200):
        for _ in range(5):
            ops.gemm(L.MTN_BF16, probs)
        lib.mtn_census_begin()
        ops.gemm(L.MTN_BF16, probs)
        lib.mtn_census_end()
        info = L.CensusLaunch()
        L.check(lib.mtn_census_info(0, C.byref(info)))
        name = f"{lib.mtn_census_variant_name(info.variant).decode()} ({info.workgroups} wgs)"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            ops.gemm(L.MTN_BF16, probs)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3, name

    for label, shapes in CASES:
        print(label)
        for env in ({}, {"MTN_GEMM_TILE": "64"}, {"MTN_GEMM_TILE": "32"}, {"MTN_GEMM_TILE": "3264"}, {"MTN_GEMM_128X_MIN_TILES": "0"}, {"MTN_GEMM_128X_MIN_TILES": "1"}):
            os.environ.update(env)
            L.reload_env()
            row = []
            for bt in (0, 1):
                probs, keep = build(shapes, bt)
                us, name = timed(probs)
                row.append(f"{'W as it lies (b_trans=1)' if bt else 'through W^T   (b_trans=0)'}: {us:7.2f} us  {name}")
            for k in env:
                del os.environ[k]
            L.reload_env()
            print(f"   {str(env) if env else 'default dispatch':36s} | " + " | ".join(row))


if __name__ == "__main__":
    main()
