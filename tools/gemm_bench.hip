// Standalone micro-benchmark of the GEMM kernels (hipEvent timing, warm and cold weights).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/gemm_bench tools/gemm_bench.hip && /tmp/gemm_bench
#include "../mtn_amd/csrc/gemm.hip"
#include "../mtn_amd/csrc/elementwise.hip"
#include <vector>
#include <cstdlib>

static void fill(void* d, size_t bytes) {
    std::vector<unsigned short> h(bytes / 2);
    for (auto& x : h) x = (unsigned short)(0x3c00 + (rand() & 0x3ff) * ((rand() & 1) ? 1 : 0) + ((rand() & 1) ? 0x8000 : 0));
    hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
}

int main() {
    struct S { int M, N, K; } shapes[] = {{640, 512, 512}, {640, 1536, 512}, {640, 2048, 512}, {640, 512, 2048}, {4096, 1024, 512}, {640, 512, 1536}};
    const int NW = 48;   // rotating weight buffers: 48 x 2 MiB+ > L2, forces cold weights
    hipStream_t st; hipStreamCreate(&st);
    for (auto s : shapes) {
        void *A, *O; std::vector<void*> W(NW), AA(NW), OO(NW);
        hipMalloc(&A, (size_t)s.M * s.K * 2); fill(A, (size_t)s.M * s.K * 2);
        hipMalloc(&O, (size_t)s.M * s.N * 2);
        for (int i = 0; i < NW; ++i) { hipMalloc(&AA[i], (size_t)s.M * s.K * 2); fill(AA[i], (size_t)s.M * s.K * 2); hipMalloc(&OO[i], (size_t)s.M * s.N * 2); }
        for (auto& w : W) { hipMalloc(&w, (size_t)s.N * s.K * 2); fill(w, (size_t)s.N * s.K * 2); }
        float* bias; hipMalloc(&bias, s.N * 4); hipMemset(bias, 0, s.N * 4);
        for (int mode = 0; mode < 3; ++mode) {       // 0 warm (same W), 1 cold (rotate W), 2 rotate W, A and the output
            for (int at = 0; at < 2; ++at) {         // 0: DMA fast path, 1: b_trans register-staged path (same math, W viewed transposed)
                mtn_gemm_problem p; memset(&p, 0, sizeof(p));
                p.A = A; p.lda = s.K; p.ldb = at ? s.N : s.K; p.M = s.M; p.N = s.N; p.K = s.K; p.b_trans = at; p.bias = bias;
                p.out_lp = O; p.ldc = s.N; p.gate_scale = 1.f;
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                const int iters = 200;
                for (int i = 0; i < 10; ++i) { p.B = W[i % NW]; mtn_gemm(MTN_BF16, 1, &p, st); }
                hipEventRecord(e0, st);
                for (int i = 0; i < iters; ++i) { p.B = W[mode ? i % NW : 0]; if (mode == 2) { p.A = AA[i % NW]; p.out_lp = OO[i % NW]; } if (mtn_gemm(MTN_BF16, 1, &p, st)) { printf("ERR %s\n", mtn_last_error()); return 1; } }
                hipEventRecord(e1, st); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                double us = ms * 1e3 / iters;
                printf("M=%4d N=%4d K=%4d %s %s: %7.2f us/launch  %7.1f TFLOP/s\n", s.M, s.N, s.K, mode == 2 ? "allcold" : mode ? "cold" : "warm", at ? "regstage(b_trans)" : "dma", us,
                       2.0 * s.M * s.N * s.K / us * 1e-6);
            }
        }
        hipFree(A); hipFree(O); for (auto w : W) hipFree(w); hipFree(bias);
    }
    return 0;
}
