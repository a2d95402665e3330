// dW-shaped (both operands contraction-major) GEMM micro-benchmark: 64x64 vs 128x128 LDS-DMA tiles.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/tt_bench.bin tools/tt_bench.hip
#include "../mtn_amd/csrc/gemm.hip"
#include "../mtn_amd/csrc/elementwise.hip"
#include <vector>
#include <cstdlib>
static void fill(void* d, size_t bytes) {
    std::vector<unsigned short> h(bytes / 2);
    for (auto& x : h) x = (unsigned short)(0x3c00 + (rand() & 0x3ff) * ((rand() & 1) ? 1 : 0) + ((rand() & 1) ? 0x8000 : 0));
    (void)hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 50;
    struct S { int M, N, K; } shapes[] = {{1536, 512, 640}, {512, 512, 640}, {2048, 512, 640}, {512, 2048, 640}, {1024, 512, 4096}, {1536, 512, 640}, {512, 512, 640}, {2048, 512, 640}};
    hipStream_t st; (void)hipStreamCreate(&st);
    std::vector<mtn_gemm_problem> probs;
    double flops = 0;
    for (auto s : shapes) {
        void *A, *B; float *O, *rs;
        (void)hipMalloc(&A, (size_t)s.K * s.M * 2); fill(A, (size_t)s.K * s.M * 2);
        (void)hipMalloc(&B, (size_t)s.K * s.N * 2); fill(B, (size_t)s.K * s.N * 2);
        (void)hipMalloc(&O, (size_t)s.M * s.N * 4); (void)hipMalloc(&rs, s.M * 4);
        mtn_gemm_problem p; memset(&p, 0, sizeof(p));
        p.A = A; p.B = B; p.lda = s.M; p.ldb = s.N; p.M = s.M; p.N = s.N; p.K = s.K; p.a_trans = 1; p.b_trans = 1;
        p.out_f32 = O; p.ldc = s.N; p.rowsum_out = rs; p.gate_scale = 1.f;
        probs.push_back(p);
        flops += 2.0 * s.M * s.N * s.K;
    }
    for (int mode = 0; mode < 2; ++mode) {
        if (mode) setenv("MTN_GEMM_TT64", "1", 1); else unsetenv("MTN_GEMM_TT64");
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int i = 0; i < 5; ++i) if (mtn_gemm(MTN_BF16, (int)probs.size(), probs.data(), st)) { printf("ERR %s\n", mtn_last_error()); return 1; }
        (void)hipEventRecord(e0, st);
        for (int i = 0; i < iters; ++i) mtn_gemm(MTN_BF16, (int)probs.size(), probs.data(), st);
        (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.2f us/launch  %.1f TFLOP/s\n", mode ? "64x64 tiles " : "128x128 tiles", ms * 1e3 / iters, flops / (ms * 1e-3 / iters) * 1e-12);
    }
    return 0;
}
