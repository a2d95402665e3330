// gemm_big_bench.hip — VERDICT r5 "Next round" item 2: a bf16 GEMM for the path's LARGE problems (M >= 2048) that is not LDS-read-bound:
// v_mfma_f32_32x32x16_bf16, 64 x 64 per wave (four accumulator tiles), 256 x 128 per workgroup, three LDS-DMA stages of 64 contraction
// elements (two in flight under the one being multiplied), output re-laid out through LDS into whole 128-byte row segments.
// C[M][N] = A[M][K] . B[N][K]^T + bias, bf16 in / fp32 accumulate / bf16 out (the forward Linear layout: the memories' K|V hoists,
// mtn.py:256-258 at 4 000-16 000 rows).  Stand-alone: timed next to the library's own dispatch (mtn_gemm -> gemm_k512 / gemm_dma128x) on the
// same operands, cold (operands rotate through > 256 MiB) and warm, and checked against a host fp64 product on sampled outputs.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_big_bench.hip -Lmtn_amd -lmtn_hip -Wl,-rpath,$PWD/mtn_amd -o tools/gemm_big_bench.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../include/mtn_hip.h"

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((address_space(3))) void lds_void_t;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)

static constexpr int BM = 256, BN = 128, BK = 64;
static constexpr int A_STAGE = BM * BK * 2, B_STAGE = BN * BK * 2, STAGE = A_STAGE + B_STAGE;      // 32 + 16 = 48 KiB
static constexpr int NSTAGE = 3, LDS_BYTES = NSTAGE * STAGE;                                      // 144 KiB
// NW waves per workgroup: 8 (wave tile 64 x 64, four accumulator tiles) or 16 (64 x 32, two).  A CU accepts about one 1 KiB load per wave
// every ~180 ns (HISTORY.md section 10), so its L2 -> LDS fill rate is its resident waves x ~5.6 GB/s: 8 waves pull ~45 GB/s, 16 ~90.

__device__ __forceinline__ bf16_t d_f2bf(float f) { uint32_t u = __float_as_uint(f); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }
static bf16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }
static float bf2f(bf16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

struct BigArgs { const bf16_t* A; const bf16_t* B; const float* bias; bf16_t* C; int M, N, K, lda, ldb, ldc, tiles_m, tiles_n;
                 int abl; unsigned long long* dbg; };   // ablation (timing only, GEMM_BIG_ABL=bits): 1 no operand loads, 2 no MFMA, 4 no output stores, 8 no LDS fragment reads

// rows of 128 bytes (64 bf16 of the contraction); 16-byte chunk c of row r sits in slot c ^ ((r >> 1) & 7): the 16 lanes one LDS cycle of a
// ds_read_b128 serves (rows {0-3, 12-15, 20-27} + 32 j of one chunk) then touch 16 different 16-byte slots of the 256-byte bank sweep
template <int NW>
__device__ __forceinline__ void dma_stage(const __amdgpu_buffer_rsrc_t rA, const __amdgpu_buffer_rsrc_t rB, unsigned char* st, const BigArgs& P,
                                          const int row0, const int col0, const int k0, const int wave, const int lane) {
    constexpr int NDMA = STAGE / 1024 / NW;
    const int rl = lane >> 3, slot = lane & 7;
#pragma unroll
    for (int j = 0; j < NDMA; ++j) {
        const int inst = j * NW + wave;                       // 1 KiB = 8 rows each; instructions 0..31: A rows, 32..47: B rows
        const bool isA = inst < A_STAGE / 1024;
        const int r = (isA ? inst : inst - A_STAGE / 1024) * 8 + rl;
        const int c = slot ^ ((r >> 1) & 7);
        const int gk = k0 + c * 8;
        unsigned voff;
        if (isA) { const int gr = row0 + r; voff = (gr < P.M && gk < P.K) ? (unsigned)gr * (unsigned)(P.lda * 2) + (unsigned)gk * 2u : 0x80000000u; }
        else { const int gc = col0 + r; voff = (gc < P.N && gk < P.K) ? (unsigned)gc * (unsigned)(P.ldb * 2) + (unsigned)gk * 2u : 0x80000000u; }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(isA ? rA : rB, (lds_void_t*)(st + inst * 1024), 16, voff, 0, 0, 0);
    }
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void gemm_big_kernel(const BigArgs P) {
    constexpr int NDMA = STAGE / 1024 / NW;                        // LDS-DMA instructions per wave and stage: 6 / 3
    constexpr int WCOLS = NW / 4, WN = BN / WCOLS, TJ = WN / 32;   // wave grid 4 x WCOLS, wave tile 64 x WN, TJ column tiles of 32
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware tile order: workgroup ids go to the 8 XCDs round-robin; one XCD takes a contiguous band of row tiles x ALL column tiles,
    // so an A row panel (256 x K) is pulled into one L2 only and B (N x K) once per XCD
    const int T = P.tiles_m * P.tiles_n;
    int idx = blockIdx.x;
    if (T >= 16) { const int c = idx & 7, r = idx >> 3, per = T >> 3, rem = T & 7; idx = c * per + (c < rem ? c : rem) + r; }
    const int tm = idx / P.tiles_n, tn = idx - tm * P.tiles_n;
    const int row0 = tm * BM, col0 = tn * BN;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)P.A, 0, (int)((size_t)P.M * P.lda * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)P.B, 0, (int)((size_t)P.N * P.ldb * 2), 0x00020000);
    const int nst = (P.K + BK - 1) / BK;
#define STAMP(k) do { if (P.dbg && tid == 0) P.dbg[(size_t)blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
    STAMP(0);
    const bool no_ld = P.abl & 1, no_mma = P.abl & 2, no_st = P.abl & 4, no_rd = P.abl & 8;
    if (!no_ld) {
        dma_stage<NW>(rA, rB, smem, P, row0, col0, 0, wave, lane);
        if (nst > 1) dma_stage<NW>(rA, rB, smem + STAGE, P, row0, col0, BK, wave, lane);
    }
    STAMP(1);
    // wave grid 4 (rows) x WCOLS (columns): wave tile 64 x WN = 2 x TJ MFMA tiles of 32 x 32.  The product is formed TRANSPOSED (first operand
    // = B rows = output columns), so that a lane's 16 accumulators of a tile are 4 runs of 4 consecutive output COLUMNS of one output row.
    const int wr = wave / WCOLS, wc = wave % WCOLS;
    const int l31 = lane & 31, kg = lane >> 5;
    f32x16_t acc[2][TJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // fragment addresses inside a stage (bytes): row r, 16-byte chunk c = 2 ks + kg
    unsigned aoff[2], boff[TJ];
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int r = wr * 64 + i * 32 + l31; aoff[i] = (unsigned)(r * 128); }
#pragma unroll
    for (int j = 0; j < TJ; ++j) { const int r = wc * WN + j * 32 + l31; boff[j] = (unsigned)(A_STAGE + r * 128); }
    const unsigned swz_a0 = (unsigned)(((wr * 64 + l31) >> 1) & 7), swz_b0 = (unsigned)(((wc * WN + l31) >> 1) & 7);      // (+32 rows: same (r >> 1) & 7)
    const unsigned lds0 = (unsigned)(size_t)smem;
    for (int s = 0; s < nst; ++s) {
        // stage s has landed once at most the NDMA instructions of stage s + 1 are still in flight (loads return in order)
        if (s + 1 < nst && !no_ld) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // everybody's share of stage s is in LDS; everybody is done reading stage s - 1
        if (s == 0) STAMP(2);
        if (s + 2 < nst && !no_ld) dma_stage<NW>(rA, rB, smem + ((s + 2) % NSTAGE) * STAGE, P, row0, col0, (s + 2) * BK, wave, lane);
        const unsigned sb = lds0 + (unsigned)((s % NSTAGE) * STAGE);
        u32x4_t fa[2][2], fb[2][TJ];                               // ping-pong fragment registers
        auto rd = [&](int buf, int ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i) { const unsigned ad = sb + aoff[i] + (((unsigned)(2 * ks + kg) ^ swz_a0) << 4); asm volatile("ds_read_b128 %0, %1" : "=v"(fa[buf][i]) : "v"(ad)); }
#pragma unroll
            for (int j = 0; j < TJ; ++j) { const unsigned ad = sb + boff[j] + (((unsigned)(2 * ks + kg) ^ swz_b0) << 4); asm volatile("ds_read_b128 %0, %1" : "=v"(fb[buf][j]) : "v"(ad)); }
        };
        if (no_rd) {
#pragma unroll
            for (int b_ = 0; b_ < 2; ++b_) {
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[b_][i] = u32x4_t{sb, sb + 1, sb + 2, sb + 3};
#pragma unroll
                for (int j = 0; j < TJ; ++j) fb[b_][j] = u32x4_t{sb, sb + 5, sb + 6, sb + 7};
            }
        } else rd(0, 0);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int cur = ks & 1;
            if (no_rd) { }
            else if (ks + 1 < BK / 16) { rd(cur ^ 1, ks + 1); asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 + TJ) : "memory"); }
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(fa[cur][i]));
#pragma unroll
            for (int j = 0; j < TJ; ++j) asm volatile("" : "+v"(fb[cur][j]));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    if (!no_mma) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8_t*)&fb[cur][j], *(const bf16x8_t*)&fa[cur][i], acc[i][j], 0, 0, 0);
        }
    }
    STAMP(3);
    // ---- epilogue: + bias, -> bf16, the wave's 64 x WN sub-tile re-laid out through its own LDS (row pitch WN * 2 + 16 B), then whole row
    // segments (WN * 2 bytes) leave as 16-byte stores.  The lane's bias quads go out together (one round trip), before the barrier.
    float4 b4s[TJ][4];
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int gc = col0 + wc * WN + j * 32 + q * 8 + kg * 4;
            gc = gc + 3 < P.N ? gc : 0;                            // (columns past N are never stored)
            b4s[j][q] = P.bias ? *(const float4*)(P.bias + gc) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    __syncthreads();                                               // the last stage is dead
    STAMP(4);
    constexpr int PITCH = WN * 2 + 16;
    unsigned char* mine = smem + wave * (64 * PITCH);
    // acc[i][j][e]: output row (within the wave tile) i * 32 + l31, output column j * 32 + (e / 4) * 8 + kg * 4 + e % 4
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cw = j * 32 + q * 8 + kg * 4;                // first of 4 consecutive columns
            const float4 b4 = b4s[j][q];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned lo = (unsigned)d_f2bf(acc[i][j][q * 4 + 0] + b4.x) | ((unsigned)d_f2bf(acc[i][j][q * 4 + 1] + b4.y) << 16);
                const unsigned hi = (unsigned)d_f2bf(acc[i][j][q * 4 + 2] + b4.z) | ((unsigned)d_f2bf(acc[i][j][q * 4 + 3] + b4.w) << 16);
                *(uint2*)(mine + (i * 32 + l31) * PITCH + cw * 2) = make_uint2(lo, hi);
            }
        }
    }
    // (same wave wrote and reads: no workgroup barrier; the LDS counter orders them)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    STAMP(5);
    constexpr int LPR = WN / 8, RPI = 64 / LPR;                   // lanes per row (16 bytes each), rows per wave-instruction
#pragma unroll
    for (int p = 0; p < 64 / RPI; ++p) {
        const int r = p * RPI + lane / LPR, c8 = lane % LPR;
        const int gr = row0 + wr * 64 + r, gc = col0 + wc * WN + c8 * 8;
        const uint4 v = *(const uint4*)(mine + r * PITCH + c8 * 16);
        if (gr < P.M && gc < P.N && !(no_st && v.x != 0x12345u)) *(uint4*)(P.C + (size_t)gr * P.ldc + gc) = v;       // (N % 8 == 0)
    }
    STAMP(6);
    if (P.dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); STAMP(7); }
}

static int g_nw = 16;
static unsigned long long* g_dbg = nullptr;
static int launch_big(const bf16_t* A, const bf16_t* B, const float* bias, bf16_t* C, int M, int N, int K, hipStream_t st) {
    static const int abl = getenv("GEMM_BIG_ABL") ? atoi(getenv("GEMM_BIG_ABL")) : 0;
    BigArgs P{A, B, bias, C, M, N, K, K, K, N, (M + BM - 1) / BM, (N + BN - 1) / BN, abl, g_dbg};
    if (g_nw == 8) hipLaunchKernelGGL(gemm_big_kernel<8>, dim3(P.tiles_m * P.tiles_n), dim3(512), LDS_BYTES, st, P);
    else hipLaunchKernelGGL(gemm_big_kernel<16>, dim3(P.tiles_m * P.tiles_n), dim3(1024), LDS_BYTES, st, P);
    return 0;
}

int main() {
    CK(hipFuncSetAttribute((const void*)gemm_big_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)gemm_big_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    hipStream_t st; CK(hipStreamCreate(&st));
    struct S { int M, N, K; const char* what; } shapes[] = {
        {8192, 1024, 512, "VERDICT shape 1 (K|V hoist of a 8 192-row memory)"}, {5120, 512, 2048, "VERDICT shape 2 (w_2-like, long contraction)"},
        {12032, 1024, 512, "cfg3 batch 64: history + caption + query K|V hoist rows"}, {4096, 1024, 512, "cfg4: one memory's K|V"},
        {2048, 512, 2048, "batch 64 feature Linear"}, {16384, 1024, 512, "batch 128"}};
    printf("# gemm_big_kernel (32x32x16 MFMA, 256x128 per workgroup, 3 LDS-DMA stages of 48 KiB; 8 waves: 64x64 per wave, 16 waves: 64x32) vs the library's dispatch (mtn_gemm), bf16, + bias, bf16 out\n");
    for (auto s : shapes) {
        const size_t ab = (size_t)s.M * s.K * 2, bb = (size_t)s.N * s.K * 2, cb = (size_t)s.M * s.N * 2;
        const int NR = (int)std::max<size_t>(2, (300ull << 20) / (ab + bb + cb) + 1);      // rotate through > 256 MiB: cold operands
        std::vector<bf16_t*> A(NR), B(NR), C(NR);
        std::vector<bf16_t> ha(ab / 2), hb(bb / 2);
        uint32_t r = 777u + s.M;
        auto rnd = [&]() { r = r * 1664525u + 1013904223u; return ((r >> 8) & 0xffff) / 65536.0f - 0.5f; };
        for (auto& v : ha) v = f2bf(rnd());
        for (auto& v : hb) v = f2bf(rnd() * 0.2f);
        std::vector<float> hbias(s.N);
        for (auto& v : hbias) v = rnd();
        float* bias; CK(hipMalloc(&bias, s.N * 4)); CK(hipMemcpy(bias, hbias.data(), s.N * 4, hipMemcpyHostToDevice));
        for (int i = 0; i < NR; ++i) {
            CK(hipMalloc(&A[i], ab)); CK(hipMalloc(&B[i], bb)); CK(hipMalloc(&C[i], cb));
            CK(hipMemcpy(A[i], ha.data(), ab, hipMemcpyHostToDevice)); CK(hipMemcpy(B[i], hb.data(), bb, hipMemcpyHostToDevice));
        }
        // correctness: 4096 sampled outputs against an fp64 product, and the library's output on the same operands
        launch_big(A[0], B[0], bias, C[0], s.M, s.N, s.K, st);
        CK(hipStreamSynchronize(st));
        std::vector<bf16_t> hc(cb / 2);
        CK(hipMemcpy(hc.data(), C[0], cb, hipMemcpyDeviceToHost));
        double worst = 0, ref_max = 0;
        for (int t = 0; t < 4096; ++t) {
            r = r * 1664525u + 1013904223u; const int m = (r >> 4) % s.M;
            r = r * 1664525u + 1013904223u; const int n = (t < 256) ? (s.N - 1 - (t % 64)) : (r >> 4) % s.N;
            double acc = hbias[n];
            for (int k = 0; k < s.K; ++k) acc += (double)bf2f(ha[(size_t)m * s.K + k]) * (double)bf2f(hb[(size_t)n * s.K + k]);
            worst = std::max(worst, fabs(acc - (double)bf2f(hc[(size_t)m * s.N + n])));
            ref_max = std::max(ref_max, fabs(acc));
        }
        auto time_it = [&](bool big, bool cold) {
            const int iters = 60;
            auto one = [&](int i) {
                const int k = cold ? i % NR : 0;
                if (big) launch_big(A[k], B[k], bias, C[k], s.M, s.N, s.K, st);
                else {
                    mtn_gemm_problem p; memset(&p, 0, sizeof(p));
                    p.A = A[k]; p.B = B[k]; p.lda = s.K; p.ldb = s.K; p.M = s.M; p.N = s.N; p.K = s.K; p.bias = bias; p.out_lp = C[k]; p.ldc = s.N; p.gate_scale = 1.f;
                    if (mtn_gemm(MTN_BF16, 1, &p, st)) { fprintf(stderr, "mtn_gemm: %s\n", mtn_last_error()); exit(1); }
                }
            };
            for (int i = 0; i < 6; ++i) one(i);
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < iters; ++i) one(i);
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, (double)ms * 1e3 / iters);
            }
            return best;
        };
        const double gf = 2.0 * s.M * s.N * s.K * 1e-9;
        g_nw = 8;
        const double t8_c = time_it(true, true), t8_w = time_it(true, false);
        g_nw = 16;
        const double tb_c = time_it(true, true), tl_c = time_it(false, true), tb_w = time_it(true, false), tl_w = time_it(false, false);
        printf("M=%5d N=%4d K=%4d (%5.1f GFLOP, %3d tiles) %-52s | big, 16 waves: cold %6.2f us %6.1f TFLOP/s, warm %6.2f us %6.1f | 8 waves: cold %6.2f us %6.1f, warm %6.2f us %6.1f | library: cold %6.2f us %6.1f TFLOP/s, warm %6.2f us %6.1f | 16 waves / lib cold %.2fx | max err %.2e of %.1f\n",
               s.M, s.N, s.K, gf, ((s.M + BM - 1) / BM) * ((s.N + BN - 1) / BN), s.what, tb_c, gf / tb_c * 1e3, tb_w, gf / tb_w * 1e3, t8_c, gf / t8_c * 1e3, t8_w, gf / t8_w * 1e3,
               tl_c, gf / tl_c * 1e3, tl_w, gf / tl_w * 1e3, tl_c / tb_c, worst, ref_max);
        if (getenv("GEMM_BIG_TIMELINE")) {
            const int T = ((s.M + BM - 1) / BM) * ((s.N + BN - 1) / BN);
            unsigned long long* d; CK(hipMalloc(&d, (size_t)T * 64));
            for (int nw : {8, 16}) {
                g_nw = nw;
                for (int rep = 0; rep < 3; ++rep) launch_big(A[rep % NR], B[rep % NR], bias, C[rep % NR], s.M, s.N, s.K, st);      // (queue ahead: the stamped launch follows others)
                g_dbg = d;
                launch_big(A[3 % NR], B[3 % NR], bias, C[3 % NR], s.M, s.N, s.K, st);
                g_dbg = nullptr;
                CK(hipStreamSynchronize(st));
                std::vector<unsigned long long> h((size_t)T * 8);
                CK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
                unsigned long long t0 = ~0ull;
                for (int w = 0; w < T; ++w) t0 = std::min(t0, h[(size_t)w * 8]);
                const char* names[8] = {"entry", "prologue DMA issued", "stage 0 landed (first barrier)", "contraction done", "bias + barrier", "tile in LDS", "stores issued", "stores drained"};
                printf("   timeline, %d waves (us after the first workgroup's entry; median / p90 / max over %d workgroups):", nw, T);
                for (int k = 0; k < 8; ++k) {
                    std::vector<double> v;
                    for (int w = 0; w < T; ++w) v.push_back((double)(h[(size_t)w * 8 + k] - t0) / 100.0);
                    std::sort(v.begin(), v.end());
                    printf("  %s %.2f / %.2f / %.2f;", names[k], v[v.size() / 2], v[(size_t)(v.size() * 0.9)], v.back());
                }
                printf("\n");
            }
            CK(hipFree(d));
        }
        fflush(stdout);
        for (int i = 0; i < NR; ++i) { CK(hipFree(A[i])); CK(hipFree(B[i])); CK(hipFree(C[i])); }
        CK(hipFree(bias));
    }
    return 0;
}
