// seam_bench.hip — what would it cost to keep the SECOND contraction of a sublayer inside the fused forward kernel?
// (BASELINE north_star: "output-proj fused per head" / "the two-GEMM FFN"; VERDICT r2 item 8: settle it with a number.)
//
// The fused forward kernel (csrc/fused.hip) leaves a workgroup = (row block, head | 256-column slice of d_ff) with its head's
// attention output o_h [rows x 64] or its slice of the FFN hidden [rows x 256] ON CHIP.  The output projection / w_2 contracts
// over ALL heads / ALL of d_ff, so finishing it in the same launch means: every workgroup multiplies its slice by the matching
// [512 x KS] panel of W_o / W_2 (on MFMA), PUBLISHES a [rows x 512] fp32 partial slab, takes a ticket, and the last arriver of a
// row block sums the 8 slabs in slice order (+ bias, + residual) — the split-K seam of the CDNA4 guide (price-list rows
// splitk-seam / publish-large / handoff-payload).  This program measures exactly that seam, stand-alone, next to the grouped GEMM
// launch it would replace (mtn_gemm from libmtn_hip.so, the launch the step runs today):
//
//   seam   : per workgroup (32 rows, slice s): slice tile -> LDS, panel fragments (coalesced loads + ds_bpermute) -> MFMA ->
//            fp32 slab by write-through (sc1) 16-byte stores -> vmcnt(0), barrier, agent-scope ticket -> last arriver: acquire,
//            8 slabs summed in order + bias + residual -> y (fp32); the ticket counter resets itself
//   launch : mtn_gemm(M = rows, N = 512, K = 8 * KS, bias, residual, fp32 out) on the same operands
// for KS = 256 (w_2: K = 2048) and KS = 64 (output projection: K = 512), rows = 640 (cfg2) and 1280 (cfg3).  The seam kernel's time
// INCLUDES staging the slice tile from memory (the fused kernel would already hold it) and a kernel launch of its own, so
// (seam - floor) is an upper bound of what the fused kernel's tail would grow by, where floor = the same kernel stopped after
// the tile staging (mode 3).  Results: profiles/r03_seam_bench.txt.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/seam_bench.hip -Lmtn_amd -lmtn_hip -Wl,-rpath,$PWD/mtn_amd -o tools/seam_bench.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../include/mtn_hip.h"

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
static constexpr int D = 512, ROWS_WG = 32, NSLICE = 8;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static bf16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }
static float bf2f(bf16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

// mode 0: everything; 1: no combine (publish + ticket only); 2: plain slab stores + __threadfence() instead of write-through;
// 3: stop after the tile staging (launch + load floor)
template <int KS>
__global__ __launch_bounds__(512) void seam_kernel(const bf16_t* __restrict__ S, int lds_, const bf16_t* __restrict__ W, int ldw,
                                                   const float* __restrict__ x, const float* __restrict__ bias, float* slabs,
                                                   unsigned* cnt, float* __restrict__ y, int rows, int mode, int same_xcd) {
    constexpr int CH = KS / 8;                       // 16-byte chunks per tile row
    constexpr int NK = KS / 32;                      // contraction steps
    __shared__ __attribute__((aligned(16))) unsigned char smem[ROWS_WG * KS * 2 + 16];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // placement 0: id % 8 = slice — an XCD streams one weight panel, the 8 slabs of a row block come from 8 XCDs (round 3);
    // placement 1 (round 4): the 8 slices of a row block on ONE XCD (workgroup id % 8 = XCD): the last
    // arriver's slab reads stay inside one L2 (guide, handoff-payload: 104-122 GB/s same-XCD with plain producer stores vs 62-70)
    int rb = blockIdx.x >> 3, slice = blockIdx.x & 7;
    if (same_xcd) { const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3; rb = (j >> 3) * 8 + xcd; slice = j & 7; }
    if (rb * ROWS_WG >= rows) return;
    const int row0 = rb * ROWS_WG;
    const int R = rows - row0 < ROWS_WG ? rows - row0 : ROWS_WG;
    // ---- the slice tile [32][KS] -> LDS (16-byte chunks swizzled with row & 7); the fused kernel holds this image already
    for (int i = tid; i < ROWS_WG * CH; i += 512) {
        const int r = i / CH, c = i - r * CH;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r < R) v = *(const uint4*)(S + (size_t)(row0 + r) * lds_ + slice * KS + c * 8);
        *(uint4*)(smem + r * (KS * 2) + (((c & ~7) | ((c ^ r) & 7)) << 4)) = v;
    }
    // ---- weight panel fragments: wave w owns output columns 64w .. 64w+63 (4 tiles of 16); coalesced: lane 4r + c reads (row r, chunk c)
    uint4 wf[4][NK];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const bf16_t* wrow = W + (size_t)(wave * 64 + nt * 16 + (lane >> 2)) * ldw + slice * KS + (lane & 3) * 8;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) wf[nt][ks] = *(const uint4*)(wrow + ks * 32);
    }
    __syncthreads();
    if (mode == 3) { if (wf[0][0].x == 0x12345678u && tid == 0) y[0] = 1.f; return; }
    {
        const int src = (4 * l15 + lg) * 4;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                wf[nt][ks].x = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)wf[nt][ks].x);
                wf[nt][ks].y = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)wf[nt][ks].y);
                wf[nt][ks].z = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)wf[nt][ks].z);
                wf[nt][ks].w = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)wf[nt][ks].w);
            }
    }
    f32x4_t acc[4][2];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt][0] = acc[nt][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        uint4 hf[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int r = mt * 16 + l15, c = ks * 4 + lg;
            hf[mt] = *(const uint4*)(smem + r * (KS * 2) + (((c & ~7) | ((c ^ r) & 7)) << 4));
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8_t*)&wf[nt][ks], *(const bf16x8_t*)&hf[mt], acc[nt][mt], 0, 0, 0);
    }
    // ---- publish the partial slab: lane holds row mt*16 + l15, four consecutive columns 64w + 16nt + 4lg .. +3
    float* slab = slabs + (size_t)(rb * NSLICE + slice) * ROWS_WG * D;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)slab, 0, ROWS_WG * D * 4, 0x00020000);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int r = mt * 16 + l15, n = wave * 64 + nt * 16 + 4 * lg;
            if (mode == 2) *(f32x4_t*)(slab + r * D + n) = acc[nt][mt];
            else {
                u32x4_t v;
                v[0] = __float_as_uint(acc[nt][mt][0]); v[1] = __float_as_uint(acc[nt][mt][1]);
                v[2] = __float_as_uint(acc[nt][mt][2]); v[3] = __float_as_uint(acc[nt][mt][3]);
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, (r * D + n) * 4, 0, 16);       // aux 16 = sc1: write-through
            }
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned* flag = (unsigned*)(smem + ROWS_WG * KS * 2);
    if (tid == 0) {
        if (mode == 2) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        *flag = __hip_atomic_fetch_add(cnt + rb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (*flag != NSLICE - 1) return;
    // ---- last arriver of the row block: acquire, sum the 8 slabs in slice order, + bias + residual
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(cnt + rb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
    }
    __syncthreads();
    if (mode == 1) return;
    const float* base = slabs + (size_t)rb * NSLICE * ROWS_WG * D;
    for (int i = tid; i < R * (D / 4); i += 512) {
        const int r = i / (D / 4), c4 = (i - r * (D / 4)) * 4;
        float4 v[NSLICE];
#pragma unroll
        for (int s = 0; s < NSLICE; ++s) v[s] = *(const float4*)(base + ((size_t)s * ROWS_WG + r) * D + c4);
        const float4 xb = *(const float4*)(x + (size_t)(row0 + r) * D + c4), bb = *(const float4*)(bias + c4);
        float4 o = v[0];
#pragma unroll
        for (int s = 1; s < NSLICE; ++s) { o.x += v[s].x; o.y += v[s].y; o.z += v[s].z; o.w += v[s].w; }
        o.x += bb.x + xb.x; o.y += bb.y + xb.y; o.z += bb.z + xb.z; o.w += bb.w + xb.w;
        *(float4*)(y + (size_t)(row0 + r) * D + c4) = o;
    }
}

template <int KS>
static float time_seam(const bf16_t* S, const bf16_t* W, const float* x, const float* bias, float* slabs, unsigned* cnt, float* y, int rows, int mode,
                       hipStream_t st, int reps, int same_xcd = 0) {
    int nrb = (rows + ROWS_WG - 1) / ROWS_WG;
    if (same_xcd) nrb = (nrb + 7) / 8 * 8;              // row blocks dealt to the 8 XCDs: padding workgroups leave at once
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(seam_kernel<KS>, dim3(nrb * NSLICE), dim3(512), 0, st, S, NSLICE * KS, W, NSLICE * KS, x, bias, slabs, cnt, y, rows, mode, same_xcd);
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) {
        hipLaunchKernelGGL(seam_kernel<KS>, dim3(nrb * NSLICE), dim3(512), 0, st, S, NSLICE * KS, W, NSLICE * KS, x, bias, slabs, cnt, y, rows, mode, same_xcd);
        if (mode == 1 || mode == 3) (void)hipMemsetAsync(cnt, 0, nrb * 4, st);      // (no last arriver to reset the tickets; costs the same in every mode-1/3 run)
    }
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms * 1e3f / reps;
}

static float time_gemm(const bf16_t* S, const bf16_t* W, const float* x, const float* bias, float* y, int rows, int K, hipStream_t st, int reps) {
    mtn_gemm_problem p;
    memset(&p, 0, sizeof(p));
    p.A = S; p.lda = K; p.B = W; p.ldb = K; p.M = rows; p.N = D; p.K = K; p.gate_scale = 1.f;
    p.bias = bias; p.residual = x; p.ldr = D; p.out_f32 = y; p.ldc = D;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) if (mtn_gemm(MTN_BF16, 1, &p, st) != MTN_OK) { fprintf(stderr, "mtn_gemm: %s\n", mtn_last_error()); exit(1); }
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) (void)mtn_gemm(MTN_BF16, 1, &p, st);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

template <int KS> static void run_case(const char* name, int rows, hipStream_t st) {
    const int K = NSLICE * KS, nrb = ((rows + ROWS_WG - 1) / ROWS_WG + 7) / 8 * 8;
    std::vector<bf16_t> hS((size_t)rows * K), hW((size_t)D * K);
    std::vector<float> hx((size_t)rows * D), hb(D);
    uint32_t seed = 12345u + rows + KS;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : hS) v = f2bf(rnd());
    for (auto& v : hW) v = f2bf(rnd() * 0.05f);
    for (auto& v : hx) v = rnd();
    for (auto& v : hb) v = rnd() * 0.1f;
    bf16_t *S, *W; float *x, *b, *slabs, *y, *yref; unsigned* cnt;
    CK(hipMalloc(&S, hS.size() * 2)); CK(hipMalloc(&W, hW.size() * 2)); CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&b, D * 4));
    CK(hipMalloc(&slabs, (size_t)nrb * NSLICE * ROWS_WG * D * 4)); CK(hipMalloc(&y, (size_t)rows * D * 4)); CK(hipMalloc(&yref, (size_t)rows * D * 4));
    CK(hipMalloc(&cnt, nrb * 4)); CK(hipMemset(cnt, 0, nrb * 4));
    CK(hipMemcpy(S, hS.data(), hS.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb.data(), D * 4, hipMemcpyHostToDevice));
    const int reps = 300;
    const float t_gemm = time_gemm(S, W, x, b, yref, rows, K, st, reps);
    const float t_floor = time_seam<KS>(S, W, x, b, slabs, cnt, y, rows, 3, st, reps);
    const float t_pub = time_seam<KS>(S, W, x, b, slabs, cnt, y, rows, 1, st, reps);
    const float t_plain = time_seam<KS>(S, W, x, b, slabs, cnt, y, rows, 2, st, reps);
    CK(hipMemset(cnt, 0, nrb * 4));
    const float t_full = time_seam<KS>(S, W, x, b, slabs, cnt, y, rows, 0, st, reps);
    CK(hipMemset(cnt, 0, nrb * 4));
    const float x_pub = time_seam<KS>(S, W, x, b, slabs, cnt, y, rows, 1, st, reps, 1);
    CK(hipMemset(cnt, 0, nrb * 4));
    const float x_plain = time_seam<KS>(S, W, x, b, slabs, cnt, y, rows, 2, st, reps, 1);
    CK(hipMemset(cnt, 0, nrb * 4));
    const float x_full = time_seam<KS>(S, W, x, b, slabs, cnt, y, rows, 0, st, reps, 1);
    // correctness of the full seam against the library GEMM (same operands; different summation order: fp32 rounding only)
    std::vector<float> a((size_t)rows * D), r((size_t)rows * D);
    CK(hipMemcpy(a.data(), y, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r.data(), yref, r.size() * 4, hipMemcpyDeviceToHost));
    double md = 0, mr = 0;
    for (size_t i = 0; i < a.size(); ++i) { md = fmax(md, fabs((double)a[i] - r[i])); mr = fmax(mr, fabs((double)r[i])); }
    printf("%-28s rows %5d K %5d | grouped GEMM launch %6.2f us | seam kernel: floor (launch + tile + panel loads) %6.2f, + MFMA + write-through slabs + ticket %6.2f,"
           " + last-arriver combine %6.2f us (plain stores + release fence instead: %6.2f) | in-kernel cost of the seam %6.2f us vs launch %6.2f us | max|diff|/max|ref| %.2e\n",
           name, rows, K, t_gemm, t_floor, t_pub, t_full, t_plain, t_full - t_floor, t_gemm, md / mr);
    printf("%-28s   ... a row block's 8 slices on ONE XCD: publish + ticket %6.2f, + combine: write-through slabs %6.2f, plain stores + release fence %6.2f us"
           " | in-kernel cost %6.2f / %6.2f us vs launch %6.2f us\n", "", x_pub, x_full, x_plain, x_full - t_floor, x_plain - t_floor, t_gemm);
    CK(hipFree(S)); CK(hipFree(W)); CK(hipFree(x)); CK(hipFree(b)); CK(hipFree(slabs)); CK(hipFree(y)); CK(hipFree(yref)); CK(hipFree(cnt));
}

int main() {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    printf("# split-K seam inside one launch vs the grouped GEMM launch it would replace (300 back-to-back launches each, operands warm in both)\n");
    for (int rows : {640, 1280}) {
        run_case<256>("w_2 (FFN second GEMM)", rows, st);
        run_case<64>("output projection", rows, st);
    }
    return 0;
}
