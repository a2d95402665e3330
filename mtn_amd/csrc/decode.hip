// decode.hip — ONE decode step of the target stream as ONE persistent launch (BASELINE configs[4]; reference data_utils.py:197-208:
// `model.decode` of every live hypothesis + `model.generator`, per generated token).
//
// What a step is (mtn_amd/decode.py, cached form): for the NEWEST position of each of the W = dialogues x beam live hypotheses, N decoder
// layers x (self-attention over the hypothesis' prefix cache, the three text cross-attentions, one attention per auto-encoder stream,
// the feed-forward) — 7 sublayers per layer at F = 2, 42 dependent sublayers for the 6-layer model — then the decoder's final LayerNorm.
// Every sublayer is LayerNorm -> projection -> attention -> output projection + residual (mtn.py:125-127, 248-267) on W <= 8 ROWS.
// Rounds 1-4 ran this on the training launches: two launches per sublayer, ~90 dependent launches of 6-10 us per step for 72 MB of
// weights (0.95 % of the HBM roofline).
//
// Here the whole pass is one grid of G <= 256 resident workgroups that walks a device-resident list of STAGES; between two stages a grid
// barrier (one agent-scope counter, relaxed polling) replaces the launch boundary:
//   * "slice" stages (q|k|v projection of the self-attention, output projections, both feed-forward Linears): every workgroup owns
//     ceil(N / G) output features of the Linear — its rows of the weight matrix are read from HBM exactly once per step, by one CU, as
//     MFMA B-fragments straight into registers, and they are REQUESTED BEFORE the workgroup waits at the barrier in front of the stage
//     (weights do not depend on the previous stage), so the weight stream hides under the synchronisation;
//   * "unit" stages (attention): workgroup (hypothesis j, head h) projects q_h = LN(x_j) W_q,h itself (its 64 rows of W_q likewise
//     prefetched), attends the memory's hoisted K|V head rows (constant per dialogue: L2-resident after the first step) or the self
//     cache, and publishes its 64 output columns.
// Activations cross workgroups through small global buffers written with agent-scope (sc1, write-through) stores and read back with
// agent-scope loads after the barrier (MI355X_MICROARCH.md, inter-workgroup visibility: the per-XCD L2s are not coherent, sc1 accesses
// are); read-only operands (weights, hoisted K|V, masks) use plain loads.  Every spin is bounded (a timeout sets sync[1] and the step's
// results are garbage — the host raises); the counter is zeroed by a memset node in front of every launch.
// Arithmetic mirrors the training kernels: LayerNorm statistics, softmax and accumulators fp32; LayerNorm output, q, probabilities'
// operands, attention output and the FFN hidden rounded to bf16 where those kernels store bf16.
#include "common.h"

#define DEC_THREADS 256
#define DEC_MAX_W 8

typedef unsigned long long u64;
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
// Agent-scope (sc1) accesses to the buffers workgroups exchange activations through, as BUFFER loads / stores with the sc1 cache-policy
// bit (aux = 16): to the compiler they are ordinary memory operations, so a thread's loads of a stage are issued back to back under ONE
// wait (the first version used relaxed agent atomics, 8 bytes each, and every one of them waited for its own round trip: 9 us per
// stage, profiles/r05_decode_*).
#define DEC_SC1 16
typedef __amdgpu_buffer_rsrc_t dec_rsrc_t;
__device__ __forceinline__ dec_rsrc_t dec_rsrc(const void* p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ uint4 ld16(dec_rsrc_t r, unsigned off) { const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, DEC_SC1); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ u64 ld8(dec_rsrc_t r, unsigned off) { const v2u_t v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, DEC_SC1); return (u64)v.x | ((u64)v.y << 32); }
__device__ __forceinline__ void st16(dec_rsrc_t r, unsigned off, uint4 v) { __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{v.x, v.y, v.z, v.w}, r, (int)off, 0, DEC_SC1); }
__device__ __forceinline__ void st8(dec_rsrc_t r, unsigned off, u64 v) { __builtin_amdgcn_raw_buffer_store_b64(v2u_t{(unsigned)v, (unsigned)(v >> 32)}, r, (int)off, 0, DEC_SC1); }
__device__ __forceinline__ unsigned ld_ag32(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct DecKernelArgs {
    mtn_decode_args a;
    const mtn_decode_stage* stages;
};

// ---- grid barrier: every payload store of this workgroup is an agent-scope (write-through) store; drain them, then one arrival
__device__ __forceinline__ void dec_grid_sync(unsigned* sync, const unsigned target) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (ld_ag32(sync) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 21)) { __hip_atomic_store(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }   // never hang the chip
        }
    }
    __syncthreads();
}

// ---- LayerNorm of one row by one wave (mtn.py:111-114: unbiased std, eps added to std): lane holds 4 consecutive floats per 256 columns;
// the result goes out as bf16 through `put(column, four values)` (an LDS image, or global memory for the final norm)
template <typename PUT>
__device__ __forceinline__ void dec_ln_row(const dec_rsrc_t rx, const unsigned row_off, const float* __restrict__ a2, const float* __restrict__ b2,
                                           const float eps, const int d, const int lane, PUT put) {
    float4 v[4];                                   // d <= 1024
    const int nv = (d + 255) >> 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (i < nv && c < d) { const uint4 q = ld16(rx, row_off + c * 4); v[i] = make_float4(__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z), __uint_as_float(q.w)); }
        else v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = fh_cross_sum(fh_row16_sum(s)) / (float)d;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (i < nv && c < d) { const float a = v[i].x - mean, b = v[i].y - mean, e = v[i].z - mean, f = v[i].w - mean; ss += (a * a + b * b) + (e * e + f * f); }
    }
    const float var = fh_cross_sum(fh_row16_sum(ss)) / (float)(d - 1);
    const float inv = 1.0f / (sqrtf(var) + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (i < nv && c < d) {
            const float4 ga = *(const float4*)(a2 + c), gb = *(const float4*)(b2 + c);
            put(c, make_float4(ga.x * (v[i].x - mean) * inv + gb.x, ga.y * (v[i].y - mean) * inv + gb.y, ga.z * (v[i].z - mean) * inv + gb.z, ga.w * (v[i].w - mean) * inv + gb.w));
        }
    }
}
__device__ __forceinline__ u64 dec_pack4(const float4 y) {
    return (u64)f32_to_bf16(y.x) | ((u64)f32_to_bf16(y.y) << 16) | ((u64)f32_to_bf16(y.z) << 32) | ((u64)f32_to_bf16(y.w) << 48);
}

// ---- the weight side of a small-M Linear on MFMA: out[row r < W][feature n] = sum_k act[r][k] w[n][k]
// A wave owns ONE 16-feature tile and a contiguous range of 32-element contraction steps of it: with T = ceil(S / 16) tiles of the
// workgroup's S features, T >= 3 -> wave w takes tile w whole; T = 2 -> two waves per tile, half the contraction each; T = 1 -> four
// quarters.  The B fragments (lane: feature n0 + lane % 16, elements k0 + (lane / 16) * 8 ..+7: 16 contiguous bytes of a weight row) of
// the wave's first 16 steps are loaded into registers by dec_w_issue() — before the barrier; further steps (not at the shapes of the
// benchmark) are loaded in the loop.
struct DecWPlan { int tile, k_lo, k_hi; };       // this wave's tile (-1: none) and contraction steps [k_lo, k_hi)
__device__ __forceinline__ DecWPlan dec_w_plan(const int S, const int K, const int wave) {
    const int T = (S + 15) >> 4, KS = K >> 5;
    DecWPlan p;
    if (T >= 3) { p.tile = wave < T ? wave : -1; p.k_lo = 0; p.k_hi = KS; }
    else if (T == 2) { p.tile = wave >> 1; const int half = (KS + 1) >> 1; p.k_lo = (wave & 1) * half; p.k_hi = min(KS, p.k_lo + half); }
    else { p.tile = 0; const int q = (KS + 3) >> 2; p.k_lo = wave * q; p.k_hi = min(KS, p.k_lo + q); }
    if (p.k_lo >= p.k_hi) p.tile = -1;
    return p;
}
struct DecWRegs { uint4 w[16]; };
__device__ __forceinline__ void dec_w_issue(DecWRegs& R, const DecWPlan& p, const bf16_t* __restrict__ w, const int n0, const int n1, const int K, const int lane) {
    const int n = n0 + p.tile * 16 + (lane & 15);
    const bool ok = p.tile >= 0 && n < n1;
    const bf16_t* row = w + (size_t)(ok ? n : n0) * K + (lane >> 4) * 8;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int ks = p.k_lo + i;
        R.w[i] = (ok && ks < p.k_hi) ? *(const uint4*)(row + ks * 32) : make_uint4(0, 0, 0, 0);
    }
}
// act: LDS image [W rows][K] bf16 with row pitch `pitch` bytes (K * 2 + 16: the W rows a ds_read_b128 touches sit in different banks)
__device__ __forceinline__ f32x4_t dec_w_mma(const DecWRegs& R, const DecWPlan& p, const bf16_t* __restrict__ w, const int n0, const int n1, const int K,
                                             const unsigned char* act, const int pitch, const int W, const int lane) {
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    if (p.tile < 0) return acc;
    const int r = lane & 15;
    const unsigned char* arow = act + (size_t)(r < W ? r : 0) * pitch + (lane >> 4) * 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int ks = p.k_lo + i;
        if (ks < p.k_hi) {                                      // (wave-uniform)
            uint4 a = *(const uint4*)(arow + ks * 64);
            if (r >= W) a = make_uint4(0, 0, 0, 0);
            mma16<bf16_t>(acc, a, R.w[i]);
        }
    }
    const int n = n0 + p.tile * 16 + r;
    for (int ks = p.k_lo + 16; ks < p.k_hi; ++ks) {             // beyond the prefetched steps
        const uint4 b = n < n1 ? *(const uint4*)(w + (size_t)n * K + ks * 32 + (lane >> 4) * 8) : make_uint4(0, 0, 0, 0);
        uint4 a = *(const uint4*)(arow + ks * 64);
        if (r >= W) a = make_uint4(0, 0, 0, 0);
        mma16<bf16_t>(acc, a, b);
    }
    return acc;                                                  // lane: rows (lane / 16) * 4 ..+3 of the activations, feature tile column lane % 16
}

// LDS layout (bytes)
#define DEC_ACT_OFF 0                 /* activations image: W x (K * 2 + 16), K <= 4096: 8 x 8208 = 65 664 */
#define DEC_RED_OFF 66048             /* partial tiles: 4 waves x 16 features x 8 rows fp32 = 2 048 */
#define DEC_Q_OFF 68096               /* unit stages: q (fp32, <= 128) */
#define DEC_SC_OFF 68608              /* scores / probabilities: <= 1024 keys fp32 */
#define DEC_PART_OFF 72704            /* PV partials: (256 / (dk / 4)) key parts x dk columns fp32 = 4 096 bytes */
#define DEC_MISC_OFF 76800            /* reductions: 16 floats */
#define DEC_LDS 76928

__device__ __forceinline__ float dec_block_max(float v, float* red, const int tid) {
    // wave max by swizzles, then across the four waves through LDS
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ float dec_block_sum(float v, float* red, const int tid) {
    v = fh_cross_sum(fh_row16_sum(v));
    __syncthreads();
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = v;
    __syncthreads();
    return (red[4] + red[5]) + (red[6] + red[7]);
}

__global__ __launch_bounds__(DEC_THREADS) void decode_step_kernel(const DecKernelArgs KA) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const mtn_decode_args& A = KA.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = gridDim.x, wg = blockIdx.x;
    const int W = A.W, d = A.d, dk = d / A.h, dff = A.d_ff;
    const int pos = *A.pos;
    const dec_rsrc_t rX = dec_rsrc(A.x, (unsigned)W * d * 4);          // [W][d] fp32 residual stream
    const dec_rsrc_t rQ = dec_rsrc(A.q, (unsigned)W * d * 2);          // [W][d] bf16
    const dec_rsrc_t rO = dec_rsrc(A.o, (unsigned)W * d * 2);
    const dec_rsrc_t rH = dec_rsrc(A.hid, (unsigned)W * dff * 2);      // [W][d_ff] bf16
    unsigned char* act = smem + DEC_ACT_OFF;
    float* red = (float*)(smem + DEC_RED_OFF);
    float* qs = (float*)(smem + DEC_Q_OFF);
    float* sc = (float*)(smem + DEC_SC_OFF);
    float* part = (float*)(smem + DEC_PART_OFF);
    float* misc = (float*)(smem + DEC_MISC_OFF);
    const float scale = rsqrtf((float)dk);
    unsigned epoch = 0;
    u64* dbg = (A.dbg && wg == 0 && tid == 0) ? (u64*)A.dbg : nullptr;       // per stage: after the barrier / operands ready / computed / stores issued

    DecWRegs R;
    DecWPlan plan;
    int n0 = 0, n1 = 0;
    // what the stage's weight prefetch needs is a function of the stage descriptor and the workgroup's place alone
    auto prefetch = [&](const mtn_decode_stage& S) {
        plan.tile = -1; n0 = n1 = 0;
        if (S.kind == MTN_DEC_SELF_QKV || S.kind == MTN_DEC_OUT || S.kind == MTN_DEC_FFN1 || S.kind == MTN_DEC_FFN2) {
            const int per = ((S.N + G - 1) / G + 3) / 4 * 4;       // a multiple of 4 features: outputs leave as 8-byte stores of four bf16
            n0 = min(S.N, wg * per); n1 = min(S.N, n0 + per);
            if (n1 > n0) { plan = dec_w_plan(n1 - n0, S.K, wave); dec_w_issue(R, plan, (const bf16_t*)S.w, n0, n1, S.K, lane); }
        } else if (S.kind == MTN_DEC_CROSS && wg < W * A.h) {
            n0 = (wg % A.h) * dk; n1 = n0 + dk;
            plan = dec_w_plan(dk, S.K, wave); dec_w_issue(R, plan, (const bf16_t*)S.w, n0, n1, S.K, lane);
        }
    };
    // the four waves' partial tiles -> LDS; then thread (feature, row) sums them in a fixed order.  red[wave][feature 0..15][row 0..7]
    auto spill = [&](const f32x4_t& acc) {
        if ((lane >> 4) < 2) {                                  // rows 0..7 (an idle wave's accumulators are zero)
#pragma unroll
            for (int k = 0; k < 4; ++k) red[(wave * 16 + (lane & 15)) * 8 + (lane >> 4) * 4 + k] = acc[k];
        }
        __syncthreads();
    };
    auto gather = [&](const int S_, const int f, const int r) -> float {       // feature f in [0, S_), row r
        const int T = (S_ + 15) >> 4, t = f >> 4, c = f & 15;
        if (T >= 3) return red[(t * 16 + c) * 8 + r];
        if (T == 2) return red[((2 * t) * 16 + c) * 8 + r] + red[((2 * t + 1) * 16 + c) * 8 + r];
        return (red[(0 * 16 + c) * 8 + r] + red[(1 * 16 + c) * 8 + r]) + (red[(2 * 16 + c) * 8 + r] + red[(3 * 16 + c) * 8 + r]);
    };

    const int n_stages = A.n_stages;
    mtn_decode_stage S = KA.stages[0];
    prefetch(S);
    for (int si = 0; si < n_stages; ++si) {
        if (si > 0) { ++epoch; dec_grid_sync(A.sync, epoch * (unsigned)G); }
        if (dbg) dbg[si * 4 + 0] = wall_clock64();
        const int K = S.K, pitch = K * 2 + 16;
        switch (S.kind) {
        case MTN_DEC_EMBED: {          // x = lut[token] * sqrt(d) + PE[pos]   (mtn.py:289, 308; eval: no dropout): column quads dealt to the workgroups
            const int per = ((d / 4 + G - 1) / G);
            const int c0 = wg * per, c1 = min(d / 4, c0 + per);
            for (int i = tid; i < W * (c1 - c0); i += DEC_THREADS) {
                const int j = i / (c1 - c0), c = 4 * (c0 + i % (c1 - c0));
                const float4 e = *(const float4*)(A.lut + (size_t)A.tokens[j] * d + c), pe = *(const float4*)(A.pe + (size_t)pos * d + c);
                st16(rX, ((unsigned)j * d + c) * 4, make_uint4(__float_as_uint(e.x * A.emb_scale + pe.x), __float_as_uint(e.y * A.emb_scale + pe.y),
                                                               __float_as_uint(e.z * A.emb_scale + pe.z), __float_as_uint(e.w * A.emb_scale + pe.w)));
            }
        } break;
        case MTN_DEC_SELF_QKV: case MTN_DEC_FFN1: {    // LayerNorm(x) of every row -> act; features n0..n1 of the Linear
            for (int j = wave; j < W; j += 4) {
                bf16_t* row = (bf16_t*)(act + (size_t)j * pitch);
                dec_ln_row(rX, (unsigned)j * d * 4, S.ln_a, S.ln_b, S.ln_eps, d, lane, [&](int c, float4 y) { *(u64*)(row + c) = dec_pack4(y); });
            }
            __syncthreads();
            if (dbg) dbg[si * 4 + 1] = wall_clock64();
            f32x4_t acc = dec_w_mma(R, plan, (const bf16_t*)S.w, n0, n1, K, act, pitch, W, lane);
            spill(acc);
            if (dbg) dbg[si * 4 + 2] = wall_clock64();
            const int Sn = n1 - n0, S4 = Sn >> 2;                          // (slices are multiples of 4 features: one 8-byte store of four bf16)
            const dec_rsrc_t rC = dec_rsrc(S.cache, (unsigned)W * A.L * 2 * d * 2);
            for (int i = tid; i < S4 * W; i += DEC_THREADS) {
                const int f = (i % S4) * 4, r = i / S4, n = n0 + f;
                float4 y;
                y.x = gather(Sn, f, r) + S.bias[n]; y.y = gather(Sn, f + 1, r) + S.bias[n + 1];
                y.z = gather(Sn, f + 2, r) + S.bias[n + 2]; y.w = gather(Sn, f + 3, r) + S.bias[n + 3];
                if (S.kind == MTN_DEC_FFN1) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); st8(rH, ((unsigned)r * S.N + n) * 2, dec_pack4(y)); }
                else if (n < d) st8(rQ, ((unsigned)r * d + n) * 2, dec_pack4(y));
                else st8(rC, (((unsigned)r * A.L + pos) * (2 * d) + (n - d)) * 2, dec_pack4(y));         // k | v of the new row into the prefix cache
            }
        } break;
        case MTN_DEC_OUT: case MTN_DEC_FFN2: {         // act = attention output (OUT) | FFN hidden (FFN2), bf16 [W][K]; + bias + residual -> x
            const dec_rsrc_t rS = S.kind == MTN_DEC_OUT ? rO : rH;
            const int K8 = K / 8;
            for (int i = tid; i < W * K8; i += DEC_THREADS) {
                const int j = i / K8, c = i % K8;
                *(uint4*)(act + (size_t)j * pitch + c * 16) = ld16(rS, ((unsigned)j * K + c * 8) * 2);
            }
            const int Sn = n1 - n0, S4 = Sn >> 2;
            // the residual quads this thread will add: asked for now, beside the operand rows
            uint4 xr[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = tid + u * DEC_THREADS;
                if (i < S4 * W) xr[u] = ld16(rX, ((unsigned)(i / S4) * d + n0 + (i % S4) * 4) * 4);
            }
            __syncthreads();
            if (dbg) dbg[si * 4 + 1] = wall_clock64();
            f32x4_t acc = dec_w_mma(R, plan, (const bf16_t*)S.w, n0, n1, K, act, pitch, W, lane);
            spill(acc);
            if (dbg) dbg[si * 4 + 2] = wall_clock64();
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = tid + u * DEC_THREADS;
                if (i < S4 * W) {
                    const int f = (i % S4) * 4, r = i / S4, n = n0 + f;
                    const float y0 = gather(Sn, f, r) + S.bias[n] + __uint_as_float(xr[u].x), y1 = gather(Sn, f + 1, r) + S.bias[n + 1] + __uint_as_float(xr[u].y);
                    const float y2 = gather(Sn, f + 2, r) + S.bias[n + 2] + __uint_as_float(xr[u].z), y3 = gather(Sn, f + 3, r) + S.bias[n + 3] + __uint_as_float(xr[u].w);
                    st16(rX, ((unsigned)r * d + n) * 4, make_uint4(__float_as_uint(y0), __float_as_uint(y1), __float_as_uint(y2), __float_as_uint(y3)));
                }
            }
        } break;
        case MTN_DEC_CROSS: case MTN_DEC_SELF_ATT: {
            if (wg >= W * A.h) break;
            const int j = wg / A.h, hd = wg % A.h;
            const bool self = S.kind == MTN_DEC_SELF_ATT;
            const int m = self ? pos + 1 : S.m;
            const dec_rsrc_t rC = dec_rsrc(S.cache, self ? (unsigned)W * A.L * 2 * d * 2 : 0u);
            // this thread's first key row (one key per thread, 16-byte pieces): requested before q is ready.  Cross: hoisted K|V rows
            // [j * m + t][2d] (read-only, plain loads); self: cache row of position t of THIS hypothesis' prefix = slot anc[j][t]
            // (agent-scope loads: the newest row was written by other workgroups one stage ago)
            uint4 kr[16];                                        // dk <= 128: 16 pieces of 8 bf16
            const int npc = dk / 8;
            {
                const int t = tid;
                if (t < m) {
                    if (self) {
                        const unsigned off = (((unsigned)A.anc[j * A.L + t] * A.L + t) * (2 * d) + hd * dk) * 2;
#pragma unroll
                        for (int c = 0; c < 16; ++c) if (c < npc) kr[c] = ld16(rC, off + c * 16);
                    } else {
                        const uint4* krow = (const uint4*)((const bf16_t*)S.kv + ((size_t)j * S.m + t) * (2 * d) + hd * dk);
#pragma unroll
                        for (int c = 0; c < 16; ++c) if (c < npc) kr[c] = krow[c];
                    }
                }
            }
            if (!self) {
                // q_h = LayerNorm(x_j) W_q,h^T + b_q,h  (the head's dk rows of W_q: prefetched), rounded to bf16 as the training kernels store q
                if (wave == 0) dec_ln_row(rX, (unsigned)j * d * 4, S.ln_a, S.ln_b, S.ln_eps, d, lane, [&](int c, float4 y) { *(u64*)((bf16_t*)act + c) = dec_pack4(y); });
                __syncthreads();
                if (dbg) dbg[si * 4 + 1] = wall_clock64();
                f32x4_t acc = dec_w_mma(R, plan, (const bf16_t*)S.w, n0, n1, K, act, pitch, 1, lane);
                spill(acc);
                if (tid < dk) qs[tid] = bf16_to_f32(f32_to_bf16(gather(dk, tid, 0) + S.bias[n0 + tid]));
            } else {
                if (tid < dk / 4) {
                    const u64 q4 = ld8(rQ, ((unsigned)j * d + hd * dk + tid * 4) * 2);
#pragma unroll
                    for (int k = 0; k < 4; ++k) qs[4 * tid + k] = bf16_to_f32((bf16_t)(q4 >> (16 * k)));
                }
                if (dbg) dbg[si * 4 + 1] = wall_clock64();
            }
            __syncthreads();
            float mx = -3.0e38f;
            for (int t = tid; t < m; t += DEC_THREADS) {
                if (t >= DEC_THREADS) {                              // keys beyond the first 256: loaded here
                    if (self) {
                        const unsigned off = (((unsigned)A.anc[j * A.L + t] * A.L + t) * (2 * d) + hd * dk) * 2;
#pragma unroll
                        for (int c = 0; c < 16; ++c) if (c < npc) kr[c] = ld16(rC, off + c * 16);
                    } else {
                        const uint4* krow = (const uint4*)((const bf16_t*)S.kv + ((size_t)j * S.m + t) * (2 * d) + hd * dk);
#pragma unroll
                        for (int c = 0; c < 16; ++c) if (c < npc) kr[c] = krow[c];
                    }
                }
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    if (c < npc) {
                        const unsigned w4[4] = {kr[c].x, kr[c].y, kr[c].z, kr[c].w};
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            s += qs[c * 8 + 2 * e] * __uint_as_float(w4[e] << 16) + qs[c * 8 + 2 * e + 1] * __uint_as_float(w4[e] & 0xffff0000u);
                    }
                }
                s *= scale;
                if (!self && S.mask && S.mask[(size_t)j * S.mask_stride + t] == 0) s = -1.0e9f;      // masked_fill(mask == 0, -1e9), mtn.py:226
                sc[t] = s;
                mx = fmaxf(mx, s);
            }
            mx = dec_block_max(mx, misc, tid);
            float sum = 0.f;
            for (int t = tid; t < m; t += DEC_THREADS) { const float e = __expf(sc[t] - mx); sc[t] = e; sum += e; }
            sum = dec_block_sum(sum, misc, tid);
            const float inv = 1.0f / sum;
            // o[c] = sum_t P[t] V[t][c], P rounded to bf16 (the training kernels feed P to the MFMA in bf16): thread = (four columns, key part)
            const int c4 = tid % (dk / 4), qt = tid / (dk / 4), nq = DEC_THREADS / (dk / 4);
            float o[4] = {0.f, 0.f, 0.f, 0.f};
            for (int t0 = qt; t0 < m; t0 += 4 * nq) {               // four keys' V quads in flight per round trip
                u64 v4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = t0 + u * nq;
                    v4[u] = 0;
                    if (t < m) {
                        if (self) v4[u] = ld8(rC, (((unsigned)A.anc[j * A.L + t] * A.L + t) * (2 * d) + d + hd * dk + c4 * 4) * 2);
                        else v4[u] = ((const u64*)((const bf16_t*)S.kv + ((size_t)j * S.m + t) * (2 * d) + d + hd * dk))[c4];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = t0 + u * nq;
                    if (t < m) {
                        const float p = bf16_to_f32(f32_to_bf16(sc[t] * inv));
#pragma unroll
                        for (int k = 0; k < 4; ++k) o[k] += p * bf16_to_f32((bf16_t)(v4[u] >> (16 * k)));
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) part[qt * dk + c4 * 4 + k] = o[k];
            __syncthreads();
            if (dbg) dbg[si * 4 + 2] = wall_clock64();
            if (tid < dk / 4) {
                float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int q = 0; q < nq; ++q) { y.x += part[q * dk + tid * 4]; y.y += part[q * dk + tid * 4 + 1]; y.z += part[q * dk + tid * 4 + 2]; y.w += part[q * dk + tid * 4 + 3]; }
                st8(rO, ((unsigned)j * d + hd * dk + tid * 4) * 2, dec_pack4(y));
            }
        } break;
        case MTN_DEC_FINAL: {          // the decoder's final LayerNorm (mtn.py:161) -> the generator's bf16 operand (read by the NEXT kernel: plain stores)
            if (wg < W && wave == 0) {
                bf16_t* row = (bf16_t*)A.out_lp + (size_t)wg * d;
                dec_ln_row(rX, (unsigned)wg * d * 4, S.ln_a, S.ln_b, S.ln_eps, d, lane, [&](int c, float4 y) { *(u64*)(row + c) = dec_pack4(y); });
            }
        } break;
        default: break;
        }
        if (dbg) dbg[si * 4 + 3] = wall_clock64();
        if (si + 1 < n_stages) { S = KA.stages[si + 1]; prefetch(S); }
    }
}

extern "C" int mtn_decode_step(const mtn_decode_args* a, const mtn_decode_stage* stages_device, int grid, void* stream) {
    MTN_CHECK_ARG(a && stages_device, "null arguments");
    MTN_CHECK_ARG(a->W >= 1 && a->W <= DEC_MAX_W, "1 .. 8 hypotheses per launch");
    MTN_CHECK_ARG(a->d >= 128 && a->d <= 1024 && (a->d == 128 || a->d == 256 || a->d == 512 || a->d == 1024), "d_model in {128, 256, 512, 1024}");
    MTN_CHECK_ARG(a->h >= 1 && a->d % a->h == 0 && (a->d / a->h == 32 || a->d / a->h == 64 || a->d / a->h == 128), "head size 32, 64 or 128");
    MTN_CHECK_ARG(a->n_stages >= 1 && a->L >= 1 && a->L <= 1024, "bad stage count / maximum length");
    MTN_CHECK_ARG(grid >= a->W * a->h && grid <= 256, "grid: at least one workgroup per (hypothesis, head), at most one per CU");
    MTN_CHECK_ARG(a->x && a->q && a->o && a->hid && a->out_lp && a->tokens && a->lut && a->pe && a->pos && a->anc && a->sync, "null buffer");
    MTN_CHECK_ARG(a->d_ff >= a->d && a->d_ff <= 4096 && a->d_ff % 32 == 0, "d_ff: a multiple of 32, at most 4096");
    hipStream_t s = (hipStream_t)stream;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)decode_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DEC_LDS) != hipSuccess) {
            mtn_set_error("mtn_decode_step: cannot opt into %d bytes of LDS", DEC_LDS);
            return MTN_ERR_LAUNCH;
        }
        attr = true;
    }
    if (hipMemsetAsync(a->sync, 0, 4, s) != hipSuccess) { mtn_set_error("mtn_decode_step: memset failed"); return MTN_ERR_LAUNCH; }   // the arrival counter (sync[1], the timeout flag, is sticky)
    DecKernelArgs KA;
    KA.a = *a;
    KA.stages = stages_device;
    hipLaunchKernelGGL(decode_step_kernel, dim3(grid), dim3(DEC_THREADS), DEC_LDS, s, KA);
    MTN_CHECK_LAUNCH();
    return MTN_OK;
}
