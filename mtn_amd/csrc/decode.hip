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
__device__ __forceinline__ u64 ld_ag(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_ag(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
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

// ---- LayerNorm of one row held 2*NP elements per lane by one wave (mtn.py:111-114: unbiased std, eps added to std) -> bf16 into LDS
template <int NP>       // u64 (= 2 floats) per lane: d = 128 * NP
__device__ __forceinline__ void dec_ln_row(const u64* xrow, const float* __restrict__ a2, const float* __restrict__ b2, const float eps,
                                           const int d, const int lane, bf16_t* dst) {
    float v[2 * NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const u64 q = ld_ag(xrow + lane + 64 * i);
        v[2 * i] = __uint_as_float((unsigned)q); v[2 * i + 1] = __uint_as_float((unsigned)(q >> 32));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2 * NP; ++i) s += v[i];
    const float mean = fh_cross_sum(fh_row16_sum(s)) / (float)d;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 2 * NP; ++i) { const float c = v[i] - mean; ss += c * c; }
    const float var = fh_cross_sum(fh_row16_sum(ss)) / (float)(d - 1);
    const float inv = 1.0f / (sqrtf(var) + eps);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int c = 2 * (lane + 64 * i);
        const float y0 = a2[c] * (v[2 * i] - mean) * inv + b2[c], y1 = a2[c + 1] * (v[2 * i + 1] - mean) * inv + b2[c + 1];
        *(unsigned*)(dst + c) = (unsigned)f32_to_bf16(y0) | ((unsigned)f32_to_bf16(y1) << 16);
    }
}
__device__ __forceinline__ void dec_ln_row_any(const u64* xrow, const float* a2, const float* b2, float eps, int d, int lane, bf16_t* dst) {
    switch (d >> 7) {
        case 1: dec_ln_row<1>(xrow, a2, b2, eps, d, lane, dst); break;
        case 2: dec_ln_row<2>(xrow, a2, b2, eps, d, lane, dst); break;
        case 4: dec_ln_row<4>(xrow, a2, b2, eps, d, lane, dst); break;
        default: dec_ln_row<8>(xrow, a2, b2, eps, d, lane, dst); break;       // d = 1024
    }
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
    const int W = A.W, d = A.d, dk = d / A.h;
    const int pos = *A.pos;
    u64* X = (u64*)A.x;                           // [W][d] fp32
    u64* Qb = (u64*)A.q;                          // [W][d] bf16
    u64* Ob = (u64*)A.o;                          // [W][d] bf16
    u64* Hb = (u64*)A.hid;                        // [W][d_ff] bf16
    unsigned char* act = smem + DEC_ACT_OFF;
    float* red = (float*)(smem + DEC_RED_OFF);
    float* qs = (float*)(smem + DEC_Q_OFF);
    float* sc = (float*)(smem + DEC_SC_OFF);
    float* part = (float*)(smem + DEC_PART_OFF);
    float* misc = (float*)(smem + DEC_MISC_OFF);
    const float scale = rsqrtf((float)dk);
    unsigned epoch = 0;

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
        const int K = S.K, pitch = K * 2 + 16;
        switch (S.kind) {
        case MTN_DEC_EMBED: {          // x = lut[token] * sqrt(d) + PE[pos]   (mtn.py:289, 308; eval: no dropout): columns dealt to the workgroups
            const int per = ((d / 2 + G - 1) / G);                 // u64 (column pairs) per workgroup
            const int c0 = wg * per, c1 = min(d / 2, c0 + per);
            for (int i = tid; i < W * (c1 - c0); i += DEC_THREADS) {
                const int j = i / (c1 - c0), c = 2 * (c0 + i % (c1 - c0));
                const float* e = A.lut + (size_t)A.tokens[j] * d + c;
                const float* pe = A.pe + (size_t)pos * d + c;
                const float y0 = e[0] * A.emb_scale + pe[0], y1 = e[1] * A.emb_scale + pe[1];
                st_ag(X + ((size_t)j * d + c) / 2, (u64)__float_as_uint(y0) | ((u64)__float_as_uint(y1) << 32));
            }
        } break;
        case MTN_DEC_SELF_QKV: case MTN_DEC_FFN1: {    // LayerNorm(x) of every row -> act; features n0..n1 of the Linear
            for (int j = wave; j < W; j += 4) dec_ln_row_any(X + (size_t)j * d / 2, S.ln_a, S.ln_b, S.ln_eps, d, lane, (bf16_t*)(act + (size_t)j * pitch));
            __syncthreads();
            f32x4_t acc = dec_w_mma(R, plan, (const bf16_t*)S.w, n0, n1, K, act, pitch, W, lane);
            spill(acc);
            const int Sn = n1 - n0, S4 = Sn >> 2;                          // (slices are multiples of 4 features: one 8-byte store of four bf16)
            for (int i = tid; i < S4 * W; i += DEC_THREADS) {
                const int f = (i % S4) * 4, r = i / S4, n = n0 + f;
                u64 pk = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float y = gather(Sn, f + k, r) + S.bias[n + k];
                    if (S.kind == MTN_DEC_FFN1) y = fmaxf(y, 0.f);
                    pk |= (u64)f32_to_bf16(y) << (16 * k);
                }
                u64* dst;
                if (S.kind == MTN_DEC_FFN1) dst = Hb + ((size_t)r * S.N + n) / 4;
                else if (n < d) dst = Qb + ((size_t)r * d + n) / 4;
                else dst = (u64*)S.cache + (((size_t)r * A.L + pos) * (2 * d) + (n - d)) / 4;            // k | v of the new row into the prefix cache
                st_ag(dst, pk);
            }
        } break;
        case MTN_DEC_OUT: case MTN_DEC_FFN2: {         // act = attention output (OUT) | FFN hidden (FFN2), bf16 [W][K]; + bias + residual -> x
            const u64* src = S.kind == MTN_DEC_OUT ? Ob : Hb;
            for (int i = tid; i < W * K / 4; i += DEC_THREADS) {
                const int j = i / (K / 4), c = i % (K / 4);
                *(u64*)(act + (size_t)j * pitch + c * 8) = ld_ag(src + (size_t)j * K / 4 + c);
            }
            __syncthreads();
            f32x4_t acc = dec_w_mma(R, plan, (const bf16_t*)S.w, n0, n1, K, act, pitch, W, lane);
            spill(acc);
            const int Sn = n1 - n0;
            for (int i = tid; i < Sn * W; i += DEC_THREADS) {
                const int f = i % Sn, r = i / Sn, n = n0 + f;
                unsigned* xp = (unsigned*)X + (size_t)r * d + n;
                const float y = gather(Sn, f, r) + S.bias[n] + __uint_as_float(ld_ag32(xp));
                __hip_atomic_store(xp, __float_as_uint(y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } break;
        case MTN_DEC_CROSS: case MTN_DEC_SELF_ATT: {
            if (wg >= W * A.h) break;
            const int j = wg / A.h, hd = wg % A.h;
            if (S.kind == MTN_DEC_CROSS) {
                // q_h = LayerNorm(x_j) W_q,h^T + b_q,h  (the head's dk rows of W_q: prefetched), rounded to bf16 as the training kernels store q
                if (wave == 0) dec_ln_row_any(X + (size_t)j * d / 2, S.ln_a, S.ln_b, S.ln_eps, d, lane, (bf16_t*)act);
                __syncthreads();
                f32x4_t acc = dec_w_mma(R, plan, (const bf16_t*)S.w, n0, n1, K, act, pitch, 1, lane);
                spill(acc);
                if (tid < dk) qs[tid] = bf16_to_f32(f32_to_bf16(gather(dk, tid, 0) + S.bias[n0 + tid]));
            } else {
                if (tid < dk / 4) {
                    const u64 q4 = ld_ag(Qb + ((size_t)j * d + hd * dk) / 4 + tid);
#pragma unroll
                    for (int k = 0; k < 4; ++k) qs[4 * tid + k] = bf16_to_f32((bf16_t)(q4 >> (16 * k)));
                }
            }
            __syncthreads();
            // scores: one key per thread.  Cross: hoisted K|V rows [j * m + t][2d] (read-only, plain loads); self: cache row of position t of
            // THIS hypothesis' prefix = slot anc[j][t] (agent-scope loads: the newest row was written by other workgroups one stage ago)
            const int m = S.kind == MTN_DEC_CROSS ? S.m : pos + 1;
            const bool self = S.kind == MTN_DEC_SELF_ATT;
            float mx = -3.0e38f;
            for (int t = tid; t < m; t += DEC_THREADS) {
                const bf16_t* krow = self ? (const bf16_t*)S.cache + ((size_t)A.anc[j * A.L + t] * A.L + t) * (2 * d) + hd * dk
                                          : (const bf16_t*)S.kv + ((size_t)j * S.m + t) * (2 * d) + hd * dk;
                float s = 0.f;
                for (int c = 0; c < dk; c += 4) {
                    const u64 k4 = self ? ld_ag((const u64*)(krow + c)) : *(const u64*)(krow + c);
#pragma unroll
                    for (int k = 0; k < 4; ++k) s += qs[c + k] * bf16_to_f32((bf16_t)(k4 >> (16 * k)));
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
            for (int t = qt; t < m; t += nq) {
                const bf16_t* vrow = self ? (const bf16_t*)S.cache + ((size_t)A.anc[j * A.L + t] * A.L + t) * (2 * d) + d + hd * dk
                                          : (const bf16_t*)S.kv + ((size_t)j * S.m + t) * (2 * d) + d + hd * dk;
                const u64 v4 = self ? ld_ag((const u64*)vrow + c4) : ((const u64*)vrow)[c4];
                const float p = bf16_to_f32(f32_to_bf16(sc[t] * inv));
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] += p * bf16_to_f32((bf16_t)(v4 >> (16 * k)));
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) part[qt * dk + c4 * 4 + k] = o[k];
            __syncthreads();
            if (tid < dk / 4) {
                u64 pk = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float y = 0.f;
                    for (int q = 0; q < nq; ++q) y += part[q * dk + tid * 4 + k];
                    pk |= (u64)f32_to_bf16(y) << (16 * k);
                }
                st_ag(Ob + ((size_t)j * d + hd * dk) / 4 + tid, pk);
            }
        } break;
        case MTN_DEC_FINAL: {          // the decoder's final LayerNorm (mtn.py:161) -> the generator's bf16 operand (read by the NEXT kernel: plain stores)
            if (wg < W && wave == 0) dec_ln_row_any(X + (size_t)wg * d / 2, S.ln_a, S.ln_b, S.ln_eps, d, lane, (bf16_t*)A.out_lp + (size_t)wg * d);
        } break;
        default: break;
        }
        if (si + 1 < n_stages) { S = KA.stages[si + 1]; __syncthreads(); prefetch(S); }
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
