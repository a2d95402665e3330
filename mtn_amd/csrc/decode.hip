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
// Here the whole pass is one grid of G <= 256 resident workgroups that walks a device-resident list of STAGES.  There is NO barrier between
// stages: every value a stage hands to the next travels as an 8-byte GRANULE {data, tag} written by ONE agent-scope (sc1, write-through)
// store, tag = (launch generation << 8 | producing stage); a consumer polls exactly the granules it needs until every tag matches
// (MI355X_MICROARCH.md, hand-off rows R2 / handoff-1to1: the data IS the flag — no drain, no flag store, no separate load afterwards).
// The first version synchronised with a grid-wide counter barrier per stage and was no faster than the launches it replaced (8.2 us per
// stage: 3 us of barrier + 0.8-2.8 us to load the operands afterwards + 1.2 us of late bias loads; profiles/r05_decode_timeline_barrier.txt).
//   * "slice" stages (q|k|v projection of the self-attention, output projections, both feed-forward Linears): every workgroup of the stage's
//     CLASS owns ceil(N / class size) output features of the Linear — its rows of the weight matrix are read from HBM exactly once per step,
//     by one CU, as MFMA B-fragments straight into registers, and they are REQUESTED BEFORE the workgroup starts polling for the stage's
//     operands (weights do not depend on the previous stage);
//   * "unit" stages (attention): workgroup (hypothesis j, head h) projects q_h = LN(x_j) W_q,h itself (its 64 rows of W_q likewise
//     prefetched), attends the memory's hoisted K|V head rows (constant per dialogue: L2-resident after the first step) or the self
//     cache, and publishes its 64 output columns.
// Three classes of workgroups — x writers (EMBED, OUT, FFN2), wide (q|k|v, FFN1), units — so that consecutive stages never share
// workgroups: while one class works the next has issued its prefetch and sits in its poll (the kernel body explains the ordering argument).
// Granule buffers: xg [W][d] (fp32 residual stream), qg [W][3d/2] (q | k | v of the newest row, bf16 pairs), og [W][d/2] (attention output),
// hg [W][d_ff/2] (FFN hidden).  A workgroup keeps ITS columns of the residual stream in LDS across stages (the slices of the three
// N = d Linears coincide).  Read-only operands (weights, hoisted K|V, masks, the cache rows of earlier steps) use plain loads.  Every
// poll is bounded (a timeout sets sync[1]: the step's results are garbage and the host raises).  The generation lives in sync[0] (sync[2] counts the workgroups that have read it): read
// by every workgroup at entry, advanced at the end by the unit that finishes row 0 — nothing to zero between launches (hipGraph replay safe).
// Arithmetic mirrors the training kernels: LayerNorm statistics, softmax and accumulators fp32; LayerNorm output, q, probabilities'
// operands, attention output and the FFN hidden rounded to bf16 where those kernels store bf16.
#include "common.h"

#define DEC_MAX_NW 16
// Waves per workgroup.  Measured on the 6-layer model, one step (profiles/r05_decode_timeline_*): 4 waves 790 us, 8 waves 602 us, 16 waves
// 747 us — sixteen waves have 128 registers each and the attention stage then spills 82 of them; eight fit (219) and halve the time a stage
// spends ISSUING its prefetch.
#define DEC_NW_USED 8
// Hypothesis rows per launch.  Every small-M Linear is a 16-row MFMA tile, so rows 9..16 ride in what used to be zero padding (round 6:
// four dialogues x beam 4 in ONE launch).  What bounds W is LDS: the activation image W x (K * 2 + 16) bytes <= DEC_ACT_BYTES and the
// normalised residual rows W x d fp32 <= 32 KiB — at d_ff = 4096 (d_model 1024) 8 rows, at the benchmark's d_ff = 2048 all 16.
#define DEC_MAX_W 16
#define DEC_ACT_BYTES 66048
// a poll gives up after DEC_POLL_TICKS of the 100 MHz wall clock (50 ms: a whole step is ~0.5 ms) — or at once when another workgroup has
// (sync[1] set): a launch that cannot have all its workgroups resident costs one timeout, not one per workgroup and stage
#define DEC_POLL_TICKS 5000000ull

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
    int n_xw, n_mid;            // workgroup classes: [0, n_xw) write x, [n_xw, n_xw + n_mid) the wide projections, the last W * h the attention units
};

// maximum over the 16 lanes of a DPP row (every lane gets it)
__device__ __forceinline__ float dec_row16_max(float v) {
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false)));     // quad_perm [1,0,3,2]
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false)));     // quad_perm [2,3,0,1]
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false)));    // row_half_mirror
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false)));    // row_mirror
    return v;
}

// Bounded polls (uniform over the workgroup: every thread reads the same words).  True = stop polling: this workgroup has waited
// DEC_POLL_TICKS (it raises sync[1]), or another one already has.
__device__ __forceinline__ bool dec_give_up(unsigned* sync, u64& t0) {
    const u64 now = wall_clock64();
    if (t0 == 0) { t0 = now; return false; }
    const bool late = now - t0 > DEC_POLL_TICKS;
    if (late && threadIdx.x == 0) __hip_atomic_store(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __syncthreads_or(late || ld_ag32(sync + 1) != 0);
}

// ---- granules
__device__ __forceinline__ void dec_pub(const dec_rsrc_t r, const unsigned index, const unsigned data, const unsigned tag) { st8(r, index * 8, (u64)data | ((u64)tag << 32)); }
__device__ __forceinline__ unsigned dec_pack2(const float a, const float b) { return (unsigned)f32_to_bf16(a) | ((unsigned)f32_to_bf16(b) << 16); }
// Poll `count` granules starting at `first` of buffer r until all carry `tag`; granule first + i goes to sink(i, data).  All threads;
// a thread reads PAIRS of adjacent granules with 16-byte loads (8-byte accesses run at 0.54-0.70 of the 16-byte rate, MI355X_MICROARCH.md),
// DEC_PPT pairs per pass — every use has an even `first` and an even `count` — and a pass is repeated until every thread's tags match.
// Round 5 read 4 single granules per thread and pass: with 16 hypothesis rows the FFN hidden (16 x 1024 granules) then took EIGHT dependent
// passes of one fabric round trip each (profiles/r06_decode_timeline_4_vs_16_rows_before.txt: FFN2 5.5 -> 21 us per stage); now two.
// (Two probes in flight half a round trip apart — to notice the arrival after 0.5-1.0 instead of 0.5-1.5 round trips — were measured and
// LOSE: the hand-offs took 0.35-0.4 us LONGER, profiles/r05_decode_poll_two_probes.txt; one probe at a time it stays.)
#define DEC_PPT 8
template <int DEC_THREADS, typename SINK>
__device__ __forceinline__ bool dec_poll(const dec_rsrc_t r, const unsigned first, const int count, const unsigned tag, unsigned* sync, SINK sink) {
    const int tid = threadIdx.x;
    const int npairs = count >> 1;
    for (int base = 0; base < npairs; base += DEC_THREADS * DEC_PPT) {
        uint4 v[DEC_PPT];
        u64 t0 = 0;
        const bool narrow = npairs - base <= DEC_THREADS * (DEC_PPT / 2);       // (uniform) half the loads suffice: the common case at <= 8 rows
        for (unsigned spins = 0;; ++spins) {
            bool ok = true;
#pragma unroll
            for (int k = 0; k < DEC_PPT; ++k) {
                if (k >= DEC_PPT / 2 && narrow) break;
                const int i = base + k * DEC_THREADS + tid;
                if (i < npairs) { v[k] = ld16(r, (first + 2 * i) * 8); ok = ok && v[k].y == tag && v[k].w == tag; }
            }
            if (__syncthreads_and(ok)) break;
            if ((spins & 63) == 63 && dec_give_up(sync, t0)) return false;       // never hang the chip
            if ((spins & 3) == 3) __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int k = 0; k < DEC_PPT; ++k) {
            if (k >= DEC_PPT / 2 && narrow) break;
            const int i = base + k * DEC_THREADS + tid;
            if (i < npairs) { sink(2 * i, v[k].x); sink(2 * i + 1, v[k].z); }
        }
    }
    return true;
}

// ---- LayerNorm of one row by one wave (mtn.py:111-114: unbiased std, eps added to std): lane holds 4 consecutive floats per 256 columns;
// the result goes out as bf16 through `put(column, four values)` (an LDS image, or global memory for the final norm)
template <typename PUT>
__device__ __forceinline__ void dec_ln_row(const float* xrow /* LDS, fp32 [d] */, const float* gains /* LDS: a_2 [d] then b_2 [d] */, const float eps, const int d, const int lane, PUT put) {
    float4 v[4];                                   // d <= 1024
    const int nv = (d + 255) >> 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (lane + 64 * i) * 4;
        v[i] = (i < nv && c < d) ? *(const float4*)(xrow + c) : make_float4(0.f, 0.f, 0.f, 0.f);
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
            const float4 ga = *(const float4*)(gains + c), gb = *(const float4*)(gains + d + c);
            put(c, make_float4(ga.x * (v[i].x - mean) * inv + gb.x, ga.y * (v[i].y - mean) * inv + gb.y, ga.z * (v[i].z - mean) * inv + gb.z, ga.w * (v[i].w - mean) * inv + gb.w));
        }
    }
}
__device__ __forceinline__ u64 dec_pack4(const float4 y) {
    return (u64)f32_to_bf16(y.x) | ((u64)f32_to_bf16(y.y) << 16) | ((u64)f32_to_bf16(y.z) << 32) | ((u64)f32_to_bf16(y.w) << 48);
}

// ---- the weight side of a small-M Linear on MFMA: out[row r < W][feature n] = sum_k act[r][k] w[n][k]
// The workgroup's S <= 64 features are T = ceil(S / 16) tiles of 16; its NW waves are dealt wpt = NW / T' to a tile (T' = T rounded up to
// a power of two) and split that tile's K / 32 contraction steps evenly.  A CU accepts a wave's loads at a fixed rate (one 1 KiB load per
// ~180 ns and wave: DESIGN.md §10), so the weight slice is requested NW / 4 times faster than by the first, four-wave version of this
// kernel, which spent 2-3 us per stage just issuing its prefetch (profiles/r05_decode_timeline_granules_v1.txt).  The B fragments (lane:
// feature n0 + lane % 16, elements k0 + (lane / 16) * 8 ..+7: 16 contiguous bytes of a weight row) of the wave's first 64 / NW steps are loaded
// into registers by dec_w_issue() — before the stage's operands exist; further steps (not at the benchmark's shapes) in the loop.
struct DecWPlan { int tile, k_lo, k_hi, wpt; };       // this wave's tile (-1: none), contraction steps [k_lo, k_hi), waves per tile
template <int DEC_NW>
__device__ __forceinline__ DecWPlan dec_w_plan(const int S, const int K, const int wave) {
    const int T = (S + 15) >> 4, KS = K >> 5;
    DecWPlan p;
    p.wpt = T <= 1 ? DEC_NW : (T == 2 ? DEC_NW / 2 : DEC_NW / 4);
    p.tile = wave / p.wpt;
    const int sub = wave % p.wpt, q = (KS + p.wpt - 1) / p.wpt;
    p.k_lo = sub * q; p.k_hi = min(KS, p.k_lo + q);
    if (p.tile >= T || p.k_lo >= p.k_hi) p.tile = -1;
    return p;
}
template <int NR> struct DecWRegs { uint4 w[NR]; };
template <int NR>
__device__ __forceinline__ void dec_w_issue(DecWRegs<NR>& R, const DecWPlan& p, const bf16_t* __restrict__ w, const int n0, const int n1, const int K, const int lane) {
    const int n = n0 + p.tile * 16 + (lane & 15);
    const bool ok = p.tile >= 0 && n < n1;
    const bf16_t* row = w + (size_t)(ok ? n : n0) * K + (lane >> 4) * 8;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int ks = p.k_lo + i;
        R.w[i] = (ok && ks < p.k_hi) ? *(const uint4*)(row + ks * 32) : make_uint4(0, 0, 0, 0);
    }
}
// act: LDS image [W rows][K] bf16 with row pitch `pitch` bytes (K * 2 + 16: the W rows a ds_read_b128 touches sit in different banks)
template <int NR>
__device__ __forceinline__ f32x4_t dec_w_mma(const DecWRegs<NR>& R, const DecWPlan& p, const bf16_t* __restrict__ w, const int n0, const int n1, const int K,
                                             const unsigned char* act, const int pitch, const int W, const int lane) {
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    if (p.tile < 0) return acc;
    const int r = lane & 15;
    const unsigned char* arow = act + (size_t)(r < W ? r : 0) * pitch + (lane >> 4) * 16;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int ks = p.k_lo + i;
        if (ks < p.k_hi) {                                      // (wave-uniform)
            uint4 a = *(const uint4*)(arow + ks * 64);
            if (r >= W) a = make_uint4(0, 0, 0, 0);
            mma16<bf16_t>(acc, a, R.w[i]);
        }
    }
    const int n = n0 + p.tile * 16 + r;
    for (int ks = p.k_lo + NR; ks < p.k_hi; ++ks) {              // beyond the prefetched steps (d_ff = 4096 only)
        const uint4 b = n < n1 ? *(const uint4*)(w + (size_t)n * K + ks * 32 + (lane >> 4) * 8) : make_uint4(0, 0, 0, 0);
        uint4 a = *(const uint4*)(arow + ks * 64);
        if (r >= W) a = make_uint4(0, 0, 0, 0);
        mma16<bf16_t>(acc, a, b);
    }
    return acc;                                                  // lane: rows (lane / 16) * 4 ..+3 of the activations, feature tile column lane % 16
}

// LDS layout (bytes)
#define DEC_ACT_OFF 0                 /* activations image: W x (K * 2 + 16) <= DEC_ACT_BYTES: 8 x 8208 (K = 4096) or 16 x 4112 (K = 2048) */
#define DEC_RED_OFF 66048             /* partial tiles: 8 waves x 16 features x 16 rows fp32 = 8 192 */
#define DEC_Q_OFF 74240               /* unit stages: q (fp32, <= 128) + the newest row's k and v of this head (self-attention): 3 x 128 floats */
#define DEC_SC_OFF 75776              /* scores / probabilities: <= 1024 keys fp32; afterwards the second level of the PV reduction */
#define DEC_PART_OFF 79872            /* PV partials: (threads / (dk / 4)) key parts x dk columns fp32 <= 16 384 */
#define DEC_MISC_OFF 96256            /* the waves' (max, sum) pairs of the softmax: 32 floats */
#define DEC_XS_OFF 96512              /* this workgroup's columns of the residual stream: W x per_x (<= 32 columns) fp32 <= 2 048 */
#define DEC_XF_OFF 98560              /* the residual rows a stage normalises: W x d fp32 <= 32 768 (W x d <= 8192) */
#define DEC_GAIN_OFF 131328           /* LayerNorm a_2 | b_2 of the stage: 2 x 1024 floats */
#define DEC_STG_OFF 139520            /* the stage list: <= 160 descriptors */
#define DEC_MAX_STAGES 160
#define DEC_LDS (DEC_STG_OFF + DEC_MAX_STAGES * (int)sizeof(mtn_decode_stage))
static_assert(DEC_LDS <= 160 * 1024, "decode_step_kernel: LDS image beyond the CU's 160 KiB");
static_assert(DEC_NW_USED * 16 * 16 * 4 <= DEC_Q_OFF - DEC_RED_OFF, "decode_step_kernel: partial-tile area");


template <int DEC_NW>
__global__ __launch_bounds__(DEC_NW * 64) void decode_step_kernel(const DecKernelArgs KA) {
    constexpr int DEC_THREADS = DEC_NW * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const mtn_decode_args& A = KA.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wg = blockIdx.x;
    const int W = A.W, d = A.d, dk = d / A.h, dff = A.d_ff;
    const int pos = *A.pos;
    // a step of this session has timed out: its granules are garbage and nothing advanced the generation — every later launch of the
    // (captured) search leaves at once; the host resets sync[] and re-runs the search on the launch-per-sublayer pass (decode.py)
    if (ld_ag32(A.sync + 1) != 0) return;
    unsigned gen0 = __hip_atomic_load(A.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" : "+v"(gen0));                                        // (the value has ARRIVED before the check-in below is issued)
    const unsigned gen = (gen0 + 1u) << 8;                               // this launch's generation (advanced at the end, once EVERY workgroup has read it)
    if (threadIdx.x == 0) __hip_atomic_fetch_add(A.sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // checked in
    const dec_rsrc_t rX = dec_rsrc(A.xg, (unsigned)W * d * 8);          // [W][d] granules: fp32 residual stream
    const dec_rsrc_t rQ = dec_rsrc(A.qg, (unsigned)W * 3 * d * 4);      // [W][3d/2] granules: q | k | v of the newest row, bf16 pairs
    const dec_rsrc_t rO = dec_rsrc(A.og, (unsigned)W * d * 4);          // [W][d/2]
    const dec_rsrc_t rH = dec_rsrc(A.hg, (unsigned)W * dff * 4);        // [W][d_ff/2]
    unsigned char* act = smem + DEC_ACT_OFF;
    float* red = (float*)(smem + DEC_RED_OFF);
    float* qs = (float*)(smem + DEC_Q_OFF);
    float* sc = (float*)(smem + DEC_SC_OFF);
    float* part = (float*)(smem + DEC_PART_OFF);
    float* misc = (float*)(smem + DEC_MISC_OFF);
    float* xs = (float*)(smem + DEC_XS_OFF);                             // [W][per_x]: this workgroup's columns of x
    float* xf = (float*)(smem + DEC_XF_OFF);                             // [W][d]
    float* gains = (float*)(smem + DEC_GAIN_OFF);                        // a_2 [d] | b_2 [d]
    const float scale = rsqrtf((float)dk);
    // Three CLASSES of workgroups, so that consecutive stages never run on the same workgroups (the stage order is wide -> unit -> x ->
    // unit -> x ... -> wide -> x): while one class works, the next class has already asked for its weights / K | V rows and sits in its
    // poll.  On one shared set of workgroups (the first versions) every stage paid its 1.6-2.1 us of prefetch ISSUE on the critical
    // path (profiles/r05_decode_timeline_8waves.txt).
    //   class 0  "x writers"  EMBED, OUT, FFN2 (N = d): each keeps ITS columns of the residual stream in LDS across the whole step
    //   class 1  "wide"       SELF_QKV (N = 3d), FFN1 (N = d_ff)
    //   class 2  "units"      CROSS, SELF_ATT: (hypothesis, head); FINAL: one row each
    const int n_xw = KA.n_xw, n_mid = KA.n_mid;
    const int cls = wg < n_xw ? 0 : (wg < n_xw + n_mid ? 1 : 2);
    const int idx = cls == 0 ? wg : (cls == 1 ? wg - n_xw : wg - n_xw - n_mid);
    const int csize = cls == 0 ? n_xw : (cls == 1 ? n_mid : W * A.h);
    const int per_x = ((d + n_xw - 1) / n_xw + 3) / 4 * 4;               // the slice of every N = d stage
    const int x0 = min(d, idx * per_x), x1 = min(d, x0 + per_x);         // (class 0)
    u64* dbg = (A.dbg && idx == 0 && tid == 0) ? (u64*)A.dbg : nullptr;       // per stage (first workgroup of the stage's class): entered / operands arrived / computed / published

    DecWRegs<64 / DEC_NW> R;                                            // 4 prefetched steps per wave on sixteen waves, 8 on eight
    DecWPlan plan;
    float4 gpre = make_float4(0.f, 0.f, 0.f, 0.f);                       // this thread's quad of the stage's LayerNorm a_2 | b_2
    float4 bpre = make_float4(0.f, 0.f, 0.f, 0.f);                       // the biases of this thread's epilogue quad
    int n0 = 0, n1 = 0;
    // everything a stage can ask for BEFORE its operands exist: the workgroup's weight rows, LayerNorm gains, biases
    auto prefetch = [&](const mtn_decode_stage& S) {
        plan.tile = -1; n0 = n1 = 0;
        const bool slice = S.kind == MTN_DEC_SELF_QKV || S.kind == MTN_DEC_OUT || S.kind == MTN_DEC_FFN1 || S.kind == MTN_DEC_FFN2;
        if (slice) {
            const int per = ((S.N + csize - 1) / csize + 3) / 4 * 4;       // a multiple of 4 features: outputs leave as pairs of granules
            n0 = min(S.N, idx * per); n1 = min(S.N, n0 + per);
            if (n1 > n0) { plan = dec_w_plan<DEC_NW>(n1 - n0, S.K, wave); dec_w_issue(R, plan, (const bf16_t*)S.w, n0, n1, S.K, lane); }
            // epilogue work items: one thread per (feature, row) where the stage writes x (one fp32 granule each), per (feature pair, row)
            // where it writes bf16 pairs — as many threads as there are granules, each with its bias in a register
            const int Sn = n1 - n0;
            if (S.kind == MTN_DEC_OUT || S.kind == MTN_DEC_FFN2) bpre.x = (Sn > 0 && tid < Sn * W) ? S.bias[n0 + tid % Sn] : 0.f;
            else if (Sn > 0 && tid < (Sn >> 1) * W) { const float2 b2 = *(const float2*)(S.bias + n0 + 2 * (tid % (Sn >> 1))); bpre.x = b2.x; bpre.y = b2.y; }
        } else if (S.kind == MTN_DEC_CROSS) {
            n0 = (idx % A.h) * dk; n1 = n0 + dk;
            plan = dec_w_plan<DEC_NW>(dk, S.K, wave); dec_w_issue(R, plan, (const bf16_t*)S.w, n0, n1, S.K, lane);
            bpre.x = tid < dk ? S.bias[n0 + tid] : 0.f;
        }
        if (S.kind == MTN_DEC_SELF_QKV || S.kind == MTN_DEC_FFN1 || S.kind == MTN_DEC_CROSS || S.kind == MTN_DEC_FINAL) {
            const int q4 = d >> 2;
            if (tid < 2 * q4) gpre = tid < q4 ? ((const float4*)S.ln_a)[tid] : ((const float4*)S.ln_b)[tid - q4];
        }
    };
    auto gains_to_lds = [&]() { if (tid < (d >> 1)) ((float4*)gains)[tid] = gpre; };       // (before the barrier that follows a poll)
    // the waves' partial tiles -> LDS; then thread (feature, row) sums a tile's partials in a fixed order.  red[wave][feature 0..15][row 0..15]
    auto spill = [&](const f32x4_t& acc) {
        *(f32x4_t*)(red + (wave * 16 + (lane & 15)) * 16 + (lane >> 4) * 4) = acc;       // rows (lane / 16) * 4 ..+3 (an idle wave's accumulators are zero)
        __syncthreads();
    };
    auto gather = [&](const int S_, const int f, const int r) -> float {       // feature f in [0, S_), row r
        const int T = (S_ + 15) >> 4, t = f >> 4, c = f & 15;
        const int wpt = T <= 1 ? DEC_NW : (T == 2 ? DEC_NW / 2 : DEC_NW / 4);
        const float* p = red + ((t * wpt) * 16 + c) * 16 + r;
        if (wpt == 2) return p[0] + p[256];
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < wpt; i += 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) s4[k] += p[(i + k) * 256];
        }
        return (s4[0] + s4[1]) + (s4[2] + s4[3]);
    };

    // the stage list -> LDS once (a descriptor read from memory at every stage entry cost 1-2 us of scalar-cache miss in front of the
    // prefetch that needs its fields: profiles/r05_decode_timeline_granules_v1.txt, column "barrier")
    const int n_stages = A.n_stages;
    mtn_decode_stage* stg = (mtn_decode_stage*)(smem + DEC_STG_OFF);
    {
        const unsigned* src = (const unsigned*)KA.stages;
        unsigned* dst = (unsigned*)stg;
        const int nw32 = n_stages * (int)(sizeof(mtn_decode_stage) / 4);
        for (int i = tid; i < nw32; i += DEC_THREADS) dst[i] = src[i];
    }
    __syncthreads();
    // a descriptor is wave-uniform: pulled out of LDS into scalar registers (26 VGPRs otherwise)
    auto stage_of = [&](const int i) {
        mtn_decode_stage D;
        const unsigned* src = (const unsigned*)(stg + i);
        unsigned* dst = (unsigned*)&D;
#pragma unroll
        for (int k = 0; k < (int)(sizeof(mtn_decode_stage) / 4); ++k) dst[k] = __builtin_amdgcn_readfirstlane(src[k]);
        return D;
    };
    auto kind_of = [&](const int i) -> int { return __builtin_amdgcn_readfirstlane(*(const int*)(stg + i)); };
    auto class_of = [](const int kind) -> int {
        return (kind == MTN_DEC_EMBED || kind == MTN_DEC_OUT || kind == MTN_DEC_FFN2) ? 0 : ((kind == MTN_DEC_SELF_QKV || kind == MTN_DEC_FFN1) ? 1 : 2);
    };
    auto next_mine = [&](int i) -> int {                                  // this class's next stage behind stage i
        for (++i; i < n_stages && class_of(kind_of(i)) != cls; ++i) {}
        return i;
    };
    auto x_tag_before = [&](int i) -> unsigned {                          // the tag of the residual stream a stage reads: its latest writer
        for (--i; i > 0 && class_of(kind_of(i)) != 0; --i) {}
        return gen | (unsigned)(i + 1);
    };
    int si = next_mine(-1);
    mtn_decode_stage S;
    if (si < n_stages) { S = stage_of(si); prefetch(S); }
    bool alive = true;
    while (si < n_stages && alive) {
        const unsigned tag = gen | (unsigned)(si + 1);
        if (dbg) dbg[si * 4 + 0] = wall_clock64();
        const int K = S.K, pitch = K * 2 + 16;
        // a workgroup without a share of this stage must not even poll: nothing orders it against the stages to come
        const bool share = S.kind == MTN_DEC_EMBED ? x1 > x0 : (cls == 2 ? true : n1 > n0);
        if (share) switch (S.kind) {
        case MTN_DEC_EMBED: {          // x = lut[token] * sqrt(d) + PE[pos]   (mtn.py:289, 308; eval: no dropout): this workgroup's columns
            const int nx = x1 - x0;
            for (int i = tid; i < W * nx; i += DEC_THREADS) {
                const int j = i / nx, c = x0 + i % nx;
                const float y = A.lut[(size_t)A.tokens[j] * d + c] * A.emb_scale + A.pe[(size_t)pos * d + c];
                xs[j * per_x + (c - x0)] = y;
                dec_pub(rX, (unsigned)j * d + c, __float_as_uint(y), tag);
            }
        } break;
        case MTN_DEC_SELF_QKV: case MTN_DEC_FFN1: {    // LayerNorm(x) of every row -> act; features n0..n1 of the Linear
            alive = dec_poll<DEC_THREADS>(rX, 0, W * d, x_tag_before(si), A.sync, [&](int i, unsigned v) { xf[i] = __uint_as_float(v); });
            gains_to_lds();
            __syncthreads();
            if (dbg) dbg[si * 4 + 1] = wall_clock64();
            for (int rw = wave; rw < W; rw += DEC_NW) {                      // one row per wave (two beyond 8 hypotheses)
                bf16_t* row = (bf16_t*)(act + (size_t)rw * pitch);
                dec_ln_row(xf + (size_t)rw * d, gains, S.ln_eps, d, lane, [&](int c, float4 y) { *(u64*)(row + c) = dec_pack4(y); });
            }
            __syncthreads();
            f32x4_t acc = dec_w_mma(R, plan, (const bf16_t*)S.w, n0, n1, K, act, pitch, W, lane);
            spill(acc);
            if (dbg) dbg[si * 4 + 2] = wall_clock64();
            const int Sn = n1 - n0, S2 = Sn >> 1;                          // (slices are multiples of 4 features)
            const dec_rsrc_t rD = S.kind == MTN_DEC_FFN1 ? rH : rQ;
            if (tid < S2 * W) {                                              // thread = (feature pair, row): one granule
                const int f = (tid % S2) * 2, r = tid / S2;
                float y0 = gather(Sn, f, r) + bpre.x, y1 = gather(Sn, f + 1, r) + bpre.y;
                if (S.kind == MTN_DEC_FFN1) { y0 = fmaxf(y0, 0.f); y1 = fmaxf(y1, 0.f); }
                dec_pub(rD, ((unsigned)r * S.N + n0 + f) >> 1, dec_pack2(y0, y1), tag);
            }
        } break;
        case MTN_DEC_OUT: case MTN_DEC_FFN2: {         // act = attention output (OUT) | FFN hidden (FFN2), bf16 pairs [W][K/2]; + bias + residual -> x
            const dec_rsrc_t rS = S.kind == MTN_DEC_OUT ? rO : rH;
            const int K2 = K >> 1;
            alive = dec_poll<DEC_THREADS>(rS, 0, W * K2, gen | (unsigned)si, A.sync, [&](int i, unsigned v) { *(unsigned*)(act + (size_t)(i / K2) * pitch + (i % K2) * 4) = v; });
            __syncthreads();
            if (dbg) dbg[si * 4 + 1] = wall_clock64();
            f32x4_t acc = dec_w_mma(R, plan, (const bf16_t*)S.w, n0, n1, K, act, pitch, W, lane);
            spill(acc);
            if (dbg) dbg[si * 4 + 2] = wall_clock64();
            const int Sn = n1 - n0;
            if (tid < Sn * W) {                                              // thread = (feature, row): one granule
                const int f = tid % Sn, r = tid / Sn;
                float* xr = xs + r * per_x + f;                              // (n0 == x0: the N = d slices coincide)
                const float y = gather(Sn, f, r) + bpre.x + *xr;
                *xr = y;
                dec_pub(rX, (unsigned)r * d + n0 + f, __float_as_uint(y), tag);
            }
        } break;
        case MTN_DEC_CROSS: case MTN_DEC_SELF_ATT: {
            const int j = idx / A.h, hd = idx % A.h;
            const bool self = S.kind == MTN_DEC_SELF_ATT;
            const int m = self ? pos + 1 : S.m;
            const int npc = dk / 8;
            const int c4 = tid % (dk / 4), qt = tid / (dk / 4), nq = DEC_THREADS / (dk / 4);
            // read-only operands of the attention, requested before x has arrived: this thread's HALF of a key row (two threads per key, 16-byte
            // pieces), its mask byte, and its V quads of the first 4 * nq keys.  Cross: hoisted K|V rows [j * m + t][2d]; self: cache rows of positions
            // < pos of THIS hypothesis' prefix (slot anc[j][t]: written by earlier launches) — the newest row arrives as granules
            uint4 kr[4];                                                  // (dk <= 64: half a row is <= 4 pieces)
            u64 vq[4];
            const int kt = tid >> 1, kh = tid & 1, nph = npc >> 1;        // this thread's key, its half, pieces per half
            unsigned char mb = 1;
            auto krow_of = [&](int t) -> const uint4* {
                return self ? (const uint4*)((const bf16_t*)S.cache + ((size_t)A.anc[j * A.L + t] * A.L + t) * (2 * d) + hd * dk)
                            : (const uint4*)((const bf16_t*)S.kv + ((size_t)j * S.m + t) * (2 * d) + hd * dk);
            };
            const int m_old = self ? pos : m;                            // keys whose rows are in memory already
            if (kt < m_old) {
                const uint4* kp = krow_of(kt) + kh * nph;
#pragma unroll
                for (int c = 0; c < 4; ++c) if (c < nph) kr[c] = kp[c];
                if (!self && S.mask) mb = S.mask[(size_t)j * S.mask_stride + kt];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = qt + u * nq;
                vq[u] = t < m_old ? ((const u64*)((const bf16_t*)krow_of(t) + d))[c4] : 0;
            }
            float* knew = qs + 128; float* vnew = qs + 256;               // self: the newest row's k and v of this head
            if (!self) {
                // q_h = LayerNorm(x_j) W_q,h^T + b_q,h  (the head's dk rows of W_q: prefetched), rounded to bf16 as the training kernels store q
                alive = dec_poll<DEC_THREADS>(rX, (unsigned)j * d, d, x_tag_before(si), A.sync, [&](int i, unsigned v) { xf[i] = __uint_as_float(v); });
                gains_to_lds();
                __syncthreads();
                if (dbg) dbg[si * 4 + 1] = wall_clock64();
                // (one wave normalises: every wave doing it into a copy of its own saves the barrier but doubles the instruction streams per SIMD —
                // measured +0.5 us per stage)
                if (wave == 0) dec_ln_row(xf, gains, S.ln_eps, d, lane, [&](int c, float4 y) { *(u64*)((bf16_t*)act + c) = dec_pack4(y); });
                __syncthreads();
                f32x4_t acc = dec_w_mma(R, plan, (const bf16_t*)S.w, n0, n1, K, act, pitch, 1, lane);
                spill(acc);
                if (tid < dk) qs[tid] = bf16_to_f32(f32_to_bf16(gather(dk, tid, 0) + bpre.x));
            } else {
                // q_h, and k_h | v_h of the newest row, arrive as the projection stage's granules (gathered below)
            }
            __syncthreads();
            if (self) {
                // three ranges of dk / 2 granules: q at column hd*dk, k at d + hd*dk, v at 2d + hd*dk of row j
                const int hp = dk / 2;
                u64 t0 = 0;
                for (unsigned spins = 0;; ++spins) {
                    bool ok = true;
                    u64 g = 0;
                    if (tid < 3 * hp) {
                        const int which = tid / hp, i = tid % hp;
                        g = ld8(rQ, (((unsigned)j * 3 * d + which * d + hd * dk) / 2 + i) * 8);
                        ok = (unsigned)(g >> 32) == (gen | (unsigned)si);
                    }
                    if (__syncthreads_and(ok)) {
                        if (tid < 3 * hp) {
                            const int which = tid / hp, i = tid % hp;
                            float* dst = which == 0 ? qs : (which == 1 ? knew : vnew);
                            dst[2 * i] = __uint_as_float(((unsigned)g) << 16); dst[2 * i + 1] = __uint_as_float(((unsigned)g) & 0xffff0000u);
                            if (which) ((unsigned*)((bf16_t*)S.cache + ((size_t)j * A.L + pos) * (2 * d) + (which - 1) * d + hd * dk))[i] = (unsigned)g;   // into the prefix cache, for the steps to come
                        }
                        break;
                    }
                    if ((spins & 63) == 63 && dec_give_up(A.sync, t0)) { alive = false; break; }
                }
                __syncthreads();
                if (dbg) dbg[si * 4 + 1] = wall_clock64();
            }
            float mx = -3.0e38f, sum = 0.f;
            for (int t = kt; t < (m + DEC_THREADS / 2 - 1) / (DEC_THREADS / 2) * (DEC_THREADS / 2); t += DEC_THREADS / 2) {       // (every pair runs the same trip count: the shuffle below)
                float s_ = 0.f;
                const bool live = t < m;
                if (live && self && t == pos) {
                    for (int c = 0; c < dk / 2; ++c) s_ += qs[kh * (dk / 2) + c] * knew[kh * (dk / 2) + c];
                } else if (live) {
                    if (t >= DEC_THREADS / 2) {                          // keys beyond the first pass: loaded here
                        const uint4* kp = krow_of(t) + kh * nph;
#pragma unroll
                        for (int c = 0; c < 4; ++c) if (c < nph) kr[c] = kp[c];
                        mb = (!self && S.mask) ? S.mask[(size_t)j * S.mask_stride + t] : 1;
                    }
                    const float* qh = qs + kh * (dk / 2);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (c < nph) {
                            const unsigned w4[4] = {kr[c].x, kr[c].y, kr[c].z, kr[c].w};
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                s_ += qh[c * 8 + 2 * e] * __uint_as_float(w4[e] << 16) + qh[c * 8 + 2 * e + 1] * __uint_as_float(w4[e] & 0xffff0000u);
                        }
                    }
                }
                s_ += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s_), 0xB1, 0xf, 0xf, false));     // the other half (lane ^ 1)
                if (live && kh == 0) {                                   // (one lane of the pair carries the key into the reduction)
                    s_ *= scale;
                    if (mb == 0) s_ = -1.0e9f;                           // masked_fill(mask == 0, -1e9), mtn.py:226
                    sc[t] = s_;
                    const float M = fmaxf(mx, s_);
                    sum = sum * __expf(mx - M) + __expf(s_ - M);
                    mx = M;
                }
            }
            // row maximum and sum of exponentials behind ONE workgroup barrier: a thread's (max, sum) pair is rescaled to the wave's maximum
            // (one exponential), the wave's pair to the row's (eight independent exponentials) — no chain of dependent ones
            {
                const float wm = fh_cross_max(dec_row16_max(mx));
                sum *= __expf(mx - wm);
                sum = fh_cross_sum(fh_row16_sum(sum));
                if (lane == 0) { misc[wave] = wm; misc[16 + wave] = sum; }
            }
            __syncthreads();                                             // (also: every score is in LDS)
            mx = misc[0];
#pragma unroll
            for (int w_ = 1; w_ < DEC_NW; ++w_) mx = fmaxf(mx, misc[w_]);
            sum = 0.f;
#pragma unroll
            for (int w_ = 0; w_ < DEC_NW; ++w_) sum += misc[16 + w_] * __expf(misc[w_] - mx);
            const float inv = 1.0f / sum;
            // o[c] = sum_t P[t] V[t][c], P rounded to bf16 (the training kernels feed P to the MFMA in bf16): thread = (four columns, key part)
            float o[4] = {0.f, 0.f, 0.f, 0.f};
            for (int u = 0; u * nq + qt < m; ++u) {
                const int t = qt + u * nq;
                const float pr = bf16_to_f32(f32_to_bf16(__expf(sc[t] - mx) * inv));
                if (self && t == pos) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] += pr * vnew[c4 * 4 + k];
                } else {
                    u64 v4 = 0;
                    if (u < 4) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) if (q == u) v4 = vq[q];
                    } else v4 = ((const u64*)((const bf16_t*)krow_of(t) + d))[c4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] += pr * bf16_to_f32((bf16_t)(v4 >> (16 * k)));
                }
            }
            // the key parts a wave holds (64 / (dk / 4) of them) by shuffles, the waves' sums through LDS in wave order
#pragma unroll
            for (int k = 0; k < 4; ++k) {                                  // (DPP / permlane, not ds_bpermute: lanes with equal lane % (dk / 4))
                if (dk == 32) o[k] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(o[k]), 0x128, 0xf, 0xf, false));    // row_ror:8
                o[k] = fh_cross_sum(o[k]);
            }
            if (lane < dk / 4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) part[wave * dk + c4 * 4 + k] = o[k];
            }
            __syncthreads();
            if (dbg) dbg[si * 4 + 2] = wall_clock64();
            if (tid < dk / 2) {
                float y0 = 0.f, y1 = 0.f;
#pragma unroll
                for (int w_ = 0; w_ < DEC_NW; ++w_) { y0 += part[w_ * dk + tid * 2]; y1 += part[w_ * dk + tid * 2 + 1]; }
                dec_pub(rO, ((unsigned)j * d + hd * dk) / 2 + tid, dec_pack2(y0, y1), tag);
            }
        } break;
        case MTN_DEC_FINAL: {          // the decoder's final LayerNorm (mtn.py:161) -> the generator's bf16 operand (read by the NEXT kernel: plain stores)
            if (idx < W) {
                alive = dec_poll<DEC_THREADS>(rX, (unsigned)idx * d, d, x_tag_before(si), A.sync, [&](int i, unsigned v) { xf[i] = __uint_as_float(v); });
                gains_to_lds();
                __syncthreads();
                if (wave == 0) {
                    bf16_t* row = (bf16_t*)A.out_lp + (size_t)idx * d;
                    dec_ln_row(xf, gains, S.ln_eps, d, lane, [&](int c, float4 y) { *(u64*)(row + c) = dec_pack4(y); });
                }
            }
        } break;
        default: break;
        }
        if (dbg) dbg[si * 4 + 3] = wall_clock64();
        __syncthreads();                                                    // (LDS images are reused by the next stage)
        const bool last = S.kind == MTN_DEC_FINAL && idx == 0;
        si = next_mine(si);
        if (si < n_stages) { S = stage_of(si); prefetch(S); }
        // the unit that normalised row 0 has seen the last x of every producer: every workgroup read the generation long ago
        if (last && tid == 0) {
            // (a workgroup dispatched late must not find the next generation: wait for all check-ins — normally long complete)
            for (unsigned spins = 0; __hip_atomic_load(A.sync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x && spins < (1u << 22); ++spins) __builtin_amdgcn_s_sleep(2);
            __hip_atomic_store(A.sync + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(A.sync, gen >> 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

extern "C" int mtn_decode_step(const mtn_decode_args* a, const mtn_decode_stage* stages_device, int grid, void* stream) {
    MTN_CHECK_ARG(a && stages_device, "null arguments");
    MTN_CHECK_ARG(a->W >= 1 && a->W <= DEC_MAX_W, "1 .. 16 hypotheses per launch");
    MTN_CHECK_ARG(a->d >= 128 && a->d <= 1024 && (a->d == 128 || a->d == 256 || a->d == 512 || a->d == 1024), "d_model in {128, 256, 512, 1024}");
    MTN_CHECK_ARG(a->h >= 1 && a->d % a->h == 0 && (a->d / a->h == 32 || a->d / a->h == 64), "head size 32 or 64");
    MTN_CHECK_ARG(a->n_stages >= 1 && a->L >= 1 && a->L <= 1024, "bad stage count / maximum length");
    MTN_CHECK_ARG(a->max_m >= 1 && a->max_m <= 1024, "max_m (the longest memory of any MTN_DEC_CROSS stage) must be 1 .. 1024: the score row lives in LDS");
    MTN_CHECK_ARG(a->d_ff >= a->d && a->d_ff <= 4096 && a->d_ff % 32 == 0, "d_ff: a multiple of 32, at most 4096");
    // Every workgroup of the launch must be RESIDENT at once (they poll each other's granules; ~150 KiB of LDS each = one per CU): the grid
    // is clamped to the device's CU count, and a launch that does not fit is refused here instead of timing out on the device.
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) {
            mtn_set_error("mtn_decode_step: cannot read the device's compute-unit count");
            return MTN_ERR_LAUNCH;
        }
        n_cu = v;
    }
    if (grid > n_cu) grid = n_cu;
    const int n_unit = a->W * a->h, n_xw = a->d / 16;
    const int n_mid = (grid - n_unit - n_xw) < 128 ? (grid - n_unit - n_xw) : 128;
    MTN_CHECK_ARG(grid <= 256 && n_unit <= 128 && n_mid >= 16, "grid: at most one workgroup per CU (and per compute unit of THIS device); W x heads <= 128 attention units + d / 16 + >= 16 more workgroups must fit");
    // a workgroup computes at most 64 output features of a Linear (four 16-feature MFMA tiles dealt to its waves; its epilogue has one
    // thread per granule): the widest slices are those of the q|k|v projection and the first FFN Linear over the n_mid wide workgroups
    const int widest = 3 * a->d > a->d_ff ? 3 * a->d : a->d_ff;
    MTN_CHECK_ARG(((widest + n_mid - 1) / n_mid + 3) / 4 * 4 <= 64 && ((a->d + n_xw - 1) / n_xw + 3) / 4 * 4 <= 32, "a workgroup's slice of a Linear exceeds 64 (wide) / 32 (d_model) features: the grid is too small for this width");
    {
        const int per_w = ((widest + n_mid - 1) / n_mid + 3) / 4 * 4, per_x = ((a->d + n_xw - 1) / n_xw + 3) / 4 * 4;
        MTN_CHECK_ARG((per_w / 2) * a->W <= DEC_NW_USED * 64 && per_x * a->W <= DEC_NW_USED * 64, "epilogue granules per workgroup exceed its threads");
    }
    MTN_CHECK_ARG(a->xg && a->qg && a->og && a->hg && a->out_lp && a->tokens && a->lut && a->pe && a->pos && a->anc && a->sync, "null buffer");
    const int kmax = a->d_ff > a->d ? a->d_ff : a->d;
    MTN_CHECK_ARG(a->W * a->d <= 8192 && a->W * (kmax * 2 + 16) <= DEC_ACT_BYTES && a->n_stages <= DEC_MAX_STAGES, "W x d_model <= 8192, W x (2 max(d_model, d_ff) + 16) <= 66048 bytes of LDS, at most 160 stages");
    hipStream_t s = (hipStream_t)stream;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)decode_step_kernel<DEC_NW_USED>, hipFuncAttributeMaxDynamicSharedMemorySize, DEC_LDS) != hipSuccess) {
            mtn_set_error("mtn_decode_step: cannot opt into %d bytes of LDS", DEC_LDS);
            return MTN_ERR_LAUNCH;
        }
        attr = true;
    }
    DecKernelArgs KA;
    KA.a = *a;
    KA.stages = stages_device;
    KA.n_xw = n_xw; KA.n_mid = n_mid;
    hipLaunchKernelGGL(decode_step_kernel<DEC_NW_USED>, dim3(n_xw + n_mid + n_unit), dim3(DEC_NW_USED * 64), DEC_LDS, s, KA);
    MTN_CHECK_LAUNCH();
    return MTN_OK;
}

// ---- test support: `n_wg` workgroups that each hold `lds_bytes` of LDS (i.e. a share of a compute unit) for `usec` microseconds and do
// nothing else.  tests/test_decode_gpu.py runs it on a second stream to make the persistent step's residency assumption FAIL on purpose.
__global__ __launch_bounds__(64) void hold_cus_kernel(const unsigned long long ticks, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u64 t0 = wall_clock64();
    unsigned acc = 0;
    while (wall_clock64() - t0 < ticks) { acc += smem[(acc * 64 + threadIdx.x) & 1023]; __builtin_amdgcn_s_sleep(32); }
    if (acc == 0xffffffffu && sink) *sink = acc;          // (keeps the LDS allocation alive)
}
extern "C" int mtn_debug_hold_cus(int n_wg, int lds_bytes, int usec, void* stream) {
    MTN_CHECK_ARG(n_wg >= 1 && n_wg <= 4096 && lds_bytes >= 1024 && lds_bytes <= 160 * 1024 && usec >= 1 && usec <= 2000000, "1..4096 workgroups, 1 KiB..160 KiB of LDS each, at most 2 s");
    if (hipFuncSetAttribute((const void*)hold_cus_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) {
        mtn_set_error("mtn_debug_hold_cus: cannot opt into %d bytes of LDS", lds_bytes);
        return MTN_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(hold_cus_kernel, dim3(n_wg), dim3(64), lds_bytes, (hipStream_t)stream, (unsigned long long)usec * 100ull, (unsigned*)nullptr);
    MTN_CHECK_LAUNCH();
    return MTN_OK;
}
