// gemm.hip — grouped MFMA GEMM for every Linear on the MTN path (mtn.py:243-244,256-258,267,273-280)
// and their backward contractions.  One launch executes up to MTN_GEMM_MAX_GROUP independent problems
// (independent sublayers of one DecoderLayer share a launch so that the grid fills 256 CUs).
//
// Tile: 64x64 outputs per 256-thread workgroup (4 waves, each a 32x32 quadrant = 2x2 MFMA 16x16 tiles),
// depth 128 bytes per row per step (64 bf16 / 32 fp32).  Operands are staged global -> registers -> LDS
// (next tile's loads are in flight during the MFMAs of the current one); a "transposed" operand (stored
// contraction-major, e.g. dY and X in dW = dY^T X) is transposed in registers in 4x4 blocks on the way to
// LDS, so the MFMA fragment reads are always 16-byte, contraction-contiguous.
// LDS rows are padded to 144 bytes: the 16 rows of a fragment read land on 16 distinct 16-byte slots.
#include <stdlib.h>
#include <type_traits>

#include "common.h"

struct GemmGroup {
    int count;
    int plain_tile_order;      // ablation (MTN_GEMM_PLAIN_TILES=1): row-major tile order, no XCD-aware mapping
    int epi_pre;               // LDS-DMA kernels: epilogue operands (bias, gate, residual) loaded ahead of the contraction (MTN_GEMM_EPI_PRE=0: off)
    // 2-D XCD map of the LDS-DMA kernels (xcd_tile2d): problem g's tiles form rg x (8 / rg) blocks, one per XCD; 0 = band map
    unsigned char xcd_rg[MTN_GEMM_MAX_GROUP];
    int tile_start[MTN_GEMM_MAX_GROUP + 1];
    mtn_gemm_problem p[MTN_GEMM_MAX_GROUP];
};

int gemm_k512_try(int count, const mtn_gemm_problem* p, int min_tiles, hipStream_t s, int* tiles_out);      // gemm_k512.hip

// Optimiser epilogue of the parameter-gradient kernels (mtn_adam_fuse): second kernel argument of the T x T kernels only.
struct AdamSlot { float *p, *m, *v; void* lp; void* lpT; int ldT, write_grad; };
struct AdamGroup {
    const float* state;
    const float* grad_scale;
    float beta1, beta2, eps;
    int any;
    int lds_epilogue;          // every slot's copies are 16-byte addressable column-wise: gemm_tt_dma128_kernel re-lays the tile out in LDS
    AdamSlot a[MTN_GEMM_MAX_GROUP];
};
__device__ __forceinline__ AdamCoef adam_coef(const AdamGroup& G) { return adam_coef(G.state, G.grad_scale, G.beta1, G.beta2, G.eps); }
// Adam on 4 consecutive columns of one row of the parameter block; g[] = the gradient (accumulator values).
// Same arithmetic, in the same order, as adam_kernel (elementwise.hip): the two paths give the same bits.
template <typename T>
__device__ __forceinline__ void adam4(const AdamSlot& A, const AdamCoef& c, bool vec, size_t o, int row, int col, int nv, const float (&g)[4]) {
    float pv[4], mv[4], vv[4];
    typedef __attribute__((ext_vector_type(4))) float nt4;
    if (vec) {
        const nt4 p_ = __builtin_nontemporal_load((const nt4*)(A.p + o)), m_ = __builtin_nontemporal_load((const nt4*)(A.m + o)),
                  v_ = __builtin_nontemporal_load((const nt4*)(A.v + o));
#pragma unroll
        for (int k = 0; k < 4; ++k) { pv[k] = p_[k]; mv[k] = m_[k]; vv[k] = v_[k]; }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) { pv[k] = k < nv ? A.p[o + k] : 0.f; mv[k] = k < nv ? A.m[o + k] : 0.f; vv[k] = k < nv ? A.v[o + k] : 0.f; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) adam_update(pv[k], mv[k], vv[k], g[k], c);
    if (vec) {
        __builtin_nontemporal_store(nt4{pv[0], pv[1], pv[2], pv[3]}, (nt4*)(A.p + o));
        __builtin_nontemporal_store(nt4{mv[0], mv[1], mv[2], mv[3]}, (nt4*)(A.m + o));
        __builtin_nontemporal_store(nt4{vv[0], vv[1], vv[2], vv[3]}, (nt4*)(A.v + o));
    } else {
        for (int k = 0; k < nv; ++k) { A.p[o + k] = pv[k]; A.m[o + k] = mv[k]; A.v[o + k] = vv[k]; }
    }
    if (A.lp) {
        T* op = (T*)A.lp + o;
        if (vec) {
            if constexpr (sizeof(T) == 2)
                *(uint2*)op = make_uint2((uint32_t)f32_to_bf16(pv[0]) | ((uint32_t)f32_to_bf16(pv[1]) << 16), (uint32_t)f32_to_bf16(pv[2]) | ((uint32_t)f32_to_bf16(pv[3]) << 16));
            else *(float4*)op = make_float4(pv[0], pv[1], pv[2], pv[3]);
        } else for (int k = 0; k < nv; ++k) op[k] = LP<T>::from_f32(pv[k]);
    }
    if (A.lpT) {
        T* tp = (T*)A.lpT + (size_t)col * A.ldT + row;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < nv) tp[(size_t)k * A.ldT] = LP<T>::from_f32(pv[k]);
    }
}

// XCD-aware tile mapping.  Workgroups are dealt to the 8 XCDs round-robin (id % 8) and every XCD has its own L2, so the
// operand panels a problem's workgroups share are fetched once PER XCD that touches them.  Local workgroup t of a problem
// (global id first_id + t) is mapped so that the workgroups of one XCD cover a contiguous band of tiles — bands of output
// rows when the row operand is the larger one (its panels are then read by one XCD each), bands of columns otherwise.
// A bijection on [0, T) for any T and any first_id; placement is only a performance assumption, never a correctness one.
__device__ __forceinline__ void xcd_tile(const GemmGroup& grp, int g, int t, int T, int tiles_m, int tiles_n, bool row_band, int& tm, int& tn) {
    int idx = t;
    if (!grp.plain_tile_order && T >= 16) {
        const int c = t & 7, r = t >> 3;                     // class (same XCD) in order of first appearance, rank inside it
        const int per = T >> 3, rem = T & 7;
        idx = c * per + (c < rem ? c : rem) + r;
    }
    (void)g;
    if (row_band) { tm = idx / tiles_n; tn = idx - tm * tiles_n; }
    else { tn = idx / tiles_m; tm = idx - tn * tiles_m; }
}

// 2-D variant for the latency-bound LDS-DMA launches (a few hundred tiles per problem).  The band map above gives an XCD a
// contiguous run of T/8 tiles in row-major order, i.e. 1-3 row panels and ALL column panels: every XCD pulls the whole B
// operand (the weight).  Here the 8 XCDs form rg row groups x cg = 8/rg column groups and XCD (i, j) takes the tiles of row band
// i and column band j, so that a problem moves cg x A + rg x B bytes through the fabric instead of ~2 x A + 8 x B (the map that
// took the fused kernel's launches from 67 to 25 MB, csrc/fused.hip).  Needs tile counts divisible by the grouping (chosen on the
// host, GemmGroup::xcd_rg) and the problem's first workgroup on XCD 0; otherwise the band map stays.
__device__ __forceinline__ void xcd_tile2d(int rg, int t, int tiles_m, int tiles_n, int& tm, int& tn) {
    const int cg = 8 / rg;
    const int c = t & 7, r = t >> 3;                         // XCD (by dispatch order), rank of the tile inside it
    const int bm = tiles_m / rg, bn = tiles_n / cg;          // the XCD's block of tiles
    const int ri = c / cg, ci = c - ri * cg;
    const int lm = r / bn, ln = r - lm * bn;
    tm = ri * bm + lm;
    tn = ci * bn + ln;
}
// host: the grouping with the least fabric traffic, cg x (M x K) + rg x (N x K); 0 when the problem does not qualify
static int pick_xcd_rg(int first_tile, int tiles_m, int tiles_n, long M, long N) {
    if ((first_tile & 7) != 0 || ((tiles_m * tiles_n) & 7) != 0) return 0;
    int best = 0;
    double cost = 1e300;
    for (int rg = 1; rg <= 8; rg *= 2) {
        const int cg = 8 / rg;
        if (tiles_m % rg != 0 || tiles_n % cg != 0) continue;
        const double c = (double)cg * M + (double)rg * N;
        if (c < cost) { cost = c; best = rg; }
    }
    return best;
}

static constexpr int TILE = 64;
static constexpr int LDSROW = 144;  // bytes per LDS tile row (128 data + 16 pad)

// ---- staging of one 64-row x 128-byte operand tile -------------------------------------------------
template <typename T, bool TR> struct Stage {
    uint4 r[(TR && sizeof(T) == 4) ? 4 : 2];

    __device__ __forceinline__ void load(const T* __restrict__ base, int ld, int R, int K, int row0, int k0, int tid) {
        constexpr int EPV = LP<T>::EPV;
        if constexpr (!TR) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                int c = tid + 256 * j;
                int gr = row0 + (c >> 3), gk = k0 + (c & 7) * EPV;
                r[j] = (gr < R && gk < K) ? *(const uint4*)(base + (size_t)gr * ld + gk) : make_uint4(0, 0, 0, 0);
            }
        } else if constexpr (sizeof(T) == 2) {  // 64 k x 64 rows, 4x4 blocks, one per thread
            int kb = tid >> 4, rb = tid & 15;
            int gr = row0 + rb * 4;
            uint2 v[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                int gk = k0 + kb * 4 + kk;
                v[kk] = (gk < K && gr < R) ? *(const uint2*)(base + (size_t)gk * ld + gr) : make_uint2(0, 0);
            }
            // 4x4 transpose of 16-bit elements: v[k] = rows r0..r3 at k  ->  out[r] = k0..k3 at row r
            r[0].x = (v[0].x & 0xffffu) | (v[1].x << 16);        r[0].y = (v[2].x & 0xffffu) | (v[3].x << 16);
            r[0].z = (v[0].x >> 16) | (v[1].x & 0xffff0000u);    r[0].w = (v[2].x >> 16) | (v[3].x & 0xffff0000u);
            r[1].x = (v[0].y & 0xffffu) | (v[1].y << 16);        r[1].y = (v[2].y & 0xffffu) | (v[3].y << 16);
            r[1].z = (v[0].y >> 16) | (v[1].y & 0xffff0000u);    r[1].w = (v[2].y >> 16) | (v[3].y & 0xffff0000u);
        } else {  // fp32: 32 k x 64 rows, 128 blocks of 4x4; threads 128..255 idle
            if (tid < 128) {
                int kb = tid >> 4, rb = tid & 15;
                int gr = row0 + rb * 4;
                uint4 v[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    int gk = k0 + kb * 4 + kk;
                    v[kk] = (gk < K && gr < R) ? *(const uint4*)(base + (size_t)gk * ld + gr) : make_uint4(0, 0, 0, 0);
                }
                r[0] = make_uint4(v[0].x, v[1].x, v[2].x, v[3].x);
                r[1] = make_uint4(v[0].y, v[1].y, v[2].y, v[3].y);
                r[2] = make_uint4(v[0].z, v[1].z, v[2].z, v[3].z);
                r[3] = make_uint4(v[0].w, v[1].w, v[2].w, v[3].w);
            }
        }
    }

    __device__ __forceinline__ void store(unsigned char* s, int tid) const {
        if constexpr (!TR) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                int c = tid + 256 * j;
                *(uint4*)(s + (c >> 3) * LDSROW + (c & 7) * 16) = r[j];
            }
        } else if constexpr (sizeof(T) == 2) {
            int kb = tid >> 4, rb = tid & 15;
            unsigned char* p = s + (rb * 4) * LDSROW + kb * 8;
            *(uint2*)(p) = make_uint2(r[0].x, r[0].y);
            *(uint2*)(p + LDSROW) = make_uint2(r[0].z, r[0].w);
            *(uint2*)(p + 2 * LDSROW) = make_uint2(r[1].x, r[1].y);
            *(uint2*)(p + 3 * LDSROW) = make_uint2(r[1].z, r[1].w);
        } else {
            if (tid < 128) {
                int kb = tid >> 4, rb = tid & 15;
                unsigned char* p = s + (rb * 4) * LDSROW + kb * 16;
#pragma unroll
                for (int q = 0; q < 4; ++q) *(uint4*)(p + q * LDSROW) = r[q];
            }
        }
    }
};

// Epilogue for 4 consecutive output columns of one row: v += bias; relu; dropout; gate; v += residual; store fp32 / lowp.
template <typename T, bool ADAM = false>
__device__ __forceinline__ void epilogue4(const mtn_gemm_problem& P, const DropState& ds, bool vec, int row, int col, int N, const f32x4_t& acc,
                                          const AdamSlot* adam = nullptr, const AdamCoef* coef = nullptr, float* v_out = nullptr, float* gate_out = nullptr,
                                          const bool pre = false, const float4 pre_bias = float4{0.f, 0.f, 0.f, 0.f},
                                          const float4 pre_res = float4{0.f, 0.f, 0.f, 0.f}, const uint2 pre_gate = uint2{0u, 0u}) {
    // pre: bias / gate / residual of these four outputs, loaded by the caller while its contraction ran (vec layouts only; by value:
    // through a pointer or a reference the caller's register arrays went to scratch memory)
    float v[4] = {acc[0], acc[1], acc[2], acc[3]};
    const size_t o = (size_t)row * P.ldc + col;
    const int nv = (col + 4 <= N) ? 4 : N - col;
    if constexpr (ADAM) {
        if (adam->p) {                                    // the accumulators are the gradient of p[row][col..col+3]
            adam4<T>(*adam, *coef, vec, o, row, col, nv, v);
            if (!adam->write_grad) return;
            if (vec) *(float4*)(P.out_f32 + o) = make_float4(v[0], v[1], v[2], v[3]);
            else for (int r = 0; r < nv; ++r) P.out_f32[o + r] = v[r];
            return;
        }
    }
    if (P.bias) {
        if (vec) { const float4 b = pre ? pre_bias : *(const float4*)(P.bias + col); v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
        else for (int r = 0; r < nv; ++r) v[r] += P.bias[col + r];
    }
    if (P.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
    if (ds.on && !P.lp_drop_after_residual) {
        const DropBase db = drop_base((uint64_t)row * (uint64_t)N + col);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = drop_keep_at(ds, db, r) ? v[r] * ds.scale : 0.f;
    }
    if (P.gate) {
        const T* gp = (const T*)P.gate + o;
        if (vec) {
            if constexpr (sizeof(T) == 2) {
                const uint2 u = pre ? pre_gate : *(const uint2*)gp;
                const float g0 = __uint_as_float(u.x << 16), g1 = __uint_as_float(u.x & 0xffff0000u), g2 = __uint_as_float(u.y << 16), g3 = __uint_as_float(u.y & 0xffff0000u);
                v[0] = g0 > 0.f ? v[0] * P.gate_scale : 0.f; v[1] = g1 > 0.f ? v[1] * P.gate_scale : 0.f;
                v[2] = g2 > 0.f ? v[2] * P.gate_scale : 0.f; v[3] = g3 > 0.f ? v[3] * P.gate_scale : 0.f;
                if (gate_out) { gate_out[0] = g0; gate_out[1] = g1; gate_out[2] = g2; gate_out[3] = g3; }
            } else {
                float4 gq = *(const float4*)gp;
                v[0] = gq.x > 0.f ? v[0] * P.gate_scale : 0.f; v[1] = gq.y > 0.f ? v[1] * P.gate_scale : 0.f;
                v[2] = gq.z > 0.f ? v[2] * P.gate_scale : 0.f; v[3] = gq.w > 0.f ? v[3] * P.gate_scale : 0.f;
            }
        } else for (int r = 0; r < nv; ++r) v[r] = LP<T>::to_f32(gp[r]) > 0.f ? v[r] * P.gate_scale : 0.f;
    }
    if (P.residual) {
        const float* rp = P.residual + (size_t)row * P.ldr + col;
        if (vec) { const float4 q = pre ? pre_res : *(const float4*)rp; v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w; }
        else for (int r = 0; r < nv; ++r) v[r] += rp[r];
    }
    if (P.out_f32) {
        if (vec) *(float4*)(P.out_f32 + o) = make_float4(v[0], v[1], v[2], v[3]);
        else for (int r = 0; r < nv; ++r) P.out_f32[o + r] = v[r];
    }
    if (v_out) { v_out[0] = v[0]; v_out[1] = v[1]; v_out[2] = v[2]; v_out[3] = v[3]; }
    if (P.out_lp) {
        T* op = (T*)P.out_lp + o;
        if (ds.on && P.lp_drop_after_residual) {          // the stored f32 value stays whole; only this copy goes through the dropout
            const DropBase db = drop_base((uint64_t)row * (uint64_t)N + col);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = drop_keep_at(ds, db, r) ? v[r] * ds.scale : 0.f;
        }
        if (vec) {
            if constexpr (sizeof(T) == 2) {
                *(uint2*)op = make_uint2((uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16), (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16));
            } else {
                *(float4*)op = make_float4(v[0], v[1], v[2], v[3]);
            }
        } else for (int r = 0; r < nv; ++r) op[r] = LP<T>::from_f32(v[r]);
    }
}


// ---------------------------------------------------------------- LayerNorm-backward epilogue (mtn_ln_epilogue, include/mtn_hip.h)
// Round 4.  A sublayer's backward ended with   [producer of dq] -> GEMM g = dq W -> LayerNorm-backward launch (5.8-6.7 us, 43 per step).
// LayerNorm backward needs two sums over the whole row of g — which no 64-column tile of the GEMM has — but both are LINEAR in dq:
//     s1 = sum_c a2_c g_c = dq . u,  u = W a2          s2 = sum_c a2_c (x_c - mean) g_c = (1 / rstd) dq . (q - c),  c = b + W b2
// (q = the saved projection output: q - c = rstd W (a2 (x - mean))).  So the kernel that produces dq writes the partial dot
// products of its own columns ({dq . u, dq . (q - c)} per row and column block: plain stores, no cross-workgroup traffic inside a
// launch), and this epilogue adds them up, applies LayerNorm backward to the accumulators and writes dx (+ the masked compute-dtype
// copy the next sublayer's backward reads, + the da2 | db2 partial rows): the LayerNorm-backward launch is gone.
static constexpr int LNE_MAX_SLOTS = 8;
static constexpr int LNE_LDS_EXTRA = 4096;      // LDS behind a kernel's operand stages: row sums of a tile (consume: 8 B per row; emit: 8 B per row and wave column)
struct LnEpiSlot {
    int mode, np;
    const float* fold;
    float* part;
    const float *x, *a2, *mean, *rstd, *dres;
    float* dx;
    void* dx_lp;
    float* colpart;
    mtn_dropout dx_lp_drop;
    float gate_inv_scale, eps;
};
struct LnEpiGroup {
    unsigned char slot_of[MTN_GEMM_MAX_GROUP];   // 0 = plain problem, k + 1 = s[k]
    LnEpiSlot s[LNE_MAX_SLOTS];
};
struct NoLn {};
template <bool ON> struct LnArg { typedef NoLn type; };
template <> struct LnArg<true> { typedef LnEpiGroup type; };

// Accumulator geometry shared by the LDS-DMA kernels: for row tile i < RT and column tile j < CT a lane holds output row
// row_base + 16 i + l15 and columns col_base + 16 j + 4 lg .. +3.  The workgroup tile has BMT rows from tile_row0 and 8 * BMT threads.
// `lds`: >= 2 * BMT floats of scratch; the caller guarantees nothing else is read from it any more once every wave is here.
template <int RT, int CT>
struct LnConsumeLoads {               // everything the consume epilogue reads from memory, as issued (see ln_consume_issue)
    float mu[RT], rs[RT];
    float4 xv[RT][CT], dv[RT][CT], av[CT];
    float p1, p2;
};
// Issue the epilogue's loads: the row statistics, x, the residual-branch gradient, the gains, and this thread's share of the tile's
// row-sum partials (row tid / 8, entries tid % 8, + 8, ...).  The kernels call this right behind their first operand stages: the
// round trip then hides under the contraction instead of following it.
template <int RT, int CT>
__device__ __forceinline__ void ln_consume_issue(const LnEpiSlot& E, LnConsumeLoads<RT, CT>& Q, const int M, const int N, const int tile_row0,
                                                 const int row_base, const int col_base, const int l15, const int lg, const int tid, const bool sums = true) {
    // (sums: this thread is one of the 8 x tile-rows threads that gather the row-sum partials; a sixteen-wave workgroup has twice as many)
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int row = row_base + 16 * i + l15, rc = row < M ? row : M - 1;
        Q.mu[i] = E.mean[rc]; Q.rs[i] = E.rstd[rc];
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            const int col = col_base + 16 * j + 4 * lg;
            Q.xv[i][j] = Q.dv[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (col_base + 16 * j < N) {
                Q.xv[i][j] = *(const float4*)(E.x + (size_t)rc * N + col);
                if (E.dres) Q.dv[i][j] = *(const float4*)(E.dres + (size_t)rc * N + col);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < CT; ++j) {
        Q.av[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col_base + 16 * j < N) Q.av[j] = *(const float4*)(E.a2 + col_base + 16 * j + 4 * lg);
    }
    Q.p1 = Q.p2 = 0.f;
    const int pr = tile_row0 + (tid >> 3);
    if (pr < M && sums) {
        const float2* pp = (const float2*)E.part + (size_t)pr * E.np;
        for (int k = tid & 7; k < E.np; k += 8) { const float2 t = pp[k]; Q.p1 += t.x; Q.p2 += t.y; }
    }
}
// The tile's row sums -> lds[2 * row], lds[2 * row + 1] (a region of their own behind the operand stages).  Called by every thread once
// its loads have landed and BEFORE a workgroup barrier that precedes the epilogue — the kernels use the barrier of their last stage,
// so the epilogue itself needs none.
template <int RT, int CT>
__device__ __forceinline__ void ln_consume_publish(const LnConsumeLoads<RT, CT>& Q, const int tid, float* lds, const bool sums = true) {
    const float p1 = fh_row8_sum(Q.p1);                    // the row's eight threads are eight consecutive lanes: fixed summation order
    const float p2 = fh_row8_sum(Q.p2);
    if ((tid & 7) == 0 && sums) *(float2*)(lds + 2 * (tid >> 3)) = make_float2(p1, p2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
template <int RT, int CT, int BMT>
__device__ __forceinline__ void ln_consume_epilogue(const LnEpiSlot& E, LnConsumeLoads<RT, CT>& Q, const f32x4_t (&acc)[RT][CT], const int M, const int N,
                                                    const int tile_row0, const int row_base, const int col_base, const int l15, const int lg, const int tid, float* lds) {
    const DropState nds = drop_init(E.dx_lp_drop);
    const float inv_d = 1.0f / (float)N;
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int row = row_base + 16 * i + l15, lr = row - tile_row0;
        const bool live = row < M;
        const float S1 = lds[2 * lr], P2 = lds[2 * lr + 1];
        const float r = Q.rs[i];
        const float std_u = fmaxf(1.0f / r - E.eps, 1e-30f);
        const float c1 = r * S1 * inv_d;
        const float c2 = P2 * r / (std_u * (float)(N - 1));            // s2 = P2 / rstd
        const int pb = ((row_base + 16 * i) >> 3) + (l15 >> 3);      // da2 | db2 partial row: 8 rows each, as mtn_layernorm_bwd
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            if (col_base + 16 * j >= N) continue;
            const int col = col_base + 16 * j + 4 * lg;
            const float g[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            const float a[4] = {Q.av[j].x, Q.av[j].y, Q.av[j].z, Q.av[j].w};
            const float x4[4] = {Q.xv[i][j].x, Q.xv[i][j].y, Q.xv[i][j].z, Q.xv[i][j].w};
            const float d4[4] = {Q.dv[i][j].x, Q.dv[i][j].y, Q.dv[i][j].z, Q.dv[i][j].w};
            float o[4], ga[4], gb[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float e = x4[k] - Q.mu[i];
                o[k] = r * (g[k] * a[k]) - c1 - c2 * e + d4[k];
                ga[k] = live ? g[k] * e * r : 0.f;
                gb[k] = live ? g[k] : 0.f;
            }
            if (live) {
                const size_t eo = (size_t)row * N + col;
                *(float4*)(E.dx + eo) = make_float4(o[0], o[1], o[2], o[3]);
                if (E.dx_lp) {
                    if (nds.on) {
                        const DropBase db = drop_base((uint64_t)eo);
#pragma unroll
                        for (int k = 0; k < 4; ++k) o[k] = drop_keep_at(nds, db, k) ? o[k] * nds.scale : 0.f;
                    }
                    store_lp4<bf16_t>((bf16_t*)E.dx_lp + eo, make_float4(o[0], o[1], o[2], o[3]));
                }
            }
            if (E.colpart) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { ga[k] = fh_row8_sum(ga[k]); gb[k] = fh_row8_sum(gb[k]); }
                if ((l15 & 7) == 0 && pb * 8 < M) {
                    float* cp = E.colpart + (size_t)pb * 2 * N + col;
                    *(float4*)cp = make_float4(ga[0], ga[1], ga[2], ga[3]);
                    *(float4*)(cp + N) = make_float4(gb[0], gb[1], gb[2], gb[3]);
                }
            }
        }
    }
}
// MTN_LN_EMIT: the row-sum partials of a GEMM that produces dq itself.  p1[i], p2[i]: this lane's sums over its own columns of
// row tile i; summed over the lane groups, then over the WCN waves that share the tile's rows (through LDS, in wave order), and
// stored as pair `blk` (the tile's 64-column block) of the row.
template <int RT, int BMT, int WCN, int GRPW = WCN>       // GRPW = waves (across the columns) that share one partial block
__device__ __forceinline__ void ln_emit_partials(const LnEpiSlot& E, float (&p1)[RT], float (&p2)[RT], const int M, const int nblk, const int blk,
                                                 const int tile_row0, const int row_base, const int wc, const int l15, const int lg, const int tid, float* lds) {
#pragma unroll
    for (int i = 0; i < RT; ++i) { p1[i] = fh_cross_sum(p1[i]); p2[i] = fh_cross_sum(p2[i]); }
    if (lg == 0) {                                         // (lds: a region of its own behind the operand stages)
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int lr = row_base - tile_row0 + 16 * i + l15;
            lds[(lr * WCN + wc) * 2] = p1[i];
            lds[(lr * WCN + wc) * 2 + 1] = p2[i];
        }
    }
    __syncthreads();
    constexpr int NB = WCN / GRPW;                         // partial blocks per tile row
    if (tid < BMT * NB) {
        const int lr = tid / NB, b = tid - lr * NB;
        if (tile_row0 + lr < M) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < GRPW; ++w) { s1 += lds[(lr * WCN + b * GRPW + w) * 2]; s2 += lds[(lr * WCN + b * GRPW + w) * 2 + 1]; }
            ((float2*)E.part)[(size_t)(tile_row0 + lr) * nblk + blk + b] = make_float2(s1, s2);
        }
    }
}

struct NoAdam {};
template <bool ON> struct AdamArg { typedef NoAdam type; };
template <> struct AdamArg<true> { typedef AdamGroup type; };

template <typename T, bool A_T, bool B_T>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmGroup grp, const typename AdamArg<A_T && B_T>::type adam) {
    constexpr int BK = 128 / (int)sizeof(T);
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TILE * LDSROW];
    unsigned char* sA = smem;
    unsigned char* sB = smem + TILE * LDSROW;

    const int tid = threadIdx.x;
    int g = 0;
    while (g + 1 < grp.count && (int)blockIdx.x >= grp.tile_start[g + 1]) ++g;
    const mtn_gemm_problem& P = grp.p[g];
    const int M = P.M, N = P.N, K = P.K;
    const int tiles_n = (N + TILE - 1) / TILE;
    const int t = (int)blockIdx.x - grp.tile_start[g];
    const int tiles_m = (M + TILE - 1) / TILE;
    int tm_, tn_;
    xcd_tile(grp, g, t, tiles_m * tiles_n, tiles_m, tiles_n, M >= N, tm_, tn_);
    const int row0 = tm_ * TILE, col0 = tn_ * TILE;
    const T* __restrict__ A = (const T*)P.A;
    const T* __restrict__ B = (const T*)P.B;
    const bool do_rowsum = (P.rowsum_out != nullptr) && (col0 == 0);

    const int lane = tid & 63, w = tid >> 6;
    const int wr = w >> 1, wc = w & 1, lg = lane >> 4, l15 = lane & 15;

    f32x4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float rsum = 0.f;

    Stage<T, A_T> stA;
    Stage<T, B_T> stB;
    stA.load(A, P.lda, M, K, row0, 0, tid);
    stB.load(B, P.ldb, N, K, col0, 0, tid);
    const DropState ds = drop_init(P.drop);        // the dropout key (scalar seed load + hashing) rides under the first tile loads

    for (int k0 = 0; k0 < K; k0 += BK) {
        stA.store(sA, tid);
        stB.store(sB, tid);
        __syncthreads();
        if (k0 + BK < K) {  // next tile's global loads fly during the MFMAs below
            stA.load(A, P.lda, M, K, row0, k0 + BK, tid);
            stB.load(B, P.ldb, N, K, col0, k0 + BK, tid);
        }
        if (do_rowsum && tid < TILE) {
            const uint4* rp = (const uint4*)(sA + tid * LDSROW);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                uint4 u = rp[q];
                if constexpr (sizeof(T) == 2) {
                    rsum += __uint_as_float(u.x << 16) + __uint_as_float(u.x & 0xffff0000u) + __uint_as_float(u.y << 16) +
                            __uint_as_float(u.y & 0xffff0000u) + __uint_as_float(u.z << 16) + __uint_as_float(u.z & 0xffff0000u) +
                            __uint_as_float(u.w << 16) + __uint_as_float(u.w & 0xffff0000u);
                } else {
                    rsum += __uint_as_float(u.x) + __uint_as_float(u.y) + __uint_as_float(u.z) + __uint_as_float(u.w);
                }
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *(const uint4*)(sA + (wr * 32 + i * 16 + l15) * LDSROW + ks * 64 + lg * 16);
                b[i] = *(const uint4*)(sB + (wc * 32 + i * 16 + l15) * LDSROW + ks * 64 + lg * 16);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma16<T>(acc[i][j], b[j], a[i]);     // transposed accumulator: see epilogue
        }
        __syncthreads();
    }

    if (do_rowsum && tid < TILE && row0 + tid < M) P.rowsum_out[row0 + tid] = rsum;

    // ---- epilogue: operands were swapped in the MFMAs, so a lane holds one output row (l15) x four consecutive columns
    const bool vec_ok = ((N | P.ldc | (P.residual ? P.ldr : 0)) & 3) == 0;
    if constexpr (A_T && B_T) {
        AdamCoef coef;
        if (adam.any) coef = adam_coef(adam);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = row0 + wr * 32 + i * 16 + l15;
            if (row >= M) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = col0 + wc * 32 + j * 16 + lg * 4;
                if (col >= N) continue;
                epilogue4<T, true>(P, ds, vec_ok && col + 3 < N, row, col, N, acc[i][j], &adam.a[g], &coef);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = row0 + wr * 32 + i * 16 + l15;
            if (row >= M) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = col0 + wc * 32 + j * 16 + lg * 4;
                if (col >= N) continue;
                epilogue4<T>(P, ds, vec_ok && col + 3 < N, row, col, N, acc[i][j]);
            }
        }
    }
}

// ====================================================================================================================
// dW = dY^T X (both operands contraction-major) with 128x128 tiles: the parameter-gradient launches carry ~2000 64x64
// workgroups each re-reading its two operand panels; at 128x128 every panel byte feeds twice the outputs, which halves
// the bytes each CU pulls (the bound, see gemm_dma_kernel).  4 waves x (4x4 MFMA tiles), 128 B of contraction per step,
// 4x4 in-register block transposes on the way to LDS exactly as in gemm_kernel<T,true,true>.
// ====================================================================================================================
template <typename T> struct StageT128 {      // one 128-row x 128-byte contraction-major operand tile
    uint4 r[sizeof(T) == 2 ? 4 : 4];
    __device__ __forceinline__ void load(const T* __restrict__ base, int ld, int R, int K, int row0, int k0, int tid) {
        if constexpr (sizeof(T) == 2) {       // 64 k x 128 rows: 16 x 32 blocks of 4x4, two per thread
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int blk = tid + 256 * j, kb = blk >> 5, rb = blk & 31;
                const int gr = row0 + rb * 4;
                uint2 v[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int gk = k0 + kb * 4 + kk;
                    v[kk] = (gk < K && gr < R) ? *(const uint2*)(base + (size_t)gk * ld + gr) : make_uint2(0, 0);
                }
                r[2 * j].x = (v[0].x & 0xffffu) | (v[1].x << 16);        r[2 * j].y = (v[2].x & 0xffffu) | (v[3].x << 16);
                r[2 * j].z = (v[0].x >> 16) | (v[1].x & 0xffff0000u);    r[2 * j].w = (v[2].x >> 16) | (v[3].x & 0xffff0000u);
                r[2 * j + 1].x = (v[0].y & 0xffffu) | (v[1].y << 16);    r[2 * j + 1].y = (v[2].y & 0xffffu) | (v[3].y << 16);
                r[2 * j + 1].z = (v[0].y >> 16) | (v[1].y & 0xffff0000u); r[2 * j + 1].w = (v[2].y >> 16) | (v[3].y & 0xffff0000u);
            }
        } else {                              // 32 k x 128 rows: 8 x 32 blocks, one per thread
            const int kb = tid >> 5, rb = tid & 31;
            const int gr = row0 + rb * 4;
            uint4 v[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int gk = k0 + kb * 4 + kk;
                v[kk] = (gk < K && gr < R) ? *(const uint4*)(base + (size_t)gk * ld + gr) : make_uint4(0, 0, 0, 0);
            }
            r[0] = make_uint4(v[0].x, v[1].x, v[2].x, v[3].x);
            r[1] = make_uint4(v[0].y, v[1].y, v[2].y, v[3].y);
            r[2] = make_uint4(v[0].z, v[1].z, v[2].z, v[3].z);
            r[3] = make_uint4(v[0].w, v[1].w, v[2].w, v[3].w);
        }
    }
    __device__ __forceinline__ void store(unsigned char* s, int tid) const {
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int blk = tid + 256 * j, kb = blk >> 5, rb = blk & 31;
                unsigned char* p = s + (rb * 4) * LDSROW + kb * 8;
                *(uint2*)(p) = make_uint2(r[2 * j].x, r[2 * j].y);
                *(uint2*)(p + LDSROW) = make_uint2(r[2 * j].z, r[2 * j].w);
                *(uint2*)(p + 2 * LDSROW) = make_uint2(r[2 * j + 1].x, r[2 * j + 1].y);
                *(uint2*)(p + 3 * LDSROW) = make_uint2(r[2 * j + 1].z, r[2 * j + 1].w);
            }
        } else {
            const int kb = tid >> 5, rb = tid & 31;
            unsigned char* p = s + (rb * 4) * LDSROW + kb * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) *(uint4*)(p + q * LDSROW) = r[q];
        }
    }
};


// ====================================================================================================================
// Fast path for row-major x row-major problems (forward Linears, and dX = dY (W^T)^T through the transposed weight
// copy): the problems on this path are SMALL (M = B*L = 640..4096 rows, K = 512..2048) and latency-bound, so the
// kernel is built around memory-level parallelism rather than MFMA issue rate:
//   * a K-chunk of 512 BYTES per row (256 bf16 / 128 fp32) per stage, two stages = 128 KiB of the CU's 160 KiB LDS;
//     for K <= 2 chunks (every d_model=512 contraction) the WHOLE K extent of both operands is in flight at once;
//   * operands go HBM/L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction), no VGPR round
//     trip; out-of-range K chunks are pushed past the buffer descriptor's bound and arrive as zeros;
//   * LDS rows are 512 B, so 16 rows of a fragment read would share one 16-byte slot: the 16-byte chunk index is
//     XOR-ed with (row & 15).  LDS-DMA writes lane-linear, so the swizzle is applied to the per-lane SOURCE address
//     and again on the fragment read (same involution);
//   * counted vmcnt + raw s_barrier: the next stage stays in flight across the barrier.
// ====================================================================================================================

typedef __attribute__((address_space(3))) void lds_void_t;

// one operand tile of ROWS rows x ROWB bytes (512 or 256): ROWS*ROWB/4096 LDS-DMA instructions per wave, each 1 KiB
// (2 rows of 512 B or 4 rows of 256 B); 16-byte slots XOR-swizzled with (row & 15) inside the row
template <typename T, int ROWS, int ROWB, int NW = 4>
__device__ __forceinline__ void dma_issue_tile(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_tile, int ld_bytes, int R,
                                               int K, int row0, int k0, int wave, int lane) {
    constexpr int EPV = LP<T>::EPV;
    constexpr int CPR = ROWB / 16;                               // 16-byte chunks per row (32 or 16)
    constexpr int RPI = 1024 / ROWB;                             // rows per wave-instruction (2 or 4)
#pragma unroll
    for (int j = 0; j < ROWS * ROWB / (1024 * NW); ++j) {
        const int row = (j * NW + wave) * RPI + lane / CPR;      // tile row written by this lane
        const int c = (lane % CPR) ^ (row & 15);                 // source chunk that lands in slot (lane % CPR)
        const int gk = k0 + c * EPV;
        int grow = row0 + row;
        grow = grow < R ? grow : R - 1;                          // rows past the end only feed outputs that are never stored
        unsigned voff = (gk < K) ? (unsigned)grow * (unsigned)ld_bytes + (unsigned)gk * (unsigned)sizeof(T) : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(lds_tile + (j * NW + wave) * 1024), 16, voff, 0, 0, 0);
    }
}

// B operand stored CONTRACTION-MAJOR ([K][N] row-major: a weight matrix W[out k][in n] used as dX = dY W — no transposed copy of
// W has to exist): the [k][BN columns] tile lands in LDS as it is (rows of BN * 2 bytes) and the MFMA fragments come out of the
// transposing LDS read ds_read_b64_tr_b16 (semantics: gemm_tt_dma_kernel above).  16-byte slots are XOR-swizzled so that the 8
// rows one LDS cycle of the read touches (k0..k0+3 of two lane groups, 8 rows apart) sit in 8 different 32-byte bank groups:
// 128-byte rows (BN = 64): even / odd rows own the two halves of the bank sweep, the slot PAIR moves with bits 1 and 3 of k;
// 64-byte rows (BN = 32): four consecutive rows own a quarter each, bit 3 of k selects the 32-byte half.
template <int BN> __device__ __forceinline__ int kn_swz(int krow) {
    if constexpr (BN == 64) return ((((krow >> 1) & 1) | (((krow >> 3) & 1) << 1)) << 1);
    else return ((krow >> 3) & 1) << 1;
}
template <int BN, int ROWB, int NW = 4>
__device__ __forceinline__ void dma_issue_tile_kn(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_tile, int ld_bytes, int N, int K,
                                                  int col0, int k0, int wave, int lane) {
    constexpr int RP = BN * 2;                                   // bytes per tile row
    constexpr int CPR = RP / 16;                                 // 16-byte chunks per row (8 or 4)
    constexpr int RPI = 1024 / RP;                               // k rows per wave-instruction (8 or 16)
#pragma unroll
    for (int j = 0; j < BN * ROWB / (1024 * NW); ++j) {
        const int krow = (j * NW + wave) * RPI + lane / CPR;
        const int c = (lane % CPR) ^ kn_swz<BN>(krow);           // source chunk that lands in slot (lane % CPR)
        const int gk = k0 + krow, gn = col0 + c * 8;
        unsigned voff = (gk < K && gn < N) ? (unsigned)gk * (unsigned)ld_bytes + (unsigned)gn * 2u : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(lds_tile + (j * NW + wave) * 1024), 16, voff, 0, 0, 0);
    }
}
// fragment for column tile n_off (multiple of 16 inside the tile) and contraction step ks (32 rows) of a [k][BN] image
template <int BN>
__device__ __forceinline__ uint4 kn_frag(const unsigned char* img, int n_off, int ks, int l15, int lg) {
    uint4 f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int krow = ks * 32 + 8 * lg + 4 * r + (l15 >> 2);
        const int col = n_off + 4 * (l15 & 3);
        const int slot = (col >> 3) ^ kn_swz<BN>(krow);
        const unsigned addr = (unsigned)(size_t)(img + krow * (BN * 2) + slot * 16 + (col & 7) * 2);
        unsigned long long v;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
        if (r == 0) { f.x = (unsigned)v; f.y = (unsigned)(v >> 32); }
        else { f.z = (unsigned)v; f.w = (unsigned)(v >> 32); }
    }
    return f;
}

// Tile BM x BN (64x64, 32x64 or 32x32), 4 waves as 2x2, each wave (BM/32) x (BN/32) MFMA 16x16 tiles.  The smaller tiles
// exist because these launches are bound by how fast ONE CU can pull its operand panels (~27 GB/s per CU measured, LDS-DMA
// and register staging alike): a [640 x 512] output is 80 workgroups of 128 KiB at 64x64 but 320 workgroups of 64 KiB at
// 32x32 — the whole chip pulls instead of a third of it.
// ROWB = bytes of contraction per row per stage: 512 (whole K <= 512-byte contractions in flight at once: best latency for
// launches of one round) or 256 (half the LDS: twice the resident workgroups, for launches that would otherwise need a
// second round).
// NBUF = 4 (round 3, long contractions in one round of workgroups): a ring of four half-size stages, stage s+3 issued while stage s
// is computed — three stages in flight instead of "refill after compute", one barrier per stage instead of two.
// NW = 8 (round 3: the 64 x 64 tile on full stages, ONE workgroup per CU): eight waves instead of four.  A wave gets one 1 KiB
// load through every ~180 ns however many it has queued (profiles/r03_fh_order.txt), so a CU's fill rate is its resident waves x
// ~5.6 GB/s: four waves pulled 22-26 GB/s where the two-workgroups-per-CU variants (eight waves) reach twice that.  Waves as 2 row
// halves x 4 column quarters (wave tile 32 x 16).
template <typename T, int BM, int BN, int DMA_ROWB, bool BTR = false, int NBUF = 2, int NW = 4, bool LNE = false>
__global__ __launch_bounds__(64 * NW) void gemm_dma_kernel(const GemmGroup grp, const typename LnArg<LNE>::type lne) {
    constexpr int BK = DMA_ROWB / (int)sizeof(T);            // 256 / 128 bf16, 128 / 64 fp32
    constexpr int WR = NW == 16 ? 4 : 2;                     // waves across the rows (NW = 16: a 4 x 4 grid of 16 x 16 wave tiles on the 64 x 64 tile)
    constexpr int WC = NW / WR;                              // waves across the columns (2 or 4)
    constexpr int TM = BM / (16 * WR), TN = BN / (16 * WC);  // MFMA tiles per wave
    constexpr int A_BYTES = BM * DMA_ROWB, B_BYTES = BN * DMA_ROWB, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int NDMA = (BM + BN) * DMA_ROWB / (1024 * NW); // LDS-DMA instructions per wave per stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    int g = 0;
    while (g + 1 < grp.count && (int)blockIdx.x >= grp.tile_start[g + 1]) ++g;
    const mtn_gemm_problem& P = grp.p[g];
    const int xcd_rg = grp.xcd_rg[g];
    const int first_tile = grp.tile_start[g];
    int lne_sl = 0;
    if constexpr (LNE) lne_sl = lne.slot_of[g];
    const int M = P.M, N = P.N, K = P.K;
    const int tiles_n = (N + BN - 1) / BN;
    const int t = (int)blockIdx.x - first_tile;
    const int tiles_m = (M + BM - 1) / BM;
    int tm_, tn_;
    if (xcd_rg) xcd_tile2d(xcd_rg, t, tiles_m, tiles_n, tm_, tn_);
    else xcd_tile(grp, g, t, tiles_m * tiles_n, tiles_m, tiles_n, M >= N, tm_, tn_);
    const int row0 = tm_ * BM, col0 = tn_ * BN;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC, lg = lane >> 4, l15 = lane & 15;
    const int lda_b = P.lda * (int)sizeof(T), ldb_b = P.ldb * (int)sizeof(T);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)P.A, 0, (M - 1) * lda_b + K * (int)sizeof(T), 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = BTR ? __builtin_amdgcn_make_buffer_rsrc((void*)P.B, 0, (K - 1) * ldb_b + N * (int)sizeof(T), 0x00020000)
                                          : __builtin_amdgcn_make_buffer_rsrc((void*)P.B, 0, (N - 1) * ldb_b + K * (int)sizeof(T), 0x00020000);
    // the B tile of a stage: row-major [BN rows][DMA_ROWB bytes of k], or (BTR: B stored [K][N]) [BK k-rows][BN columns]
    auto issue_b = [&](unsigned char* dst, int k0) {
        if constexpr (BTR) dma_issue_tile_kn<BN, DMA_ROWB, NW>(rB, dst, ldb_b, N, K, col0, k0, wave, lane);
        else dma_issue_tile<T, BN, DMA_ROWB, NW>(rB, dst, ldb_b, N, K, col0, k0, wave, lane);
    };

    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nstages = (K + BK - 1) / BK;
    auto issue_stage = [&](int st) {
        unsigned char* dst = smem + (st & (NBUF - 1)) * STAGE_BYTES;
        dma_issue_tile<T, BM, DMA_ROWB, NW>(rA, dst, lda_b, M, K, row0, st * BK, wave, lane);
        issue_b(dst + A_BYTES, st * BK);
    };
    // prologue: NBUF = 2: up to two stages in flight; NBUF = 4: up to three (NDMA LDS-DMA instructions per wave per stage, always)
    for (int st = 0; st < (NBUF == 2 ? 2 : NBUF - 1) && st < nstages; ++st) issue_stage(st);
    // LayerNorm epilogue (LNE): its loads go out right behind the first stages, so that their round trip hides under the contraction.
    // The counted waits below stay as they are: with these loads in the queue a wait for stage s also waits for the loads issued
    // before the stages behind s — at most a few instructions' worth of over-waiting in the first two iterations, exact afterwards.
    typename std::conditional<LNE, LnConsumeLoads<TM, TN>, NoLn>::type lnq;
    float4 emit_u[TN], emit_c[TN];
    int lne_mode = 0;
    // the ordinary epilogue's operands (bias, gate, residual of this lane's outputs) likewise: one memory round trip less behind the
    // last MFMA (bf16 launches with 16-byte layouts; MTN_GEMM_EPI_PRE=0: load them in the epilogue as before)
    const bool vec_ok = ((N | P.ldc | (P.residual ? P.ldr : 0)) & 3) == 0;
    float4 pre_b[TN], pre_r[TM][TN];
    uint2 pre_g[TM][TN];
    bool pre_on = sizeof(T) == 2 && vec_ok && grp.epi_pre;
    if constexpr (LNE) {
        const int sl = lne_sl;
        if (sl != 0) {
            const LnEpiSlot& E = lne.s[sl - 1];
            lne_mode = E.mode;
            if (E.mode == MTN_LN_CONSUME) ln_consume_issue<TM, TN>(E, lnq, M, N, row0, row0 + wr * (BM / WR), col0 + wc * (BN / WC), l15, lg, tid, tid < 8 * BM);
            else {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = col0 + wc * (BN / WC) + j * 16 + lg * 4;
                    const int cc = col < N ? col : 0;
                    emit_u[j] = *(const float4*)(E.fold + cc);
                    emit_c[j] = *(const float4*)(E.fold + N + cc);
                }
            }
        }
    }
    if (lne_mode == MTN_LN_CONSUME) pre_on = false;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int colj = col0 + wc * (BN / WC) + j * 16 + lg * 4, col = colj + 3 < N ? colj : 0;
        pre_b[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pre_on && P.bias) pre_b[j] = *(const float4*)(P.bias + col);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rowi = row0 + wr * (BM / WR) + i * 16 + l15, row = rowi < M ? rowi : M - 1;
            pre_r[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
            pre_g[i][j] = make_uint2(0u, 0u);
            if (pre_on && P.residual) pre_r[i][j] = *(const float4*)(P.residual + (size_t)row * P.ldr + col);
            if constexpr (sizeof(T) == 2) {
                if (pre_on && P.gate) pre_g[i][j] = *(const uint2*)((const T*)P.gate + (size_t)row * P.ldc + col);
            }
        }
    }
    const DropState ds = drop_init(P.drop);        // scalar seed load + key hashing ride under the operand DMA
    for (int s = 0; s < nstages; ++s) {
        // stage s landed; the stages behind it may still fly (NBUF = 2: one, NBUF = 4: two): counted vmcnt
        const int ahead = nstages - 1 - s;
        static_assert(NDMA == 16 || NDMA == 12 || NDMA == 8 || NDMA == 6 || NDMA == 4 || NDMA == 2, "unexpected stage size");
        if (NBUF > 2 && ahead >= 2) {
            static_assert(NBUF == 2 || 2 * NDMA <= 16, "vmcnt immediates end at 63, keep the pipeline's count small");
            if constexpr (NDMA == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if constexpr (NDMA == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else if (ahead >= 1) {
            if constexpr (NDMA == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if constexpr (NDMA == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if constexpr (NDMA == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (NDMA == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if constexpr (NDMA == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (LNE) {                   // last stage: everything has landed, the epilogue's loads included — its row sums
                if (lne_mode == MTN_LN_CONSUME) ln_consume_publish<TM, TN>(lnq, tid, (float*)(smem + NBUF * STAGE_BYTES), tid < 8 * BM);     // ride on this barrier
            }
        }
        __builtin_amdgcn_s_barrier();
        if constexpr (NBUF > 2) {                  // every wave is done with stage s-1: its buffer takes stage s + NBUF - 1
            if (s + NBUF - 1 < nstages) issue_stage(s + NBUF - 1);
        }
        const unsigned char* sA = smem + (s & (NBUF - 1)) * STAGE_BYTES;
        const unsigned char* sB = sA + A_BYTES;
        const int kleft = K - s * BK;
        int ksteps = kleft >= BK ? DMA_ROWB / 64 : (kleft * (int)sizeof(T) + 63) / 64;   // 64-byte contraction steps holding data
        {
            // Every fragment read is INLINE ASM, for two reasons.  (1) The compiler puts s_waitcnt vmcnt(0) in front of any LDS read
            // it can see while LDS-DMA instructions are outstanding (it cannot tell the stage being read from the one being filled):
            // with plain loads here the refill of stage s+2 was drained before stage s+1 — landed long ago — could be touched, one
            // exposed DMA latency per stage beyond the second (round 3: found on the persistent K = 512 kernel, csrc/gemm_k512.hip).
            // (2) The transposing read of the BTR form has no builtin.  The compiler neither counts nor overlaps asm reads, so the
            // loop is software-pipelined by hand: the fragments of step ks+1 are in flight while the MFMAs of step ks issue
            // (ping-pong registers, one explicit lgkmcnt(0) per step).
            u32x4_t a0[TM], b0[TN], a1[TM], b1[TN];
            auto load = [&](int ks, u32x4_t* a, u32x4_t* b) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int ra = wr * (BM / WR) + i * 16 + l15;
                    const unsigned addr = (unsigned)(size_t)(sA + ra * DMA_ROWB + (((ks * 4 + lg) ^ (ra & 15)) << 4));
                    asm volatile("ds_read_b128 %0, %1" : "=v"(a[i]) : "v"(addr));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (BTR) {
                        const uint4 f = kn_frag<BN>(sB, wc * (BN / WC) + j * 16, ks, l15, lg);
                        b[j] = u32x4_t{f.x, f.y, f.z, f.w};
                    } else {
                        const int rb = wc * (BN / WC) + j * 16 + l15;
                        const unsigned addr = (unsigned)(size_t)(sB + rb * DMA_ROWB + (((ks * 4 + lg) ^ (rb & 15)) << 4));
                        asm volatile("ds_read_b128 %0, %1" : "=v"(b[j]) : "v"(addr));
                    }
                }
            };
            auto mfmas = [&](const u32x4_t* a, const u32x4_t* b) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) mma16<T>(acc[i][j], as_uint4(b[j]), as_uint4(a[i]));     // transposed accumulator: see epilogue
            };
            auto landed = [&](u32x4_t* a, u32x4_t* b) {                // (MTN_LANDED: common.h)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < TM; ++i) MTN_LANDED(a[i]);
#pragma unroll
                for (int j = 0; j < TN; ++j) MTN_LANDED(b[j]);
                __builtin_amdgcn_sched_barrier(0);
            };
            if (ksteps > 0) load(0, a0, b0);
            for (int ks = 0; ks < ksteps; ks += 2) {
                landed(a0, b0);
                if (ks + 1 < ksteps) load(ks + 1, a1, b1);
                __builtin_amdgcn_sched_barrier(0);
                mfmas(a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                if (ks + 1 < ksteps) {
                    landed(a1, b1);
                    if (ks + 2 < ksteps) load(ks + 2, a0, b0);
                    __builtin_amdgcn_sched_barrier(0);
                    mfmas(a1, b1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if constexpr (NBUF == 2) {
            if (s + 2 < nstages) {       // refill this buffer with stage s+2 once every wave is done reading it
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                issue_stage(s + 2);
            }
        }
    }

    // ---- epilogue.  The MFMAs were issued with the operands swapped (acc = W-tile x X-tile^T), so a lane holds, for ONE
    //      output row m = l15, FOUR CONSECUTIVE output columns n = 4*lg + r of each 16x16 tile: bias, residual, gate and
    //      both outputs move as 8/16-byte vectors (4x fewer memory instructions than the row-per-register layout).
    if constexpr (LNE) {
        static_assert(64 * NW >= 8 * BM, "the LayerNorm epilogue deals a tile's rows to groups of eight threads");
        const int sl = lne_sl;
        if (sl != 0) {
            const LnEpiSlot& E = lne.s[sl - 1];
            if (lne_mode == MTN_LN_CONSUME) {
                ln_consume_epilogue<TM, TN, BM>(E, lnq, acc, M, N, row0, row0 + wr * (BM / WR), col0 + wc * (BN / WC), l15, lg, tid, (float*)(smem + NBUF * STAGE_BYTES));
                return;
            }
            float p1[TM], p2[TM];
            // MTN_LN_EMIT: the ordinary epilogue, and the two dot products of the stored values on the way
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                p1[i] = p2[i] = 0.f;
                const int row = row0 + wr * (BM / WR) + i * 16 + l15;
                if (row >= M) continue;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = col0 + wc * (BN / WC) + j * 16 + lg * 4;
                    if (col >= N) continue;
                    float v[4], gt[4] = {0.f, 0.f, 0.f, 0.f};
                    epilogue4<T>(P, ds, vec_ok && col + 3 < N, row, col, N, acc[i][j], nullptr, nullptr, v, gt, pre_on && col + 3 < N, pre_b[j], pre_r[i][j], pre_g[i][j]);
                    const float4 u4 = emit_u[j], c4 = emit_c[j];
                    const float uu[4] = {u4.x, u4.y, u4.z, u4.w}, cc[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float vq = LP<T>::to_f32(LP<T>::from_f32(v[k]));     // the value the consumer GEMM will read
                        p1[i] += vq * uu[k];
                        p2[i] += vq * (gt[k] * E.gate_inv_scale - cc[k]);
                    }
                }
            }
            ln_emit_partials<TM, BM, WC>(E, p1, p2, M, N / 64, col0 / 64, row0, row0 + wr * (BM / WR), wc, l15, lg, tid, (float*)(smem + NBUF * STAGE_BYTES));
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = row0 + wr * (BM / WR) + i * 16 + l15;
        if (row >= M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = col0 + wc * (BN / WC) + j * 16 + lg * 4;
            if (col >= N) continue;
            epilogue4<T>(P, ds, vec_ok && col + 3 < N, row, col, N, acc[i][j], nullptr, nullptr, nullptr, nullptr, pre_on && col + 3 < N, pre_b[j], pre_r[i][j], pre_g[i][j]);
        }
    }
}

// ====================================================================================================================
// dW = dY^T X on LDS-DMA (bf16).  Both operands are contraction-major in memory ([rows of the batch][features]), which is
// exactly what LDS-DMA can copy (lane-linear 16-byte chunks, no transposition in flight): the [k][n] tiles land in LDS
// as they are and the MFMA fragments are read with the transposing LDS read ds_read_b64_tr_b16.  Semantics measured on
// gfx950 (tools/tr_probe.hip): inside each 16-lane group, result lane i element j = element (i%4) of the 8-byte chunk
// supplied by lane 4j + i/4.  So lane t of a group supplies &T[k0 + t/4][n0 + 4*(t%4)] and lane i receives T[k0..k0+3][n0+i]:
// a 4(k) x 16(n) block delivered column-per-lane; two reads (k0 and k0+4) make one 16x16x32 fragment with the standard
// slot order (lane group g covers k = 8g..8g+7).
// Stage = 128 contraction rows x (64+64) features = 32 KiB, two stages = 64 KiB -> two workgroups per CU.
// Bias gradients (row sums of dY^T) ride on one extra MFMA per fragment against an all-ones operand.
// ====================================================================================================================
static constexpr int TTD_KS = 128;                        // contraction rows per stage
static constexpr int TTD_TILE_BYTES = TTD_KS * 128;       // one operand tile per stage: 128 k x 64 features x 2 B
static constexpr int TTD_LDS = 4 * TTD_TILE_BYTES;        // 2 operands x 2 stages

__device__ __forceinline__ void ttd_issue_tile(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_tile, int ld_bytes, int R, int K,
                                               int col0, int k0, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int krow = (j * 4 + wave) * 8 + (lane >> 3);           // row of the tile written by this lane
        const int p = lane & 7;                                      // 16-byte slot inside the 128-byte row
        const int c = p ^ ((krow >> 1) & 7);                         // source chunk (XOR swizzle: the 4 rows of a transposing read hit different banks)
        const int gk = k0 + krow, gn = col0 + c * 8;
        unsigned voff = (gk < K && gn < R) ? (unsigned)gk * (unsigned)ld_bytes + (unsigned)gn * 2u : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(lds_tile + (j * 4 + wave) * 1024), 16, voff, 0, 0, 0);
    }
}

// fragment for feature tile n_off (multiple of 16) and contraction step ks (32 rows) of a [k][64 features] image
__device__ __forceinline__ uint4 ttd_frag(const unsigned char* img, int n_off, int ks, int l15, int lg) {
    uint4 f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int krow = ks * 32 + 8 * lg + 4 * r + (l15 >> 2);
        const int col = n_off + 4 * (l15 & 3);                       // element column
        const int slot = (col >> 3) ^ ((krow >> 1) & 7);             // swizzled 16-byte slot
        const unsigned addr = (unsigned)(size_t)(img + krow * 128 + slot * 16 + (col & 7) * 2);
        unsigned long long v;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
        if (r == 0) { f.x = (unsigned)v; f.y = (unsigned)(v >> 32); }
        else { f.z = (unsigned)v; f.w = (unsigned)(v >> 32); }
    }
    return f;
}

__global__ __launch_bounds__(256) void gemm_tt_dma_kernel(const GemmGroup grp, const AdamGroup adam) {
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    int g = 0;
    while (g + 1 < grp.count && (int)blockIdx.x >= grp.tile_start[g + 1]) ++g;
    const mtn_gemm_problem& P = grp.p[g];
    const int M = P.M, N = P.N, K = P.K;
    const int tiles_n = (N + 63) / 64;
    const int t = (int)blockIdx.x - grp.tile_start[g];
    const int tiles_m = (M + 64 - 1) / 64;
    int tm_, tn_;
    xcd_tile(grp, g, t, tiles_m * tiles_n, tiles_m, tiles_n, M >= N, tm_, tn_);
    const int row0 = tm_ * 64, col0 = tn_ * 64;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, lg = lane >> 4, l15 = lane & 15;
    const int lda_b = P.lda * 2, ldb_b = P.ldb * 2;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)P.A, 0, (K - 1) * lda_b + M * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)P.B, 0, (K - 1) * ldb_b + N * 2, 0x00020000);
    const bool do_rowsum = (P.rowsum_out != nullptr) && (col0 == 0);

    f32x4_t acc[2][2], rs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        rs[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    const uint4 ones = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);   // bf16 1.0 x 8

    const int nstages = (K + TTD_KS - 1) / TTD_KS;
    ttd_issue_tile(rA, smem, lda_b, M, K, row0, 0, wave, lane);
    ttd_issue_tile(rB, smem + TTD_TILE_BYTES, ldb_b, N, K, col0, 0, wave, lane);
    if (nstages > 1) {
        ttd_issue_tile(rA, smem + 2 * TTD_TILE_BYTES, lda_b, M, K, row0, TTD_KS, wave, lane);
        ttd_issue_tile(rB, smem + 3 * TTD_TILE_BYTES, ldb_b, N, K, col0, TTD_KS, wave, lane);
    }
    for (int s = 0; s < nstages; ++s) {
        if (s + 1 < nstages) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // 8 LDS-DMA instructions per wave per stage
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned char* sA = smem + (s & 1) * 2 * TTD_TILE_BYTES;
        const unsigned char* sB = sA + TTD_TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < TTD_KS / 32; ++ks) {
            uint4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = ttd_frag(sA, wr * 32 + i * 16, ks, l15, lg);
                b[i] = ttd_frag(sB, wc * 32 + i * 16, ks, l15, lg);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j) mma16<T>(acc[i][j], b[j], a[i]);     // transposed accumulator (vector epilogue)
                if (do_rowsum && wc == 0) mma16<T>(rs[i], ones, a[i]);
            }
        }
        if (s + 2 < nstages) {
            __builtin_amdgcn_s_barrier();
            unsigned char* dst = smem + (s & 1) * 2 * TTD_TILE_BYTES;
            ttd_issue_tile(rA, dst, lda_b, M, K, row0, (s + 2) * TTD_KS, wave, lane);
            ttd_issue_tile(rB, dst + TTD_TILE_BYTES, ldb_b, N, K, col0, (s + 2) * TTD_KS, wave, lane);
        }
    }
    if (do_rowsum && wc == 0 && lg == 0) {       // every accumulator row of rs[] holds the same sums; column l15 = output row
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = row0 + wr * 32 + i * 16 + l15;
            if (row < M) P.rowsum_out[row] = rs[i][0];
        }
    }
    const DropState ds = drop_init(P.drop);
    const bool vec_ok = ((N | P.ldc | (P.residual ? P.ldr : 0)) & 3) == 0;
    AdamCoef coef;
    if (adam.any) coef = adam_coef(adam);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = row0 + wr * 32 + i * 16 + l15;
        if (row >= M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = col0 + wc * 32 + j * 16 + lg * 4;
            if (col >= N) continue;
            epilogue4<T, true>(P, ds, vec_ok && col + 3 < N, row, col, N, acc[i][j], &adam.a[g], &coef);
        }
    }
}

// ====================================================================================================================
// The same contraction with 128 x 128 output tiles for the large parameter-gradient launches: every operand byte pulled
// into LDS feeds twice the outputs of the 64 x 64 kernel, which is what bounds these launches (L2 -> LDS fill).  Stage = 64
// contraction rows x (128 + 128) features = 32 KiB, two stages = 64 KiB -> two workgroups per CU; 4 waves x (4 x 4) MFMA
// tiles.  Rows of the [k][128 features] image are 256 B = one full sweep of the LDS banks, so the 16-byte slots are
// XOR-swizzled with s(k) = 2*(k & 3) + 8*((k >> 3) & 1): the 8 rows one transposing read touches per LDS cycle
// (k0..k0+3 of lane groups 0 and 1, 8 rows apart) then sit in 8 different 32-byte bank groups.
// ====================================================================================================================
static constexpr int TTB_KS = 64;                         // contraction rows per stage
static constexpr int TTB_TILE_BYTES = TTB_KS * 256;       // one operand tile per stage: 64 k x 128 features x 2 B
static constexpr int TTB_LDS = 4 * TTB_TILE_BYTES;        // 2 operands x 2 stages = 64 KiB

__device__ __forceinline__ int ttb_swz(int krow) { return ((krow & 3) << 1) | (((krow >> 3) & 1) << 3); }

__device__ __forceinline__ void ttb_issue_tile(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_tile, int ld_bytes, int R, int K,
                                               int col0, int k0, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {                                    // 16 instructions per tile, 4 per wave: 4 rows of 256 B each
        const int inst = j * 4 + wave;
        const int krow = inst * 4 + (lane >> 4);
        const int p = lane & 15;                                     // 16-byte slot inside the 256-byte row
        const int c = p ^ ttb_swz(krow);                             // source chunk that lands in slot p
        const int gk = k0 + krow, gn = col0 + c * 8;
        unsigned voff = (gk < K && gn < R) ? (unsigned)gk * (unsigned)ld_bytes + (unsigned)gn * 2u : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(lds_tile + inst * 1024), 16, voff, 0, 0, 0);
    }
}

// fragment for feature tile n_off (multiple of 16) and contraction step ks (32 rows) of a [k][128 features] image
__device__ __forceinline__ uint4 ttb_frag(const unsigned char* img, int n_off, int ks, int l15, int lg) {
    uint4 f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int krow = ks * 32 + 8 * lg + 4 * r + (l15 >> 2);
        const int col = n_off + 4 * (l15 & 3);
        const int slot = (col >> 3) ^ ttb_swz(krow);
        const unsigned addr = (unsigned)(size_t)(img + krow * 256 + slot * 16 + (col & 7) * 2);
        unsigned long long v;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
        if (r == 0) { f.x = (unsigned)v; f.y = (unsigned)(v >> 32); }
        else { f.z = (unsigned)v; f.w = (unsigned)(v >> 32); }
    }
    return f;
}

// Bias of the Linear whose weight gradient this problem is (table form, round 5): db = the row sums this tile produces anyway, so the
// bias's Adam update is applied right there (p, m, v, compute-dtype copy at the bias's flat offset) instead of by a later pass.
struct BiasAdam { float *p, *m, *v; bf16_t* lp; };
// One 128 x 128 tile of dW = dY^T X (+ optimiser epilogue): shared by the kernel-argument form and the table form below.
__device__ __forceinline__ void tt128_tile(const mtn_gemm_problem& P, const AdamSlot& S, const bool lds_epilogue, const AdamCoef& coef,
                                           const int row0, const int col0, unsigned char* smem, const BiasAdam bz = BiasAdam{nullptr, nullptr, nullptr, nullptr}) {
    typedef bf16_t T;
    const int tid = threadIdx.x;
    const int M = P.M, N = P.N, K = P.K;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, lg = lane >> 4, l15 = lane & 15;
    const int lda_b = P.lda * 2, ldb_b = P.ldb * 2;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)P.A, 0, (K - 1) * lda_b + M * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)P.B, 0, (K - 1) * ldb_b + N * 2, 0x00020000);
    const bool do_rowsum = (P.rowsum_out != nullptr) && (col0 == 0);

    f32x4_t acc[4][4], rs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        rs[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    const uint4 ones = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);

    const int nstages = (K + TTB_KS - 1) / TTB_KS;
    ttb_issue_tile(rA, smem, lda_b, M, K, row0, 0, wave, lane);
    ttb_issue_tile(rB, smem + TTB_TILE_BYTES, ldb_b, N, K, col0, 0, wave, lane);
    if (nstages > 1) {
        ttb_issue_tile(rA, smem + 2 * TTB_TILE_BYTES, lda_b, M, K, row0, TTB_KS, wave, lane);
        ttb_issue_tile(rB, smem + 3 * TTB_TILE_BYTES, ldb_b, N, K, col0, TTB_KS, wave, lane);
    }
    for (int s = 0; s < nstages; ++s) {
        if (s + 1 < nstages) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // 8 LDS-DMA instructions per wave per stage
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned char* sA = smem + (s & 1) * 2 * TTB_TILE_BYTES;
        const unsigned char* sB = sA + TTB_TILE_BYTES;
        // Both contraction steps of the stage are read up front (2 x 16 transposing reads; they are inline asm, the compiler does
        // not overlap them with anything): the MFMAs of step 0 issue once at most 15 reads are outstanding (LDS returns in
        // order: all 16 of step 0 have landed), those of step 1 hide under them.
        static_assert(TTB_KS / 32 == 2, "two contraction steps per stage");
        uint4 a[2][4], b[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[ks][i] = ttb_frag(sA, wr * 64 + i * 16, ks, l15, lg);
                b[ks][i] = ttb_frag(sB, wc * 64 + i * 16, ks, l15, lg);
            }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks == 0) asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) mma16<T>(acc[i][j], b[ks][j], a[ks][i]);     // transposed accumulator (vector epilogue)
                if (do_rowsum && wc == 0) mma16<T>(rs[i], ones, a[ks][i]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (s + 2 < nstages) {
            __builtin_amdgcn_s_barrier();
            unsigned char* dst = smem + (s & 1) * 2 * TTB_TILE_BYTES;
            ttb_issue_tile(rA, dst, lda_b, M, K, row0, (s + 2) * TTB_KS, wave, lane);
            ttb_issue_tile(rB, dst + TTB_TILE_BYTES, ldb_b, N, K, col0, (s + 2) * TTB_KS, wave, lane);
        }
    }
    if (do_rowsum && wc == 0 && lg == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = row0 + wr * 64 + i * 16 + l15;
            if (row < M) {
                P.rowsum_out[row] = rs[i][0];
                if (bz.p) {                          // same arithmetic as adam_chunks_kernel on the stored gradient: the same bits
                    float pv = bz.p[row], mv = bz.m[row], vv = bz.v[row];
                    adam_update(pv, mv, vv, rs[i][0], coef);
                    bz.p[row] = pv; bz.m[row] = mv; bz.v[row] = vv;
                    if (bz.lp) bz.lp[row] = f32_to_bf16(pv);
                }
            }
        }
    }
    const DropState ds = drop_init(P.drop);
    const bool vec_ok = ((N | P.ldc | (P.residual ? P.ldr : 0)) & 3) == 0;
    if (lds_epilogue && S.p) {
        // Optimiser epilogue through LDS (the operand stages are dead): the accumulators hold the gradient as 64-byte row
        // pieces per lane group, which would turn the seven parameter streams into partial-line traffic.  Half a tile at a
        // time (64 rows): gradient -> LDS [64][132] fp32; every thread then owns whole float4s of full 512-byte rows of
        // p / m / v (coalesced loads and stores; plain ones measured 0.3 % of a step faster than non-temporal) and of the compute-dtype copy, and parks the new weights as
        // bf16 in a second LDS image [128 cols][72] from which the TRANSPOSED copy leaves as 128-byte column runs.
        float* sg = (float*)smem;                                  // 64 x 132 floats = 33 792 B
        bf16_t* st = (bf16_t*)(smem + 64 * 132 * 4);               // 128 x 72 bf16   = 18 432 B
        typedef __attribute__((ext_vector_type(4))) float nt4;
        const int c4 = (tid & 31) * 4, rb = tid >> 5;              // this thread: columns c4..c4+3 of rows rb*8 .. rb*8+7 of a half
        const bool col_ok = col0 + c4 < N;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // all 24 16-byte loads of this thread's 8 rows are in flight before anything waits on them — and before the
            // gradient goes through LDS (a load of the next row cannot be hoisted above this row's stores by the compiler: p, m,
            // v may alias for all it knows; one HBM latency per row made the epilogue latency-bound, 50 us per tile)
            nt4 p_[8], m_[8], v_[8];
            bool ok[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int row = row0 + h * 64 + rb * 8 + q;
                ok[q] = col_ok && row < M;
                if (ok[q]) {
                    const size_t o = (size_t)row * P.ldc + col0 + c4;
                    p_[q] = *(const nt4*)(S.p + o);
                    m_[q] = *(const nt4*)(S.m + o);
                    v_[q] = *(const nt4*)(S.v + o);
                }
            }
            __syncthreads();
            if (wr == h) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        *(f32x4_t*)(sg + (i * 16 + l15) * 132 + wc * 64 + j * 16 + lg * 4) = acc[i][j];
            }
            __syncthreads();
            uint32_t pk[4][4];                                     // the new weights as bf16: [column][row pair]
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = rb * 8 + q;
                const int row = row0 + h * 64 + r;
                float pv[4] = {0.f, 0.f, 0.f, 0.f};
                if (ok[q]) {
                    const size_t o = (size_t)row * P.ldc + col0 + c4;
                    const f32x4_t gq = *(const f32x4_t*)(sg + r * 132 + c4);
                    float mv[4], vv[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) { pv[k] = p_[q][k]; mv[k] = m_[q][k]; vv[k] = v_[q][k]; adam_update(pv[k], mv[k], vv[k], gq[k], coef); }
                    *(nt4*)(S.p + o) = nt4{pv[0], pv[1], pv[2], pv[3]};
                    *(nt4*)(S.m + o) = nt4{mv[0], mv[1], mv[2], mv[3]};
                    *(nt4*)(S.v + o) = nt4{vv[0], vv[1], vv[2], vv[3]};
                    if (S.write_grad) *(float4*)(P.out_f32 + o) = make_float4(gq[0], gq[1], gq[2], gq[3]);
                    if (S.lp)
                        *(uint2*)((T*)S.lp + o) = make_uint2((uint32_t)f32_to_bf16(pv[0]) | ((uint32_t)f32_to_bf16(pv[1]) << 16),
                                                             (uint32_t)f32_to_bf16(pv[2]) | ((uint32_t)f32_to_bf16(pv[3]) << 16));
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t bits = (uint32_t)f32_to_bf16(pv[k]);
                    if (q & 1) pk[k][q >> 1] |= bits << 16; else pk[k][q >> 1] = bits;
                }
            }
            if (S.lpT) {                                            // (tile-uniform: only W_o keeps a transposed copy)
#pragma unroll
                for (int k = 0; k < 4; ++k) *(uint4*)(st + (c4 + k) * 72 + rb * 8) = make_uint4(pk[k][0], pk[k][1], pk[k][2], pk[k][3]);
                __syncthreads();
                const int r8 = (tid & 7) * 8;                       // 8 consecutive rows = 16 bytes of one column of the tile
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const int c = (tid >> 3) + ps * 32;
                    const int row = row0 + h * 64 + r8;
                    if (col0 + c < N && row < M)                    // M % 8 == 0: the 8 rows are in or out together
                        *(uint4*)((T*)S.lpT + (size_t)(col0 + c) * S.ldT + row) = *(const uint4*)(st + c * 72 + r8);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = row0 + wr * 64 + i * 16 + l15;
        if (row >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = col0 + wc * 64 + j * 16 + lg * 4;
            if (col >= N) continue;
            epilogue4<T, true>(P, ds, vec_ok && col + 3 < N, row, col, N, acc[i][j], &S, &coef);
        }
    }
}


__global__ __launch_bounds__(256, 2) void gemm_tt_dma128_kernel(const GemmGroup grp, const AdamGroup adam) {
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    int g = 0;
    while (g + 1 < grp.count && (int)blockIdx.x >= grp.tile_start[g + 1]) ++g;
    const mtn_gemm_problem& P = grp.p[g];
    const int M = P.M, N = P.N, K = P.K;
    const int tiles_n = (N + 127) / 128;
    const int t = (int)blockIdx.x - grp.tile_start[g];
    const int tiles_m = (M + 128 - 1) / 128;
    int tm_, tn_;
    xcd_tile(grp, g, t, tiles_m * tiles_n, tiles_m, tiles_n, M >= N, tm_, tn_);
    const int row0 = tm_ * 128, col0 = tn_ * 128;
    AdamCoef coef;
    if (adam.any) coef = adam_coef(adam);
    tt128_tile(P, adam.a[g], adam.lds_epilogue != 0, coef, row0, col0, smem);
}

// --------------------------------------------------------------------------------------------------------------------
// Table form: ALL parameter-gradient problems of a backward pass (~200 at cfg2) in ONE launch.  The problem list does not
// fit a kernel argument, so it lives in device memory (mtn_gemm_tt_table stages it through pinned memory): a header, the
// problems, and a tile map (grid index -> problem, tile) in which consecutive problems have very different contraction
// lengths.  Why: with the optimiser epilogue a tile is a compute phase (LDS-DMA pulls + MFMA) followed by a streaming phase
// (26 B/param of HBM traffic).  Launch by launch, all tiles are in the same phase at the same time and the two costs add;
// in one long launch the resident workgroups drift apart and one tile's streaming overlaps its neighbours' contractions.
// --------------------------------------------------------------------------------------------------------------------
struct TTProblem {
    const void* A; const void* B;
    float* out; float* rowsum;
    float *p, *m, *v; void* lp; void* lpT;
    int lda, ldb, M, N, K, ldc, ldT, write_grad;
    int tiles_m, tiles_n, first_tile, pad_;
    long bias_off;                             // >= 0: flat offset of the bias whose gradient `rowsum` is -> its Adam update rides on the tile (round 5)
};
// Front tiles (round 5, mtn_tt_aux): the optimiser for what no dW epilogue covers — the LayerNorm gains / biases (their gradient = the
// sum of the partial rows the LayerNorm-backward kernels left: the finalize pass, same summation order, + Adam) and chunks of the flat
// buffers whose gradient is complete before the launch (embedding tables).  They sit at the FRONT of the grid: the first resident
// dW tiles are all in their contraction phase then and leave HBM idle.  Replaces the ln_bwd_finalize and adam_chunks launches.
struct TTLnUnit { const float* partial; int nparts, d, col0, pad_; long a_off, b_off; };      // 256 columns of one LayerNorm's [da2 | db2]
struct TTHeader {
    const float* state; const float* grad_scale;
    float beta1, beta2, eps;
    int any, lds_epilogue, n_problems, n_tiles, plain_tile_order;
    long problems_off, tilemap_off;            // byte offsets from the header
    int n_front, n_ln_units, n_chunks, pad_;
    long ln_off, chunk_off_off, chunk_len_off; // byte offsets from the header: TTLnUnit[], long[], int[]
    float *fp, *fg, *fm, *fv; bf16_t* flp;     // the flat parameter / gradient / moment buffers and the compute-dtype copy
};
__device__ __forceinline__ void tt_front_tile(const TTHeader* __restrict__ hdr, const int idx) {
    const unsigned char* base = (const unsigned char*)hdr;
    const AdamCoef coef = adam_coef(hdr->state, hdr->grad_scale, hdr->beta1, hdr->beta2, hdr->eps);
    if (idx < hdr->n_ln_units) {
        // ln_bwd_finalize_kernel's arithmetic for one column (16 interleaved groups of partial rows, four chains each, the same
        // combination order), then Adam on that one LayerNorm gain / bias
        const TTLnUnit U = ((const TTLnUnit*)(base + hdr->ln_off))[idx];
        const int c = U.col0 + (int)threadIdx.x;
        if (c >= 2 * U.d) return;
        const float* p = U.partial + c;
        const size_t ld = (size_t)2 * U.d;
        float red[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            int i = q;
            for (; i + 112 < U.nparts; i += 128) {
                const float a0 = p[(size_t)(i + 0) * ld], a1 = p[(size_t)(i + 16) * ld], a2 = p[(size_t)(i + 32) * ld], a3 = p[(size_t)(i + 48) * ld];
                const float a4 = p[(size_t)(i + 64) * ld], a5 = p[(size_t)(i + 80) * ld], a6 = p[(size_t)(i + 96) * ld], a7 = p[(size_t)(i + 112) * ld];
                s0 += a0; s1 += a1; s2 += a2; s3 += a3; s0 += a4; s1 += a5; s2 += a6; s3 += a7;
            }
            for (; i + 48 < U.nparts; i += 64) {
                s0 += p[(size_t)(i + 0) * ld]; s1 += p[(size_t)(i + 16) * ld];
                s2 += p[(size_t)(i + 32) * ld]; s3 += p[(size_t)(i + 48) * ld];
            }
            for (; i < U.nparts; i += 16) s0 += p[(size_t)i * ld];
            red[q] = (s0 + s1) + (s2 + s3);
        }
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k += 4) t += (red[k] + red[k + 1]) + (red[k + 2] + red[k + 3]);
        const long o = c < U.d ? U.a_off + c : U.b_off + (c - U.d);
        hdr->fg[o] = t;
        float pv = hdr->fp[o], mv = hdr->fm[o], vv = hdr->fv[o];
        adam_update(pv, mv, vv, t, coef);
        hdr->fp[o] = pv; hdr->fm[o] = mv; hdr->fv[o] = vv;
        if (hdr->flp) hdr->flp[o] = f32_to_bf16(pv);
        return;
    }
    const int ci = idx - hdr->n_ln_units;
    if (ci >= hdr->n_chunks) return;                               // (the front is padded to a multiple of 8 tiles: the XCD map of the dW tiles)
    // adam_chunks_kernel's body: a whole chunk (<= 4096 elements) in flight before the first update
    const long cb = ((const long*)(base + hdr->chunk_off_off))[ci];
    const int n4 = ((const int*)(base + hdr->chunk_len_off))[ci] >> 2;
    float* __restrict__ p = hdr->fp; const float* __restrict__ g = hdr->fg; float* __restrict__ m = hdr->fm; float* __restrict__ v = hdr->fv;
    float4 pv[4], gv[4], mv[4], vv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int q = threadIdx.x + 256 * u;
        const long i = cb + (long)(q < n4 ? q : 0) * 4;
        pv[u] = *(const float4*)(p + i); gv[u] = *(const float4*)(g + i); mv[u] = *(const float4*)(m + i); vv[u] = *(const float4*)(v + i);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int q = threadIdx.x + 256 * u;
        if (q >= n4) continue;
        const long i = cb + (long)q * 4;
        float* pp = &pv[u].x; float* gp = &gv[u].x; float* mp = &mv[u].x; float* vp = &vv[u].x;
#pragma unroll
        for (int k = 0; k < 4; ++k) adam_update(pp[k], mp[k], vp[k], gp[k], coef);
        *(float4*)(p + i) = pv[u]; *(float4*)(m + i) = mv[u]; *(float4*)(v + i) = vv[u];
        if (hdr->flp) {
            uint2 w;
            w.x = (uint32_t)f32_to_bf16(pv[u].x) | ((uint32_t)f32_to_bf16(pv[u].y) << 16);
            w.y = (uint32_t)f32_to_bf16(pv[u].z) | ((uint32_t)f32_to_bf16(pv[u].w) << 16);
            *(uint2*)(hdr->flp + i) = w;
        }
    }
}
__global__ __launch_bounds__(256, 2) void gemm_tt_dma128_table_kernel(const TTHeader* __restrict__ hdr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned char* base = (const unsigned char*)hdr;
    const uint32_t* tilemap = (const uint32_t*)(base + hdr->tilemap_off);
    const uint32_t e = tilemap[blockIdx.x];                       // problem << 12 | tile inside the problem, or 0x80000000 | optimiser unit
    if (e & 0x80000000u) { tt_front_tile(hdr, (int)(e & 0x7fffffffu)); return; }
    const int g = (int)(e >> 12), t = (int)(e & 0xfff);
    const TTProblem& Q = ((const TTProblem*)(base + hdr->problems_off))[g];
    mtn_gemm_problem P;
    P.A = Q.A; P.B = Q.B; P.lda = Q.lda; P.ldb = Q.ldb; P.M = Q.M; P.N = Q.N; P.K = Q.K; P.a_trans = 1; P.b_trans = 1;
    P.bias = nullptr; P.relu = 0; P.drop.p = 0.f; P.drop.salt = 0; P.drop.seed = nullptr; P.gate = nullptr; P.gate_scale = 1.f;
    P.residual = nullptr; P.ldr = 0; P.out_f32 = Q.out; P.out_lp = nullptr; P.ldc = Q.ldc; P.lp_drop_after_residual = 0; P.rowsum_out = Q.rowsum; P.adam = nullptr;
    AdamSlot S;
    S.p = Q.p; S.m = Q.m; S.v = Q.v; S.lp = Q.lp; S.lpT = Q.lpT; S.ldT = Q.ldT; S.write_grad = Q.write_grad;
    // the problem's tiles are consecutive grid indices starting at first_tile: same XCD-aware band mapping as xcd_tile()
    const int T = Q.tiles_m * Q.tiles_n;
    int idx = t;
    if (!hdr->plain_tile_order && T >= 16) {
        const int c = t & 7, r = t >> 3;                  // class (same XCD) in order of first appearance, rank inside it
        const int per = T >> 3, rem = T & 7;
        idx = c * per + (c < rem ? c : rem) + r;
    }
    int tm_, tn_;
    if (Q.M >= Q.N) { tm_ = idx / Q.tiles_n; tn_ = idx - tm_ * Q.tiles_n; }
    else { tn_ = idx / Q.tiles_m; tm_ = idx - tn_ * Q.tiles_m; }
    AdamCoef coef;
    if (hdr->any) coef = adam_coef(hdr->state, hdr->grad_scale, hdr->beta1, hdr->beta2, hdr->eps);
    BiasAdam bz{nullptr, nullptr, nullptr, nullptr};
    if (Q.bias_off >= 0) { bz.p = hdr->fp + Q.bias_off; bz.m = hdr->fm + Q.bias_off; bz.v = hdr->fv + Q.bias_off; bz.lp = hdr->flp ? hdr->flp + Q.bias_off : nullptr; }
    tt128_tile(P, S, hdr->lds_epilogue != 0, coef, tm_ * 128, tn_ * 128, smem, bz);
}

// ====================================================================================================================
// Row-major x row-major (both operands contraction-contiguous) with 128 x 128 output tiles, for launches with enough tiles
// to fill the chip (memory K/V projections of all layers, the loss head, the memory-gradient groups): half the operand bytes
// per output of the 64 x 64 kernels.  Stage = (128 + 128) rows x 64 contraction elements (128 B per row) = 32 KiB, two stages,
// two workgroups per CU, 4 waves x (4 x 4) MFMA tiles.  128-byte rows: two rows per sweep of the LDS banks, 16-byte slots
// XOR-swizzled with (row & 7) -> the 16 lanes one ds_read_b128 cycle serves hit 16 different slots.
// ====================================================================================================================
static constexpr int NTB_BK = 64;                          // contraction elements per stage (bf16)
static constexpr int NTB_TILE_BYTES = 128 * 128;           // one operand tile per stage: 128 rows x 128 B
static constexpr int NTB_LDS = 4 * NTB_TILE_BYTES;         // 64 KiB

__device__ __forceinline__ void ntb_issue_tile(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_tile, int ld_bytes, int R, int K,
                                               int row0, int k0, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {                                    // 16 instructions per tile, 4 per wave: 8 rows of 128 B each
        const int inst = j * 4 + wave;
        const int row = inst * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (row & 7);                        // source chunk that lands in slot (lane & 7)
        const int gk = k0 + c * 8;
        int grow = row0 + row;
        grow = grow < R ? grow : R - 1;                              // rows past the end only feed outputs that are never stored
        unsigned voff = (gk < K) ? (unsigned)grow * (unsigned)ld_bytes + (unsigned)gk * 2u : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(lds_tile + inst * 1024), 16, voff, 0, 0, 0);
    }
}

__global__ __launch_bounds__(256, 2) void gemm_dma128_kernel(const GemmGroup grp) {
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    int g = 0;
    while (g + 1 < grp.count && (int)blockIdx.x >= grp.tile_start[g + 1]) ++g;
    const mtn_gemm_problem& P = grp.p[g];
    const int M = P.M, N = P.N, K = P.K;
    const int tiles_n = (N + 127) / 128;
    const int t = (int)blockIdx.x - grp.tile_start[g];
    const int tiles_m = (M + 128 - 1) / 128;
    int tm_, tn_;
    xcd_tile(grp, g, t, tiles_m * tiles_n, tiles_m, tiles_n, M >= N, tm_, tn_);
    const int row0 = tm_ * 128, col0 = tn_ * 128;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, lg = lane >> 4, l15 = lane & 15;
    const int lda_b = P.lda * 2, ldb_b = P.ldb * 2;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)P.A, 0, (M - 1) * lda_b + K * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)P.B, 0, (N - 1) * ldb_b + K * 2, 0x00020000);

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nstages = (K + NTB_BK - 1) / NTB_BK;
    ntb_issue_tile(rA, smem, lda_b, M, K, row0, 0, wave, lane);
    ntb_issue_tile(rB, smem + NTB_TILE_BYTES, ldb_b, N, K, col0, 0, wave, lane);
    if (nstages > 1) {
        ntb_issue_tile(rA, smem + 2 * NTB_TILE_BYTES, lda_b, M, K, row0, NTB_BK, wave, lane);
        ntb_issue_tile(rB, smem + 3 * NTB_TILE_BYTES, ldb_b, N, K, col0, NTB_BK, wave, lane);
    }
    const DropState ds = drop_init(P.drop);
    for (int s = 0; s < nstages; ++s) {
        if (s + 1 < nstages) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // 8 LDS-DMA instructions per wave per stage
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned char* sA = smem + (s & 1) * 2 * NTB_TILE_BYTES;
        const unsigned char* sB = sA + NTB_TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {                               // 32 contraction elements = one 16-byte chunk per lane group
            uint4 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ra = wr * 64 + i * 16 + l15, rb = wc * 64 + i * 16 + l15;
                a[i] = *(const uint4*)(sA + ra * 128 + (((ks * 4 + lg) ^ (ra & 7)) << 4));
                b[i] = *(const uint4*)(sB + rb * 128 + (((ks * 4 + lg) ^ (rb & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mma16<T>(acc[i][j], b[j], a[i]);     // transposed accumulator (vector epilogue)
        }
        if (s + 2 < nstages) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            unsigned char* dst = smem + (s & 1) * 2 * NTB_TILE_BYTES;
            ntb_issue_tile(rA, dst, lda_b, M, K, row0, (s + 2) * NTB_BK, wave, lane);
            ntb_issue_tile(rB, dst + NTB_TILE_BYTES, ldb_b, N, K, col0, (s + 2) * NTB_BK, wave, lane);
        }
    }
    const bool vec_ok = ((N | P.ldc | (P.residual ? P.ldr : 0)) & 3) == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = row0 + wr * 64 + i * 16 + l15;
        if (row >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = col0 + wc * 64 + j * 16 + lg * 4;
            if (col >= N) continue;
            epilogue4<T>(P, ds, vec_ok && col + 3 < N, row, col, N, acc[i][j]);
        }
    }
}

// ---------------------------------------------------------------- launch census (measurement support, bench.py)
// While recording, every mtn_gemm call keeps a copy of its problem list; mtn_census_replay re-issues a recorded launch so
// ====================================================================================================================
// 128 x 128 output tiles, 512 threads, FOUR stages of 64 contraction elements in LDS (round 3).  For launches of one round of big
// tiles with a long contraction — the memory-gradient GEMM dmem = dkv Wkv ([8 064 x 512 x 1 024]: 252 tiles) was 1 008 tiles of
// 64 x 64 on half stages, 31 us: every CU pulled 1 MiB where 512 KiB do.  A row-major [128 rows][64 k]; B either row-major
// [128 n][64 k] or (BTR) as the weight lies, [64 k][128 n] (256-byte rows, the table kernel's loader and transposing fragment
// reads).  8 waves = 2 row halves x 4 column quarters, wave tile 64 x 32 (4 x 2 MFMA tiles).  Stage s+3 is issued while stage s
// is computed: three stages (96 KiB) in flight per workgroup, counted vmcnt, ONE barrier per stage; every fragment read is
// inline asm (see gemm_dma_kernel: a visible LDS read would drain the stages in flight).
// ====================================================================================================================
static constexpr int G8_BK = 64;
static constexpr int G8_OP_BYTES = 128 * 128;              // one operand's stage: 128 rows x 128 B, or 64 k-rows x 256 B
static constexpr int G8_STAGE = 2 * G8_OP_BYTES;           // 32 KiB
static constexpr int G8_NST = 4;
static constexpr int G8_LDS = G8_NST * G8_STAGE;           // 128 KiB

// NW = 16 waves (1 024 threads, wave tile 32 x 32) pull the stages faster than 8 (a CU's fill rate grows with its resident waves).
template <bool BTR, int NW = 16, bool LNE = false>
__global__ __launch_bounds__(64 * NW) void gemm_dma128x_kernel(const GemmGroup grp, const typename LnArg<LNE>::type lne) {
    constexpr int WR = NW / 4;                                     // waves down the rows (2 or 4); four across the columns
    constexpr int TM = 128 / (16 * WR);                            // MFMA tiles per wave down the rows (4 or 2)
    constexpr int IPW = 16 / NW;                                   // LDS-DMA instructions per wave per operand per stage (2 or 1)
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    int g = 0;
    while (g + 1 < grp.count && (int)blockIdx.x >= grp.tile_start[g + 1]) ++g;
    const mtn_gemm_problem& P = grp.p[g];
    const int M = P.M, N = P.N, K = P.K;
    const int tiles_n = (N + 127) / 128, tiles_m = (M + 127) / 128;
    const int t = (int)blockIdx.x - grp.tile_start[g];
    int tm_, tn_;
    xcd_tile(grp, g, t, tiles_m * tiles_n, tiles_m, tiles_n, M >= N, tm_, tn_);
    const int row0 = tm_ * 128, col0 = tn_ * 128;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3, lg = lane >> 4, l15 = lane & 15;
    const int lda_b = P.lda * 2, ldb_b = P.ldb * 2;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)P.A, 0, (M - 1) * lda_b + K * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = BTR ? __builtin_amdgcn_make_buffer_rsrc((void*)P.B, 0, (K - 1) * ldb_b + N * 2, 0x00020000)
                                          : __builtin_amdgcn_make_buffer_rsrc((void*)P.B, 0, (N - 1) * ldb_b + K * 2, 0x00020000);
    // 2 * IPW LDS-DMA instructions per wave per stage, always: IPW for A (8 rows of 128 B each), IPW for B
    auto issue_rows = [&](__amdgpu_buffer_rsrc_t rs, unsigned char* dst, int ld_bytes, int R, int r0, int k0) {
#pragma unroll
        for (int j = 0; j < IPW; ++j) {
            const int inst = j * NW + wave;
            const int row = inst * 8 + (lane >> 3);
            const int c = (lane & 7) ^ (row & 7);                    // source chunk that lands in slot (lane & 7)
            const int gk = k0 + c * 8;
            int grow = r0 + row;
            grow = grow < R ? grow : R - 1;                          // rows past the end only feed outputs that are never stored
            unsigned voff = (gk < K) ? (unsigned)grow * (unsigned)ld_bytes + (unsigned)gk * 2u : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(dst + inst * 1024), 16, voff, 0, 0, 0);
        }
    };
    auto issue_kn = [&](unsigned char* dst, int k0) {               // B as it lies: [64 k][128 n], 4 k-rows of 256 B per instruction
#pragma unroll
        for (int j = 0; j < IPW; ++j) {
            const int inst = j * NW + wave;
            const int krow = inst * 4 + (lane >> 4);
            const int c = (lane & 15) ^ ttb_swz(krow);
            const int gk = k0 + krow, gn = col0 + c * 8;
            unsigned voff = (gk < K && gn < N) ? (unsigned)gk * (unsigned)ldb_b + (unsigned)gn * 2u : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void_t*)(dst + inst * 1024), 16, voff, 0, 0, 0);
        }
    };
    auto issue = [&](int st) {
        unsigned char* dst = smem + (st & (G8_NST - 1)) * G8_STAGE;
        issue_rows(rA, dst, lda_b, M, row0, st * G8_BK);
        if constexpr (BTR) issue_kn(dst + G8_OP_BYTES, st * G8_BK);
        else issue_rows(rB, dst + G8_OP_BYTES, ldb_b, N, col0, st * G8_BK);
    };

    f32x4_t acc[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nstages = (K + G8_BK - 1) / G8_BK;
    for (int st = 0; st < 3 && st < nstages; ++st) issue(st);
    // LayerNorm-backward epilogue of the problems that carry one (MTN_LN_CONSUME; see gemm_dma_kernel): loads behind the first stages
    typename std::conditional<LNE, LnConsumeLoads<TM, 2>, NoLn>::type lnq;
    int lne_slot = 0;
    if constexpr (LNE) {
        static_assert(64 * NW == 8 * 128, "the LayerNorm epilogue deals a tile's rows to groups of eight threads");
        lne_slot = lne.slot_of[g];
        if (lne_slot != 0) ln_consume_issue<TM, 2>(lne.s[lne_slot - 1], lnq, M, N, row0, row0 + wr * (16 * TM), col0 + wc * 32, l15, lg, tid);
    }
    const DropState ds = drop_init(P.drop);
    for (int s = 0; s < nstages; ++s) {
        const int ahead = nstages - 1 - s;                           // stages issued behind this one: min(ahead, 2) may still fly
        if (ahead >= 2) { if constexpr (IPW == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else if (ahead == 1) { if constexpr (IPW == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
        else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (LNE) { if (lne_slot != 0) ln_consume_publish<TM, 2>(lnq, tid, (float*)(smem + G8_LDS)); }       // (see gemm_dma_kernel)
        }
        __builtin_amdgcn_s_barrier();                                // stage s is in LDS for every wave; every wave is done with stage s-1
        if (s + 3 < nstages) issue(s + 3);                           // ... whose buffer takes stage s+3
        const unsigned char* sA = smem + (s & (G8_NST - 1)) * G8_STAGE;
        const unsigned char* sB = sA + G8_OP_BYTES;
        u32x4_t a0[TM], b0[2], a1[TM], b1[2];
        auto load = [&](int ks, u32x4_t* a, u32x4_t* b) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int ra = wr * (16 * TM) + i * 16 + l15;
                const unsigned addr = (unsigned)(size_t)(sA + ra * 128 + (((ks * 4 + lg) ^ (ra & 7)) << 4));
                asm volatile("ds_read_b128 %0, %1" : "=v"(a[i]) : "v"(addr));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if constexpr (BTR) {
                    const uint4 f = ttb_frag(sB, wc * 32 + j * 16, ks, l15, lg);
                    b[j] = u32x4_t{f.x, f.y, f.z, f.w};
                } else {
                    const int rb = wc * 32 + j * 16 + l15;
                    const unsigned addr = (unsigned)(size_t)(sB + rb * 128 + (((ks * 4 + lg) ^ (rb & 7)) << 4));
                    asm volatile("ds_read_b128 %0, %1" : "=v"(b[j]) : "v"(addr));
                }
            }
        };
        auto landed = [&](u32x4_t* a, u32x4_t* b) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < TM; ++i) MTN_LANDED(a[i]);
#pragma unroll
            for (int j = 0; j < 2; ++j) MTN_LANDED(b[j]);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto mfmas = [&](const u32x4_t* a, const u32x4_t* b) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma16<T>(acc[i][j], as_uint4(b[j]), as_uint4(a[i]));     // transposed accumulator (vector epilogue)
        };
        load(0, a0, b0);
        landed(a0, b0);
        load(1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        landed(a1, b1);
        mfmas(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
    }
    const bool vec_ok = ((N | P.ldc | (P.residual ? P.ldr : 0)) & 3) == 0;
    if constexpr (LNE) {
        if (lne_slot != 0) {
            ln_consume_epilogue<TM, 2, 128>(lne.s[lne_slot - 1], lnq, acc, M, N, row0, row0 + wr * (16 * TM), col0 + wc * 32, l15, lg, tid, (float*)(smem + G8_LDS));
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = row0 + wr * (16 * TM) + i * 16 + l15;
        if (row >= M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = col0 + wc * 32 + j * 16 + lg * 4;
            if (col >= N) continue;
            epilogue4<T>(P, ds, vec_ok && col + 3 < N, row, col, N, acc[i][j]);
        }
    }
}


// that the caller can time each of the step's GEMM launches with HIP events on the launch stream.
#include <algorithm>
#include <vector>
struct CensusEntry { int dtype, count, variant, tiles; mtn_gemm_problem p[MTN_GEMM_MAX_GROUP]; mtn_adam_fuse adam[MTN_GEMM_MAX_GROUP]; mtn_ln_epilogue ln[MTN_GEMM_MAX_GROUP]; int table; };
struct CensusTable {                                                                            // a table-form launch (any number of problems)
    std::vector<mtn_gemm_problem> p; std::vector<mtn_adam_fuse> adam;
    bool has_aux; mtn_tt_aux aux; std::vector<mtn_tt_ln_unit> ln; std::vector<long> chunk_off; std::vector<int> chunk_len;    // its front tiles (mtn_tt_aux)
};
static std::vector<CensusEntry> g_census;
static std::vector<CensusTable> g_census_tables;
static bool g_census_on = false;
static int g_variant = 0, g_variant_tiles = 0;     // set by launch_gemm: which kernel the dispatch picked
enum { V_REG_NN = 0, V_REG_NT, V_REG_TN, V_REG_TT, V_DMA64, V_DMA3264, V_DMA32, V_TT_DMA, V_TT128, V_TT_DMA128, V_DMA128, V_DMA64H, V_DMA32H, V_TT_TABLE, V_K512, V_DMA128X, V_COUNT };
static const char* const g_variant_name[V_COUNT] = {
    "gemm_kernel<N,N> 64x64 reg-staged", "gemm_kernel<N,T>", "gemm_kernel<T,N>", "gemm_kernel<T,T> 64x64 reg-staged",
    "gemm_dma_kernel<64,64>", "gemm_dma_kernel<32,64>", "gemm_dma_kernel<32,32>", "gemm_tt_dma_kernel", "(gemm_tt128_kernel: removed in round 5)",
    "gemm_tt_dma128_kernel", "gemm_dma128_kernel", "gemm_dma_kernel<64,64> half stages", "gemm_dma_kernel<32,32> half stages", "gemm_tt_dma128_table_kernel", "gemm_k512_kernel", "gemm_dma128x_kernel (128x128, four stages)"};

template <typename T, int BM, int BN, int ROWB, bool BTR = false, int NBUF = 2, int NW = 4, bool LNE = false>
static int launch_dma_impl(const GemmGroup& grp, const typename LnArg<LNE>::type& lne, int tiles, hipStream_t s) {
    g_variant = (BM == 64) ? (ROWB == 512 ? V_DMA64 : V_DMA64H) : (BN == 64 ? V_DMA3264 : (ROWB == 512 ? V_DMA32 : V_DMA32H));
    g_variant_tiles = tiles;
    constexpr int LDS = NBUF * (BM + BN) * ROWB + (LNE ? LNE_LDS_EXTRA : 0);       // (+ the LayerNorm epilogue's row-sum scratch)
    static bool attr_set = false;          // > 64 KiB of dynamic LDS needs the opt-in once per kernel
    if (!attr_set && LDS > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_dma_kernel<T, BM, BN, ROWB, BTR, NBUF, NW, LNE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) { mtn_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return MTN_ERR_LAUNCH; }
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_dma_kernel<T, BM, BN, ROWB, BTR, NBUF, NW, LNE>), dim3(tiles), dim3(64 * NW), LDS, s, grp, lne);
    MTN_CHECK_LAUNCH();
    return MTN_OK;
}
// `lne` (launches with a LayerNorm epilogue, mtn_ln_epilogue): only the tile shapes the step's launches take are instantiated with it —
// bf16, a square tile whose rows are dealt to groups of eight threads (64 x 64 on eight waves, 32 x 32 on four), either B layout
template <typename T, int BM, int BN, int ROWB, bool BTR = false, int NBUF = 2, int NW = 4>
static int launch_dma(const GemmGroup& grp, int tiles, hipStream_t s, const LnEpiGroup* lne = nullptr) {
    if (lne) {
        if constexpr (sizeof(T) == 2 && NBUF == 2 && (64 * NW == 8 * BM || (NW == 16 && BM == 64)) && BM == BN)
            return launch_dma_impl<T, BM, BN, ROWB, BTR, NBUF, NW, true>(grp, *lne, tiles, s);
        mtn_set_error("mtn_gemm: no LayerNorm-epilogue form of this kernel (tile %d x %d, stage %d B, %d waves)", BM, BN, ROWB, NW);
        return MTN_ERR_ARG;
    }
    return launch_dma_impl<T, BM, BN, ROWB, BTR, NBUF, NW, false>(grp, NoLn{}, tiles, s);
}

// row-major B, or (btr: bf16 only) B stored [K][N]
template <typename T, int BM, int BN, int ROWB, int NBUF = 2, int NW = 4>
static int launch_dma_any(const GemmGroup& grp, int tiles, bool btr, hipStream_t s, const LnEpiGroup* lne = nullptr) {
    if constexpr (sizeof(T) == 2) {
        if (btr) return launch_dma<T, BM, BN, ROWB, true, NBUF, NW>(grp, tiles, s, lne);
    }
    return launch_dma<T, BM, BN, ROWB, false, NBUF, NW>(grp, tiles, s, lne);
}

// tile_start[] for a given tile shape; returns the total
static int retile(GemmGroup& grp, int bm, int bn, bool xcd2d = false) {
    int tiles = 0;
    const bool on = xcd2d && !grp.plain_tile_order;
    for (int i = 0; i < grp.count; ++i) {
        grp.tile_start[i] = tiles;
        const int tm = (grp.p[i].M + bm - 1) / bm, tn = (grp.p[i].N + bn - 1) / bn;
        grp.xcd_rg[i] = on ? (unsigned char)pick_xcd_rg(tiles, tm, tn, grp.p[i].M, grp.p[i].N) : 0;
        tiles += tm * tn;
    }
    for (int i = grp.count; i <= MTN_GEMM_MAX_GROUP; ++i) grp.tile_start[i] = tiles;
    return tiles;
}

template <typename T>
static int launch_gemm(const GemmGroup& grp, const AdamGroup& adam, int total_tiles, bool at, bool bt, bool dma_ok, hipStream_t s,
                       const LnEpiGroup* lne = nullptr, bool lne_emit = false) {
    dim3 grid(total_tiles), block(256);
    g_variant_tiles = total_tiles;
    g_variant = at ? (bt ? V_REG_TT : V_REG_TN) : (bt ? V_REG_NT : V_REG_NN);
    if constexpr (sizeof(T) == 2) {
        // 128x128 row-major tiles: measured 330-350 TFLOP/s on the memory K/V launches against 370-400 for the 64x64
        // register-staged kernel at 5 workgroups per CU (K = 512 is too short to amortise a 16-fragment epilogue at two
        // workgroups per CU) -> opt-in (MTN_GEMM_NTB_MIN_TILES=<tiles>), kept for larger contractions
        if (!at && !bt && !lne && MTN_ENV("MTN_GEMM_NTB_MIN_TILES") != nullptr) {
            bool ok = true;
            int t128 = 0;
            for (int i = 0; i < grp.count; ++i) {
                const mtn_gemm_problem& q = grp.p[i];
                ok = ok && !q.rowsum_out && q.K % 8 == 0 && (long)q.M * q.lda * 2 < (1L << 31) && (long)q.N * q.ldb * 2 < (1L << 31);
                t128 += ((q.M + 127) / 128) * ((q.N + 127) / 128);
            }
            const char* tmin = MTN_ENV("MTN_GEMM_NTB_MIN_TILES");
            if (ok && t128 >= atoi(tmin)) {
                static bool attr_set = false;
                if (!attr_set) {
                    (void)hipFuncSetAttribute((const void*)gemm_dma128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NTB_LDS);
                    attr_set = true;
                }
                GemmGroup g2 = grp;
                const int tiles = retile(g2, 128, 128);
                g_variant = V_DMA128; g_variant_tiles = tiles;
                hipLaunchKernelGGL(gemm_dma128_kernel, dim3(tiles), block, NTB_LDS, s, g2);
                MTN_CHECK_LAUNCH();
                return MTN_OK;
            }
        }
    }
    if constexpr (sizeof(T) == 2) {
        // one round (or a few) of 128 x 128 tiles with four stages in flight: launches with >= 192 such tiles and a contraction of
        // >= 768 (at K = 512 the register-staged 64 x 64 kernel is as fast: 18.4 vs 19.0 us on the generator logits), either B layout:
        // the memory-gradient GEMM 23.6 -> 17.3 us per launch, step +0.9 % (profiles/r03_w_gemm128x_ab.txt)
        const char* xmin = MTN_ENV("MTN_GEMM_128X_MIN_TILES");
        const int x_min = xmin ? atoi(xmin) : 192;
        if (!at && x_min > 0 && !lne_emit && (!lne || bt)) {
            bool ok = true;
            int t128 = 0, kmax = 0;
            for (int i = 0; i < grp.count; ++i) {
                const mtn_gemm_problem& q = grp.p[i];
                ok = ok && !q.rowsum_out && q.K % 8 == 0 && q.K >= 256 && q.N % 8 == 0 && q.lda % 8 == 0 && q.ldb % 8 == 0 &&
                     (long)q.M * q.lda * 2 < (1L << 31) && (long)(bt ? q.K : q.N) * q.ldb * 2 < (1L << 31);
                t128 += ((q.M + 127) / 128) * ((q.N + 127) / 128);
                kmax = q.K > kmax ? q.K : kmax;
            }
            static int attr_state = 0;                                  // 0 = not tried, 1 = the 128 KiB dynamic-LDS opt-in holds, -1 = refused
            if (ok && t128 >= x_min && (xmin || kmax >= 768) && attr_state == 0) {
                const bool good =
                    hipFuncSetAttribute((const void*)gemm_dma128x_kernel<false, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS) == hipSuccess &&
                    hipFuncSetAttribute((const void*)gemm_dma128x_kernel<true, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS) == hipSuccess &&
                    hipFuncSetAttribute((const void*)gemm_dma128x_kernel<false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS) == hipSuccess &&
                    hipFuncSetAttribute((const void*)gemm_dma128x_kernel<true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS) == hipSuccess &&
                    hipFuncSetAttribute((const void*)gemm_dma128x_kernel<true, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS + LNE_LDS_EXTRA) == hipSuccess;
                attr_state = good ? 1 : -1;                              // refused: this and every later launch take the 64 x 64 / register-staged kernels below
                if (!good) (void)hipGetLastError();
            }
            if (ok && t128 >= x_min && (xmin || kmax >= 768) && attr_state == 1) {            // (the memory gradient shares its launch with the K = 512 dX of the same group)
                GemmGroup g2 = grp;
                const int tiles = retile(g2, 128, 128);
                g_variant = V_DMA128X; g_variant_tiles = tiles;
                if (lne) hipLaunchKernelGGL((gemm_dma128x_kernel<true, 16, true>), dim3(tiles), dim3(1024), G8_LDS + LNE_LDS_EXTRA, s, g2, *lne);
                else {                 // (sixteen waves: +0.5 % against eight, round 3)
                    if (bt) hipLaunchKernelGGL((gemm_dma128x_kernel<true, 16>), dim3(tiles), dim3(1024), G8_LDS, s, g2, NoLn{});
                    else hipLaunchKernelGGL((gemm_dma128x_kernel<false, 16>), dim3(tiles), dim3(1024), G8_LDS, s, g2, NoLn{});
                }
                MTN_CHECK_LAUNCH();
                return MTN_OK;
            }
        }
    }
    // B stored contraction-major ([K][N], b_trans = 1: dX = dY W with W as the forward pass keeps it): same kernel, the B tile is
    // brought as [k][columns] and its fragments come out of the transposing LDS read — bf16 only, N and ldb multiples of 8
    bool btr_ok = bt && sizeof(T) == 2;
    for (int i = 0; i < grp.count && btr_ok; ++i)
        btr_ok = grp.p[i].N % 8 == 0 && grp.p[i].ldb % 8 == 0 && (long)grp.p[i].K * grp.p[i].ldb * 2 < (1L << 31);
    if (lne && !(!at && dma_ok && (!bt || btr_ok))) {
        mtn_set_error("mtn_gemm: a LayerNorm epilogue needs the LDS-DMA kernel (bf16, a_trans = 0, aligned operands, a grid of at most 640 / 4096 tiles)");
        return MTN_ERR_ARG;
    }
    if (!at && dma_ok && (!bt || btr_ok)) {
        // Tile choice by the bytes ONE CU has to pull (the bound of these launches, ~27 GB/s per CU): 64x64 tiles run one
        // workgroup per CU in ceil(tiles/256) rounds of (64+64)*K bytes; 32x32 tiles spread 4x the workgroups of half the
        // size, two per CU (their DMA latencies overlap: x0.75, fitted on tools/gemm_bench.hip).  MTN_GEMM_TILE forces one.
        GemmGroup g2 = grp;
        const char* force = MTN_ENV("MTN_GEMM_TILE");
        int f = force ? atoi(force) : 0;
        if (lne_emit) f = 64;                          // the emitted row-sum partials are per 64-column block
        if (lne && f == 3264) f = 0;
        double b64 = 0, b32 = 0, wg32max = 0;
        int t64 = 0;
        for (int i = 0; i < grp.count; ++i) {
            const mtn_gemm_problem& q = grp.p[i];
            const double kb = (double)q.K * sizeof(T);
            const int n64 = ((q.M + 63) / 64) * ((q.N + 63) / 64), n32 = ((q.M + 31) / 32) * ((q.N + 31) / 32);
            t64 += n64;
            b64 += n64 * 128.0 * kb;
            b32 += n32 * 64.0 * kb;
            if (64.0 * kb > wg32max) wg32max = 64.0 * kb;
        }
        double c64 = (double)((t64 + 255) / 256) * (b64 / t64);
        // 257..512 tiles of 64x64 run in ONE round on half-size stages (two workgroups per CU, like the 32x32 tiles: x0.75)
        if (t64 > 256 && t64 <= 512) c64 = 0.75 * b64 / 256.0;   // +0.3 % cfg2, +0.5 % at 64 samples (profiles/r03_o_c64h_ab.txt)
        double c32 = 0.75 * b32 / 256.0;
        if (c32 < wg32max) c32 = wg32max;
        // half-size stages (256 B of contraction per row) double the resident workgroups: taken when the launch would
        // otherwise need a second round (64x64: one workgroup per CU at 128 KiB; 32x32: two at 64 KiB)
        const bool half_ok = true;
        const bool half_force = MTN_ENV("MTN_GEMM_FORCE_HALF") != nullptr;      // tests
        {   // eight waves pull a 64 x 64 workgroup's bytes faster than four did when the 0.75 above was fitted: the 64 x 64 cost is
            // weighted 0.8 (sweep in profiles/r03_ac_c64_scale_sweep.txt: cfg2 +0.3 %, batch 64 +2.1 % against 1.0; 0.5 loses)
            c64 *= 0.8;
        }
        // (a ring of four half-size stages for long contractions was measured at -0.8 % on the cfg2 step, profiles/r03_x_deep_ring_ab.txt: not kept)
        if (f == 64 || (!f && c64 <= c32)) {
            const int t = retile(g2, 64, 64, true);
            if (half_force || (half_ok && t > 256)) {
                return launch_dma_any<T, 64, 64, 256, 2, 8>(g2, t, bt, s, lne);       // (eight waves: four were the rounds 1-2 form)
            }
            if constexpr (sizeof(T) == 2) {
                // sixteen waves (a 4 x 4 grid of 16 x 16 wave tiles): cfg2 step +0.6 %, batch 64 +0.4 % against eight (profiles/r04_s_nw16_ab.txt).
                // MTN_GEMM_NW16 = 0 none, 1 plain launches only, 2 launches with a LayerNorm epilogue only (default 3: both)
                const char* n16 = MTN_ENV("MTN_GEMM_NW16");
                const int m16 = n16 ? atoi(n16) : 3;
                if (((m16 & 1) && !lne) || ((m16 & 2) && lne)) return launch_dma_any<T, 64, 64, 512, 2, 16>(g2, t, bt, s, lne);
            }
            return launch_dma_any<T, 64, 64, 512, 2, 8>(g2, t, bt, s, lne);     // eight waves: see the kernel
        }
        if (f == 3264) return launch_dma_any<T, 32, 64, 512>(g2, retile(g2, 32, 64, true), bt, s);
        const int t = retile(g2, 32, 32, true);
        if (half_force || (half_ok && t > 1024)) return launch_dma_any<T, 32, 32, 256>(g2, t, bt, s, lne);   // (measured in the step: 640 tiles 10.7 vs 10.1 us, 1280 tiles 12.7 vs 13.8 us)
        return launch_dma_any<T, 32, 32, 512>(g2, t, bt, s, lne);
    } else if (!at && !bt) hipLaunchKernelGGL((gemm_kernel<T, false, false>), grid, block, 0, s, grp, NoAdam{});
    else if (!at && bt) hipLaunchKernelGGL((gemm_kernel<T, false, true>), grid, block, 0, s, grp, NoAdam{});
    else if (at && bt) {
        bool ttd = sizeof(T) == 2;      // LDS-DMA + transposing LDS reads (bf16)
        for (int i = 0; i < grp.count; ++i)
            ttd = ttd && grp.p[i].M % 8 == 0 && grp.p[i].N % 8 == 0 && (long)grp.p[i].K * grp.p[i].lda * 2 < (1L << 31) &&
                  (long)grp.p[i].K * grp.p[i].ldb * 2 < (1L << 31);
        bool ttb = ttd;                    // 128x128 tiles when every problem fills them
        int t128 = 0;
        for (int i = 0; i < grp.count; ++i) {
            ttb = ttb && grp.p[i].M >= 128 && grp.p[i].N >= 128;
            t128 += ((grp.p[i].M + 127) / 128) * ((grp.p[i].N + 127) / 128);
        }
        const char* tmin = MTN_ENV("MTN_GEMM_TTB_MIN_TILES");
        ttb = ttb && (t128 >= (tmin ? atoi(tmin) : 192) || adam.any);          // the coalesced optimiser epilogue lives in the 128-tile kernel
        if (ttb) {
            if constexpr (sizeof(T) == 2) {
                static bool attr_set = false;
                if (!attr_set) {
                    (void)hipFuncSetAttribute((const void*)gemm_tt_dma128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TTB_LDS);
                    attr_set = true;
                }
                GemmGroup g2 = grp;
                const int tiles = retile(g2, 128, 128);
                g_variant = V_TT_DMA128; g_variant_tiles = tiles;
                AdamGroup a2 = adam;
                a2.lds_epilogue = adam.any;
                for (int i = 0; i < grp.count; ++i) {
                    const AdamSlot& sl = adam.a[i];
                    if (!sl.p) continue;
                    if (grp.p[i].ldc % 4 != 0 || (sl.lpT && (sl.ldT % 8 != 0 || (((uintptr_t)sl.lpT) & 15) != 0)) || (((uintptr_t)sl.p) & 15) != 0 ||
                        (sl.lp && (((uintptr_t)sl.lp) & 7) != 0))
                        a2.lds_epilogue = 0;
                }
                hipLaunchKernelGGL(gemm_tt_dma128_kernel, dim3(tiles), block, TTB_LDS, s, g2, a2);
            }
        } else if (ttd) {
            g_variant = V_TT_DMA;
            if constexpr (sizeof(T) == 2) hipLaunchKernelGGL(gemm_tt_dma_kernel, grid, block, TTD_LDS, s, grp, adam);
        } else hipLaunchKernelGGL((gemm_kernel<T, true, true>), grid, block, 0, s, grp, adam);
    }
    else hipLaunchKernelGGL((gemm_kernel<T, true, false>), grid, block, 0, s, grp, NoAdam{});
    MTN_CHECK_LAUNCH();
    return MTN_OK;
}

extern "C" int mtn_gemm(int dtype, int count, const mtn_gemm_problem* problems, void* stream) {
    MTN_CHECK_ARG(dtype == MTN_F32 || dtype == MTN_BF16, "bad dtype");
    MTN_CHECK_ARG(count >= 1 && count <= MTN_GEMM_MAX_GROUP && problems, "bad problem count");
    GemmGroup grp;
    AdamGroup adam;
    LnEpiGroup lne;
    memset(&grp, 0, sizeof(grp));
    memset(&adam, 0, sizeof(adam));
    memset(&lne, 0, sizeof(lne));
    int n_lne = 0;
    bool lne_emit = false;
    grp.count = count;
    grp.plain_tile_order = 0;
    grp.epi_pre = !(MTN_ENV("MTN_GEMM_EPI_PRE") && MTN_ENV("MTN_GEMM_EPI_PRE")[0] == '0');
    int tiles = 0;
    const int align = (dtype == MTN_BF16) ? 8 : 4;
    for (int i = 0; i < count; ++i) {
        const mtn_gemm_problem& p = problems[i];
        MTN_CHECK_ARG(p.A && p.B && (p.out_f32 || p.out_lp || (p.ln && p.ln->mode == MTN_LN_CONSUME)), "null operand/output");
        MTN_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "empty problem");
        MTN_CHECK_ARG(p.a_trans == problems[0].a_trans && p.b_trans == problems[0].b_trans, "mixed layouts in one group");
        MTN_CHECK_ARG(p.lda % align == 0 && p.ldb % align == 0, "lda/ldb must be multiples of 16 bytes");
        MTN_CHECK_ARG(p.a_trans || p.K % align == 0, "K must be a multiple of 16 bytes for a row-major A");
        MTN_CHECK_ARG(p.b_trans || p.K % align == 0, "K must be a multiple of 16 bytes for a row-major B");
        MTN_CHECK_ARG(!p.a_trans || p.M % 4 == 0, "M must be a multiple of 4 for a transposed A");
        MTN_CHECK_ARG(!p.b_trans || p.N % 4 == 0, "N must be a multiple of 4 for a transposed B");
        MTN_CHECK_ARG((((uintptr_t)p.A) & 15) == 0 && (((uintptr_t)p.B) & 15) == 0, "A/B must be 16-byte aligned");
        grp.tile_start[i] = tiles;
        tiles += ((p.M + TILE - 1) / TILE) * ((p.N + TILE - 1) / TILE);
        grp.p[i] = p;
        grp.p[i].adam = nullptr;                  // host pointer: never dereferenced on the device
        grp.p[i].ln = nullptr;
        if (p.ln && p.ln->mode != 0) {
            const mtn_ln_epilogue& e = *p.ln;
            MTN_CHECK_ARG(dtype == MTN_BF16 && !p.a_trans && !p.adam && p.b_trans, "LayerNorm epilogue: bf16, dX = dY W problems");
            MTN_CHECK_ARG(n_lne < LNE_MAX_SLOTS, "LayerNorm epilogue: too many problems with one in this launch");
            MTN_CHECK_ARG(e.part, "LayerNorm epilogue: null row-sum partial buffer");
            LnEpiSlot& sl = lne.s[n_lne];
            sl.mode = e.mode; sl.np = e.np; sl.fold = e.fold; sl.part = e.part;
            sl.x = e.x; sl.a2 = e.a2; sl.mean = e.mean; sl.rstd = e.rstd; sl.dres = e.dres; sl.dx = e.dx; sl.dx_lp = e.dx_lp;
            sl.colpart = e.colpart; sl.dx_lp_drop = e.dx_lp_drop; sl.gate_inv_scale = e.gate_inv_scale; sl.eps = e.eps;
            if (e.mode == MTN_LN_EMIT) {
                MTN_CHECK_ARG(e.fold && p.N % 64 == 0 && p.out_lp && (((uintptr_t)e.fold) & 15) == 0, "LayerNorm epilogue (emit): fold vectors, N % 64 == 0, a compute-dtype output");
                lne_emit = true;
            } else {
                MTN_CHECK_ARG(e.mode == MTN_LN_CONSUME, "LayerNorm epilogue: bad mode");
                MTN_CHECK_ARG(e.x && e.a2 && e.mean && e.rstd && e.dx && e.np >= 1, "LayerNorm epilogue (consume): null input");
                MTN_CHECK_ARG(p.N % 16 == 0 && !p.bias && !p.relu && !p.gate && !p.residual && p.drop.p == 0.f, "LayerNorm epilogue (consume): plain g = dY W with N % 16 == 0");
                MTN_CHECK_ARG(((((uintptr_t)e.x) | ((uintptr_t)e.a2) | ((uintptr_t)e.dx) | ((uintptr_t)e.dres) | ((uintptr_t)e.colpart) | ((uintptr_t)e.part)) & 15) == 0 &&
                              (((uintptr_t)e.dx_lp) & 7) == 0, "LayerNorm epilogue (consume): misaligned buffer");
            }
            lne.slot_of[i] = (unsigned char)(++n_lne);
        }
        if (p.adam) {
            const mtn_adam_fuse& f = *p.adam;
            MTN_CHECK_ARG(p.a_trans && p.b_trans, "the optimiser epilogue rides on parameter-gradient GEMMs (a_trans = b_trans = 1) only");
            MTN_CHECK_ARG(f.p && f.m && f.v && f.state && p.out_f32, "optimiser epilogue: p, m, v, state and out_f32 are required");
            MTN_CHECK_ARG(!p.bias && !p.relu && !p.gate && !p.residual && !p.out_lp && p.drop.p == 0.f, "optimiser epilogue: C must be the whole, plain gradient");
            MTN_CHECK_ARG(!f.p_lpT || f.ldT >= p.M, "optimiser epilogue: ldT is the row stride of the transposed copy (>= M)");
            if (!adam.any) {
                adam.state = f.state; adam.grad_scale = f.grad_scale; adam.beta1 = f.beta1; adam.beta2 = f.beta2; adam.eps = f.eps;
                adam.any = 1;
            }
            MTN_CHECK_ARG(adam.state == f.state && adam.grad_scale == f.grad_scale && adam.beta1 == f.beta1 && adam.beta2 == f.beta2 && adam.eps == f.eps,
                          "optimiser epilogue: one set of hyper-parameters per launch");
            AdamSlot& a = adam.a[i];
            a.p = f.p; a.m = f.m; a.v = f.v; a.lp = f.p_lp; a.lpT = f.p_lpT; a.ldT = f.ldT; a.write_grad = f.write_grad;
        }
    }
    for (int i = count; i <= MTN_GEMM_MAX_GROUP; ++i) grp.tile_start[i] = tiles;
    hipStream_t s = (hipStream_t)stream;
    // LDS-DMA path: row-major operands below 2 GiB, no row-sum side output, and a grid that fits the chip in about two
    // rounds (128 KiB of LDS = one workgroup per CU): large grids are throughput-bound and do better on the
    // register-staged kernel at 5 workgroups per CU.
    // (b_trans = 1: the register-staged <N,T> fallback is 1.5x slower than <N,N> — 33.7 vs 21.7 us on the memories' dX launch,
    //  the LDS-DMA kernel with half stages 24.2: tools/nt_gemm_probe.py — so the LDS-DMA kernel keeps those too)
    // (a launch with a LayerNorm epilogue stays on the LDS-DMA kernels whatever its grid: the epilogue lives there)
    bool dma_ok = !problems[0].rowsum_out && (n_lne > 0 || tiles <= (problems[0].b_trans && !problems[0].a_trans ? 4096 : 640));
    const long esz = (dtype == MTN_BF16) ? 2 : 4;
    for (int i = 0; i < count; ++i) {
        const mtn_gemm_problem& p = problems[i];
        if (p.rowsum_out || (long)p.M * p.lda * esz >= (1L << 31) || (long)p.N * p.ldb * esz >= (1L << 31)) dma_ok = false;
    }
    // many large problems with the short contraction K = 512 (the memories' K|V projections for all layers): csrc/gemm_k512.hip
    int rc = 1, k512_tiles = 0;
    const int k512_min = 512;
    const int took = (dtype == MTN_BF16 && k512_min > 0 && !n_lne) ? gemm_k512_try(count, problems, k512_min, s, &k512_tiles) : 0;
    if (took < 0) { mtn_set_error("gemm_k512_kernel: launch failed"); return MTN_ERR_LAUNCH; }
    if (took == 1) { g_variant = V_K512; g_variant_tiles = k512_tiles; rc = MTN_OK; }
    else
        rc = (dtype == MTN_BF16) ? launch_gemm<bf16_t>(grp, adam, tiles, problems[0].a_trans, problems[0].b_trans, dma_ok, s, n_lne ? &lne : nullptr, lne_emit)
                                 : launch_gemm<float>(grp, adam, tiles, problems[0].a_trans, problems[0].b_trans, dma_ok, s);
    if (g_census_on && rc == MTN_OK) {
        CensusEntry e;
        memset(&e, 0, sizeof(e));
        e.dtype = dtype; e.count = count; e.variant = g_variant; e.tiles = g_variant_tiles; e.table = -1;
        for (int i = 0; i < count; ++i) {
            e.p[i] = problems[i];
            if (problems[i].adam) e.adam[i] = *problems[i].adam;       // the caller's descriptor dies with the call
            if (problems[i].ln) e.ln[i] = *problems[i].ln;
        }
        g_census.push_back(e);
    }
    return rc;
}

// Host side of the table form.  The table is staged through pinned memory and copied with the launch stream; eager calls
// cycle through a small ring of (pinned, device) slot pairs guarded by events, a call made while the stream is being
// captured into a hipGraph gets a slot pair of its own that is never reused (the graph's memcpy node reads it at every replay).
#define MTN_TT_SLOT_BYTES (256 * 1024)
struct TTSlot { unsigned char* host; unsigned char* dev; hipEvent_t ev; bool fresh; };
static TTSlot g_tt_ring[8];
static int g_tt_next = 0;
static int tt_slot_alloc(TTSlot& sl) {
    // allocation calls are "unsafe" under a global-mode stream capture (they would invalidate it): relax the mode for this thread
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    (void)hipThreadExchangeStreamCaptureMode(&mode);
    struct Restore { hipStreamCaptureMode m; ~Restore() { (void)hipThreadExchangeStreamCaptureMode(&m); } } restore{mode};
    if (hipHostMalloc((void**)&sl.host, MTN_TT_SLOT_BYTES, hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void**)&sl.dev, MTN_TT_SLOT_BYTES) != hipSuccess ||
        hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming) != hipSuccess) {
        mtn_set_error("mtn_gemm_tt_table: cannot allocate a staging slot: %s", hipGetErrorString(hipGetLastError()));
        return MTN_ERR_LAUNCH;
    }
    sl.fresh = true;
    return MTN_OK;
}

extern "C" int mtn_gemm_tt_table(int dtype, int count, const mtn_gemm_problem* problems, void* stream) {
    return mtn_gemm_tt_table_aux(dtype, count, problems, nullptr, stream);
}

extern "C" int mtn_gemm_tt_table_aux(int dtype, int count, const mtn_gemm_problem* problems, const mtn_tt_aux* aux, void* stream) {
    MTN_CHECK_ARG(dtype == MTN_BF16, "the table form is bf16 only");
    MTN_CHECK_ARG(count >= 1 && count <= 1024 && problems, "bad problem count");
    hipStream_t s = (hipStream_t)stream;
    long tiles = 0;
    for (int i = 0; i < count; ++i) {
        const mtn_gemm_problem& p = problems[i];
        MTN_CHECK_ARG(p.A && p.B && p.out_f32 && p.M > 0 && p.N > 0 && p.K > 0, "null operand/output or empty problem");
        MTN_CHECK_ARG(p.a_trans && p.b_trans, "the table form takes parameter-gradient problems (a_trans = b_trans = 1) only");
        MTN_CHECK_ARG(!p.bias && !p.relu && !p.gate && !p.residual && !p.out_lp && p.drop.p == 0.f, "plain C = A^T B only");
        MTN_CHECK_ARG(p.M % 8 == 0 && p.N % 8 == 0 && p.lda % 8 == 0 && p.ldb % 8 == 0 && p.ldc % 4 == 0, "M, N, lda, ldb multiples of 8; ldc of 4");
        MTN_CHECK_ARG((long)p.K * p.lda * 2 < (1L << 31) && (long)p.K * p.ldb * 2 < (1L << 31), "operands must be below 2 GiB");
        MTN_CHECK_ARG((((uintptr_t)p.A) & 15) == 0 && (((uintptr_t)p.B) & 15) == 0 && (((uintptr_t)p.out_f32) & 15) == 0, "16-byte aligned bases");
        const long t = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
        MTN_CHECK_ARG(t <= 4095, "at most 4095 tiles per problem");
        tiles += t;
    }
    // front tiles (mtn_tt_aux): 256 columns of a LayerNorm's [da2 | db2] per unit, one chunk per tile; padded to a multiple of 8 so
    // that dW tile i still runs on XCD i % 8 (the tile map below deals problems to XCDs by grid index)
    int n_ln_units = 0, n_chunks = 0, n_front = 0;
    if (aux) {
        MTN_CHECK_ARG(aux->p && aux->g && aux->m && aux->v && aux->n_flat > 0, "mtn_tt_aux: flat buffers are required");
        MTN_CHECK_ARG(aux->n_ln >= 0 && aux->n_chunks >= 0 && (aux->n_ln == 0 || aux->ln) && (aux->n_chunks == 0 || (aux->chunk_off && aux->chunk_len)), "mtn_tt_aux: bad lists");
        for (int i = 0; i < aux->n_ln; ++i) {
            const mtn_tt_ln_unit& u = aux->ln[i];
            MTN_CHECK_ARG(u.partial && u.nparts > 0 && u.d > 0 && u.a_off >= 0 && u.b_off >= 0 && u.a_off + u.d <= aux->n_flat && u.b_off + u.d <= aux->n_flat,
                          "mtn_tt_aux: LayerNorm unit outside the flat buffers");
            n_ln_units += (2 * u.d + 255) / 256;
        }
        for (int i = 0; i < aux->n_chunks; ++i)
            MTN_CHECK_ARG(aux->chunk_off[i] >= 0 && aux->chunk_len[i] > 0 && aux->chunk_len[i] <= 4096 && aux->chunk_off[i] % 4 == 0 && aux->chunk_len[i] % 4 == 0 &&
                          aux->chunk_off[i] + aux->chunk_len[i] <= aux->n_flat, "mtn_tt_aux: chunks are <= 4096 elements, offsets and lengths multiples of 4");
        n_chunks = aux->n_chunks;
        n_front = (n_ln_units + n_chunks + 7) / 8 * 8;
    }
    const long ln_bytes = (long)n_ln_units * sizeof(TTLnUnit);
    const long need = ((long)sizeof(TTHeader) + (long)count * sizeof(TTProblem) + (tiles + n_front) * 4 + 15) / 16 * 16 + ln_bytes + (long)n_chunks * 12 + 16;
    MTN_CHECK_ARG(need <= MTN_TT_SLOT_BYTES, "problem list too large for one table");
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    TTSlot own, *sl;
    if (cap != hipStreamCaptureStatusNone) {
        if (tt_slot_alloc(own) != MTN_OK) return MTN_ERR_LAUNCH;       // lives as long as the process: the captured graph reads it
        sl = &own;
    } else {
        sl = &g_tt_ring[g_tt_next];
        g_tt_next = (g_tt_next + 1) % 8;
        if (!sl->host) { if (tt_slot_alloc(*sl) != MTN_OK) return MTN_ERR_LAUNCH; }
        else if (!sl->fresh) (void)hipEventSynchronize(sl->ev);          // its previous copy has been consumed
    }
    TTHeader* H = (TTHeader*)sl->host;
    memset(H, 0, sizeof(*H));
    H->n_problems = count; H->n_tiles = (int)tiles;
    H->problems_off = sizeof(TTHeader);
    H->tilemap_off = sizeof(TTHeader) + (long)count * sizeof(TTProblem);
    TTProblem* Q = (TTProblem*)(sl->host + H->problems_off);
    uint32_t* map = (uint32_t*)(sl->host + H->tilemap_off);
    bool lds_ok = true;
    if (aux) {
        H->n_front = n_front; H->n_ln_units = n_ln_units; H->n_chunks = n_chunks;
        H->fp = aux->p; H->fg = aux->g; H->fm = aux->m; H->fv = aux->v; H->flp = (bf16_t*)aux->lp;
        H->ln_off = ((long)sizeof(TTHeader) + (long)count * sizeof(TTProblem) + (tiles + n_front) * 4 + 15) / 16 * 16;
        H->chunk_off_off = H->ln_off + ln_bytes;                                    // (ln_bytes is a multiple of 8: longs stay aligned)
        H->chunk_len_off = H->chunk_off_off + (long)n_chunks * 8;
        TTLnUnit* U = (TTLnUnit*)(sl->host + H->ln_off);
        int k = 0;
        for (int i = 0; i < aux->n_ln; ++i)
            for (int c0 = 0; c0 < 2 * aux->ln[i].d; c0 += 256) {
                TTLnUnit& u = U[k++];
                memset(&u, 0, sizeof(u));
                u.partial = aux->ln[i].partial; u.nparts = aux->ln[i].nparts; u.d = aux->ln[i].d; u.col0 = c0; u.a_off = aux->ln[i].a_off; u.b_off = aux->ln[i].b_off;
            }
        if (n_chunks) {
            memcpy(sl->host + H->chunk_off_off, aux->chunk_off, (size_t)n_chunks * 8);
            memcpy(sl->host + H->chunk_len_off, aux->chunk_len, (size_t)n_chunks * 4);
        }
    }
    int first = 0;
    for (int i = 0; i < count; ++i) {
        const mtn_gemm_problem& p = problems[i];
        TTProblem& q = Q[i];
        memset(&q, 0, sizeof(q));
        q.A = p.A; q.B = p.B; q.out = p.out_f32; q.rowsum = p.rowsum_out;
        q.bias_off = -1;
        if (aux && aux->bias_adam && p.rowsum_out && p.rowsum_out >= aux->g && p.rowsum_out + p.M <= aux->g + aux->n_flat) q.bias_off = (long)(p.rowsum_out - aux->g);
        q.lda = p.lda; q.ldb = p.ldb; q.M = p.M; q.N = p.N; q.K = p.K; q.ldc = p.ldc;
        q.tiles_m = (p.M + 127) / 128; q.tiles_n = (p.N + 127) / 128; q.first_tile = first;
        if (p.adam) {
            const mtn_adam_fuse& f = *p.adam;
            MTN_CHECK_ARG(f.p && f.m && f.v && f.state, "optimiser epilogue: p, m, v and state are required");
            MTN_CHECK_ARG(!f.p_lpT || f.ldT >= p.M, "optimiser epilogue: ldT is the row stride of the transposed copy (>= M)");
            if (!H->any) { H->state = f.state; H->grad_scale = f.grad_scale; H->beta1 = f.beta1; H->beta2 = f.beta2; H->eps = f.eps; H->any = 1; }
            MTN_CHECK_ARG(H->state == f.state && H->grad_scale == f.grad_scale && H->beta1 == f.beta1 && H->beta2 == f.beta2 && H->eps == f.eps,
                          "optimiser epilogue: one set of hyper-parameters per launch");
            q.p = f.p; q.m = f.m; q.v = f.v; q.lp = f.p_lp; q.lpT = f.p_lpT; q.ldT = f.ldT; q.write_grad = f.write_grad;
            if ((f.p_lpT && (f.ldT % 8 != 0 || (((uintptr_t)f.p_lpT) & 15) != 0)) || (((uintptr_t)f.p) & 15) != 0 || (f.p_lp && (((uintptr_t)f.p_lp) & 7) != 0))
                lds_ok = false;
        }
        first += q.tiles_m * q.tiles_n;
    }
    if (aux && (n_front > 0 || aux->bias_adam)) {
        MTN_CHECK_ARG(aux->state, "mtn_tt_aux: the optimiser state is required");
        if (!H->any) { H->state = aux->state; H->grad_scale = aux->grad_scale; H->beta1 = aux->beta1; H->beta2 = aux->beta2; H->eps = aux->eps; H->any = 1; }
        MTN_CHECK_ARG(H->state == aux->state && H->grad_scale == aux->grad_scale && H->beta1 == aux->beta1 && H->beta2 == aux->beta2 && H->eps == aux->eps,
                      "mtn_tt_aux: one set of optimiser hyper-parameters per launch");
    }
    H->lds_epilogue = (H->any && lds_ok) ? 1 : 0;
    // Tile map.  Workgroups are dealt to the 8 XCDs round-robin by grid index and each XCD has its own L2, so a problem whose
    // tiles are spread over all XCDs has its operand panels fetched from HBM by each of them (measured: 1.5 GB of operand
    // traffic per launch for 0.44 GB of operands).  Here every problem is given to ONE XCD: grid index i holds a tile of the
    // queue of XCD i % 8; problems are dealt to the queues longest-processing-time-first on an estimate of their cost
    // (tiles x (contraction rows + the epilogue's bytes in row equivalents)), which balances the queues to within one problem.
    {
        H->plain_tile_order = 1;                                          // inside a problem: plain order, one L2 sees them all
        std::vector<int> order(count);
        std::vector<double> cost(count);
        for (int i = 0; i < count; ++i) {
            order[i] = i;
            cost[i] = (double)Q[i].tiles_m * Q[i].tiles_n * (Q[i].K + (Q[i].p ? 896.0 : 128.0));
        }
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] > cost[b]; });
        std::vector<int> owner(count);
        double load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < count; ++k) {
            int c = 0;
            for (int x = 1; x < 8; ++x) if (load[x] < load[c]) c = x;
            owner[order[k]] = c;
            load[c] += cost[order[k]];
        }
        std::vector<uint32_t> queue[8];                                    // in the caller's order: long and short contractions alternate
        for (int i = 0; i < count; ++i)
            for (int k = 0; k < Q[i].tiles_m * Q[i].tiles_n; ++k) queue[owner[i]].push_back(((uint32_t)i << 12) | (uint32_t)k);
        size_t head[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (long i = 0; i < tiles; ++i) {
            int c = (int)(i & 7);
            if (head[c] >= queue[c].size()) {                              // this XCD's queue is done: help the one with most left
                int best = -1;
                size_t left = 0;
                for (int x = 0; x < 8; ++x)
                    if (queue[x].size() - head[x] > left) { left = queue[x].size() - head[x]; best = x; }
                c = best;
            }
            map[i] = queue[c][head[c]++];
        }
    }
    if (n_front > 0) {
        // The optimiser units sit at the FRONT of the grid (a multiple of 8 of them, so that a dW tile keeps its grid index mod 8 and with it
        // its XCD).  Measured, same box, cfg2 step (profiles/r05_d_tail_head_ab.txt): all at the front 3.559 -> 3.529 ms (the first
        // resident dW tiles are in their contraction phase and leave HBM idle); dealt in between the dW tiles +-0 (they compete with
        // the tiles' streaming phases); all at the end -14 us.
        std::vector<uint32_t> dmap(map, map + tiles);
        long pos = 0;
        for (long au = 0; au < n_front; ++au) map[pos++] = 0x80000000u | (uint32_t)au;
        for (long di = 0; di < tiles; ++di) map[pos++] = dmap[di];
    }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_tt_dma128_table_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TTB_LDS);
        attr_set = true;
    }
    hipError_t e;
    if (cap != hipStreamCaptureStatusNone) {
        // under capture the table is constant for the life of the graph: copy it NOW (synchronously, outside the graph) instead
        // of recording a copy node that every replay would re-execute (5.5 us per step)
        hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
        (void)hipThreadExchangeStreamCaptureMode(&mode);
        static hipStream_t copy_stream = nullptr;                 // non-blocking: no implicit dependency on the capturing stream
        e = copy_stream ? hipSuccess : hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMemcpyAsync(sl->dev, sl->host, (size_t)need, hipMemcpyHostToDevice, copy_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(copy_stream);
        (void)hipThreadExchangeStreamCaptureMode(&mode);
    } else {
        e = hipMemcpyAsync(sl->dev, sl->host, (size_t)need, hipMemcpyHostToDevice, s);
    }
    if (e != hipSuccess) { mtn_set_error("mtn_gemm_tt_table: staging copy failed: %s", hipGetErrorString(e)); return MTN_ERR_LAUNCH; }
    hipLaunchKernelGGL(gemm_tt_dma128_table_kernel, dim3((unsigned)(tiles + n_front)), dim3(256), TTB_LDS, s, (const TTHeader*)sl->dev);
    MTN_CHECK_LAUNCH();
    if (cap == hipStreamCaptureStatusNone) { (void)hipEventRecord(sl->ev, s); sl->fresh = false; }
    if (g_census_on) {
        CensusTable t;
        t.p.assign(problems, problems + count);
        t.adam.resize(count);
        for (int i = 0; i < count; ++i)
            if (problems[i].adam) t.adam[i] = *problems[i].adam;
        t.has_aux = aux != nullptr;
        if (aux) {
            t.aux = *aux;
            t.ln.assign(aux->ln, aux->ln + aux->n_ln);
            t.chunk_off.assign(aux->chunk_off, aux->chunk_off + aux->n_chunks);
            t.chunk_len.assign(aux->chunk_len, aux->chunk_len + aux->n_chunks);
        }
        CensusEntry e;
        memset(&e, 0, sizeof(e));
        e.dtype = dtype; e.count = count; e.variant = V_TT_TABLE; e.tiles = (int)tiles; e.table = (int)g_census_tables.size();
        g_census_tables.push_back(t);
        g_census.push_back(e);
    }
    return MTN_OK;
}

extern "C" int mtn_census_begin(void) { g_census.clear(); g_census_tables.clear(); g_census_on = true; return MTN_OK; }
extern "C" int mtn_census_end(void) { g_census_on = false; return (int)g_census.size(); }
extern "C" const char* mtn_census_variant_name(int variant) { return (variant >= 0 && variant < V_COUNT) ? g_variant_name[variant] : "?"; }

extern "C" int mtn_census_info(int i, mtn_census_launch* out) {
    MTN_CHECK_ARG(i >= 0 && i < (int)g_census.size() && out, "bad census index");
    const CensusEntry& e = g_census[i];
    const double esz = (e.dtype == MTN_BF16) ? 2.0 : 4.0;
    memset(out, 0, sizeof(*out));
    out->dtype = e.dtype; out->count = e.count; out->variant = e.variant; out->workgroups = e.tiles;
    if (e.table >= 0) {
        // algorithmic bytes: operands once, and per output element either the stored gradient (4 B) or the optimiser
        // epilogue's streams: p, m, v read and written (24 B) + the two compute-dtype copies (2 x 2 B)
        const CensusTable& t = g_census_tables[e.table];
        for (int k = 0; k < e.count; ++k) {
            const mtn_gemm_problem& p = t.p[k];
            out->flops += 2.0 * p.M * p.N * p.K;
            double per = 4.0;
            if (p.adam) per = 24.0 + (t.adam[k].p_lp ? esz : 0.0) + (t.adam[k].p_lpT ? esz : 0.0) + (t.adam[k].write_grad ? 4.0 : 0.0);
            out->bytes += ((double)p.M + p.N) * p.K * esz + (double)p.M * p.N * per;
            if (k < 4) { out->M[k] = p.M; out->N[k] = p.N; out->K[k] = p.K; }
        }
        if (t.has_aux) {
            // front tiles: chunks move p, g, m, v in (16 B) and p, m, v + the compute-dtype copy out (12 B + esz) per element; a LayerNorm
            // unit reads its partial rows and does the same for 2d elements (+ the gradient written); a fused bias: 12 B in, 16 B + esz out
            for (size_t c = 0; c < t.chunk_len.size(); ++c) out->bytes += (double)t.chunk_len[c] * (28.0 + (t.aux.lp ? esz : 0.0));
            for (size_t u = 0; u < t.ln.size(); ++u) out->bytes += 2.0 * t.ln[u].d * (4.0 * t.ln[u].nparts + 28.0 + (t.aux.lp ? esz : 0.0));
            if (t.aux.bias_adam)
                for (int k = 0; k < e.count; ++k)
                    if (t.p[k].rowsum_out && t.p[k].rowsum_out >= t.aux.g && t.p[k].rowsum_out + t.p[k].M <= t.aux.g + t.aux.n_flat)
                        out->bytes += (double)t.p[k].M * (28.0 + (t.aux.lp ? esz : 0.0));
        }
        return MTN_OK;
    }
    for (int k = 0; k < e.count; ++k) {
        const mtn_gemm_problem& p = e.p[k];
        out->flops += 2.0 * p.M * p.N * p.K;
        if (p.ln && e.ln[k].mode == MTN_LN_CONSUME)
            out->bytes += ((double)p.M + p.N) * p.K * esz + (double)p.M * p.N * (8.0 + (e.ln[k].dres ? 4.0 : 0.0) + (e.ln[k].dx_lp ? esz : 0.0));
        else
        out->bytes += ((double)p.M + p.N) * p.K * esz + (double)p.M * p.N * ((p.out_f32 ? 4.0 : 0.0) + (p.out_lp ? esz : 0.0)) +
                      (p.residual ? (double)p.M * p.N * 4.0 : 0.0);
        if (k < 4) { out->M[k] = p.M; out->N[k] = p.N; out->K[k] = p.K; }
    }
    return MTN_OK;
}

extern "C" int mtn_census_replay(int i, int reps, void* stream) {
    MTN_CHECK_ARG(i >= 0 && i < (int)g_census.size() && reps >= 1, "bad census index");
    const bool was = g_census_on;
    g_census_on = false;
    CensusEntry e = g_census[i];
    if (e.table >= 0) {
        CensusTable t = g_census_tables[e.table];
        for (int k = 0; k < e.count; ++k)
            if (t.p[k].adam) t.p[k].adam = &t.adam[k];
        g_census_on = false;
        int rc = MTN_OK;
        if (t.has_aux) { t.aux.ln = t.ln.data(); t.aux.chunk_off = t.chunk_off.data(); t.aux.chunk_len = t.chunk_len.data(); }
        for (int r = 0; r < reps && rc == MTN_OK; ++r) rc = mtn_gemm_tt_table_aux(e.dtype, e.count, t.p.data(), t.has_aux ? &t.aux : nullptr, stream);
        g_census_on = was;
        return rc;
    }
    for (int k = 0; k < e.count; ++k)
    {
        if (e.p[k].adam) e.p[k].adam = &e.adam[k];      // replays re-apply the optimiser epilogue (same cost; parameters drift by <= lr per replay)
        if (e.p[k].ln) e.p[k].ln = &e.ln[k];            // ... and the LayerNorm epilogue (idempotent: it only reads the partials)
    }
    int rc = MTN_OK;
    for (int r = 0; r < reps && rc == MTN_OK; ++r) rc = mtn_gemm(e.dtype, e.count, e.p, stream);
    g_census_on = was;
    return rc;
}


// ---------------------------------------------------------------- on-box MFMA peak (measurement support, bench.py)
// Register-only MFMA issue on independent accumulator chains, 8 workgroups of 4 waves per CU: the achievable dense bf16 rate of
// THIS box at its sustained clock — the measured denominator next to the 2.5 PFLOP/s spec.  Two instruction shapes: the
// 16x16x32 form every kernel of the path uses (8 chains per wave) and the 32x32x16 form (4 chains of 16 accumulators), which is
// the one /opt/skills/guides/MI355X_MICROARCH.md quotes 2 495 TFLOP/s for (half the operand bytes per FLOP).
typedef float f32x16_t __attribute__((ext_vector_type(16)));
template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_spin_kernel(float* out, int iters) {
    bf16x8_t a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (float)(threadIdx.x + i)); b[i] = (__bf16)(0.002f * (float)((int)threadIdx.x - i)); }
    if (SHAPE == 16) {
        f32x4_t c[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[j], 0, 0, 0);
        }
        f32x4_t s = c[0] + c[1] + c[2] + c[3] + c[4] + c[5] + c[6] + c[7];
        if (s[0] == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = s[1];
    } else {
        f32x16_t c[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) c[j][k] = 0.f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[j], 0, 0, 0);
        }
        f32x16_t s = c[0] + c[1] + c[2] + c[3];
        if (s[0] == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = s[1] + s[15];
    }
}

static int measure_mfma(int shape, int iters, float* scratch, hipStream_t s, double* tflops) {
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { mtn_set_error("hipEventCreate failed"); return MTN_ERR_LAUNCH; }
    const int wgs = 256 * 8;
    double best = 0.0;
    for (int rep = 0; rep < 10; ++rep) {              // the clock ramps over the first repetitions: report the best
        (void)hipEventRecord(e0, s);
        if (shape == 16) hipLaunchKernelGGL(mfma_spin_kernel<16>, dim3(wgs), dim3(256), 0, s, scratch, iters);
        else hipLaunchKernelGGL(mfma_spin_kernel<32>, dim3(wgs), dim3(256), 0, s, scratch, iters);
        (void)hipEventRecord(e1, s);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double per_iter = shape == 16 ? 8 * (2.0 * 16 * 16 * 32) : 4 * (2.0 * 32 * 32 * 16);
        const double tf = (double)wgs * 4 * iters * per_iter / (ms * 1e-3) / 1e12;
        if (tf > best) best = tf;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    MTN_CHECK_LAUNCH();
    *tflops = best;
    return MTN_OK;
}

extern "C" int mtn_measure_mfma_peak(int iters, float* scratch, void* stream, double* tflops) {
    MTN_CHECK_ARG(iters > 0 && scratch && tflops, "bad arguments");
    double a = 0.0, b = 0.0;
    int rc = measure_mfma(16, iters, scratch, (hipStream_t)stream, &a);
    if (rc == MTN_OK) rc = measure_mfma(32, iters, scratch, (hipStream_t)stream, &b);
    *tflops = a > b ? a : b;
    return rc;
}

extern "C" int mtn_measure_mfma_peak_shapes(int iters, float* scratch, void* stream, double* tflops_16x16x32, double* tflops_32x32x16) {
    MTN_CHECK_ARG(iters > 0 && scratch && tflops_16x16x32 && tflops_32x32x16, "bad arguments");
    int rc = measure_mfma(16, iters, scratch, (hipStream_t)stream, tflops_16x16x32);
    if (rc == MTN_OK) rc = measure_mfma(32, iters, scratch, (hipStream_t)stream, tflops_32x32x16);
    return rc;
}
