// layernorm.hip — MTN's LayerNorm variant (mtn.py:103-114): y = a2*(x-mean)/(std_unbiased+eps)+b2.
// HBM/L2-bound row kernels: one 64-lane wave per row, float4 loads, wave-shuffle reductions.
#include "common.h"

// ------------------------------------------------------------------------------------------ forward
// Grouped: one launch normalises the rows of up to MTN_LN_MAX_GROUP independent streams (the lockstep sublayer groups of
// a DecoderLayer: x and the two auto-encoder streams), each with its own gain/bias.
struct LnFwdGroup {
    int count;
    int block_start[MTN_LN_MAX_GROUP + 1];
    mtn_ln_fwd_desc d[MTN_LN_MAX_GROUP];
};

// value of 4 consecutive columns of a (possibly synthesised) input row
__device__ __forceinline__ float4 ln_src(const mtn_ln_fwd_desc& D, const DropState& ds, int row, int c, long tok, int pos) {
    float4 v;
    if (D.tokens) {
        v = *(const float4*)(D.lut + (size_t)tok * D.d + c);
        v.x *= D.emb_scale; v.y *= D.emb_scale; v.z *= D.emb_scale; v.w *= D.emb_scale;
    } else {
        v = *(const float4*)(D.x + (size_t)row * D.d + c);
    }
    if (D.pe) {
        const float4 p = *(const float4*)(D.pe + (size_t)pos * D.d + c);
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        if (ds.on) {
            const uint64_t idx = (uint64_t)row * D.d + c;
            v.x = drop_keep(ds, idx) ? v.x * ds.scale : 0.f;
            v.y = drop_keep(ds, idx + 1) ? v.y * ds.scale : 0.f;
            v.z = drop_keep(ds, idx + 2) ? v.z * ds.scale : 0.f;
            v.w = drop_keep(ds, idx + 3) ? v.w * ds.scale : 0.f;
        }
    }
    return v;
}

template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const LnFwdGroup grp) {
    int g = 0;
    while (g + 1 < grp.count && (int)blockIdx.x >= grp.block_start[g + 1]) ++g;
    const mtn_ln_fwd_desc& D = grp.d[g];
    const int d = D.d;
    const int lane = threadIdx.x & 63;
    const int row = ((int)blockIdx.x - grp.block_start[g]) * 4 + (threadIdx.x >> 6);
    if (row >= D.rows) return;
    const DropState ds = drop_init(D.drop);
    const long tok = D.tokens ? D.tokens[row] : 0;
    const int pos = D.pe ? row % D.seq_len : 0;
    T* y_lp = (T*)D.y_lp;
    float mean = 0.f, rstd = 1.f;
    if (!D.no_ln) {
        float s = 0.f;
        for (int c = lane * 4; c < d; c += 256) {
            float4 v = ln_src(D, ds, row, c, tok, pos);
            s += (v.x + v.y) + (v.z + v.w);
        }
        mean = wave_sum(s) / (float)d;
        float q = 0.f;
        for (int c = lane * 4; c < d; c += 256) {
            float4 v = ln_src(D, ds, row, c, tok, pos);
            float e0 = v.x - mean, e1 = v.y - mean, e2 = v.z - mean, e3 = v.w - mean;
            q += (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
        }
        const float std_u = sqrtf(wave_sum(q) / (float)(d - 1));
        rstd = 1.0f / (std_u + D.eps);
        if (lane == 0) {
            if (D.mean) D.mean[row] = mean;
            if (D.rstd) D.rstd[row] = rstd;
        }
    }
    for (int c = lane * 4; c < d; c += 256) {
        float4 v = ln_src(D, ds, row, c, tok, pos);
        if (D.x_out) *(float4*)(D.x_out + (size_t)row * d + c) = v;
        float4 o = v;
        if (!D.no_ln) {
            float4 ga = *(const float4*)(D.a2 + c);
            float4 b = *(const float4*)(D.b2 + c);
            o.x = ga.x * (v.x - mean) * rstd + b.x;
            o.y = ga.y * (v.y - mean) * rstd + b.y;
            o.z = ga.z * (v.z - mean) * rstd + b.z;
            o.w = ga.w * (v.w - mean) * rstd + b.w;
        }
        if (D.y_f32) *(float4*)(D.y_f32 + (size_t)row * d + c) = o;
        if (y_lp) store_lp4<T>(y_lp + (size_t)row * d + c, o);
    }
}

// d <= 256*MAXV: the (possibly synthesised) row is read ONCE into registers; mean, variance and output come from there
// (the generic kernel above re-reads the row for each of its three passes: three dependent memory round trips).
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void ln_fwd_small_kernel(const LnFwdGroup grp) {
    int g = 0;
    while (g + 1 < grp.count && (int)blockIdx.x >= grp.block_start[g + 1]) ++g;
    const mtn_ln_fwd_desc& D = grp.d[g];
    const int d = D.d;
    const int lane = threadIdx.x & 63;
    const int row = ((int)blockIdx.x - grp.block_start[g]) * 4 + (threadIdx.x >> 6);
    if (row >= D.rows) return;
    const DropState ds = drop_init(D.drop);
    const long tok = D.tokens ? D.tokens[row] : 0;
    const int pos = D.pe ? row % D.seq_len : 0;
    float4 v[MAXV], ga[MAXV], gb[MAXV];
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c = lane * 4 + 256 * j;
        if (c < d) {
            v[j] = ln_src(D, ds, row, c, tok, pos);
            if (!D.no_ln) { ga[j] = *(const float4*)(D.a2 + c); gb[j] = *(const float4*)(D.b2 + c); }
        }
    }
    float mean = 0.f, rstd = 1.f;
    if (!D.no_ln) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < MAXV; ++j)
            if (lane * 4 + 256 * j < d) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        mean = wave_sum(s) / (float)d;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < MAXV; ++j)
            if (lane * 4 + 256 * j < d) {
                const float e0 = v[j].x - mean, e1 = v[j].y - mean, e2 = v[j].z - mean, e3 = v[j].w - mean;
                q += (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
            }
        const float std_u = sqrtf(wave_sum(q) / (float)(d - 1));
        rstd = 1.0f / (std_u + D.eps);
        if (lane == 0) {
            if (D.mean) D.mean[row] = mean;
            if (D.rstd) D.rstd[row] = rstd;
        }
    }
    T* y_lp = (T*)D.y_lp;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c = lane * 4 + 256 * j;
        if (c < d) {
            if (D.x_out) *(float4*)(D.x_out + (size_t)row * d + c) = v[j];
            float4 o = v[j];
            if (!D.no_ln) {
                o.x = ga[j].x * (v[j].x - mean) * rstd + gb[j].x;
                o.y = ga[j].y * (v[j].y - mean) * rstd + gb[j].y;
                o.z = ga[j].z * (v[j].z - mean) * rstd + gb[j].z;
                o.w = ga[j].w * (v[j].w - mean) * rstd + gb[j].w;
            }
            if (D.y_f32) *(float4*)(D.y_f32 + (size_t)row * d + c) = o;
            if (y_lp) store_lp4<T>(y_lp + (size_t)row * d + c, o);
        }
    }
}

extern "C" int mtn_layernorm_fwd_group(int dtype, int count, const mtn_ln_fwd_desc* descs, void* stream) {
    MTN_CHECK_ARG(dtype == MTN_F32 || dtype == MTN_BF16, "bad dtype");
    MTN_CHECK_ARG(count >= 1 && count <= MTN_LN_MAX_GROUP && descs, "bad group");
    LnFwdGroup grp;
    memset(&grp, 0, sizeof(grp));
    grp.count = count;
    int blocks = 0;
    for (int i = 0; i < count; ++i) {
        const mtn_ln_fwd_desc& D = descs[i];
        MTN_CHECK_ARG(D.rows > 0 && D.d >= 4 && D.d % 4 == 0, "rows>0 and d%4==0 required");
        MTN_CHECK_ARG((D.x || (D.tokens && D.lut)) && (D.no_ln || (D.a2 && D.b2)), "null input");
        MTN_CHECK_ARG(!D.pe || D.seq_len > 0, "positional encoding needs seq_len");
        grp.block_start[i] = blocks;
        blocks += (D.rows + 3) / 4;
        grp.d[i] = D;
    }
    for (int i = count; i <= MTN_LN_MAX_GROUP; ++i) grp.block_start[i] = blocks;
    hipStream_t s = (hipStream_t)stream;
    int dmax = 0;
    for (int i = 0; i < count; ++i) dmax = descs[i].d > dmax ? descs[i].d : dmax;
    if (dmax <= 512) {
        if (dtype == MTN_BF16) hipLaunchKernelGGL((ln_fwd_small_kernel<bf16_t, 2>), dim3(blocks), dim3(256), 0, s, grp);
        else hipLaunchKernelGGL((ln_fwd_small_kernel<float, 2>), dim3(blocks), dim3(256), 0, s, grp);
    } else if (dtype == MTN_BF16) hipLaunchKernelGGL((ln_fwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, grp);
    else hipLaunchKernelGGL((ln_fwd_kernel<float>), dim3(blocks), dim3(256), 0, s, grp);
    MTN_CHECK_LAUNCH();
    return MTN_OK;
}

extern "C" int mtn_layernorm_fwd(int dtype, int rows, int d, float eps, const float* x, const float* a2, const float* b2,
                                 float* y_f32, void* y_lp, float* mean, float* rstd, void* stream) {
    mtn_ln_fwd_desc D;
    memset(&D, 0, sizeof(D));
    D.rows = rows; D.d = d; D.eps = eps; D.x = x; D.a2 = a2; D.b2 = b2; D.y_f32 = y_f32; D.y_lp = y_lp; D.mean = mean; D.rstd = rstd;
    return mtn_layernorm_fwd_group(dtype, 1, &D, stream);
}

// ------------------------------------------------------------------------------------------ backward
// With h_i = g_i*a2_i, xc_i = x_i-mean, r = 1/(std+eps), s1 = sum h, s2 = sum h*xc:
//   dx_i = r*h_i - r*s1/d - s2*r^2/(std*(d-1)) * xc_i      (+ dres_i)
//   da2_i = sum_rows g_i*xc_i*r        db2_i = sum_rows g_i
// dx is on the critical path of backward: 2 rows per wave, 8 rows per 256-thread workgroup, so a 640-row stream
// fills 80 workgroups.  Parameter gradients are off the critical path: each workgroup reduces its rows through
// LDS and writes ONE partial row [2d]; a grouped finalize kernel (many LayerNorms per launch, fixed summation
// order => deterministic, no atomics) turns partials into da2/db2 — see mtn_layernorm_bwd_finalize().
static constexpr int LN_BWD_ROWS_PER_WAVE = 2;
static constexpr int LN_BWD_ROWS_PER_BLOCK = 4 * LN_BWD_ROWS_PER_WAVE;
static constexpr int LN_MAXV = 8;  // float4 per lane -> d <= 2048

struct LnBwdGroup {
    int count;
    int block_start[MTN_LN_MAX_GROUP + 1];
    mtn_ln_bwd_desc d[MTN_LN_MAX_GROUP];
};

__global__ __launch_bounds__(256) void ln_bwd_kernel(const LnBwdGroup grp) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [4 waves][2][d]
    int gi = 0;
    while (gi + 1 < grp.count && (int)blockIdx.x >= grp.block_start[gi + 1]) ++gi;
    const mtn_ln_bwd_desc& D = grp.d[gi];
    const int rows = D.rows, d = D.d;
    const float eps = D.eps;
    const float* __restrict__ x = D.x;
    const float* __restrict__ a2 = D.a2;
    const float* __restrict__ mean = D.mean;
    const float* __restrict__ rstd = D.rstd;
    const float* __restrict__ g = D.g;
    const float* __restrict__ dres = D.dres;
    float* __restrict__ dx = D.dx;
    float* __restrict__ partial = D.partial;
    const DropState nds = drop_init(D.dx_lp_drop);          // optional: dx again, as the next backward stage consumes it
    const int blk = (int)blockIdx.x - grp.block_start[gi];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blk * LN_BWD_ROWS_PER_BLOCK + wave * LN_BWD_ROWS_PER_WAVE;
    float4 ga[LN_MAXV], gb[LN_MAXV];
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) ga[j] = gb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float inv_d = 1.0f / (float)d;
#pragma unroll
    for (int rr = 0; rr < LN_BWD_ROWS_PER_WAVE; ++rr) {
        const int row = r0 + rr;
        if (row >= rows) break;
        const float mu = mean[row], r = rstd[row];
        const float std_u = fmaxf(1.0f / r - eps, 1e-30f);
        const float* xr = x + (size_t)row * d;
        const float* gr = g + (size_t)row * d;
        float4 hv[LN_MAXV], ev[LN_MAXV];
        float s1 = 0.f, s2 = 0.f;
        unsigned keep = 0xffffffffu;               // hand-off dropout mask of this lane's elements: integer work that does not
        if (D.dx_lp && nds.on) {                   // depend on the data, placed ahead of the loads' first use
            keep = 0u;
#pragma unroll
            for (int j = 0; j < LN_MAXV; ++j) {
                const int c = lane * 4 + 256 * j;
                if (c < d) {
                    const size_t e = (size_t)row * d + c;
#pragma unroll
                    for (int k = 0; k < 4; ++k) keep |= (drop_keep(nds, e + k) ? 1u : 0u) << (j * 4 + k);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < LN_MAXV; ++j) {
            const int c = lane * 4 + 256 * j;
            if (c < d) {
                float4 xv = *(const float4*)(xr + c), gv = *(const float4*)(gr + c), av = *(const float4*)(a2 + c);
                float4 h = make_float4(gv.x * av.x, gv.y * av.y, gv.z * av.z, gv.w * av.w);
                float4 e = make_float4(xv.x - mu, xv.y - mu, xv.z - mu, xv.w - mu);
                s1 += (h.x + h.y) + (h.z + h.w);
                s2 += (h.x * e.x + h.y * e.y) + (h.z * e.z + h.w * e.w);
                ga[j].x += gv.x * e.x * r; ga[j].y += gv.y * e.y * r; ga[j].z += gv.z * e.z * r; ga[j].w += gv.w * e.w * r;
                gb[j].x += gv.x; gb[j].y += gv.y; gb[j].z += gv.z; gb[j].w += gv.w;
                hv[j] = h; ev[j] = e;
            }
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        const float c1 = r * s1 * inv_d;
        const float c2 = s2 * r * r / (std_u * (float)(d - 1));
#pragma unroll
        for (int j = 0; j < LN_MAXV; ++j) {
            const int c = lane * 4 + 256 * j;
            if (c < d) {
                float4 o;
                o.x = r * hv[j].x - c1 - c2 * ev[j].x;
                o.y = r * hv[j].y - c1 - c2 * ev[j].y;
                o.z = r * hv[j].z - c1 - c2 * ev[j].z;
                o.w = r * hv[j].w - c1 - c2 * ev[j].w;
                if (dres) {
                    float4 dv = *(const float4*)(dres + (size_t)row * d + c);
                    o.x += dv.x; o.y += dv.y; o.z += dv.z; o.w += dv.w;
                }
                *(float4*)(dx + (size_t)row * d + c) = o;
                if (D.dx_lp) {
                    const size_t e = (size_t)row * d + c;
                    if (nds.on) {
                        const unsigned kb = keep >> (j * 4);
                        o.x = (kb & 1u) ? o.x * nds.scale : 0.f;
                        o.y = (kb & 2u) ? o.y * nds.scale : 0.f;
                        o.z = (kb & 4u) ? o.z * nds.scale : 0.f;
                        o.w = (kb & 8u) ? o.w * nds.scale : 0.f;
                    }
                    if (D.dx_lp_dtype == MTN_BF16) store_lp4<bf16_t>((bf16_t*)D.dx_lp + e, o);
                    else store_lp4<float>((float*)D.dx_lp + e, o);
                }
            }
        }
    }
    if (partial == nullptr) return;
    float* sa = sm + (size_t)wave * 2 * d;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        const int c = lane * 4 + 256 * j;
        if (c < d) {
            *(float4*)(sa + c) = ga[j];
            *(float4*)(sa + d + c) = gb[j];
        }
    }
    __syncthreads();
    float* pp = partial + (size_t)blk * 2 * d;
    for (int c = threadIdx.x * 4; c < 2 * d; c += 1024) {
        float4 a = *(const float4*)(sm + c), b = *(const float4*)(sm + 2 * d + c);
        float4 e = *(const float4*)(sm + 4 * d + c), f = *(const float4*)(sm + 6 * d + c);
        *(float4*)(pp + c) = make_float4((a.x + b.x) + (e.x + f.x), (a.y + b.y) + (e.y + f.y), (a.z + b.z) + (e.z + f.z), (a.w + b.w) + (e.w + f.w));
    }
}

// The same kernel for d <= 256*MAXV with ALL global loads of the wave's two rows issued before any arithmetic (the generic
// kernel above walks the rows one after the other: two dependent memory round trips on the critical path of backward).
// NWV = 8 (round 3): eight waves of ONE row each instead of four of two — a wave gets one load through every ~100-180 ns however many
// it has queued, so the 14 loads of a two-row wave were 1.4-2.5 us of issue on a 6.6 us launch.
template <int MAXV, int NWV = 8>
__global__ __launch_bounds__(64 * NWV) void ln_bwd_small_kernel(const LnBwdGroup grp) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [NWV waves][2][d]
    int gi = 0;
    while (gi + 1 < grp.count && (int)blockIdx.x >= grp.block_start[gi + 1]) ++gi;
    const mtn_ln_bwd_desc& D = grp.d[gi];
    const int rows = D.rows, d = D.d;
    const float eps = D.eps;
    const int blk = (int)blockIdx.x - grp.block_start[gi];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int R = LN_BWD_ROWS_PER_BLOCK / NWV;
    const int r0 = blk * LN_BWD_ROWS_PER_BLOCK + wave * R;
    float4 xv[R][MAXV], gv[R][MAXV], dv[R][MAXV], av[MAXV];
    float mu[R], rs[R];
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
        const int row = r0 + rr < rows ? r0 + rr : rows - 1;           // clamped: rows past the end are computed, not stored
        mu[rr] = D.mean[row]; rs[rr] = D.rstd[row];
#pragma unroll
        for (int j = 0; j < MAXV; ++j) {
            const int c = lane * 4 + 256 * j;
            if (c < d) {
                xv[rr][j] = *(const float4*)(D.x + (size_t)row * d + c);
                gv[rr][j] = *(const float4*)(D.g + (size_t)row * d + c);
                dv[rr][j] = D.dres ? *(const float4*)(D.dres + (size_t)row * d + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c = lane * 4 + 256 * j;
        if (c < d) av[j] = *(const float4*)(D.a2 + c);
    }
    const DropState nds = drop_init(D.dx_lp_drop);
    unsigned keep[R];
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
        keep[rr] = 0xffffffffu;
        if (D.dx_lp && nds.on) {
            keep[rr] = 0u;
#pragma unroll
            for (int j = 0; j < MAXV; ++j) {
                const int c = lane * 4 + 256 * j;
                if (c < d) {
                    const size_t e = (size_t)(r0 + rr) * d + c;
#pragma unroll
                    for (int k = 0; k < 4; ++k) keep[rr] |= (drop_keep(nds, e + k) ? 1u : 0u) << (j * 4 + k);
                }
            }
        }
    }
    float4 ga[MAXV], gb[MAXV];
#pragma unroll
    for (int j = 0; j < MAXV; ++j) ga[j] = gb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float inv_d = 1.0f / (float)d;
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
        const int row = r0 + rr;
        const bool live = row < rows;
        const float r = rs[rr];
        const float std_u = fmaxf(1.0f / r - eps, 1e-30f);
        float4 hv[MAXV], ev[MAXV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < MAXV; ++j) {
            const int c = lane * 4 + 256 * j;
            if (c < d) {
                const float4 x4 = xv[rr][j], g4 = gv[rr][j], a4 = av[j];
                float4 h = make_float4(g4.x * a4.x, g4.y * a4.y, g4.z * a4.z, g4.w * a4.w);
                float4 e = make_float4(x4.x - mu[rr], x4.y - mu[rr], x4.z - mu[rr], x4.w - mu[rr]);
                s1 += (h.x + h.y) + (h.z + h.w);
                s2 += (h.x * e.x + h.y * e.y) + (h.z * e.z + h.w * e.w);
                if (live) {
                    ga[j].x += g4.x * e.x * r; ga[j].y += g4.y * e.y * r; ga[j].z += g4.z * e.z * r; ga[j].w += g4.w * e.w * r;
                    gb[j].x += g4.x; gb[j].y += g4.y; gb[j].z += g4.z; gb[j].w += g4.w;
                }
                hv[j] = h; ev[j] = e;
            }
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        const float c1 = r * s1 * inv_d;
        const float c2 = s2 * r * r / (std_u * (float)(d - 1));
        if (!live) continue;
#pragma unroll
        for (int j = 0; j < MAXV; ++j) {
            const int c = lane * 4 + 256 * j;
            if (c < d) {
                float4 o;
                o.x = r * hv[j].x - c1 - c2 * ev[j].x + dv[rr][j].x;
                o.y = r * hv[j].y - c1 - c2 * ev[j].y + dv[rr][j].y;
                o.z = r * hv[j].z - c1 - c2 * ev[j].z + dv[rr][j].z;
                o.w = r * hv[j].w - c1 - c2 * ev[j].w + dv[rr][j].w;
                *(float4*)(D.dx + (size_t)row * d + c) = o;
                if (D.dx_lp) {
                    const size_t e = (size_t)row * d + c;
                    if (nds.on) {
                        const unsigned kb = keep[rr] >> (j * 4);
                        o.x = (kb & 1u) ? o.x * nds.scale : 0.f;
                        o.y = (kb & 2u) ? o.y * nds.scale : 0.f;
                        o.z = (kb & 4u) ? o.z * nds.scale : 0.f;
                        o.w = (kb & 8u) ? o.w * nds.scale : 0.f;
                    }
                    if (D.dx_lp_dtype == MTN_BF16) store_lp4<bf16_t>((bf16_t*)D.dx_lp + e, o);
                    else store_lp4<float>((float*)D.dx_lp + e, o);
                }
            }
        }
    }
    if (D.partial == nullptr) return;
    float* sa = sm + (size_t)wave * 2 * d;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c = lane * 4 + 256 * j;
        if (c < d) {
            *(float4*)(sa + c) = ga[j];
            *(float4*)(sa + d + c) = gb[j];
        }
    }
    __syncthreads();
    float* pp = D.partial + (size_t)blk * 2 * d;
    for (int c = threadIdx.x * 4; c < 2 * d; c += 256 * NWV) {
        float4 a = *(const float4*)(sm + c), b = *(const float4*)(sm + 2 * d + c);
        float4 e = *(const float4*)(sm + 4 * d + c), f = *(const float4*)(sm + 6 * d + c);
        float4 t = make_float4((a.x + b.x) + (e.x + f.x), (a.y + b.y) + (e.y + f.y), (a.z + b.z) + (e.z + f.z), (a.w + b.w) + (e.w + f.w));
        if constexpr (NWV == 8) {
            a = *(const float4*)(sm + 8 * d + c); b = *(const float4*)(sm + 10 * d + c);
            e = *(const float4*)(sm + 12 * d + c); f = *(const float4*)(sm + 14 * d + c);
            t.x += (a.x + b.x) + (e.x + f.x); t.y += (a.y + b.y) + (e.y + f.y); t.z += (a.z + b.z) + (e.z + f.z); t.w += (a.w + b.w) + (e.w + f.w);
        }
        *(float4*)(pp + c) = t;
    }
}

// Grouped finalize: blockIdx.y = LayerNorm index, blockIdx.x covers the 2d columns (first d -> da2, next d -> db2).
struct LnFinalizeGroup {
    int count;
    mtn_ln_finalize_desc d[MTN_LN_FINALIZE_MAX_GROUP];
};
__global__ __launch_bounds__(1024) void ln_bwd_finalize_kernel(const LnFinalizeGroup grp) {
    // 64 columns per workgroup; the partial rows are dealt to 16 thread groups (fixed assignment and summation order:
    // deterministic), combined through LDS.  (A 4096-row stream leaves 512 partial rows: with 4 groups each thread walked 128 of
    // them in dependent batches — 14 us per launch; 16 groups cut the chain to a quarter.)
    __shared__ float red[16][64];
    const mtn_ln_finalize_desc& D = grp.d[blockIdx.y];
    const int d = D.d;
    const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (c < 2 * d) {
        const float* p = D.partial + c;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int i = q;
        for (; i + 112 < D.nparts; i += 128) {                   // eight independent loads per round (long streams: 512 partial rows)
            const float a0 = p[(size_t)(i + 0) * 2 * d], a1 = p[(size_t)(i + 16) * 2 * d], a2 = p[(size_t)(i + 32) * 2 * d], a3 = p[(size_t)(i + 48) * 2 * d];
            const float a4 = p[(size_t)(i + 64) * 2 * d], a5 = p[(size_t)(i + 80) * 2 * d], a6 = p[(size_t)(i + 96) * 2 * d], a7 = p[(size_t)(i + 112) * 2 * d];
            s0 += a0; s1 += a1; s2 += a2; s3 += a3; s0 += a4; s1 += a5; s2 += a6; s3 += a7;
        }
        for (; i + 48 < D.nparts; i += 64) {
            s0 += p[(size_t)(i + 0) * 2 * d]; s1 += p[(size_t)(i + 16) * 2 * d];
            s2 += p[(size_t)(i + 32) * 2 * d]; s3 += p[(size_t)(i + 48) * 2 * d];
        }
        for (; i < D.nparts; i += 16) s0 += p[(size_t)i * 2 * d];
        s = (s0 + s1) + (s2 + s3);
    }
    red[q][cl] = s;
    __syncthreads();
    if (q == 0 && c < 2 * d) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k += 4) t += (red[k][cl] + red[k + 1][cl]) + (red[k + 2][cl] + red[k + 3][cl]);
        if (c < d) { if (D.da2) D.da2[c] = t; }
        else if (D.db2) D.db2[c - d] = t;
    }
}

static inline int ln_bwd_blocks(int rows) { return (rows + LN_BWD_ROWS_PER_BLOCK - 1) / LN_BWD_ROWS_PER_BLOCK; }

extern "C" long mtn_layernorm_bwd_partial_floats(int rows, int d) { return (long)ln_bwd_blocks(rows) * 2 * d; }
extern "C" int mtn_layernorm_bwd_nparts(int rows) { return ln_bwd_blocks(rows); }

extern "C" int mtn_layernorm_bwd_finalize(int count, const mtn_ln_finalize_desc* descs, void* stream) {
    MTN_CHECK_ARG(count >= 1 && descs, "bad arguments");
    hipStream_t s = (hipStream_t)stream;
    for (int base = 0; base < count; base += MTN_LN_FINALIZE_MAX_GROUP) {
        LnFinalizeGroup grp;
        grp.count = count - base < MTN_LN_FINALIZE_MAX_GROUP ? count - base : MTN_LN_FINALIZE_MAX_GROUP;
        int dmax = 0;
        for (int i = 0; i < grp.count; ++i) {
            grp.d[i] = descs[base + i];
            MTN_CHECK_ARG(grp.d[i].partial && grp.d[i].nparts > 0 && grp.d[i].d > 0, "bad finalize descriptor");
            if (grp.d[i].d > dmax) dmax = grp.d[i].d;
        }
        hipLaunchKernelGGL(ln_bwd_finalize_kernel, dim3((2 * dmax + 63) / 64, grp.count), dim3(1024), 0, s, grp);
        MTN_CHECK_LAUNCH();
    }
    return MTN_OK;
}

extern "C" int mtn_layernorm_bwd_group(int count, const mtn_ln_bwd_desc* descs, void* stream) {
    MTN_CHECK_ARG(count >= 1 && count <= MTN_LN_MAX_GROUP && descs, "bad group");
    LnBwdGroup grp;
    memset(&grp, 0, sizeof(grp));
    grp.count = count;
    int blocks = 0, dmax = 0;
    bool any_partial = false;
    for (int i = 0; i < count; ++i) {
        const mtn_ln_bwd_desc& D = descs[i];
        MTN_CHECK_ARG(D.rows > 0 && D.d >= 4 && D.d % 4 == 0 && D.d <= 256 * LN_MAXV, "rows>0, d%4==0, d<=2048 required");
        MTN_CHECK_ARG(D.x && D.a2 && D.mean && D.rstd && D.g && D.dx, "null input");
        grp.block_start[i] = blocks;
        blocks += ln_bwd_blocks(D.rows);
        grp.d[i] = D;
        if (D.d > dmax) dmax = D.d;
        any_partial |= (D.partial != nullptr);
    }
    for (int i = count; i <= MTN_LN_MAX_GROUP; ++i) grp.block_start[i] = blocks;
    const size_t lds = any_partial ? sizeof(float) * 8 * (size_t)dmax : 0;
    if (dmax <= 512) hipLaunchKernelGGL((ln_bwd_small_kernel<2, 8>), dim3(blocks), dim3(512), 2 * lds, (hipStream_t)stream, grp);
    else hipLaunchKernelGGL(ln_bwd_kernel, dim3(blocks), dim3(256), lds, (hipStream_t)stream, grp);
    MTN_CHECK_LAUNCH();
    return MTN_OK;
}

extern "C" int mtn_layernorm_bwd(int rows, int d, float eps, const float* x, const float* a2, const float* mean,
                                 const float* rstd, const float* g, const float* dres, float* dx, float* da2, float* db2,
                                 float* partial, void* stream) {
    MTN_CHECK_ARG(partial || (!da2 && !db2), "parameter gradients need the partial scratch buffer");
    mtn_ln_bwd_desc D = {rows, d, eps, x, a2, mean, rstd, g, dres, dx, partial};
    if (int rc = mtn_layernorm_bwd_group(1, &D, stream)) return rc;
    if (da2 || db2) {   // immediate finalize; callers that batch many LayerNorms pass NULL here and call mtn_layernorm_bwd_finalize later
        mtn_ln_finalize_desc F = {partial, ln_bwd_blocks(rows), d, da2, db2};
        return mtn_layernorm_bwd_finalize(1, &F, stream);
    }
    return MTN_OK;
}

// ------------------------------------------------------------------------------------------ fold vectors (mtn_ln_fold)
// u[k] = sum_c W[k][c] a2[c],  c[k] = bias[k] + sum_c W[k][c] b2[c]  for every Linear W [K, d] that follows a LayerNorm (a2, b2):
// what lets LayerNorm backward ride in the epilogue of g = dq W (mtn_ln_epilogue, include/mtn_hip.h).  Weights change every step,
// so this runs once per step: one launch for the whole model (~88 MB of bf16 weights at BASELINE configs[1]), a wave per row,
// 64 rows per 512-thread workgroup, the row as ONE 16-byte load per lane per 512 columns.
template <int NJ>      // 512-column chunks of a row a lane may hold (1: d <= 512)
__global__ __launch_bounds__(512) void ln_fold_kernel(const mtn_ln_fold_desc* __restrict__ descs, const int* __restrict__ block_desc, const int d) {
    // 64 rows per 512-thread workgroup (8 waves x 8 rows, all eight row loads of a wave in flight at once)
    // (round 5: the step's other head launches — zero fill of the glue gradients, schedule tick, dropout seed — were merged into this
    //  launch as extra workgroups and measured SLOWER, +10-13 us per step in either dispatch order: profiles/r05_d_tail_head_ab.txt)
    const int fb = (int)blockIdx.x;
    const mtn_ln_fold_desc D = descs[block_desc[fb]];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k0 = (fb - D.block_start) * 64 + wave * 8;
    if (k0 >= D.K) return;
    const bf16_t* w = (const bf16_t*)D.w;
    uint4 wv[8][NJ];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int k = k0 + r < D.K ? k0 + r : D.K - 1;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = (lane + 64 * j) * 8;
            wv[r][j] = c < d ? *(const uint4*)(w + (size_t)k * d + c) : make_uint4(0, 0, 0, 0);
        }
    }
    float su[8], sc[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) su[r] = sc[r] = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = (lane + 64 * j) * 8;
        if (c >= d) break;
        const float4 a0 = *(const float4*)(D.a2 + c), a1 = *(const float4*)(D.a2 + c + 4);
        const float4 b0 = *(const float4*)(D.b2 + c), b1 = *(const float4*)(D.b2 + c + 4);
        const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const uint32_t q[4] = {wv[r][j].x, wv[r][j].y, wv[r][j].z, wv[r][j].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lo = __uint_as_float(q[e] << 16), hi = __uint_as_float(q[e] & 0xffff0000u);
                su[r] += lo * av[2 * e] + hi * av[2 * e + 1];
                sc[r] += lo * bv[2 * e] + hi * bv[2 * e + 1];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const float u = fh_cross_sum(fh_row16_sum(su[r])), c = fh_cross_sum(fh_row16_sum(sc[r]));      // (DPP + lane swaps: no LDS traffic)
        if (lane == 0 && k0 + r < D.K) {
            D.out[k0 + r] = u;
            D.out[D.K + k0 + r] = c + (D.bias ? D.bias[k0 + r] : 0.f);
        }
    }
}
extern "C" int mtn_ln_fold(int dtype, const mtn_ln_fold_desc* descs_device, const int* block_desc, int total_blocks, int d, void* stream) {
    MTN_CHECK_ARG(dtype == MTN_BF16, "fold vectors exist for the bf16 path only");
    MTN_CHECK_ARG(descs_device && block_desc && total_blocks > 0, "null descriptor table");
    MTN_CHECK_ARG(d > 0 && d % 8 == 0 && d <= 2048, "d must be a multiple of 8, at most 2048");
    if (d <= 512) hipLaunchKernelGGL(ln_fold_kernel<1>, dim3(total_blocks), dim3(512), 0, (hipStream_t)stream, descs_device, block_desc, d);
    else hipLaunchKernelGGL(ln_fold_kernel<4>, dim3(total_blocks), dim3(512), 0, (hipStream_t)stream, descs_device, block_desc, d);
    MTN_CHECK_LAUNCH();
    return MTN_OK;
}

// ------------------------------------------------------------------------------------------ embedding backward
struct EmbedBwdGroup {
    int count;
    int block_start[MTN_LN_MAX_GROUP + 1];
    mtn_embed_bwd_desc d[MTN_LN_MAX_GROUP];
};
__global__ __launch_bounds__(256) void embed_bwd_kernel(const EmbedBwdGroup grp) {
    int g = 0;
    while (g + 1 < grp.count && (int)blockIdx.x >= grp.block_start[g + 1]) ++g;
    const mtn_embed_bwd_desc& D = grp.d[g];
    const int lane = threadIdx.x & 63;
    const int row = ((int)blockIdx.x - grp.block_start[g]) * 4 + (threadIdx.x >> 6);
    if (row >= D.rows) return;
    const DropState ds = drop_init(D.drop);
    const long tok = D.tokens[row];
    float* dst = D.dlut + (size_t)tok * D.d;
    const float* src = D.dx + (size_t)row * D.d;
    for (int c = lane; c < D.d; c += 64) {
        float v = src[c] * D.emb_scale;
        if (ds.on) v = drop_keep(ds, (uint64_t)row * D.d + c) ? v * ds.scale : 0.f;
        atomicAdd(dst + c, v);
    }
}
// Deterministic variant (no atomics, the same bits on every run).  A workgroup of EMB_W = 8 waves owns 8 vocabulary entries of one
// table, one per wave.  The token lists of all streams that use the table — one virtual list, stream order then row order —
// are staged through LDS in chunks (coalesced, once per workgroup and pass) and scanned with 64-token ballots:
//   pass 1  each wave counts the occurrences of its entry and, while there are at most EMB_HEAVY of them, adds those rows of dx
//           on the spot, in list order, through the stream's scale and dropout mask (up to eight rows' loads in flight); an
//           entry that stays at or below EMB_HEAVY writes dlut[v] += sum, once (otherwise the partial sum is dropped);
//   pass 2  a frequent entry (the pad id of a ragged batch, '.', '?', ...) would serialise thousands of rows in one wave, so
//           all 8 waves take it together: wave w scans ballots w, w+8, ... of every chunk, the 8 partial sums meet in LDS
//           and are added in wave order.
// Which wave adds which row depends only on the token values and shapes, so the rounding is the same on every run.
#define EMB_CH 8192                        // tokens per staged chunk (int32 in LDS)
#define EMB_W 8                            // waves (= vocabulary entries) per workgroup: 512 threads keep 256 VGPRs per lane (1 024 spilled), ~50 KB of LDS: three workgroups per CU
#define EMB_HEAVY 24
#define EMB_MLP 6
struct EmbedDetGroup {
    int n_lut;
    float* dlut[MTN_LN_MAX_GROUP];
    int V[MTN_LN_MAX_GROUP], d[MTN_LN_MAX_GROUP], n[MTN_LN_MAX_GROUP], total[MTN_LN_MAX_GROUP];
    int idx[MTN_LN_MAX_GROUP][MTN_LN_MAX_GROUP];
    int first[MTN_LN_MAX_GROUP][MTN_LN_MAX_GROUP + 1];      // first virtual token of each stream of the table
    mtn_embed_bwd_desc s[MTN_LN_MAX_GROUP];
};
struct EmbStream {                          // one stream of the table, as the workgroup keeps it in LDS
    const long* tokens;
    const float* dx;
    float emb_scale;
    int first;
    DropState ds;
};
// add the rows of dx whose virtual token indices are the set bits of m (offset base) to acc, lowest index first
__device__ __forceinline__ void emb_add_rows(unsigned long long m, int base, const EmbStream* st, int ns, int d, int c0, int lane,
                                             float (&acc)[8]) {
    while (m) {
        int g[EMB_MLP], n = 0;
#pragma unroll
        for (int k = 0; k < EMB_MLP; ++k)
            if (m) { g[k] = base + __builtin_ctzll(m); m &= m - 1; n = k + 1; }
        float x[EMB_MLP][8];
        int row[EMB_MLP], str[EMB_MLP];
#pragma unroll
        for (int k = 0; k < EMB_MLP; ++k) {
            if (k < n) {
                int si = 0;
                while (si + 1 < ns && g[k] >= st[si + 1].first) ++si;
                str[k] = si;
                row[k] = g[k] - st[si].first;
                const float* src = st[si].dx + (size_t)row[k] * d + c0;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int c = lane * 4 + 256 * j;
                    if (c0 + c + 3 < d) {
                        const float4 q = *reinterpret_cast<const float4*>(src + c);
                        x[k][4 * j] = q.x; x[k][4 * j + 1] = q.y; x[k][4 * j + 2] = q.z; x[k][4 * j + 3] = q.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[k][4 * j + e] = 0.f;
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < EMB_MLP; ++k) {
            if (k < n) {
                const EmbStream& S = st[str[k]];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = c0 + lane * 4 + 256 * (j >> 2) + (j & 3);
                    float y = x[k][j] * S.emb_scale;
                    if (S.ds.on) y = (c < d && drop_keep(S.ds, (uint64_t)row[k] * d + c)) ? y * S.ds.scale : 0.f;
                    acc[j] += y;
                }
            }
        }
    }
}
// the same for an explicit list of n virtual token indices (ascending): up to EMB_MLP rows in flight per batch, added in list order.
// Round 4: an entry's occurrences used to be added ballot by ballot while scanning — one dependent global round trip per ballot
// that had a hit, ~10 of them for the most frequent entry of a uniform batch (40+ us per launch in the step); collecting the hits
// first makes that ceil(n / EMB_MLP) round trips.
__device__ __forceinline__ void emb_add_list(const int* list, const int n_all, const EmbStream* st, int ns, int d, int c0, int lane, float (&acc)[8]) {
    for (int i0 = 0; i0 < n_all; i0 += EMB_MLP) {
        const int n = n_all - i0 < EMB_MLP ? n_all - i0 : EMB_MLP;
        float x[EMB_MLP][8];
        int row[EMB_MLP], str[EMB_MLP];
#pragma unroll
        for (int k = 0; k < EMB_MLP; ++k) {
            if (k < n) {
                const int g = list[i0 + k];
                int si = 0;
                while (si + 1 < ns && g >= st[si + 1].first) ++si;
                str[k] = si;
                row[k] = g - st[si].first;
                const float* src = st[si].dx + (size_t)row[k] * d + c0;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int c = lane * 4 + 256 * j;
                    if (c0 + c + 3 < d) {
                        const float4 q = *reinterpret_cast<const float4*>(src + c);
                        x[k][4 * j] = q.x; x[k][4 * j + 1] = q.y; x[k][4 * j + 2] = q.z; x[k][4 * j + 3] = q.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[k][4 * j + e] = 0.f;
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < EMB_MLP; ++k) {
            if (k < n) {
                const EmbStream& S = st[str[k]];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = c0 + lane * 4 + 256 * (j >> 2) + (j & 3);
                    float y = x[k][j] * S.emb_scale;
                    if (S.ds.on) y = (c < d && drop_keep(S.ds, (uint64_t)row[k] * d + c)) ? y * S.ds.scale : 0.f;
                    acc[j] += y;
                }
            }
        }
    }
}
__global__ __launch_bounds__(64 * EMB_W, 4) void embed_bwd_det_kernel(const EmbedDetGroup grp) {
    __shared__ int tok[EMB_CH];
    __shared__ int hits[EMB_W][EMB_HEAVY];
    __shared__ float part_sum[EMB_W][512];
    __shared__ EmbStream st[MTN_LN_MAX_GROUP];
    __shared__ int count[EMB_W];
    const int t = blockIdx.y;
    if (blockIdx.x * EMB_W >= grp.V[t]) return;               // table shorter than the longest one in the launch (uniform)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int v = blockIdx.x * EMB_W + wave;
    const bool live = v < grp.V[t];
    const int d = grp.d[t], total = grp.total[t], ns = grp.n[t];
    if (threadIdx.x < MTN_LN_MAX_GROUP) {
#pragma unroll
        for (int i = 0; i < MTN_LN_MAX_GROUP; ++i)           // static indices into the kernel argument
            if ((int)threadIdx.x == i && i < ns) {
                int which = 0;
#pragma unroll
                for (int q = 0; q < MTN_LN_MAX_GROUP; ++q)
                    if (q == t) which = grp.idx[q][i];
#pragma unroll
                for (int q = 0; q < MTN_LN_MAX_GROUP; ++q)
                    if (q == which) {
                        st[i].tokens = grp.s[q].tokens; st[i].dx = grp.s[q].dx; st[i].emb_scale = grp.s[q].emb_scale;
                        st[i].ds = drop_init(grp.s[q].drop);
                    }
#pragma unroll
                for (int q = 0; q < MTN_LN_MAX_GROUP; ++q)
                    if (q == t) st[i].first = grp.first[q][i];
            }
    }
    const bool one_chunk = total <= EMB_CH;
    auto stage = [&](int ch, int cnt) {                    // stream by stream: plain strided copies, no per-token search
        __syncthreads();
        for (int si = 0; si < ns; ++si) {
            const int first = st[si].first, last = si + 1 < ns ? st[si + 1].first : total;
            const int lo = max(first, ch), hi = min(last, ch + cnt);
            const long* src = st[si].tokens - first;
            for (int g = lo + (int)threadIdx.x; g < hi; g += 64 * EMB_W) tok[g - ch] = (int)src[g];
        }
        __syncthreads();
    };
    __syncthreads();
    for (int c0 = 0; c0 < d; c0 += 512) {                  // 8 columns per lane per pass (one pass for d <= 512)
        // passes 1+2 in one scan: count the occurrences of this wave's entry and, as long as there are at most EMB_HEAVY of
        // them, add their rows on the spot (four ballots' tokens are read from LDS before the first is tested)
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        int mine = 0;
        for (int ch = 0; ch < total; ch += EMB_CH) {
            const int cnt = min(EMB_CH, total - ch);
            if (!one_chunk || c0 == 0) stage(ch, cnt);
            if (live)
                for (int b = 0; b < cnt; b += 256) {
                    int t4[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) t4[k] = b + k * 64 + lane < cnt ? tok[b + k * 64 + lane] : -1;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        unsigned long long m = __ballot(t4[k] == v);
                        while (m) {                                  // (wave-uniform) note the hit; the rows are fetched once the scan is done
                            if (mine < EMB_HEAVY && lane == 0) hits[wave][mine] = ch + b + k * 64 + __builtin_ctzll(m);
                            ++mine;
                            m &= m - 1;
                        }
                    }
                }
        }
        const bool light = live && mine > 0 && mine <= EMB_HEAVY;
        if (light) {
            __builtin_amdgcn_wave_barrier();
            emb_add_list(hits[wave], mine, st, ns, d, c0, lane, acc);
            float* dst = grp.dlut[t] + (size_t)v * d + c0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = lane * 4 + 256 * (j >> 2) + (j & 3);
                if (c0 + c < d) dst[c] += acc[j];
            }
        }
        if (lane == 0) count[wave] = live ? mine : 0;
        __syncthreads();
        // pass 2: frequent entries, all EMB_W waves on one entry at a time
        for (int h = 0; h < EMB_W; ++h) {
            if (count[h] <= EMB_HEAVY) continue;           // uniform over the workgroup
            const int vh = blockIdx.x * EMB_W + h;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
            for (int ch = 0; ch < total; ch += EMB_CH) {
                const int cnt = min(EMB_CH, total - ch);
                if (!one_chunk) stage(ch, cnt);
                for (int b = wave * 64; b < cnt; b += EMB_W * 64)
                    emb_add_rows(__ballot(b + lane < cnt && tok[b + lane] == vh), ch + b, st, ns, d, c0, lane, acc);
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 8; ++j) part_sum[wave][lane * 4 + 256 * (j >> 2) + (j & 3)] = acc[j];
            __syncthreads();
            if (threadIdx.x < 512 && c0 + (int)threadIdx.x < d) {
                float sum = 0.f;
#pragma unroll
                for (int q = 0; q < EMB_W; ++q) sum += part_sum[q][threadIdx.x];
                grp.dlut[t][(size_t)vh * d + c0 + threadIdx.x] += sum;
            }
        }
    }
}

extern "C" int mtn_embed_bwd_group(int count, const mtn_embed_bwd_desc* descs, void* stream) {
    MTN_CHECK_ARG(count >= 1 && count <= MTN_LN_MAX_GROUP && descs, "bad group");
    // Default since round 4: the atomic-free kernel (18 us of a 3.7 ms cfg2 step) — the whole train step is then bitwise reproducible
    // run to run.  MTN_EMBED_DETERMINISTIC=0 selects the float-atomic scatter (faster on ragged batches with many pad tokens).
    const char* det_env = MTN_ENV("MTN_EMBED_DETERMINISTIC");
    bool det = !(det_env && det_env[0] == '0');
    for (int i = 0; i < count; ++i) det = det && descs[i].lut_rows > 0;
    if (det) {
        EmbedDetGroup g;
        memset(&g, 0, sizeof(g));
        int vmax = 0;
        for (int i = 0; i < count; ++i) {
            MTN_CHECK_ARG(descs[i].rows > 0 && descs[i].d > 0 && descs[i].tokens && descs[i].dx && descs[i].dlut, "bad descriptor");
            MTN_CHECK_ARG(descs[i].d % 4 == 0, "d must be a multiple of 4");
            g.s[i] = descs[i];
            int t = 0;
            while (t < g.n_lut && g.dlut[t] != descs[i].dlut) ++t;
            if (t == g.n_lut) { g.dlut[t] = descs[i].dlut; g.V[t] = descs[i].lut_rows; g.d[t] = descs[i].d; ++g.n_lut; }
            MTN_CHECK_ARG(g.V[t] == descs[i].lut_rows && g.d[t] == descs[i].d, "streams of one table disagree on its shape");
            g.idx[t][g.n[t]] = i;
            g.first[t][g.n[t]] = g.total[t];
            g.total[t] += descs[i].rows;
            g.first[t][++g.n[t]] = g.total[t];
            if (g.V[t] > vmax) vmax = g.V[t];
        }
        hipLaunchKernelGGL(embed_bwd_det_kernel, dim3((vmax + EMB_W - 1) / EMB_W, g.n_lut), dim3(64 * EMB_W), 0, (hipStream_t)stream, g);
        MTN_CHECK_LAUNCH();
        return MTN_OK;
    }
    EmbedBwdGroup grp;
    memset(&grp, 0, sizeof(grp));
    grp.count = count;
    int blocks = 0;
    for (int i = 0; i < count; ++i) {
        MTN_CHECK_ARG(descs[i].rows > 0 && descs[i].d > 0 && descs[i].tokens && descs[i].dx && descs[i].dlut, "bad descriptor");
        grp.block_start[i] = blocks;
        blocks += (descs[i].rows + 3) / 4;
        grp.d[i] = descs[i];
    }
    for (int i = count; i <= MTN_LN_MAX_GROUP; ++i) grp.block_start[i] = blocks;
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grp);
    MTN_CHECK_LAUNCH();
    return MTN_OK;
}
