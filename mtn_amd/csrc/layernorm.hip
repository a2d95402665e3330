// layernorm.hip — MTN's LayerNorm variant (mtn.py:103-114): y = a2*(x-mean)/(std_unbiased+eps)+b2.
// HBM/L2-bound row kernels: one 64-lane wave per row, float4 loads, wave-shuffle reductions.
#include "common.h"

// ------------------------------------------------------------------------------------------ forward
template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(int rows, int d, float eps, const float* __restrict__ x,
                                                     const float* __restrict__ a2, const float* __restrict__ b2,
                                                     float* __restrict__ y_f32, T* __restrict__ y_lp,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * d;
    float s = 0.f;
    for (int c = lane * 4; c < d; c += 256) {
        float4 v = *(const float4*)(xr + c);
        s += (v.x + v.y) + (v.z + v.w);
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
    for (int c = lane * 4; c < d; c += 256) {
        float4 v = *(const float4*)(xr + c);
        float e0 = v.x - mean, e1 = v.y - mean, e2 = v.z - mean, e3 = v.w - mean;
        q += (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
    }
    const float std_u = sqrtf(wave_sum(q) / (float)(d - 1));
    const float rstd = 1.0f / (std_u + eps);
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
    for (int c = lane * 4; c < d; c += 256) {
        float4 v = *(const float4*)(xr + c);
        float4 g = *(const float4*)(a2 + c);
        float4 b = *(const float4*)(b2 + c);
        float4 o;
        o.x = g.x * (v.x - mean) * rstd + b.x;
        o.y = g.y * (v.y - mean) * rstd + b.y;
        o.z = g.z * (v.z - mean) * rstd + b.z;
        o.w = g.w * (v.w - mean) * rstd + b.w;
        if (y_f32) *(float4*)(y_f32 + (size_t)row * d + c) = o;
        if (y_lp) {
            T* yp = y_lp + (size_t)row * d + c;
            if constexpr (sizeof(T) == 2) {
                uint2 pk;
                pk.x = (uint32_t)f32_to_bf16(o.x) | ((uint32_t)f32_to_bf16(o.y) << 16);
                pk.y = (uint32_t)f32_to_bf16(o.z) | ((uint32_t)f32_to_bf16(o.w) << 16);
                *(uint2*)yp = pk;
            } else {
                *(float4*)yp = o;
            }
        }
    }
}

extern "C" int mtn_layernorm_fwd(int dtype, int rows, int d, float eps, const float* x, const float* a2, const float* b2,
                                 float* y_f32, void* y_lp, float* mean, float* rstd, void* stream) {
    MTN_CHECK_ARG(dtype == MTN_F32 || dtype == MTN_BF16, "bad dtype");
    MTN_CHECK_ARG(rows > 0 && d >= 4 && d % 4 == 0, "rows>0 and d%4==0 required");
    MTN_CHECK_ARG(x && a2 && b2, "null input");
    dim3 grid((rows + 3) / 4), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MTN_BF16)
        hipLaunchKernelGGL((ln_fwd_kernel<bf16_t>), grid, block, 0, s, rows, d, eps, x, a2, b2, y_f32, (bf16_t*)y_lp, mean, rstd);
    else
        hipLaunchKernelGGL((ln_fwd_kernel<float>), grid, block, 0, s, rows, d, eps, x, a2, b2, y_f32, (float*)y_lp, mean, rstd);
    MTN_CHECK_LAUNCH();
    return MTN_OK;
}

// ------------------------------------------------------------------------------------------ backward
// With h_i = g_i*a2_i, xc_i = x_i-mean, r = 1/(std+eps), s1 = sum h, s2 = sum h*xc:
//   dx_i = r*h_i - r*s1/d - s2*r^2/(std*(d-1)) * xc_i      (+ dres_i)
//   da2_i = sum_rows g_i*xc_i*r        db2_i = sum_rows g_i
// Parameter gradients: each wave accumulates its rows in registers (lane owns columns lane*4+256*j),
// writes one partial row; ln_bwd_finalize sums the partials (deterministic, no atomics).
static constexpr int LN_BWD_ROWS_PER_WAVE = 16;
static constexpr int LN_MAXV = 8;  // float4 per lane -> d <= 2048

__global__ __launch_bounds__(256) void ln_bwd_kernel(int rows, int d, float eps, const float* __restrict__ x,
                                                     const float* __restrict__ a2, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ g,
                                                     const float* __restrict__ dres, float* __restrict__ dx,
                                                     float* __restrict__ partial) {
    const int lane = threadIdx.x & 63;
    const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int r0 = wave_global * LN_BWD_ROWS_PER_WAVE;
    float4 ga[LN_MAXV], gb[LN_MAXV];
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) ga[j] = gb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float inv_d = 1.0f / (float)d;
    for (int rr = 0; rr < LN_BWD_ROWS_PER_WAVE; ++rr) {
        const int row = r0 + rr;
        if (row >= rows) break;
        const float mu = mean[row], r = rstd[row];
        const float std_u = fmaxf(1.0f / r - eps, 1e-30f);
        const float* xr = x + (size_t)row * d;
        const float* gr = g + (size_t)row * d;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < LN_MAXV; ++j) {
            const int c = lane * 4 + 256 * j;
            if (c < d) {
                float4 xv = *(const float4*)(xr + c), gv = *(const float4*)(gr + c), av = *(const float4*)(a2 + c);
                float h0 = gv.x * av.x, h1 = gv.y * av.y, h2 = gv.z * av.z, h3 = gv.w * av.w;
                float e0 = xv.x - mu, e1 = xv.y - mu, e2 = xv.z - mu, e3 = xv.w - mu;
                s1 += (h0 + h1) + (h2 + h3);
                s2 += (h0 * e0 + h1 * e1) + (h2 * e2 + h3 * e3);
                ga[j].x += gv.x * e0 * r; ga[j].y += gv.y * e1 * r; ga[j].z += gv.z * e2 * r; ga[j].w += gv.w * e3 * r;
                gb[j].x += gv.x; gb[j].y += gv.y; gb[j].z += gv.z; gb[j].w += gv.w;
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
                float4 xv = *(const float4*)(xr + c), gv = *(const float4*)(gr + c), av = *(const float4*)(a2 + c);
                float4 o;
                o.x = r * gv.x * av.x - c1 - c2 * (xv.x - mu);
                o.y = r * gv.y * av.y - c1 - c2 * (xv.y - mu);
                o.z = r * gv.z * av.z - c1 - c2 * (xv.z - mu);
                o.w = r * gv.w * av.w - c1 - c2 * (xv.w - mu);
                if (dres) {
                    float4 dv = *(const float4*)(dres + (size_t)row * d + c);
                    o.x += dv.x; o.y += dv.y; o.z += dv.z; o.w += dv.w;
                }
                *(float4*)(dx + (size_t)row * d + c) = o;
            }
        }
    }
    float* pa = partial + (size_t)wave_global * 2 * d;
    float* pb = pa + d;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        const int c = lane * 4 + 256 * j;
        if (c < d) {
            *(float4*)(pa + c) = ga[j];
            *(float4*)(pb + c) = gb[j];
        }
    }
}

__global__ __launch_bounds__(256) void ln_bwd_finalize(int nparts, int d, const float* __restrict__ partial,
                                                       float* __restrict__ da2, float* __restrict__ db2) {
    const int c = blockIdx.x * 256 + threadIdx.x;  // column over [0, 2d): first d -> da2, next d -> db2
    if (c >= 2 * d) return;
    float s = 0.f;
    for (int p = 0; p < nparts; ++p) s += partial[(size_t)p * 2 * d + c];
    if (c < d) { if (da2) da2[c] = s; }
    else if (db2) db2[c - d] = s;
}

static inline int ln_bwd_waves(int rows) { return (rows + LN_BWD_ROWS_PER_WAVE - 1) / LN_BWD_ROWS_PER_WAVE; }

extern "C" long mtn_layernorm_bwd_partial_floats(int rows, int d) {
    int blocks = (ln_bwd_waves(rows) + 3) / 4;
    return (long)blocks * 4 * 2 * d;
}

extern "C" int mtn_layernorm_bwd(int rows, int d, float eps, const float* x, const float* a2, const float* mean,
                                 const float* rstd, const float* g, const float* dres, float* dx, float* da2, float* db2,
                                 float* partial, void* stream) {
    MTN_CHECK_ARG(rows > 0 && d >= 4 && d % 4 == 0 && d <= 256 * LN_MAXV, "rows>0, d%4==0, d<=2048 required");
    MTN_CHECK_ARG(x && a2 && mean && rstd && g && dx && partial, "null input");
    hipStream_t s = (hipStream_t)stream;
    const int blocks = (ln_bwd_waves(rows) + 3) / 4;
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(blocks), dim3(256), 0, s, rows, d, eps, x, a2, mean, rstd, g, dres, dx, partial);
    MTN_CHECK_LAUNCH();
    if (da2 || db2) {
        hipLaunchKernelGGL(ln_bwd_finalize, dim3((2 * d + 255) / 256), dim3(256), 0, s, blocks * 4, d, partial, da2, db2);
        MTN_CHECK_LAUNCH();
    }
    return MTN_OK;
}
