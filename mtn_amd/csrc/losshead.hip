// losshead.hip — generator log-softmax + label-smoothed KL divergence, fused per row.
//   Generator.forward      : log_softmax(proj(x))                          (mtn.py:62-69)
//   LabelSmoothing.forward : KLDivLoss(sum)(logp, smoothed one-hot)        (label_smoothing.py:20-32)
//   SimpleLossCompute      : sum_i coef_i * KL_i / norm_i                  (data_utils.py:133-144)
// The (rows, vocab) log-probabilities and target distribution are never materialised: with z the logits, lse the row
// log-sum-exp, t the target, eps = smoothing/(V-2), conf = 1-smoothing and td the smoothed target row,
//     KL_row = sum_j td_j log td_j - sum_j td_j z_j + lse * sum_j td_j
// needs only lse, S = sum_j z_j, z_t and z_pad.  Backward: dz_j = g * (softmax_j * sum_j td_j - td_j).
// The reference's quirk is kept (label_smoothing.py:29): <pad> rows are zeroed only if the sum of their row indices is
// positive, i.e. a lone <pad> target at flat row 0 keeps td = eps everywhere but the <pad> column.
// One 64-lane wave per row, float4 loads, wave-shuffle reductions; HBM/L2-bound (12 KB per row at V = 3000).
#include "common.h"

struct RowInfo {
    int seg, local;      // segment (stream) and row inside it
    long t;
    float scale;         // coef / norm
    bool zero_row, t_is_pad;
};

__device__ __forceinline__ RowInfo row_info(const mtn_losshead_args& A, int row, int lane) {
    RowInfo r;
    int base = 0;
    r.seg = 0;
    for (int s = 0; s < A.n_seg; ++s) {
        if (row >= base && row < base + A.rows[s]) { r.seg = s; break; }
        base += A.rows[s];
    }
    r.local = row - base;
    r.t = A.target[r.seg][r.local];
    r.scale = A.coef[r.seg] / *A.norm[r.seg];
    r.t_is_pad = (r.t == A.pad);
    r.zero_row = false;
    if (r.t_is_pad) {
        if (r.local > 0) r.zero_row = true;                 // its own index makes the index sum positive
        else {                                              // flat row 0: zeroed only if another <pad> row exists
            int any = 0;
            for (int i = 1 + lane; i < A.rows[r.seg]; i += 64) any |= (A.target[r.seg][i] == A.pad);
            r.zero_row = __any(any);
        }
    }
    return r;
}

__global__ __launch_bounds__(256) void losshead_fwd_kernel(const mtn_losshead_args A, int total_rows) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= total_rows) return;
    const RowInfo r = row_info(A, row, lane);
    const float* z = A.logits + (size_t)row * A.ldz;
    const int V = A.V;
    float mx = -INFINITY, S = 0.f;
    for (int c = lane * 4; c < V; c += 256) {
        float4 v = *(const float4*)(z + c);                 // V % 4 == 0 (checked on the host)
        mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
        S += (v.x + v.y) + (v.z + v.w);
    }
    mx = wave_max(mx);
    S = wave_sum(S);
    float se = 0.f;
    for (int c = lane * 4; c < V; c += 256) {
        float4 v = *(const float4*)(z + c);
        se += (__expf(v.x - mx) + __expf(v.y - mx)) + (__expf(v.z - mx) + __expf(v.w - mx));
    }
    const float lse = mx + __logf(wave_sum(se));
    if (lane == 0) {
        A.lse[row] = lse;
        float loss = 0.f;
        if (!r.zero_row) {
            const float eps = A.smoothing / (float)(V - 2), conf = 1.0f - A.smoothing;
            const float zt = z[r.t], zp = z[A.pad];
            float n_eps, sum_tdz, sum_td, sum_tdlog;
            if (!r.t_is_pad) {
                n_eps = (float)(V - 2);
                sum_tdz = eps * (S - zt - zp) + conf * zt;
                sum_td = n_eps * eps + conf;
                sum_tdlog = (eps > 0.f ? n_eps * eps * __logf(eps) : 0.f) + (conf > 0.f ? conf * __logf(conf) : 0.f);
            } else {
                n_eps = (float)(V - 1);
                sum_tdz = eps * (S - zp);
                sum_td = n_eps * eps;
                sum_tdlog = eps > 0.f ? n_eps * eps * __logf(eps) : 0.f;
            }
            loss = (sum_tdlog - sum_tdz + lse * sum_td) * r.scale;
        }
        A.rowloss[row] = loss;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void losshead_bwd_kernel(const mtn_losshead_args A, int total_rows) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= total_rows) return;
    const RowInfo r = row_info(A, row, lane);
    const float* z = A.logits + (size_t)row * A.ldz;
    T* dz = (T*)A.dlogits + (size_t)row * A.ldd;
    const int V = A.V;
    const float g = (*A.gloss) * r.scale;
    const float eps = A.smoothing / (float)(V - 2), conf = 1.0f - A.smoothing;
    const float sum_td = r.zero_row ? 0.f : (r.t_is_pad ? (float)(V - 1) * eps : (float)(V - 2) * eps + conf);
    const float lse = A.lse[row];
    for (int c = lane * 4; c < A.ldd; c += 256) {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < V && !r.zero_row) {
            float4 v = *(const float4*)(z + c);
            float* ov = &o.x;
            const float* vv = &v.x;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = c + k;
                float td = eps;
                if (j == A.pad) td = 0.f;
                else if (j == r.t) td = conf;
                ov[k] = g * (__expf(vv[k] - lse) * sum_td - td);
            }
        }
        store_lp4<T>(dz + c, o);
    }
}

extern "C" int mtn_losshead_fwd(const mtn_losshead_args* A, void* stream) {
    MTN_CHECK_ARG(A && A->n_seg >= 1 && A->n_seg <= MTN_LOSSHEAD_MAX_SEG, "bad segment count");
    MTN_CHECK_ARG(A->V >= 4 && A->V % 4 == 0 && A->ldz % 4 == 0 && A->pad >= 0 && A->pad < A->V, "V and ldz must be multiples of 4");
    MTN_CHECK_ARG(A->logits && A->lse && A->rowloss, "null buffer");
    int total = 0;
    for (int s = 0; s < A->n_seg; ++s) { MTN_CHECK_ARG(A->rows[s] > 0 && A->target[s] && A->norm[s], "bad segment"); total += A->rows[s]; }
    hipLaunchKernelGGL(losshead_fwd_kernel, dim3((total + 3) / 4), dim3(256), 0, (hipStream_t)stream, *A, total);
    MTN_CHECK_LAUNCH();
    return MTN_OK;
}

extern "C" int mtn_losshead_bwd(int dtype, const mtn_losshead_args* A, void* stream) {
    MTN_CHECK_ARG(dtype == MTN_F32 || dtype == MTN_BF16, "bad dtype");
    MTN_CHECK_ARG(A && A->n_seg >= 1 && A->n_seg <= MTN_LOSSHEAD_MAX_SEG, "bad segment count");
    MTN_CHECK_ARG(A->V % 4 == 0 && A->ldz % 4 == 0 && A->ldd % 4 == 0 && A->ldd >= A->V, "V, ldz, ldd must be multiples of 4");
    MTN_CHECK_ARG(A->logits && A->lse && A->gloss && A->dlogits, "null buffer");
    int total = 0;
    for (int s = 0; s < A->n_seg; ++s) total += A->rows[s];
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MTN_BF16) hipLaunchKernelGGL((losshead_bwd_kernel<bf16_t>), dim3((total + 3) / 4), dim3(256), 0, st, *A, total);
    else hipLaunchKernelGGL((losshead_bwd_kernel<float>), dim3((total + 3) / 4), dim3(256), 0, st, *A, total);
    MTN_CHECK_LAUNCH();
    return MTN_OK;
}
