// elementwise.hip — HBM-bound helpers on the path: casts, dropout-backward, fused Adam/Noam
// (train.py:190, data_utils.py:92-117).  16-byte vector accesses, grid-stride loops.
#include <stdarg.h>
#include <stdlib.h>

#include "common.h"

// ---------------------------------------------------------------- error plumbing
static thread_local char g_err[512] = "";
void mtn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* mtn_last_error(void) { return g_err; }
extern "C" int mtn_version(void) { return 113; }

// ---------------------------------------------------------------- environment switches (common.h: MTN_ENV)
// The library is entered from the caller's thread AND from autograd's device thread (backward), so a site's cache may be filled by
// two threads at once: the refresh runs under a mutex and the generation is published with release / read with acquire, so a reader
// that sees the current generation also sees the value written for it.  (A site's value buffer is only rewritten on a generation
// change: mtn_reload_env() must not race with launches that are reading switches — the tests call it between steps.)
#include <mutex>
static int g_env_gen = 0;
static std::mutex g_env_mutex;
extern "C" int mtn_reload_env(void) { return __atomic_add_fetch(&g_env_gen, 1, __ATOMIC_ACQ_REL); }
const char* mtn_env_lookup(MtnEnvVar* v) {
    const int g = __atomic_load_n(&g_env_gen, __ATOMIC_ACQUIRE);
    if (__atomic_load_n(&v->gen, __ATOMIC_ACQUIRE) != g) {
        std::lock_guard<std::mutex> lock(g_env_mutex);
        if (v->gen != g) {
            const char* e = getenv(v->name);
            v->set = e != nullptr;
            if (e) { strncpy(v->val, e, sizeof(v->val) - 1); v->val[sizeof(v->val) - 1] = 0; }
            __atomic_store_n(&v->gen, g, __ATOMIC_RELEASE);
        }
    }
    return v->set ? v->val : nullptr;
}

static inline int grid_for(long n_vec) {
    long b = (n_vec + 255) / 256;
    if (b > 2048) b = 2048;  // 256 CUs x 8 blocks, grid-stride the rest
    if (b < 1) b = 1;
    return (int)b;
}

// ---------------------------------------------------------------- cast / dropout-backward
struct CastGroup {
    int count;
    mtn_cast_desc d[MTN_CAST_MAX_GROUP];
};

template <typename T>
__global__ __launch_bounds__(256) void cast_kernel(const CastGroup grp) {
    const mtn_cast_desc& D = grp.d[blockIdx.y];
    const long n = D.n;
    const float* __restrict__ src = D.src;
    T* __restrict__ dst = (T*)D.dst;
    const DropState ds = drop_init(D.drop);
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 v = *(const float4*)(src + i * 4);
        if (ds.on) {
            v.x = drop_keep(ds, (uint64_t)(i * 4 + 0)) ? v.x * ds.scale : 0.f;
            v.y = drop_keep(ds, (uint64_t)(i * 4 + 1)) ? v.y * ds.scale : 0.f;
            v.z = drop_keep(ds, (uint64_t)(i * 4 + 2)) ? v.z * ds.scale : 0.f;
            v.w = drop_keep(ds, (uint64_t)(i * 4 + 3)) ? v.w * ds.scale : 0.f;
        }
        if (D.gate) {
            const float4 gt = *(const float4*)(D.gate + i * 4);
            v.x = gt.x > 0.f ? v.x : 0.f; v.y = gt.y > 0.f ? v.y : 0.f; v.z = gt.z > 0.f ? v.z : 0.f; v.w = gt.w > 0.f ? v.w : 0.f;
        }
        if constexpr (sizeof(T) == 2) {
            uint2 u;
            u.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
            u.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
            *(uint2*)(dst + i * 4) = u;
        } else {
            *(float4*)(dst + i * 4) = v;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {  // tail
        long i = (n4 << 2) + threadIdx.x;
        float v = src[i];
        if (ds.on) v = drop_keep(ds, (uint64_t)i) ? v * ds.scale : 0.f;
        if (D.gate && !(D.gate[i] > 0.f)) v = 0.f;
        dst[i] = LP<T>::from_f32(v);
    }
}

extern "C" int mtn_cast_group(int dtype, int count, const mtn_cast_desc* descs, void* stream) {
    MTN_CHECK_ARG(dtype == MTN_F32 || dtype == MTN_BF16, "bad dtype");
    MTN_CHECK_ARG(count >= 1 && count <= MTN_CAST_MAX_GROUP && descs, "bad group");
    CastGroup grp;
    memset(&grp, 0, sizeof(grp));
    grp.count = count;
    long nmax = 0;
    for (int i = 0; i < count; ++i) {
        MTN_CHECK_ARG(descs[i].n > 0 && descs[i].src && descs[i].dst, "bad cast descriptor");
        grp.d[i] = descs[i];
        if (descs[i].n > nmax) nmax = descs[i].n;
    }
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(grid_for(nmax >> 2), count);
    if (dtype == MTN_BF16) hipLaunchKernelGGL((cast_kernel<bf16_t>), grid, dim3(256), 0, s, grp);
    else hipLaunchKernelGGL((cast_kernel<float>), grid, dim3(256), 0, s, grp);
    MTN_CHECK_LAUNCH();
    return MTN_OK;
}

static int launch_cast(int dtype, long n, const float* src, void* dst, mtn_dropout drop, void* stream) {
    mtn_cast_desc D = {n, src, dst, drop, nullptr};
    return mtn_cast_group(dtype, 1, &D, stream);
}

extern "C" int mtn_cast_f32_to_lp(int dtype, long n, const float* src, void* dst, void* stream) {
    mtn_dropout none = {0.f, 0u, nullptr};
    return launch_cast(dtype, n, src, dst, none, stream);
}
extern "C" int mtn_dropout_bwd_to_lp(int dtype, long n, const float* src, mtn_dropout drop, void* dst, void* stream) {
    return launch_cast(dtype, n, src, dst, drop, stream);
}

// ---------------------------------------------------------------- transposed weight copies
// dX = dY W needs W with the contraction index contiguous: every 2-D path weight [rows, cols] keeps a transposed
// compute-dtype copy [cols, rows] at the same offset of a second flat buffer, refreshed once per optimiser step
// (2 x 213 MB of HBM traffic for the 106.65 M-parameter model: ~0.1 ms, against ~150 small GEMMs that then run on the
// LDS-DMA fast path).  One launch covers all matrices through a device-resident descriptor table.
template <typename T>
__global__ __launch_bounds__(256) void transpose_group_kernel(const T* __restrict__ src, T* __restrict__ dst,
                                                              const mtn_transpose_desc* __restrict__ descs, int count) {
    // 64x64 tile through LDS; 16-byte global accesses on both sides when the matrix allows it (8 bf16 / 4 fp32 per lane)
    constexpr int V = 16 / (int)sizeof(T);
    __shared__ __attribute__((aligned(16))) T tile[64][64 + V];
    int lo = 0, hi = count - 1;
    const int b = blockIdx.x;
    while (lo < hi) {                       // last descriptor whose tile_start <= b
        int mid = (lo + hi + 1) >> 1;
        if (descs[mid].tile_start <= b) lo = mid; else hi = mid - 1;
    }
    const mtn_transpose_desc D = descs[lo];
    const int tiles_c = (D.cols + 63) >> 6;
    const int t = b - D.tile_start;
    const int r0 = (t / tiles_c) * 64, c0 = (t % tiles_c) * 64;
    const T* s = src + D.off;
    T* d = dst + D.dst_off;
    const bool full = (r0 + 64 <= D.rows) && (c0 + 64 <= D.cols) && (D.rows % V == 0) && (D.cols % V == 0) && (D.off % V == 0) && (D.dst_off % V == 0);
    if (full) {
        constexpr int VPR = 64 / V;                      // vectors per tile row (8 bf16 / 16 fp32)
        for (int i = threadIdx.x; i < 64 * VPR; i += 256) {
            const int r = i / VPR, cv = (i % VPR) * V;
            *(uint4*)&tile[r][cv] = *(const uint4*)(s + (size_t)(r0 + r) * D.cols + c0 + cv);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 64 * VPR; i += 256) {
            const int c = i / VPR, rv = (i % VPR) * V;   // output row c0 + c, output columns r0 + rv ..
            __attribute__((aligned(16))) T tmp[V];
#pragma unroll
            for (int k = 0; k < V; ++k) tmp[k] = tile[rv + k][c];
            *(uint4*)(d + (size_t)(c0 + c) * D.rows + r0 + rv) = *(const uint4*)tmp;
        }
        return;
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty + 4 * i, c = c0 + tx;
        if (r < D.rows && c < D.cols) tile[ty + 4 * i][tx] = s[(size_t)r * D.cols + c];
    }
    __syncthreads();
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty + 4 * i, r = r0 + tx;
        if (r < D.rows && c < D.cols) d[(size_t)c * D.rows + r] = tile[tx][ty + 4 * i];
    }
}

extern "C" int mtn_transpose_group(int dtype, const void* src, void* dst, const mtn_transpose_desc* descs_device, int count,
                                   int total_tiles, void* stream) {
    MTN_CHECK_ARG(dtype == MTN_F32 || dtype == MTN_BF16, "bad dtype");
    MTN_CHECK_ARG(src && dst && descs_device && count > 0 && total_tiles > 0, "bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MTN_BF16)
        hipLaunchKernelGGL((transpose_group_kernel<bf16_t>), dim3(total_tiles), dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, descs_device, count);
    else
        hipLaunchKernelGGL((transpose_group_kernel<float>), dim3(total_tiles), dim3(256), 0, s, (const float*)src, (float*)dst, descs_device, count);
    MTN_CHECK_LAUNCH();
    return MTN_OK;
}

// ---------------------------------------------------------------- Noam schedule + Adam
__global__ void noam_tick_kernel(float* state, float factor, float model_size, float warmup, float beta1, float beta2) {
    if (threadIdx.x == 0 && blockIdx.x == 0) noam_tick_body(state, factor, model_size, warmup, beta1, beta2);   // data_utils.py:111-117
}
extern "C" int mtn_noam_tick(float* state, float factor, int model_size, int warmup, float beta1, float beta2, void* stream) {
    MTN_CHECK_ARG(state && model_size > 0 && warmup > 0, "bad arguments");
    hipLaunchKernelGGL(noam_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, factor, (float)model_size, (float)warmup, beta1, beta2);
    MTN_CHECK_LAUNCH();
    return MTN_OK;
}

// torch.optim.Adam semantics (adam_update, common.h).  30 B/param of HBM traffic with the bf16 copy.
template <typename T>
__global__ __launch_bounds__(256) void adam_kernel(long n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, T* __restrict__ p_lp, const float* __restrict__ state,
                                                   const float* __restrict__ grad_scale, float beta1, float beta2, float eps) {
    const AdamCoef c = adam_coef(state, grad_scale, beta1, beta2, eps);
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        // the 16 B/param fp32 streams are touched once per step: non-temporal, so that they do not evict the bf16 weights
        // (the forward/backward working set) from L2 / Infinity Cache
        typedef __attribute__((ext_vector_type(4))) float nt4;
        nt4 p_ = __builtin_nontemporal_load((const nt4*)(p + i * 4)), g_ = __builtin_nontemporal_load((const nt4*)(g + i * 4));
        nt4 m_ = __builtin_nontemporal_load((const nt4*)(m + i * 4)), v_ = __builtin_nontemporal_load((const nt4*)(v + i * 4));
        float4 pv = make_float4(p_[0], p_[1], p_[2], p_[3]), gv = make_float4(g_[0], g_[1], g_[2], g_[3]);
        float4 mv = make_float4(m_[0], m_[1], m_[2], m_[3]), vv = make_float4(v_[0], v_[1], v_[2], v_[3]);
        float* pp = &pv.x; float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) adam_update(pp[k], mp[k], vp[k], gp[k], c);
        __builtin_nontemporal_store(nt4{pv.x, pv.y, pv.z, pv.w}, (nt4*)(p + i * 4));
        __builtin_nontemporal_store(nt4{mv.x, mv.y, mv.z, mv.w}, (nt4*)(m + i * 4));
        __builtin_nontemporal_store(nt4{vv.x, vv.y, vv.z, vv.w}, (nt4*)(v + i * 4));
        if (p_lp) {
            if constexpr (sizeof(T) == 2) {
                uint2 u;
                u.x = (uint32_t)f32_to_bf16(pv.x) | ((uint32_t)f32_to_bf16(pv.y) << 16);
                u.y = (uint32_t)f32_to_bf16(pv.z) | ((uint32_t)f32_to_bf16(pv.w) << 16);
                *(uint2*)(p_lp + i * 4) = u;
            } else {
                *(float4*)(p_lp + i * 4) = pv;
            }
        }
    }
}

// The same update over a list of chunks of the flat buffers (what is left once the parameter-gradient GEMMs have applied
// the optimiser to the weight matrices in their epilogues: biases, LayerNorm gains, embedding tables, the loss head).
// chunk c = elements [off[c], off[c] + len[c]) with off, len multiples of 4 and len <= 4096; one workgroup per chunk.
template <typename T>
__global__ __launch_bounds__(256) void adam_chunks_kernel(const long* __restrict__ off, const int* __restrict__ len, float* __restrict__ p,
                                                          const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                          T* __restrict__ p_lp, const float* __restrict__ state,
                                                          const float* __restrict__ grad_scale, float beta1, float beta2, float eps) {
    const long base = off[blockIdx.x];
    const int n4 = len[blockIdx.x] >> 2;
    // a whole chunk (<= 4096 elements = four float4 per thread and stream) is in flight before the first update: the loop that loaded,
    // updated and stored one float4 at a time made four dependent memory round trips per workgroup (38 -> 3x us per step on the
    // 4.4 M elements the GEMM epilogues leave: biases, LayerNorm gains, embedding tables, the loss head)
    float4 pv[4], gv[4], mv[4], vv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int q = threadIdx.x + 256 * u;
        const long i = base + (long)(q < n4 ? q : 0) * 4;
        pv[u] = *(const float4*)(p + i); gv[u] = *(const float4*)(g + i); mv[u] = *(const float4*)(m + i); vv[u] = *(const float4*)(v + i);
    }
    const AdamCoef c = adam_coef(state, grad_scale, beta1, beta2, eps);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int q = threadIdx.x + 256 * u;
        if (q >= n4) continue;
        const long i = base + (long)q * 4;
        float* pp = &pv[u].x; float* gp = &gv[u].x; float* mp = &mv[u].x; float* vp = &vv[u].x;
#pragma unroll
        for (int k = 0; k < 4; ++k) adam_update(pp[k], mp[k], vp[k], gp[k], c);
        *(float4*)(p + i) = pv[u]; *(float4*)(m + i) = mv[u]; *(float4*)(v + i) = vv[u];
        if (p_lp) {
            if constexpr (sizeof(T) == 2) {
                uint2 w;
                w.x = (uint32_t)f32_to_bf16(pv[u].x) | ((uint32_t)f32_to_bf16(pv[u].y) << 16);
                w.y = (uint32_t)f32_to_bf16(pv[u].z) | ((uint32_t)f32_to_bf16(pv[u].w) << 16);
                *(uint2*)(p_lp + i) = w;
            } else {
                *(float4*)(p_lp + i) = pv[u];
            }
        }
    }
}

extern "C" int mtn_adam_step_chunks(int dtype, int n_chunks, const long* off, const int* len, float* p, const float* g, float* m, float* v,
                                    void* p_lp, const float* state, const float* grad_scale, float beta1, float beta2, float eps, void* stream) {
    MTN_CHECK_ARG(dtype == MTN_F32 || dtype == MTN_BF16, "bad dtype");
    MTN_CHECK_ARG(n_chunks > 0 && off && len && p && g && m && v && state, "chunk list and buffers must be non-null");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MTN_BF16)
        hipLaunchKernelGGL((adam_chunks_kernel<bf16_t>), dim3(n_chunks), dim3(256), 0, s, off, len, p, g, m, v, (bf16_t*)p_lp, state, grad_scale, beta1, beta2, eps);
    else
        hipLaunchKernelGGL((adam_chunks_kernel<float>), dim3(n_chunks), dim3(256), 0, s, off, len, p, g, m, v, (float*)p_lp, state, grad_scale, beta1, beta2, eps);
    MTN_CHECK_LAUNCH();
    return MTN_OK;
}

extern "C" int mtn_adam_step(int dtype, long n, float* p, const float* g, float* m, float* v, void* p_lp, const float* state,
                             const float* grad_scale, float beta1, float beta2, float eps, void* stream) {
    MTN_CHECK_ARG(dtype == MTN_F32 || dtype == MTN_BF16, "bad dtype");
    MTN_CHECK_ARG(n > 0 && n % 4 == 0 && p && g && m && v && state, "n must be a positive multiple of 4; buffers non-null");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MTN_BF16)
        hipLaunchKernelGGL((adam_kernel<bf16_t>), dim3(grid_for(n >> 2)), dim3(256), 0, s, n, p, g, m, v, (bf16_t*)p_lp, state, grad_scale, beta1, beta2, eps);
    else
        hipLaunchKernelGGL((adam_kernel<float>), dim3(grid_for(n >> 2)), dim3(256), 0, s, n, p, g, m, v, (float*)p_lp, state, grad_scale, beta1, beta2, eps);
    MTN_CHECK_LAUNCH();
    return MTN_OK;
}

// ---------------------------------------------------------------- on-box HBM ceiling (measurement support, bench.py)
// A 16-byte-per-lane streaming copy src -> dst of `bytes` bytes (caller's buffers, each >= bytes and well beyond the 256 MiB
// Infinity Cache): what THIS box's HBM delivers to a kernel that does nothing but move bytes — the measured denominator
// beside the 8 TB/s spec for the HBM-bound launches (parameter-gradient + optimiser table launch, adam_kernel).  Reports the
// best over a few variants (non-temporal / plain, 1 / 4 loads in flight per thread, 8-32 workgroups per CU) as (bytes read +
// bytes written) / time.
template <bool NT, int U>
__global__ __launch_bounds__(256) void hbm_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long n16) {
    typedef __attribute__((ext_vector_type(4))) unsigned nt4;
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {   // U independent 16-byte loads in flight per thread
        nt4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load((const nt4*)(src + i + u * stride)) : *(const nt4*)(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT) __builtin_nontemporal_store(v[u], (nt4*)(dst + i + u * stride));
            else *(nt4*)(dst + i + u * stride) = v[u];
        }
    }
    for (; i < n16; i += stride) *(nt4*)(dst + i) = *(const nt4*)(src + i);
}

extern "C" int mtn_measure_hbm_peak(const void* src, void* dst, long bytes, void* stream, double* gbps) {
    MTN_CHECK_ARG(src && dst && gbps && bytes >= (1L << 20) && bytes % 16 == 0, "bad arguments");
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { mtn_set_error("hipEventCreate failed"); return MTN_ERR_LAUNCH; }
    const long n16 = bytes / 16;
    double best = 0.0;
    // a ceiling is the best the box does: non-temporal / plain accesses, 1 / 4 loads in flight per thread, 8 / 16 / 32 workgroups per CU
    for (int variant = 0; variant < 4; ++variant)
        for (int wpc = 8; wpc <= 32; wpc *= 2)
            for (int rep = 0; rep < 4; ++rep) {
                const dim3 grid(256 * wpc), block(256);
                (void)hipEventRecord(e0, s);
                switch (variant) {
                    case 0: hipLaunchKernelGGL((hbm_copy_kernel<true, 1>), grid, block, 0, s, (const uint4*)src, (uint4*)dst, n16); break;
                    case 1: hipLaunchKernelGGL((hbm_copy_kernel<true, 4>), grid, block, 0, s, (const uint4*)src, (uint4*)dst, n16); break;
                    case 2: hipLaunchKernelGGL((hbm_copy_kernel<false, 1>), grid, block, 0, s, (const uint4*)src, (uint4*)dst, n16); break;
                    default: hipLaunchKernelGGL((hbm_copy_kernel<false, 4>), grid, block, 0, s, (const uint4*)src, (uint4*)dst, n16); break;
                }
                (void)hipEventRecord(e1, s);
                (void)hipEventSynchronize(e1);
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                const double g = 2.0 * (double)bytes / (ms * 1e-3) / 1e9;
                if (g > best) best = g;
            }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    MTN_CHECK_LAUNCH();
    *gbps = best;
    return MTN_OK;
}

// ---------------------------------------------------------------- CU-masked streams (measurement support)
// A stream whose kernels may only run on the first `n_cus` compute units the runtime enumerates (hipExtStreamCreateWithCUMask;
// the mask bits are dealt to the XCDs round-robin, so the allowed CUs are spread evenly over the 8 dies), optionally at low
// priority: tools/overlap_cu_mask_probe.py uses it to measure whether an HBM-bound pass (parameter gradients + optimiser) can
// hide under the latency-bound backward chain.  The caller owns the stream (mtn_stream_destroy).
extern "C" int mtn_stream_create_cu_masked(int n_cus, int low_priority, void** stream_out) {
    MTN_CHECK_ARG(stream_out && n_cus >= 1 && n_cus <= 256, "n_cus must be in [1, 256]");
    uint32_t mask[8];
    for (int w = 0; w < 8; ++w) {
        const int lo = w * 32;
        mask[w] = n_cus >= lo + 32 ? 0xffffffffu : (n_cus > lo ? ((1u << (n_cus - lo)) - 1u) : 0u);
    }
    hipStream_t s = nullptr;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
    if (e != hipSuccess) { mtn_set_error("hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e)); return MTN_ERR_LAUNCH; }
    (void)low_priority;      // (the CU-mask constructor takes no priority: a masked stream runs at the default priority)
    *stream_out = (void*)s;
    return MTN_OK;
}
extern "C" int mtn_stream_destroy(void* stream) {
    MTN_CHECK_ARG(stream, "null stream");
    if (hipStreamDestroy((hipStream_t)stream) != hipSuccess) { mtn_set_error("hipStreamDestroy failed"); return MTN_ERR_LAUNCH; }
    return MTN_OK;
}
