// common.h — shared device/host helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mtn_hip.h"

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// ---------------------------------------------------------------- error plumbing (host)
void mtn_set_error(const char* fmt, ...);
#define MTN_CHECK_ARG(cond, msg)                         \
    do {                                                 \
        if (!(cond)) {                                   \
            mtn_set_error("%s: %s", __func__, msg);      \
            return MTN_ERR_ARG;                          \
        }                                                \
    } while (0)
#define MTN_CHECK_LAUNCH()                                                        \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            mtn_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__)); \
            return MTN_ERR_LAUNCH;                                                \
        }                                                                         \
    } while (0)

// ---------------------------------------------------------------- environment switches (host)
// Development / test switches are read through a per-site cache instead of getenv() on every launch (a linear scan of the
// environment, ~10 of them on the GEMM dispatch path: ~2 us per launch on the eager paths).  A process that changes such a
// variable after its first use calls mtn_reload_env() (include/mtn_hip.h); the tests do.
struct MtnEnvVar { const char* name; int gen; bool set; char val[56]; };
const char* mtn_env_lookup(MtnEnvVar* v);      // nullptr when unset (elementwise.hip)
#define MTN_ENV(NAME) ([]() -> const char* { static MtnEnvVar v_ = {NAME, -1, false, {0}}; return mtn_env_lookup(&v_); }())

// ---------------------------------------------------------------- bf16 <-> f32
__device__ __forceinline__ float bf16_to_f32(bf16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {  // round-to-nearest-even, NaN preserved
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

template <typename T> struct LP;  // low-precision element traits
template <> struct LP<float> {
    static constexpr int EPV = 4;     // elements per 16-byte vector
    static constexpr int KSTEP = 16;  // contraction elements consumed by one 16-byte fragment pair
    __device__ static __forceinline__ float to_f32(float v) { return v; }
    __device__ static __forceinline__ float from_f32(float v) { return v; }
};
template <> struct LP<bf16_t> {
    static constexpr int EPV = 8;
    static constexpr int KSTEP = 32;
    __device__ static __forceinline__ float to_f32(bf16_t v) { return bf16_to_f32(v); }
    __device__ static __forceinline__ bf16_t from_f32(float v) { return f32_to_bf16(v); }
};

// One 16-byte fragment pair -> 16x16 accumulator.  A fragment lane l holds row (l&15), contraction slots
// (l>>4)*EPV .. +EPV-1 of the KSTEP-wide step; B likewise for column (l&15).  The slot->k map is the same
// permutation on both operands, so any k-order inside the 16 bytes is fine.
// 16 bytes of MFMA operand read by inline asm (ds_read_b128 / ds_read_b64_tr_b16): a native vector so that it can be an asm "+v"
// operand, and the pass that ties its consumers to the wait behind the read (MTN_LANDED)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
__device__ __forceinline__ uint4 as_uint4(const u32x4_t& v) { return make_uint4(v.x, v.y, v.z, v.w); }
// The compiler believes an asm read's result is there when the asm statement ends; anything it derives from it could be computed
// before the s_waitcnt that really delivers it.  Passing the register through an empty asm right after the wait makes every
// consumer depend on a statement that cannot move above the wait.
#define MTN_LANDED(v) asm volatile("" : "+v"(v))
template <typename T> __device__ __forceinline__ void mma16(f32x4_t& acc, const uint4& a, const uint4& b);
template <> __device__ __forceinline__ void mma16<bf16_t>(f32x4_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8_t*)&a, *(const bf16x8_t*)&b, acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma16<float>(f32x4_t& acc, const uint4& a, const uint4& b) {
    // exact-fp32 MFMA (v_mfma_f32_16x16x4_f32): lane group g supplies k = g per instruction
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}

template <typename T> __device__ __forceinline__ void store_lp4(T* p, float4 v);      // 4 consecutive elements, 8/16-byte aligned
template <> __device__ __forceinline__ void store_lp4<float>(float* p, float4 v) { *(float4*)p = v; }
template <> __device__ __forceinline__ void store_lp4<bf16_t>(bf16_t* p, float4 v) {
    uint2 u;
    u.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
    u.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
    *(uint2*)p = u;
}

// ---------------------------------------------------------------- dropout keep-mask
// Counter-based: keep(idx) is a pure function of (seed, salt, idx).
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
struct DropState {
    uint32_t k0, k1, thresh;  // thresh = p * 2^24
    float scale;              // 1/(1-p)
    bool on;
};
__device__ __forceinline__ DropState drop_init(const mtn_dropout& d) {
    DropState s;
    s.on = (d.p > 0.f) && (d.seed != nullptr);
    s.k0 = s.k1 = s.thresh = 0; s.scale = 1.f;
    if (s.on) {
        // the seed as a SCALAR load with its own wait (lgkmcnt): as a vector load its consumer — the key hashing right below — made the
        // compiler wait for vmcnt(0), i.e. for every load and LDS-DMA the kernel had in flight (the seed is the youngest of them)
        uint64_t sd;
        asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(sd) : "s"(d.seed) : "memory");
        s.k0 = mix32((uint32_t)sd ^ (d.salt * 0x9E3779B9u));
        s.k1 = mix32((uint32_t)(sd >> 32) + d.salt * 0x85EBCA6Bu + 0x165667B1u);
        s.thresh = (uint32_t)(d.p * 16777216.0f);
        s.scale = 1.0f / (1.0f - d.p);
    }
    return s;
}
// keep(idx) = lowbias32((lo(idx) ^ k0) + hi(idx) * DROP_HI_MUL + k1), top 24 bits against p * 2^24: one 32-bit mixer per element
// (two integer multiplies — they run at quarter rate on the vector ALU, and the attention kernels hash 16 elements per lane per tile).
// BOTH key words enter BEFORE the mixer (round 4).  Until round 3 k1 was XORed onto the mixer's output: two streams with equal k0 —
// seeds that differ only in their high word — were the same 24-bit values up to a fixed XOR, and their keep bits correlated
// (rho = -0.11 at p = 0.1, tests/test_dropout_stream_gpu.py).  Same instruction count: an add in front instead of an xor behind.
static constexpr uint32_t DROP_HI_MUL = 0x9E3779B1u;
__device__ __forceinline__ bool drop_keep(const DropState& s, uint64_t idx) {
    const uint32_t lo = (uint32_t)idx, hi = (uint32_t)(idx >> 32);
    const uint32_t r = mix32((lo ^ s.k0) + hi * DROP_HI_MUL + s.k1);
    return (r >> 8) >= s.thresh;
}
// The same function for idx = base + off with a (wave-)uniform 64-bit base and a 32-bit offset: the high word's term is prepared once
struct DropBase { uint32_t lo, hic; };
__device__ __forceinline__ DropBase drop_base(uint64_t base) {
    DropBase b;
    b.lo = (uint32_t)base;
    b.hic = (uint32_t)(base >> 32) * DROP_HI_MUL;
    return b;
}
__device__ __forceinline__ bool drop_keep_at(const DropState& s, const DropBase& b, uint32_t off) {
    const uint32_t lo = b.lo + off;
    const uint32_t r = mix32((lo ^ s.k0) + (b.hic + s.k1) + (lo < b.lo ? DROP_HI_MUL : 0u));
    return (r >> 8) >= s.thresh;
}

// ---------------------------------------------------------------- Noam schedule tick (data_utils.py:111-117), one thread
// state = [step, lr, 1 - b1^step, 1 - b2^step]; shared by noam_tick_kernel and the step-head launch (same bits)
__device__ __forceinline__ void noam_tick_body(float* state, float factor, float model_size, float warmup, float beta1, float beta2) {
    const float step = state[0] + 1.0f;
    state[0] = step;
    state[1] = factor * rsqrtf(model_size) * fminf(rsqrtf(step), step * powf(warmup, -1.5f));
    state[2] = 1.0f - powf(beta1, step);
    state[3] = 1.0f - powf(beta2, step);
}

// ---------------------------------------------------------------- Adam (torch.optim.Adam semantics), one element
// m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps).
// Roundings pinned with explicit fma so that every kernel that applies the update (adam_kernel, adam_chunks_kernel, the
// optimiser epilogue of the parameter-gradient GEMMs) produces the same bits.
struct AdamCoef { float step_size, inv_sqrt_bc2, gs, beta1, beta2, eps; };
__device__ __forceinline__ AdamCoef adam_coef(const float* state, const float* grad_scale, float beta1, float beta2, float eps) {
    AdamCoef c;
    const float lr = state[1], bc1 = state[2], bc2 = state[3];
    c.step_size = lr / bc1; c.inv_sqrt_bc2 = rsqrtf(bc2);
    c.gs = grad_scale ? *grad_scale : 1.0f;
    c.beta1 = beta1; c.beta2 = beta2; c.eps = eps;
    return c;
}
__device__ __forceinline__ void adam_update(float& p, float& m, float& v, float g, const AdamCoef& c) {
    const float gk = g * c.gs;
    m = __builtin_fmaf(c.beta1, m, (1.0f - c.beta1) * gk);
    v = __builtin_fmaf(c.beta2, v, ((1.0f - c.beta2) * gk) * gk);
    const float denom = __builtin_fmaf(sqrtf(v), c.inv_sqrt_bc2, c.eps);
    p = __builtin_fmaf(-c.step_size, m / denom, p);
}

// ---------------------------------------------------------------- wave reductions (64 lanes)
// sum over the 16 lanes of a DPP row (every lane gets the total)
__device__ __forceinline__ float fh_row16_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));   // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false));   // row_mirror
    return v;
}
// combine across the four 16-lane rows of the wave (lanes with equal lane & 15)
__device__ __forceinline__ float fh_cross_max(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float fh_cross_sum(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// sum over the 8 lanes of each half of a DPP row (lanes with equal lane >> 3; every lane gets its half's total)
__device__ __forceinline__ float fh_row8_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));   // row_half_mirror
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
