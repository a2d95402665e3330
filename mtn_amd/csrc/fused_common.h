// fused_common.h — constants and device helpers shared by the fused forward (fused.hip) and backward (fused_bwd.hip) kernels of a
// sublayer group: bf16, d_model = 512, d_k = 64, 512-thread workgroups of 8 waves.
#pragma once
#include <stdlib.h>

#include "common.h"

static constexpr int FH_D = 512;          // d_model
static constexpr int FH_DK = 64;          // head width
static constexpr int FH_ROWB = FH_D * 2;  // bytes per row of the xn / memory images
static constexpr int FH_HROWB = FH_DK * 2;  // bytes per row of the Q / K / V head images
static constexpr int FH_MASKB = 8;        // mask bytes a thread stages at most (4096 per 512-thread workgroup)
static constexpr int FH_THREADS = 512;    // 8 waves: column block (wave & 3) x half of the contraction (wave >> 2)
enum { FH_SELF = 0, FH_CROSS_READY = 1, FH_CROSS_RAW = 2, FH_FFN = 3 };
#define FH_MAX_MEMBERS (2 * MTN_SUBLAYER_MAX_GROUP)


typedef __attribute__((address_space(3))) void fh_lds_void_t;

// LDS-DMA issued by INLINE ASM (round 3).  With the builtin the compiler knows that LDS-DMA is outstanding and puts
// s_waitcnt vmcnt(0) in front of every LDS read it can see afterwards (it cannot tell the image being filled from the bytes being
// read) — in the fused forward kernel that drained the weight fragments before the LayerNorm could read its gains from LDS.  Issued
// by asm the compiler does not see it: every reader of a DMA'd image sits behind an explicit s_waitcnt vmcnt + barrier in the
// source (and the compiler's own counted waits for register loads only get more conservative: loads return in order).
// Buffer resource words: base[31:0], base[47:32] (stride 0), num_records (bytes), flags 0x00020000 — as make_buffer_rsrc builds them.
typedef __attribute__((ext_vector_type(4))) int fh_rsrc_t;
__device__ __forceinline__ fh_rsrc_t fh_make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)(size_t)base;
    fh_rsrc_t r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}
// 64 lanes x 16 bytes -> LDS bytes [lds_addr, lds_addr + 1024) (lds_addr wave-uniform); voff = byte offset per lane, >= num_records reads zeros
// (m0 is on the clobber list so that the compiler re-materialises it if it ever holds something there; clang warns that it does not
//  PRESERVE reserved registers across an asm statement, which is exactly the contract wanted here)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void fh_dma16(fh_rsrc_t rsrc, unsigned lds_addr, unsigned voff) {
    lds_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);        // an "s" operand is not made uniform for us
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc) : "memory", "m0");
}
#pragma clang diagnostic pop
typedef __attribute__((ext_vector_type(2))) __bf16 fh_bf16x2;
typedef __attribute__((ext_vector_type(2))) float fh_f32x2;

__device__ __forceinline__ uint32_t fh_pack2(float a, float b) {        // v_cvt_pk_bf16_f32: round-to-nearest-even
    const fh_bf16x2 r = __builtin_convertvector(fh_f32x2{a, b}, fh_bf16x2);
    return *(const uint32_t*)&r;
}
// (fh_row16_sum, fh_cross_max, fh_cross_sum: common.h)

__device__ __forceinline__ uint4 fh_xfrag(const unsigned char* img, int row, int chunk) {          // [row][512] image
    return *(const uint4*)(img + row * FH_ROWB + ((chunk ^ (row & 15)) << 4));
}
__device__ __forceinline__ uint4 fh_hfrag(const unsigned char* img, int row, int chunk) {          // [row][64] image, Q / K
    return *(const uint4*)(img + row * FH_HROWB + ((chunk ^ (row & 7)) << 4));
}
// A-operand fragment of V^T from the row-major V image: head columns n_off + (lane & 15), keys row0 + 8*lg .. +7
// (ds_read_b64_tr_b16, semantics as in gemm.hip ttd_frag; 16-byte slots swizzled with (row >> 1) & 7)
__device__ __forceinline__ uint4 fh_vfrag(const unsigned char* img, int row0, int n_off, int l15, int lg) {
    uint4 f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int krow = row0 + 8 * lg + 4 * r + (l15 >> 2);
        const int col = n_off + 4 * (l15 & 3);
        const int slot = (col >> 3) ^ ((krow >> 1) & 7);
        const unsigned addr = (unsigned)(size_t)(img + krow * FH_HROWB + slot * 16 + (col & 7) * 2);
        unsigned long long v;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
        if (r == 0) { f.x = (unsigned)v; f.y = (unsigned)(v >> 32); }
        else { f.z = (unsigned)v; f.w = (unsigned)(v >> 32); }
    }
    return f;
}

// Counted waits of the fused kernels.  They wait with `s_waitcnt vmcnt(N)`, N = the vector-memory operations the compiler emits TODAY behind
// the LDS-DMA images / row loads a stage needs (each site spells its count out and static_asserts what it depends on).  -DMTN_SAFE_WAITS
// turns every one of them into a full wait: the twin library tests/test_counted_waits_gpu.py compares the shipped one with, bit for bit.
#ifdef MTN_SAFE_WAITS
#define FH_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define FH_WAIT_VM_LGKM0(n) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#else
#define FH_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define FH_WAIT_VM_LGKM0(n) asm volatile("s_waitcnt vmcnt(" #n ") lgkmcnt(0)" ::: "memory")
#endif
