// attn_mfma.h — the MFMA scaled-dot-product attention forward (mtn.py:221-231) as a device function, shared by the stand-alone
// attention kernel (attention.hip) and the fused projection+attention kernel (fused.hip).
#pragma once
#include "common.h"

template <typename T> __device__ __forceinline__ float4 load4(const T* p);
template <> __device__ __forceinline__ float4 load4<float>(const float* p) { return *(const float4*)p; }
template <> __device__ __forceinline__ float4 load4<bf16_t>(const bf16_t* p) {
    uint2 u = *(const uint2*)p;
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}
template <typename T> __device__ __forceinline__ void store4(T* p, float4 v);
template <> __device__ __forceinline__ void store4<float>(float* p, float4 v) { *(float4*)p = v; }
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, float4 v) {
    uint2 u;
    u.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
    u.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
    *(uint2*)p = u;
}
__device__ __forceinline__ float dot4(float4 a, float4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

static constexpr int MQ = 32;   // query rows per wave (2 MFMA column tiles)
static constexpr int MK = 64;   // keys per tile      (4 MFMA row tiles)

template <typename T> __device__ __forceinline__ uint4 load_frag(const T* base, bool valid) {
    return valid ? *(const uint4*)base : make_uint4(0, 0, 0, 0);
}

// Row pad of a transposed image and the lane -> 4x4 block map of its staging.  ds_write_b64 is served in groups of 16 contiguous
// lanes over 32 banks: with 16 column blocks per row (d_k = 64) a group takes 4 column blocks x 4 row blocks, and with a pitch of
// 8 (mod 32) bytes their 8-byte pieces (column block * 4 rows * pitch + row block * 8) fall on 16 different bank pairs; the plain
// map (16 column blocks of one row block per group) with a pitch of 16 (mod 32) put them on 2 (SQ_LDS_BANK_CONFLICT 0.6-0.76).
template <typename T> __host__ __device__ constexpr int tpad() { return sizeof(T) == 2 ? 8 : 16; }
__device__ __forceinline__ void tblk(int blk, int cb_n, int& kb, int& cb) {
    if (cb_n == 16) {
        cb = (blk & 3) | (((blk >> 4) & 3) << 2);
        kb = ((blk >> 2) & 3) | ((blk >> 6) << 2);
    } else {
        kb = blk / cb_n;
        cb = blk - kb * cb_n;
    }
}

// 4x4 block transpose staging: src rows [r0, r0+nrows) x ncols (row stride ld, rows clamped to rmax-1) -> dst[col][row]
template <typename T>
__device__ __forceinline__ void stage_transposed(unsigned char* dst, int dst_row_bytes, const T* src, int ld, int r0, int nrows,
                                                 int rmax, int ncols, int lane) {
    const int cb_n = ncols >> 2, nblk = (nrows >> 2) * cb_n;
    for (int blk = lane; blk < nblk; blk += 64) {
        int kb, cb;
        tblk(blk, cb_n, kb, cb);
        if constexpr (sizeof(T) == 2) {
            uint2 v[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                int r = r0 + kb * 4 + kk;
                r = r < rmax ? r : rmax - 1;
                v[kk] = *(const uint2*)(src + (size_t)r * ld + cb * 4);
            }
            unsigned char* p = dst + (size_t)(cb * 4) * dst_row_bytes + kb * 8;
            *(uint2*)(p) = make_uint2((v[0].x & 0xffffu) | (v[1].x << 16), (v[2].x & 0xffffu) | (v[3].x << 16));
            *(uint2*)(p + dst_row_bytes) = make_uint2((v[0].x >> 16) | (v[1].x & 0xffff0000u), (v[2].x >> 16) | (v[3].x & 0xffff0000u));
            *(uint2*)(p + 2 * dst_row_bytes) = make_uint2((v[0].y & 0xffffu) | (v[1].y << 16), (v[2].y & 0xffffu) | (v[3].y << 16));
            *(uint2*)(p + 3 * dst_row_bytes) = make_uint2((v[0].y >> 16) | (v[1].y & 0xffff0000u), (v[2].y >> 16) | (v[3].y & 0xffff0000u));
        } else {
            uint4 v[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                int r = r0 + kb * 4 + kk;
                r = r < rmax ? r : rmax - 1;
                v[kk] = *(const uint4*)(src + (size_t)r * ld + cb * 4);
            }
            unsigned char* p = dst + (size_t)(cb * 4) * dst_row_bytes + kb * 16;
            *(uint4*)(p) = make_uint4(v[0].x, v[1].x, v[2].x, v[3].x);
            *(uint4*)(p + dst_row_bytes) = make_uint4(v[0].y, v[1].y, v[2].y, v[3].y);
            *(uint4*)(p + 2 * dst_row_bytes) = make_uint4(v[0].z, v[1].z, v[2].z, v[3].z);
            *(uint4*)(p + 3 * dst_row_bytes) = make_uint4(v[0].w, v[1].w, v[2].w, v[3].w);
        }
    }
}

// Same staging split in two so the global loads can be issued ahead of the code that needs the LDS image:
// NB = blocks per lane = ceil((nrows/4)*(ncols/4)/64), compile time.
template <typename T, int NB> struct TStage {
    uint4 v[NB][sizeof(T) == 2 ? 2 : 4];
    __device__ __forceinline__ void load(const T* src, int ld, int r0, int nrows, int rmax, int ncols, int lane) {
        const int cb_n = ncols >> 2, nblk = (nrows >> 2) * cb_n;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int blk = lane + 64 * i;
            if (blk < nblk) {
                int kb, cb;
        tblk(blk, cb_n, kb, cb);
                if constexpr (sizeof(T) == 2) {
                    uint2 t[4];
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        int r = r0 + kb * 4 + kk;
                        r = r < rmax ? r : rmax - 1;
                        t[kk] = *(const uint2*)(src + (size_t)r * ld + cb * 4);
                    }
                    v[i][0] = make_uint4(t[0].x, t[0].y, t[1].x, t[1].y);
                    v[i][1] = make_uint4(t[2].x, t[2].y, t[3].x, t[3].y);
                } else {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        int r = r0 + kb * 4 + kk;
                        r = r < rmax ? r : rmax - 1;
                        v[i][kk] = *(const uint4*)(src + (size_t)r * ld + cb * 4);
                    }
                }
            }
        }
    }
    __device__ __forceinline__ void store(unsigned char* dst, int dst_row_bytes, int nrows, int ncols, int lane) const {
        const int cb_n = ncols >> 2, nblk = (nrows >> 2) * cb_n;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int blk = lane + 64 * i;
            if (blk < nblk) {
                int kb, cb;
        tblk(blk, cb_n, kb, cb);
                if constexpr (sizeof(T) == 2) {
                    const uint32_t a0 = v[i][0].x, a1 = v[i][0].y, b0 = v[i][0].z, b1 = v[i][0].w;   // rows k0 (a), k1 (b)
                    const uint32_t c0 = v[i][1].x, c1 = v[i][1].y, d0 = v[i][1].z, d1 = v[i][1].w;   // rows k2 (c), k3 (d)
                    unsigned char* p = dst + (size_t)(cb * 4) * dst_row_bytes + kb * 8;
                    *(uint2*)(p) = make_uint2((a0 & 0xffffu) | (b0 << 16), (c0 & 0xffffu) | (d0 << 16));
                    *(uint2*)(p + dst_row_bytes) = make_uint2((a0 >> 16) | (b0 & 0xffff0000u), (c0 >> 16) | (d0 & 0xffff0000u));
                    *(uint2*)(p + 2 * dst_row_bytes) = make_uint2((a1 & 0xffffu) | (b1 << 16), (c1 & 0xffffu) | (d1 << 16));
                    *(uint2*)(p + 3 * dst_row_bytes) = make_uint2((a1 >> 16) | (b1 & 0xffff0000u), (c1 >> 16) | (d1 & 0xffff0000u));
                } else {
                    unsigned char* p = dst + (size_t)(cb * 4) * dst_row_bytes + kb * 16;
                    *(uint4*)(p) = make_uint4(v[i][0].x, v[i][1].x, v[i][2].x, v[i][3].x);
                    *(uint4*)(p + dst_row_bytes) = make_uint4(v[i][0].y, v[i][1].y, v[i][2].y, v[i][3].y);
                    *(uint4*)(p + 2 * dst_row_bytes) = make_uint4(v[i][0].z, v[i][1].z, v[i][2].z, v[i][3].z);
                    *(uint4*)(p + 3 * dst_row_bytes) = make_uint4(v[i][0].w, v[i][1].w, v[i][2].w, v[i][3].w);
                }
            }
        }
    }
};

// C tiles (4 values per lane each) -> one B/A-operand fragment of a contraction step (bf16: two tiles, fp32: one)
template <typename T> __device__ __forceinline__ uint4 frag_from_c(const f32x4_t& lo, const f32x4_t& hi);
template <> __device__ __forceinline__ uint4 frag_from_c<bf16_t>(const f32x4_t& lo, const f32x4_t& hi) {
    return make_uint4((uint32_t)f32_to_bf16(lo[0]) | ((uint32_t)f32_to_bf16(lo[1]) << 16), (uint32_t)f32_to_bf16(lo[2]) | ((uint32_t)f32_to_bf16(lo[3]) << 16),
                      (uint32_t)f32_to_bf16(hi[0]) | ((uint32_t)f32_to_bf16(hi[1]) << 16), (uint32_t)f32_to_bf16(hi[2]) | ((uint32_t)f32_to_bf16(hi[3]) << 16));
}
template <> __device__ __forceinline__ uint4 frag_from_c<float>(const f32x4_t& lo, const f32x4_t&) {
    return make_uint4(__float_as_uint(lo[0]), __float_as_uint(lo[1]), __float_as_uint(lo[2]), __float_as_uint(lo[3]));
}
// transposed-LDS fragment: row `row` of the [col][key] image, contraction step u (bf16: keys 32u+4lg.. and 32u+16+4lg..; fp32: 16u+4lg..)
template <typename T> __device__ __forceinline__ uint4 frag_from_tlds(const unsigned char* img, int row_bytes, int row, int u, int lg) {
    const unsigned char* p = img + (size_t)row * row_bytes;
    if constexpr (sizeof(T) == 2) {
        uint2 a = *(const uint2*)(p + (u * 32 + 4 * lg) * 2), b = *(const uint2*)(p + (u * 32 + 16 + 4 * lg) * 2);
        return make_uint4(a.x, a.y, b.x, b.y);
    } else {
        return *(const uint4*)(p + (u * 16 + 4 * lg) * 4);
    }
}

// NW = waves per workgroup: 1, or 4 with the key tiles dealt round-robin to the waves (long memories: the per-tile chain of
// dependent memory round trips runs 4-wide) and the partial (max, sum, O) combined through LDS at the end.
// `wave` = this wave's index among the NW waves that share the item (b, hh, q0); `fsm` = LDS scratch of that set of waves
// (NW * DK * VT_ROW bytes, or the combine area if larger); with NW > 1 every wave of the set must make the call (the combine
// uses __syncthreads, so the set is the whole workgroup).  Returns early for waves that have nothing to do.
template <typename T, int DK, int NW>
__device__ __forceinline__ void attn_fwd_mfma_body(const mtn_attn_args& A, const int b, const int hh, const int q0, const int wave,
                                                   unsigned char* fsm) {
    constexpr int EPV = LP<T>::EPV, KSTEP = LP<T>::KSTEP;
    constexpr int NKS = (DK + KSTEP - 1) / KSTEP;     // contraction steps over the head dimension
    constexpr int NDT = DK / 16;                      // 16-column tiles of the head dimension
    constexpr int TPK = KSTEP / 16;                   // 16-key C tiles per contraction step over keys (bf16 2, fp32 1)
    constexpr int NU = MK / KSTEP;                    // contraction steps per key tile
    constexpr int VT_ROW = MK * (int)sizeof(T) + tpad<T>();  // bytes per row of the transposed V image
    const int lane = threadIdx.x & 63, l15 = lane & 15, lg = lane >> 4;
    unsigned char* vt = fsm + (size_t)wave * DK * VT_ROW;
    const int a = A.a, m = A.m;
    // a member whose keys fit one tile needs one wave: the other waves of the set leave at once and wave 0 finishes without
    // the combine
    const bool solo = (NW == 1) || (m <= MK);
    if (NW > 1 && solo && wave > 0) return;
    const float scale = rsqrtf((float)DK);
    const T* qg = (const T*)A.q + (size_t)b * a * A.ldq + hh * DK;
    const T* kg = (const T*)A.k + (size_t)b * m * A.ldkv + hh * DK;
    const T* vg = (const T*)A.v + (size_t)b * m * A.ldkv + hh * DK;
    const DropState ds = drop_init(A.drop);
    const DropBase dbase = drop_base((uint64_t)(b * A.h + hh) * (uint64_t)a * (uint64_t)m);   // P-dropout index of (q, key) = base + q * m + key

    uint4 qf[2][NKS];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        int q = q0 + qt * 16 + l15;
        q = q < a ? q : a - 1;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[qt][ks] = load_frag<T>(qg + (size_t)q * A.ldq + ks * KSTEP + lg * EPV, ks * KSTEP + lg * EPV < DK);
    }
    f32x4_t ot[NDT][2];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) ot[dt][qt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float mrun[2] = {-INFINITY, -INFINITY}, lrun[2] = {0.f, 0.f};

    constexpr int NBV = ((MK / 4) * (DK / 4) + 63) / 64;
    for (int j0 = wave * MK; j0 < m; j0 += NW * MK) {
        // ---- all global loads of the tile go out first: K fragments, V (for the transposed image), mask bytes
        uint4 kf[4][NKS];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            int key = j0 + kt * 16 + l15;
            key = key < m ? key : m - 1;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) kf[kt][ks] = load_frag<T>(kg + (size_t)key * A.ldkv + ks * KSTEP + lg * EPV, ks * KSTEP + lg * EPV < DK);
        }
        TStage<T, NBV> vst;
        vst.load(vg, A.ldkv, j0, MK, m, DK, lane);
        uint32_t mk[2][4];           // 4 mask bytes (keys 4lg..4lg+3 of tile kt) per query column; 1 = keep
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            int q = q0 + qt * 16 + l15;
            q = q < a ? q : a - 1;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                uint32_t w = 0x01010101u;
                if (A.mask && (qt == 0 || A.mask_sq != 0)) {
                    w = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int key = j0 + kt * 16 + 4 * lg + r;
                        key = key < m ? key : m - 1;
                        w |= (uint32_t)(A.mask[(size_t)b * A.mask_sb + (size_t)q * A.mask_sq + key] != 0) << (8 * r);
                    }
                } else if (A.mask) {
                    w = mk[0][kt];   // key-padding mask: same for every query row
                }
                mk[qt][kt] = w;
            }
        }
        // ---- S^T tile = K Q^T
        f32x4_t st[4][2];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                st[kt][qt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) mma16<T>(st[kt][qt], kf[kt][ks], qf[qt][ks]);
            }
        // ---- V^T image for this key tile (wave-private LDS: LDS operations of one wave complete in order, only the
        //      compiler needs a fence)
        __builtin_amdgcn_wave_barrier();
        vst.store(vt, VT_ROW, MK, DK, lane);
        // ---- mask, scale, online softmax (per query column = per lane&15)
        float alpha[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            int q = q0 + qt * 16 + l15;
            const int qc = q < a ? q : a - 1;
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = j0 + kt * 16 + 4 * lg + r;
                    float sv = st[kt][qt][r] * scale;
                    if (key >= m) sv = -INFINITY;
                    else if (((mk[qt][kt] >> (8 * r)) & 0xffu) == 0) sv = -1e9f;
                    st[kt][qt][r] = sv;
                    mx = fmaxf(mx, sv);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mn = fmaxf(mrun[qt], mx);
            float psum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float pv = __expf(st[kt][qt][r] - mn);     // exp(-inf) = 0 for tile padding
                    psum += pv;
                    if (ds.on) {
                        const int key = j0 + kt * 16 + 4 * lg + r;
                        pv = drop_keep_at(ds, dbase, (uint32_t)(qc * m + key)) ? pv * ds.scale : 0.f;
                    }
                    st[kt][qt][r] = pv;
                }
            psum += __shfl_xor(psum, 16, 64);
            psum += __shfl_xor(psum, 32, 64);
            alpha[qt] = (mrun[qt] == -INFINITY) ? 0.f : __expf(mrun[qt] - mn);
            lrun[qt] = lrun[qt] * alpha[qt] + psum;
            mrun[qt] = mn;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- O^T = alpha * O^T + V^T P^T
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                ot[dt][qt][0] *= alpha[qt]; ot[dt][qt][1] *= alpha[qt]; ot[dt][qt][2] *= alpha[qt]; ot[dt][qt][3] *= alpha[qt];
            }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            uint4 pf[2];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) pf[qt] = frag_from_c<T>(st[u * TPK][qt], st[u * TPK + TPK - 1][qt]);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const uint4 vf = frag_from_tlds<T>(vt, VT_ROW, dt * 16 + l15, u, lg);
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) mma16<T>(ot[dt][qt], vf, pf[qt]);
            }
        }
    }
    // ---- epilogue: lane holds O^T[dcol = dt*16 + 4*lg + r][q = qt*16 + l15]
    T* og = (T*)A.o + (size_t)b * a * A.ldo + hh * DK;
    if (solo) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int q = q0 + qt * 16 + l15;
            if (q < a) {
                const float inv = 1.0f / lrun[qt];
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    const f32x4_t o = ot[dt][qt];
                    store4<T>(og + (size_t)q * A.ldo + dt * 16 + 4 * lg, make_float4(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv));
                }
                if (A.lse && lg == 0) {
                    float* stp = A.lse + 2 * ((size_t)(b * A.h + hh) * a + q);
                    stp[0] = mrun[qt];
                    stp[1] = inv;
                }
            }
        }
    } else {
        // partial results of the NW waves -> LDS, then every thread combines a few (q, 4 columns) entries
        // (the combine area re-uses the waves' V^T images: every wave must be done with its image first)
        __syncthreads();
        float* cO = (float*)fsm;                                   // [NW][DK][33]
        float* cm = cO + NW * DK * 33;                             // [NW][32]
        float* cl = cm + NW * 32;                                  // [NW][32]
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) cO[(wave * DK + dt * 16 + 4 * lg + r) * 33 + qt * 16 + l15] = ot[dt][qt][r];
            if (lg == 0) { cm[wave * 32 + qt * 16 + l15] = mrun[qt]; cl[wave * 32 + qt * 16 + l15] = lrun[qt]; }
        }
        __syncthreads();
        for (int idx = wave * 64 + lane; idx < 32 * (DK / 4); idx += 64 * NW) {
            const int ql = idx & 31, d4 = (idx >> 5) * 4;
            const int q = q0 + ql;
            if (q >= a) continue;
            float M = cm[ql];
#pragma unroll
            for (int w = 1; w < NW; ++w) M = fmaxf(M, cm[w * 32 + ql]);
            float den = 0.f, num[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const float f = __expf(cm[w * 32 + ql] - M);          // waves without a tile hold max = -inf -> factor 0
                den += f * cl[w * 32 + ql];
#pragma unroll
                for (int r = 0; r < 4; ++r) num[r] += f * cO[(w * DK + d4 + r) * 33 + ql];
            }
            const float inv = 1.0f / den;
            store4<T>(og + (size_t)q * A.ldo + d4, make_float4(num[0] * inv, num[1] * inv, num[2] * inv, num[3] * inv));
            if (A.lse && d4 == 0) {
                float* stp = A.lse + 2 * ((size_t)(b * A.h + hh) * a + q);
                stp[0] = M;
                stp[1] = inv;
            }
        }
    }
}
