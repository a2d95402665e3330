// attention.hip — scaled-dot-product attention core (mtn.py:221-231), forward and backward.
// One workgroup per (batch row, head).  K/V tiles are staged through LDS (fp32, rows padded to dk+4 so that
// 16-byte reads of 16 consecutive rows hit distinct banks), scores are tiled over keys with an online
// softmax (running row max / sum, wave-shuffle reductions), masked scores take the reference's -1e9
// (a fully masked row is uniform, not NaN), probabilities are dropped out with the counter-based mask.
// Query lengths on this path are short (T,Q <= ~54 in DSTC7; 20 in the bench configs), so the contraction
// runs on the VALU with 4x-register-tiled LDS reads; the d_model x d_model projections around it are MFMA.
#include <stdlib.h>

#include "common.h"

static constexpr int AQ = 32;       // query rows per forward workgroup
static constexpr int MT_F = 64;     // keys per forward tile
static constexpr int MT_B = 32;     // keys per backward tile
static constexpr int AMAX_B = 64;   // max query rows in backward (whole query block lives in LDS)

#include "attn_mfma.h"


// ------------------------------------------------------------------------------------------ forward
struct AttnGroup {
    int count;
    mtn_attn_args a[MTN_ATTN_MAX_GROUP];
};

template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnGroup G) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const mtn_attn_args& A = G.a[blockIdx.z];
    if ((int)blockIdx.x >= A.B * A.h || (int)blockIdx.y * AQ >= A.a) return;
    const int dk = A.dk, ldr = dk + 4, a = A.a, m = A.m;
    float* Qs = sm;                    // [AQ][ldr]  (pre-scaled by 1/sqrt(dk))
    float* Os = Qs + AQ * ldr;         // [AQ][ldr]
    float* Ks = Os + AQ * ldr;         // [MT_F][ldr]
    float* Vs = Ks + MT_F * ldr;       // [MT_F][ldr]
    float* Ss = Vs + MT_F * ldr;       // [AQ][MT_F+4]
    float* mrow = Ss + AQ * (MT_F + 4);
    float* lrow = mrow + AQ;
    float* arow = lrow + AQ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / A.h, hh = blockIdx.x % A.h, q0 = blockIdx.y * AQ;
    const int dk4 = dk >> 2;
    const float scale = rsqrtf((float)dk);
    const T* qg = (const T*)A.q + (size_t)b * a * A.ldq + hh * dk;
    const T* kg = (const T*)A.k + (size_t)b * m * A.ldkv + hh * dk;
    const T* vg = (const T*)A.v + (size_t)b * m * A.ldkv + hh * dk;
    const DropState ds = drop_init(A.drop);

    for (int idx = tid; idx < AQ * dk4; idx += 256) {
        int i = idx / dk4, c = (idx % dk4) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q0 + i < a) {
            v = load4<T>(qg + (size_t)(q0 + i) * A.ldq + c);
            v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        }
        *(float4*)(Qs + i * ldr + c) = v;
        *(float4*)(Os + i * ldr + c) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid < AQ) { mrow[tid] = -INFINITY; lrow[tid] = 0.f; arow[tid] = 0.f; }

    for (int j0 = 0; j0 < m; j0 += MT_F) {
        __syncthreads();
        for (int idx = tid; idx < MT_F * dk4; idx += 256) {
            int j = idx / dk4, c = (idx % dk4) * 4;
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
            if (j0 + j < m) {
                kv = load4<T>(kg + (size_t)(j0 + j) * A.ldkv + c);
                vv = load4<T>(vg + (size_t)(j0 + j) * A.ldkv + c);
            }
            *(float4*)(Ks + j * ldr + c) = kv;
            *(float4*)(Vs + j * ldr + c) = vv;
        }
        __syncthreads();
        // scores: task = (block of 4 query rows, key j); 8*64 = 512 tasks
        for (int task = tid; task < (AQ / 4) * MT_F; task += 256) {
            const int j = task % MT_F, i0 = (task / MT_F) * 4;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            const float* kp = Ks + j * ldr;
            const float* qp = Qs + i0 * ldr;
#pragma unroll 2
            for (int c = 0; c < dk; c += 4) {
                float4 kv = *(const float4*)(kp + c);
                s0 += dot4(*(const float4*)(qp + c), kv);
                s1 += dot4(*(const float4*)(qp + ldr + c), kv);
                s2 += dot4(*(const float4*)(qp + 2 * ldr + c), kv);
                s3 += dot4(*(const float4*)(qp + 3 * ldr + c), kv);
            }
            float sv[4] = {s0, s1, s2, s3};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qi = q0 + i0 + r;
                float s = sv[r];
                if (j0 + j >= m || qi >= a) s = -INFINITY;  // tile padding: excluded from the softmax
                else if (A.mask && A.mask[(size_t)b * A.mask_sb + (size_t)qi * A.mask_sq + j0 + j] == 0) s = -1e9f;
                Ss[(i0 + r) * (MT_F + 4) + j] = s;
            }
        }
        __syncthreads();
        // online softmax: each wave owns rows wave, wave+4, ...; lane = key within the tile
        for (int i = wave; i < AQ; i += 4) {
            const float s = Ss[i * (MT_F + 4) + lane];
            const float mo = mrow[i];
            const float mn = fmaxf(mo, wave_max(s));
            float p = (mn == -INFINITY) ? 0.f : __expf(s - mn);
            const float psum = wave_sum(p);
            const float alpha = (mo == -INFINITY) ? 0.f : __expf(mo - mn);
            if (ds.on) {
                const uint64_t idx = ((uint64_t)(b * A.h + hh) * a + (q0 + i)) * (uint64_t)m + (j0 + lane);
                p = drop_keep(ds, idx) ? p * ds.scale : 0.f;
            }
            Ss[i * (MT_F + 4) + lane] = p;
            if (lane == 0) { mrow[i] = mn; lrow[i] = lrow[i] * alpha + psum; arow[i] = alpha; }
        }
        __syncthreads();
        // O = alpha*O + P V : task = (block of 4 rows, 4 columns)
        for (int task = tid; task < (AQ / 4) * dk4; task += 256) {
            const int c = (task % dk4) * 4, i0 = (task / dk4) * 4;
            float4 acc[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float4 o = *(const float4*)(Os + (i0 + r) * ldr + c);
                const float al = arow[i0 + r];
                acc[r] = make_float4(o.x * al, o.y * al, o.z * al, o.w * al);
            }
#pragma unroll 2
            for (int j = 0; j < MT_F; j += 4) {
                float4 v0 = *(const float4*)(Vs + j * ldr + c), v1 = *(const float4*)(Vs + (j + 1) * ldr + c);
                float4 v2 = *(const float4*)(Vs + (j + 2) * ldr + c), v3 = *(const float4*)(Vs + (j + 3) * ldr + c);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float4 p = *(const float4*)(Ss + (i0 + r) * (MT_F + 4) + j);
                    acc[r].x += p.x * v0.x + p.y * v1.x + p.z * v2.x + p.w * v3.x;
                    acc[r].y += p.x * v0.y + p.y * v1.y + p.z * v2.y + p.w * v3.y;
                    acc[r].z += p.x * v0.z + p.y * v1.z + p.z * v2.z + p.w * v3.z;
                    acc[r].w += p.x * v0.w + p.y * v1.w + p.z * v2.w + p.w * v3.w;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) *(float4*)(Os + (i0 + r) * ldr + c) = acc[r];
        }
    }
    __syncthreads();
    T* og = (T*)A.o + (size_t)b * a * A.ldo + hh * dk;
    for (int idx = tid; idx < AQ * dk4; idx += 256) {
        int i = idx / dk4, c = (idx % dk4) * 4;
        if (q0 + i < a) {
            const float inv = 1.0f / lrow[i];
            float4 o = *(const float4*)(Os + i * ldr + c);
            store4<T>(og + (size_t)(q0 + i) * A.ldo + c, make_float4(o.x * inv, o.y * inv, o.z * inv, o.w * inv));
        }
    }
    if (A.lse && tid < AQ && q0 + tid < a) {   // row max and row sum kept apart: max may be -1e9 (fully masked row)
        float* st = A.lse + 2 * ((size_t)(b * A.h + hh) * a + q0 + tid);
        st[0] = mrow[tid];
        st[1] = 1.0f / lrow[tid];
    }
}

// ------------------------------------------------------------------------------------------ backward
// P_ij = exp(S_ij - lse_i); dP = dO V^T (through the same dropout mask); D_i = sum_c dO_ic O_ic;
// dS = P*(dP - D), zero where the score was masked; dV = Pdrop^T dO; dQ = scale * dS K; dK = dS^T (scale*Q).
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const AttnGroup G) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const mtn_attn_args& A = G.a[blockIdx.z];
    if ((int)blockIdx.x >= A.B * A.h) return;
    const int dk = A.dk, ldr = dk + 4, a = A.a, m = A.m;
    const int ap = (a + 3) & ~3;
    float* Qs = sm;                  // [ap][ldr] pre-scaled
    float* dOs = Qs + ap * ldr;      // [ap][ldr]
    float* dQs = dOs + ap * ldr;     // [ap][ldr]
    float* Ks = dQs + ap * ldr;      // [MT_B][ldr]
    float* Vs = Ks + MT_B * ldr;     // [MT_B][ldr]
    float* Ps = Vs + MT_B * ldr;     // [ap][MT_B+4]  dropped-out probabilities
    float* dSs = Ps + ap * (MT_B + 4);  // [ap][MT_B+4]
    float* Dr = dSs + ap * (MT_B + 4);  // [ap]
    float* Ls = Dr + ap;                // [ap] row max
    float* Li = Ls + ap;                // [ap] 1/row sum
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / A.h, hh = blockIdx.x % A.h;
    const int dk4 = dk >> 2;
    const float scale = rsqrtf((float)dk);
    const T* qg = (const T*)A.q + (size_t)b * a * A.ldq + hh * dk;
    const T* kg = (const T*)A.k + (size_t)b * m * A.ldkv + hh * dk;
    const T* vg = (const T*)A.v + (size_t)b * m * A.ldkv + hh * dk;
    const T* og = (const T*)A.o + (size_t)b * a * A.ldo + hh * dk;
    const T* dog = (const T*)A.d_o + (size_t)b * a * A.ldo + hh * dk;
    const DropState ds = drop_init(A.drop);

    for (int idx = tid; idx < ap * dk4; idx += 256) {
        int i = idx / dk4, c = (idx % dk4) * 4;
        float4 qv = make_float4(0.f, 0.f, 0.f, 0.f), dv = qv;
        if (i < a) {
            qv = load4<T>(qg + (size_t)i * A.ldq + c);
            qv.x *= scale; qv.y *= scale; qv.z *= scale; qv.w *= scale;
            dv = load4<T>(dog + (size_t)i * A.ldo + c);
        }
        *(float4*)(Qs + i * ldr + c) = qv;
        *(float4*)(dOs + i * ldr + c) = dv;
        *(float4*)(dQs + i * ldr + c) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int i = wave; i < ap; i += 4) {  // D_i = sum_c dO*O, one wave per row
        float s = 0.f;
        if (i < a)
            for (int c = lane; c < dk; c += 64)
                s += LP<T>::to_f32(dog[(size_t)i * A.ldo + c]) * LP<T>::to_f32(og[(size_t)i * A.ldo + c]);
        s = wave_sum(s);
        if (lane == 0) {
            Dr[i] = s;
            const float* st = A.lse + 2 * ((size_t)(b * A.h + hh) * a + i);
            Ls[i] = (i < a) ? st[0] : 0.f;
            Li[i] = (i < a) ? st[1] : 0.f;
        }
    }

    T* dqg = (T*)A.dq + (size_t)b * a * A.ldq + hh * dk;
    T* dkg = (T*)A.dk_out + (size_t)b * m * A.ldkv + hh * dk;
    T* dvg = (T*)A.dv_out + (size_t)b * m * A.ldkv + hh * dk;

    for (int j0 = 0; j0 < m; j0 += MT_B) {
        __syncthreads();
        for (int idx = tid; idx < MT_B * dk4; idx += 256) {
            int j = idx / dk4, c = (idx % dk4) * 4;
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
            if (j0 + j < m) {
                kv = load4<T>(kg + (size_t)(j0 + j) * A.ldkv + c);
                vv = load4<T>(vg + (size_t)(j0 + j) * A.ldkv + c);
            }
            *(float4*)(Ks + j * ldr + c) = kv;
            *(float4*)(Vs + j * ldr + c) = vv;
        }
        __syncthreads();
        // P and dS for the tile: task = (block of 4 query rows, key j)
        for (int task = tid; task < (ap / 4) * MT_B; task += 256) {
            const int j = task % MT_B, i0 = (task / MT_B) * 4;
            float s[4] = {0.f, 0.f, 0.f, 0.f}, dp[4] = {0.f, 0.f, 0.f, 0.f};
            const float* kp = Ks + j * ldr;
            const float* vp = Vs + j * ldr;
#pragma unroll 2
            for (int c = 0; c < dk; c += 4) {
                float4 kv = *(const float4*)(kp + c), vv = *(const float4*)(vp + c);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[r] += dot4(*(const float4*)(Qs + (i0 + r) * ldr + c), kv);
                    dp[r] += dot4(*(const float4*)(dOs + (i0 + r) * ldr + c), vv);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qi = i0 + r;
                float pd = 0.f, dsv = 0.f;
                if (j0 + j < m && qi < a) {
                    const bool keep_score = !(A.mask && A.mask[(size_t)b * A.mask_sb + (size_t)qi * A.mask_sq + j0 + j] == 0);
                    const float sc = keep_score ? s[r] : -1e9f;
                    const float p = __expf(sc - Ls[qi]) * Li[qi];
                    float dpd = dp[r];
                    pd = p;
                    if (ds.on) {
                        const uint64_t idx = ((uint64_t)(b * A.h + hh) * a + qi) * (uint64_t)m + (j0 + j);
                        const bool kp_ = drop_keep(ds, idx);
                        pd = kp_ ? p * ds.scale : 0.f;
                        dpd = kp_ ? dpd * ds.scale : 0.f;
                    }
                    dsv = keep_score ? p * (dpd - Dr[qi]) : 0.f;
                }
                Ps[qi * (MT_B + 4) + j] = pd;
                dSs[qi * (MT_B + 4) + j] = dsv;
            }
        }
        __syncthreads();
        // dV[j] = sum_i Pd[i][j] dO[i],  dK[j] = sum_i dS[i][j] Qs[i]   : task = (key j, 4 columns)
        for (int task = tid; task < MT_B * dk4; task += 256) {
            const int c = (task % dk4) * 4, j = task / dk4;
            if (j0 + j < m) {
                float4 av = make_float4(0.f, 0.f, 0.f, 0.f), ak = av;
#pragma unroll 4
                for (int i = 0; i < ap; ++i) {
                    const float p = Ps[i * (MT_B + 4) + j], d = dSs[i * (MT_B + 4) + j];
                    float4 dv = *(const float4*)(dOs + i * ldr + c), qv = *(const float4*)(Qs + i * ldr + c);
                    av.x += p * dv.x; av.y += p * dv.y; av.z += p * dv.z; av.w += p * dv.w;
                    ak.x += d * qv.x; ak.y += d * qv.y; ak.z += d * qv.z; ak.w += d * qv.w;
                }
                store4<T>(dvg + (size_t)(j0 + j) * A.ldkv + c, av);
                store4<T>(dkg + (size_t)(j0 + j) * A.ldkv + c, ak);
            }
        }
        // dQ[i] += sum_j dS[i][j] K[j]   : task = (row i, 4 columns)
        for (int task = tid; task < ap * dk4; task += 256) {
            const int c = (task % dk4) * 4, i = task / dk4;
            float4 acc = *(const float4*)(dQs + i * ldr + c);
#pragma unroll 2
            for (int j = 0; j < MT_B; j += 4) {
                float4 d = *(const float4*)(dSs + i * (MT_B + 4) + j);
                float4 k0 = *(const float4*)(Ks + j * ldr + c), k1 = *(const float4*)(Ks + (j + 1) * ldr + c);
                float4 k2 = *(const float4*)(Ks + (j + 2) * ldr + c), k3 = *(const float4*)(Ks + (j + 3) * ldr + c);
                acc.x += d.x * k0.x + d.y * k1.x + d.z * k2.x + d.w * k3.x;
                acc.y += d.x * k0.y + d.y * k1.y + d.z * k2.y + d.w * k3.y;
                acc.z += d.x * k0.z + d.y * k1.z + d.z * k2.z + d.w * k3.z;
                acc.w += d.x * k0.w + d.y * k1.w + d.z * k2.w + d.w * k3.w;
            }
            *(float4*)(dQs + i * ldr + c) = acc;
        }
    }
    __syncthreads();
    for (int idx = tid; idx < ap * dk4; idx += 256) {
        int i = idx / dk4, c = (idx % dk4) * 4;
        if (i < a) {
            float4 v = *(const float4*)(dQs + i * ldr + c);
            store4<T>(dqg + (size_t)i * A.ldq + c, make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale));
        }
    }
}

// ====================================================================================================================
// MFMA forward.  One wave per (batch row, head, 32 query rows).  Scores are computed TRANSPOSED, S^T = K Q^T
// (A = K rows, B = Q rows, both contraction-contiguous straight from global memory as 16-byte fragments), so that in the
// MFMA C layout a lane holds, for ONE query column (lane & 15), the keys 4*(lane>>4)+r of every 16-key tile:
//   * the softmax row max / row sum are in-register reductions + two wave shuffles (xor 16, 32);
//   * P^T is already the B-operand fragment of O^T = V^T P^T (contraction slot (lane>>4, j) <-> key 16*(j>>2)+4*(lane>>4)+(j&3)),
//     no cross-lane traffic to repack it;
// V is staged through LDS transposed (4x4 in-register block transposes) so that the A-operand fragment (V^T rows = head
// columns, 4 consecutive keys per slot group) is an 8/16-byte LDS read.  fp32 mode runs the same code on the exact-fp32 MFMA.
// ====================================================================================================================

template <typename T, int DK, int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_fwd_mfma_kernel(const AttnGroup G) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];   // per wave: V^T image [DK][VT_ROW]; then the combine area
    const mtn_attn_args& A = G.a[blockIdx.z];
    if ((int)blockIdx.x >= A.B * A.h || (int)blockIdx.y * MQ >= A.a) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    attn_fwd_mfma_body<T, DK, NW>(A, blockIdx.x / A.h, blockIdx.x % A.h, blockIdx.y * MQ, wave, fsm);
}

template <typename T, int DK, int NW> static int launch_fwd_mfma_nw(const AttnGroup& G, dim3 grid, hipStream_t s) {
    const size_t vt = (size_t)DK * (MK * sizeof(T) + tpad<T>());
    if constexpr (NW == 1) {
        hipLaunchKernelGGL((attn_fwd_mfma_kernel<T, DK, 1>), grid, dim3(64), vt, s, G);
    } else {
        const size_t comb = sizeof(float) * (NW * DK * 33 + 2 * NW * 32);
        const size_t lds = NW * vt > comb ? NW * vt : comb;        // the combine area aliases the V^T images
        if (int rc = set_lds(attn_fwd_mfma_kernel<T, DK, NW>, lds)) return rc;
        hipLaunchKernelGGL((attn_fwd_mfma_kernel<T, DK, NW>), grid, dim3(64 * NW), lds, s, G);
    }
    return MTN_OK;
}
// nw = waves per workgroup = key tiles of the longest memory, capped at 4 (1, 2 or 4)
template <typename T, int DK> static int launch_fwd_mfma(const AttnGroup& G, dim3 grid, int nw, hipStream_t s) {
    if (nw <= 1) return launch_fwd_mfma_nw<T, DK, 1>(G, grid, s);
    if (nw == 2) return launch_fwd_mfma_nw<T, DK, 2>(G, grid, s);
    if (nw <= 4) return launch_fwd_mfma_nw<T, DK, 4>(G, grid, s);
    return launch_fwd_mfma_nw<T, DK, 8>(G, grid, s);
}
template <typename T> static int dispatch_fwd_mfma(int dk, const AttnGroup& G, dim3 grid, int nw, hipStream_t s) {
    switch (dk) {
        case 16: return launch_fwd_mfma<T, 16>(G, grid, nw, s);
        case 32: return launch_fwd_mfma<T, 32>(G, grid, nw, s);
        case 64: return launch_fwd_mfma<T, 64>(G, grid, nw, s);
        case 128: return launch_fwd_mfma<T, 128>(G, grid, nw, s);
    }
    return -1;
}

// ====================================================================================================================
// MFMA backward (query blocks of <= 32 rows: T,Q <= 32 — longer queries take the VALU kernel above).  One wave per
// (batch row, head), keys in tiles of 32.  Here scores are computed UN-transposed, S = Q K^T and dP = dO V^T (all four
// operands contraction-contiguous from global memory), so that in the C layout a lane holds one KEY column (lane & 15)
// and the query rows 4*(lane>>4)+r: P and dS are then directly the B-operand fragments of the two contractions over the
// query index, dV^T = dO^T Pdrop and dK^T = Q^T dS (A operands dO^T, Q^T come from transposed LDS images built once).
// dQ^T = K^T dS^T contracts over keys: dS makes one round trip through LDS ([q][key] image) to come back with the query
// index on lane & 15, and K^T is staged transposed per key tile.
// ====================================================================================================================
static constexpr int BQ = 32;   // query rows per block
static constexpr int BK = 32;   // keys per tile

template <typename T, int DK, int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_bwd_mfma_kernel(const AttnGroup G) {
    constexpr int EPV = LP<T>::EPV, KSTEP = LP<T>::KSTEP;
    constexpr int NKS = (DK + KSTEP - 1) / KSTEP;
    constexpr int NDT = DK / 16;
    constexpr int TPK = KSTEP / 16;                    // C tiles per contraction step over a 32-long index (bf16 2, fp32 1)
    constexpr int NU = 32 / KSTEP;                     // contraction steps over 32 queries / 32 keys (bf16 1, fp32 2)
    constexpr int ROWB = 32 * (int)sizeof(T) + tpad<T>();     // bytes per row of the transposed images (32 entries + pad)
    extern __shared__ __attribute__((aligned(16))) unsigned char bsm[];
    const int lane = threadIdx.x & 63, l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char* dOt = bsm;                 // [DK][ROWB]  dO^T  (row = head column, entries = query rows)   shared by the waves
    unsigned char* Qt = dOt + DK * ROWB;     // [DK][ROWB]  Q^T                                               shared
    float* Ds = (float*)(Qt + DK * ROWB);    // [32]        D_q = sum_c dO[q][c] O[q][c]                      shared
    unsigned char* Kt = (unsigned char*)(Ds + 32) + (size_t)wave * (DK + 32) * ROWB;   // [DK][ROWB] K^T of the wave's current key tile
    unsigned char* dSs = Kt + DK * ROWB;     // [32][ROWB]  dS  (row = query, entries = keys of the tile)      per wave
    const mtn_attn_args& A = G.a[blockIdx.z];
    if ((int)blockIdx.x >= A.B * A.h) return;
    const int b = blockIdx.x / A.h, hh = blockIdx.x % A.h;
    const int a = A.a, m = A.m;
    // one pass covers query rows qo .. qo+an-1 of every sample (an <= 32); sequences longer than 32 rows take several passes,
    // the later ones adding their dK / dV to what the earlier ones stored (mtn_attention_bwd_group)
    const int qo = A.q0, an = A.qn > 0 ? A.qn : (a - A.q0 < BQ ? a - A.q0 : BQ);
    const bool solo = (NW == 1) || (m <= BK);          // one key tile: one wave (see the forward kernel)
    if (NW > 1 && solo && wave > 0) return;
    const float scale = rsqrtf((float)DK);
    const T* qg = (const T*)A.q + ((size_t)b * a + qo) * A.ldq + hh * DK;
    const T* kg = (const T*)A.k + (size_t)b * m * A.ldkv + hh * DK;
    const T* vg = (const T*)A.v + (size_t)b * m * A.ldkv + hh * DK;
    const T* og = (const T*)A.o + ((size_t)b * a + qo) * A.ldo + hh * DK;
    const T* dog = (const T*)A.d_o + ((size_t)b * a + qo) * A.ldo + hh * DK;
    const DropState ds = drop_init(A.drop);
    const DropBase dbase = drop_base((uint64_t)(b * A.h + hh) * (uint64_t)a * (uint64_t)m);   // P-dropout index of (q, key) = base + q * m + key

    // ---- prologue: A-operand fragments of Q and dO (rows = queries), transposed images, D_q
    uint4 qf[2][NKS], dof[2][NKS];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        int q = qt * 16 + l15;
        q = q < an ? q : an - 1;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const bool ok = ks * KSTEP + lg * EPV < DK;
            qf[qt][ks] = load_frag<T>(qg + (size_t)q * A.ldq + ks * KSTEP + lg * EPV, ok);
            dof[qt][ks] = load_frag<T>(dog + (size_t)q * A.ldo + ks * KSTEP + lg * EPV, ok);
        }
    }
    constexpr int NBT = ((32 / 4) * (DK / 4) + 63) / 64;     // 4x4 blocks per lane for a 32-row transposed image
    if (wave == 0) {                                          // the shared images are built by wave 0 (small: 2 x 32 x DK)
        TStage<T, NBT> sdo, sq;
        sdo.load(dog, A.ldo, 0, BQ, an, DK, lane);
        sq.load(qg, A.ldq, 0, BQ, an, DK, lane);
        sdo.store(dOt, ROWB, BQ, DK, lane);
        sq.store(Qt, ROWB, BQ, DK, lane);
    }
    if (wave == 0) {   // D_q: lanes 2q and 2q+1 each sum half of row q
        int q = lane >> 1;
        const int half = lane & 1;
        float sacc = 0.f;
        if (q < an) {
            const T* dp = dog + (size_t)q * A.ldo + half * (DK / 2);
            const T* op = og + (size_t)q * A.ldo + half * (DK / 2);
#pragma unroll
            for (int c = 0; c < DK / 2; c += 4) {
                float4 x = load4<T>(dp + c), y = load4<T>(op + c);
                sacc += dot4(x, y);
            }
        }
        sacc += __shfl_xor(sacc, 1, 64);
        if (half == 0) Ds[q] = sacc;
    }
    __syncthreads();
    float mxq[2][4], invq[2][4], Dq[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = qt * 16 + 4 * lg + r;
            const int qc = q < an ? q : an - 1;
            const float* stp = A.lse + 2 * ((size_t)(b * A.h + hh) * a + qo + qc);
            mxq[qt][r] = stp[0];
            invq[qt][r] = stp[1];
            Dq[qt][r] = Ds[q];
        }
    f32x4_t dqt[NDT][2];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) dqt[dt][qt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    T* dqg = (T*)A.dq + ((size_t)b * a + qo) * A.ldq + hh * DK;
    T* dkg = (T*)A.dk_out + (size_t)b * m * A.ldkv + hh * DK;
    T* dvg = (T*)A.dv_out + (size_t)b * m * A.ldkv + hh * DK;

    for (int j0 = wave * BK; j0 < m; j0 += NW * BK) {
        // ---- all global loads of the tile first: K/V fragments, K for the transposed image, mask bytes
        uint4 kf[2][NKS], vf[2][NKS];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            int key = j0 + kt * 16 + l15;
            key = key < m ? key : m - 1;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const bool ok = ks * KSTEP + lg * EPV < DK;
                kf[kt][ks] = load_frag<T>(kg + (size_t)key * A.ldkv + ks * KSTEP + lg * EPV, ok);
                vf[kt][ks] = load_frag<T>(vg + (size_t)key * A.ldkv + ks * KSTEP + lg * EPV, ok);
            }
        }
        TStage<T, NBT> skt;
        skt.load(kg, A.ldkv, j0, BK, m, DK, lane);
        uint32_t mk[2][2];           // mask bytes of query rows 4lg..4lg+3 (tile qt) for key column kt*16+l15; 1 = keep
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            int key = j0 + kt * 16 + l15;
            key = key < m ? key : m - 1;
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                uint32_t w = 0x01010101u;
                if (A.mask) {
                    if (A.mask_sq == 0) {
                        const uint32_t bq = A.mask[(size_t)b * A.mask_sb + key] != 0;
                        w = bq * 0x01010101u;
                    } else {
                        w = 0;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            int q = qt * 16 + 4 * lg + r;
                            q = q < an ? q : an - 1;
                            w |= (uint32_t)(A.mask[(size_t)b * A.mask_sb + (size_t)(qo + q) * A.mask_sq + key] != 0) << (8 * r);
                        }
                    }
                }
                mk[qt][kt] = w;
            }
        }
        // ---- S = Q K^T, dP = dO V^T   (C layout: rows q = qt*16+4lg+r, column key = kt*16 + l15)
        f32x4_t sc[2][2], dp[2][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                sc[qt][kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                dp[qt][kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    mma16<T>(sc[qt][kt], qf[qt][ks], kf[kt][ks]);
                    mma16<T>(dp[qt][kt], dof[qt][ks], vf[kt][ks]);
                }
            }
        __builtin_amdgcn_wave_barrier();                   // wave-private Kt / dSs: in-order LDS, compiler fence only
        skt.store(Kt, ROWB, BK, DK, lane);
        // ---- P, dS in registers (sc <- dropped-out P, dp <- dS); dS also goes to LDS as [q][key]
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const int key = j0 + kt * 16 + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = qt * 16 + 4 * lg + r;
                    float pd = 0.f, dsv = 0.f;
                    if (key < m && q < an) {
                        const bool keep_score = ((mk[qt][kt] >> (8 * r)) & 0xffu) != 0;
                        const float sv = keep_score ? sc[qt][kt][r] * scale : -1e9f;
                        const float p = __expf(sv - mxq[qt][r]) * invq[qt][r];
                        float dpd = dp[qt][kt][r];
                        pd = p;
                        if (ds.on) {
                            const bool kp = drop_keep_at(ds, dbase, (uint32_t)((qo + q) * m + key));
                            pd = kp ? p * ds.scale : 0.f;
                            dpd = kp ? dpd * ds.scale : 0.f;
                        }
                        dsv = keep_score ? p * (dpd - Dq[qt][r]) : 0.f;
                    }
                    sc[qt][kt][r] = pd;
                    dp[qt][kt][r] = dsv;
                    *(T*)(dSs + (size_t)q * ROWB + (kt * 16 + l15) * sizeof(T)) = LP<T>::from_f32(dsv);
                }
            }
        // ---- dV^T = dO^T Pdrop, dK^T = Q^T dS (contraction over the 32 query rows), written per key tile
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const int key = j0 + kt * 16 + l15;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                f32x4_t av = f32x4_t{0.f, 0.f, 0.f, 0.f}, ak = av;
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const uint4 pf = frag_from_c<T>(sc[u * TPK][kt], sc[u * TPK + TPK - 1][kt]);
                    const uint4 sf = frag_from_c<T>(dp[u * TPK][kt], dp[u * TPK + TPK - 1][kt]);
                    mma16<T>(av, frag_from_tlds<T>(dOt, ROWB, dt * 16 + l15, u, lg), pf);
                    mma16<T>(ak, frag_from_tlds<T>(Qt, ROWB, dt * 16 + l15, u, lg), sf);
                }
                if (key < m) {   // lane holds rows (head columns) dt*16 + 4lg + r of key column `key`
                    T* pv = dvg + (size_t)key * A.ldkv + dt * 16 + 4 * lg;
                    T* pk = dkg + (size_t)key * A.ldkv + dt * 16 + 4 * lg;
                    float4 ov = make_float4(av[0], av[1], av[2], av[3]);
                    float4 ok_ = make_float4(ak[0] * scale, ak[1] * scale, ak[2] * scale, ak[3] * scale);
                    if (A.kv_acc && !(A.kv_last && !A.kv_accum)) {
                        // several query-block passes with an fp32 workspace: the sums of the passes stay in fp32 and are rounded
                        // to the output type once, by the last pass
                        const size_t HD = (size_t)A.h * DK;
                        float* wk = A.kv_acc + ((size_t)b * m + key) * HD + hh * DK + dt * 16 + 4 * lg;
                        float* wv = wk + (size_t)A.B * m * HD;
                        if (A.kv_accum) {
                            const float4 pv0 = *(const float4*)wv, pk0 = *(const float4*)wk;
                            ov.x += pv0.x; ov.y += pv0.y; ov.z += pv0.z; ov.w += pv0.w;
                            ok_.x += pk0.x; ok_.y += pk0.y; ok_.z += pk0.z; ok_.w += pk0.w;
                        }
                        if (A.kv_last) { store4<T>(pv, ov); store4<T>(pk, ok_); }
                        else { *(float4*)wv = ov; *(float4*)wk = ok_; }
                    } else {
                        if (A.kv_accum) {                   // a later query-block pass: add to the earlier passes' sums
                            const float4 pv0 = load4<T>(pv), pk0 = load4<T>(pk);
                            ov.x += pv0.x; ov.y += pv0.y; ov.z += pv0.z; ov.w += pv0.w;
                            ok_.x += pk0.x; ok_.y += pk0.y; ok_.z += pk0.z; ok_.w += pk0.w;
                        }
                        store4<T>(pv, ov);
                        store4<T>(pk, ok_);
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();                   // Kt and dSs are complete (same wave)
        // ---- dQ^T += K^T dS^T (contraction over the 32 keys of the tile)
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            uint4 sf[2];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) sf[qt] = frag_from_tlds<T>(dSs, ROWB, qt * 16 + l15, u, lg);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const uint4 kfT = frag_from_tlds<T>(Kt, ROWB, dt * 16 + l15, u, lg);
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) mma16<T>(dqt[dt][qt], kfT, sf[qt]);
            }
        }
    }
    // ---- dQ: lane holds dQ^T[d = dt*16 + 4lg + r][q = qt*16 + l15]
    if (solo) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int q = qt * 16 + l15;
            if (q < an) {
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    const f32x4_t v = dqt[dt][qt];
                    store4<T>(dqg + (size_t)q * A.ldq + dt * 16 + 4 * lg, make_float4(v[0] * scale, v[1] * scale, v[2] * scale, v[3] * scale));
                }
            }
        }
    } else {   // sum the waves' partial dQ^T through LDS
        __syncthreads();                                    // the partial sums re-use the waves' K^T / dS images
        float* cq = (float*)(Ds + 32);                                                        // [NW][DK][33]
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) cq[(wave * DK + dt * 16 + 4 * lg + r) * 33 + qt * 16 + l15] = dqt[dt][qt][r];
        __syncthreads();
        for (int idx = threadIdx.x; idx < 32 * (DK / 4); idx += 64 * NW) {
            const int q = idx & 31, d4 = (idx >> 5) * 4;
            if (q >= an) continue;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < NW; ++w)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += cq[(w * DK + d4 + r) * 33 + q];
            store4<T>(dqg + (size_t)q * A.ldq + d4, make_float4(v[0] * scale, v[1] * scale, v[2] * scale, v[3] * scale));
        }
    }
}

template <typename T, int DK, int NW> static int launch_bwd_mfma_nw(const AttnGroup& G, dim3 grid, hipStream_t s) {
    const size_t rowb = 32 * sizeof(T) + tpad<T>();
    const size_t shared = 2 * (size_t)DK * rowb + 32 * sizeof(float), per_wave = ((size_t)DK + 32) * rowb;
    if constexpr (NW == 1) {
        hipLaunchKernelGGL((attn_bwd_mfma_kernel<T, DK, 1>), grid, dim3(64), shared + per_wave, s, G);
    } else {
        const size_t comb = sizeof(float) * NW * DK * 33;
        const size_t lds = shared + (NW * per_wave > comb ? NW * per_wave : comb);   // the dQ combine area aliases the per-wave images
        if (int rc = set_lds(attn_bwd_mfma_kernel<T, DK, NW>, lds)) return rc;
        hipLaunchKernelGGL((attn_bwd_mfma_kernel<T, DK, NW>), grid, dim3(64 * NW), lds, s, G);
    }
    return MTN_OK;
}
template <typename T, int DK> static int launch_bwd_mfma(const AttnGroup& G, dim3 grid, int nw, hipStream_t s) {
    if (nw <= 1) return launch_bwd_mfma_nw<T, DK, 1>(G, grid, s);
    if (nw == 2) return launch_bwd_mfma_nw<T, DK, 2>(G, grid, s);
    if (nw <= 4) return launch_bwd_mfma_nw<T, DK, 4>(G, grid, s);
    return launch_bwd_mfma_nw<T, DK, 8>(G, grid, s);
}
template <typename T> static int dispatch_bwd_mfma(int dk, const AttnGroup& G, dim3 grid, int nw, hipStream_t s) {
    switch (dk) {
        case 16: return launch_bwd_mfma<T, 16>(G, grid, nw, s);
        case 32: return launch_bwd_mfma<T, 32>(G, grid, nw, s);
        case 64: return launch_bwd_mfma<T, 64>(G, grid, nw, s);
        case 128: return launch_bwd_mfma<T, 128>(G, grid, nw, s);
    }
    return -1;
}

// ------------------------------------------------------------------------------------------ host
static size_t fwd_lds_bytes(int dk) { return sizeof(float) * ((size_t)(2 * AQ + 2 * MT_F) * (dk + 4) + (size_t)AQ * (MT_F + 4) + 3 * AQ); }
static size_t bwd_lds_bytes(int a, int dk) {
    size_t ap = (a + 3) & ~3;
    return sizeof(float) * ((3 * ap + 2 * MT_B) * (size_t)(dk + 4) + 2 * ap * (MT_B + 4) + 3 * ap);
}

static int check_attn(const mtn_attn_args* A, bool bwd) {
    MTN_CHECK_ARG(A, "null args");
    MTN_CHECK_ARG(A->B > 0 && A->h > 0 && A->a > 0 && A->m > 0, "empty attention problem");
    MTN_CHECK_ARG(A->dk >= 4 && A->dk % 4 == 0 && A->dk <= 128, "dk must be a multiple of 4, <= 128");
    MTN_CHECK_ARG(A->ldq % 4 == 0 && A->ldkv % 4 == 0 && A->ldo % 4 == 0, "row strides must be multiples of 4");
    MTN_CHECK_ARG(A->q && A->k && A->v && A->o, "null tensor");
    if (bwd) {
        MTN_CHECK_ARG(A->d_o && A->dq && A->dk_out && A->dv_out && A->lse, "null backward tensor");
    }
    return MTN_OK;
}

template <typename K> static int set_lds(K kernel, size_t bytes) {
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) { mtn_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return MTN_ERR_LAUNCH; }
    }
    return MTN_OK;
}

extern "C" int mtn_attention_fwd_group(int dtype, int count, const mtn_attn_args* args, void* stream) {
    MTN_CHECK_ARG(dtype == MTN_F32 || dtype == MTN_BF16, "bad dtype");
    MTN_CHECK_ARG(count >= 1 && count <= MTN_ATTN_MAX_GROUP && args, "bad group");
    AttnGroup G;
    memset(&G, 0, sizeof(G));
    G.count = count;
    size_t lds = 0;
    int gx = 0, gy = 0;
    for (int i = 0; i < count; ++i) {
        if (int rc = check_attn(&args[i], false)) return rc;
        G.a[i] = args[i];
        size_t l = fwd_lds_bytes(args[i].dk);
        if (l > lds) lds = l;
        if (args[i].B * args[i].h > gx) gx = args[i].B * args[i].h;
        if ((args[i].a + AQ - 1) / AQ > gy) gy = (args[i].a + AQ - 1) / AQ;
    }
    hipStream_t s = (hipStream_t)stream;
    {   // MFMA path: every member has the same head size in {16,32,64,128} and 8-byte aligned rows
        bool ok = true;
        const int dk = args[0].dk;
        int gym = 0;
        for (int i = 0; i < count; ++i) {
            ok = ok && args[i].dk == dk && (dk == 16 || dk == 32 || dk == 64 || dk == 128) && args[i].ldq % 8 == 0 && args[i].ldkv % 8 == 0 && args[i].ldo % 4 == 0;
            if ((args[i].a + MQ - 1) / MQ > gym) gym = (args[i].a + MQ - 1) / MQ;
        }
        if (ok) {
            dim3 gridm(gx, gym, count);
            int nw = 1;                                 // a memory longer than one key tile: 2, 4 or 8 waves share the tiles
            int wgs = 0;
            for (int i = 0; i < count; ++i) { const int t = (args[i].m + MK - 1) / MK; nw = t > nw ? t : nw; wgs += args[i].B * args[i].h * gym; }
            // 8 waves (one or two tiles each) when the launch leaves most of the chip idle anyway (small batch x long memory)
            nw = nw <= 2 ? nw : ((nw > 4 && wgs <= 256) ? 8 : 4);
            int rc = (dtype == MTN_BF16) ? dispatch_fwd_mfma<bf16_t>(dk, G, gridm, nw, s) : dispatch_fwd_mfma<float>(dk, G, gridm, nw, s);
            if (rc == MTN_OK) { MTN_CHECK_LAUNCH(); return MTN_OK; }
        }
    }
    dim3 grid(gx, gy, count), block(256);
    if (dtype == MTN_BF16) {
        if (int rc = set_lds(attn_fwd_kernel<bf16_t>, lds)) return rc;
        hipLaunchKernelGGL((attn_fwd_kernel<bf16_t>), grid, block, lds, s, G);
    } else {
        if (int rc = set_lds(attn_fwd_kernel<float>, lds)) return rc;
        hipLaunchKernelGGL((attn_fwd_kernel<float>), grid, block, lds, s, G);
    }
    MTN_CHECK_LAUNCH();
    return MTN_OK;
}

extern "C" int mtn_attention_bwd_group(int dtype, int count, const mtn_attn_args* args, void* stream) {
    MTN_CHECK_ARG(dtype == MTN_F32 || dtype == MTN_BF16, "bad dtype");
    MTN_CHECK_ARG(count >= 1 && count <= MTN_ATTN_MAX_GROUP && args, "bad group");
    AttnGroup G;
    memset(&G, 0, sizeof(G));
    G.count = count;
    size_t lds = 0;
    int gx = 0;
    for (int i = 0; i < count; ++i) {
        if (int rc = check_attn(&args[i], true)) return rc;
        G.a[i] = args[i];
        size_t l = bwd_lds_bytes(args[i].a < AMAX_B ? args[i].a : AMAX_B, args[i].dk);      // (VALU fallback only)
        if (l > lds) lds = l;
        if (args[i].B * args[i].h > gx) gx = args[i].B * args[i].h;
    }
    MTN_CHECK_ARG(lds <= 160 * 1024, "attention backward tile does not fit LDS");
    dim3 grid(gx, 1, count), block(256);
    hipStream_t s = (hipStream_t)stream;
    {   // MFMA path: same supported head size everywhere; query rows in passes of 32 (a pass takes the members that still
        // have rows left; passes after the first add their dK / dV to the stored sums)
        bool ok = true;
        const int dk = args[0].dk;
        int amax = 0;
        for (int i = 0; i < count; ++i) {
            ok = ok && args[i].dk == dk && (dk == 16 || dk == 32 || dk == 64 || dk == 128) && args[i].ldq % 8 == 0 &&
                 args[i].ldkv % 8 == 0 && args[i].ldo % 8 == 0;
            amax = args[i].a > amax ? args[i].a : amax;
        }
        if (ok) {
            int rc = MTN_OK;
            for (int q0 = 0; q0 < amax && rc == MTN_OK; q0 += BQ) {
                AttnGroup P;
                memset(&P, 0, sizeof(P));
                int nw = 1, wgs = 0, px = 0;
                for (int i = 0; i < count; ++i) {
                    if (args[i].a <= q0) continue;
                    mtn_attn_args& t = P.a[P.count++];
                    t = args[i];
                    // the fp32 dK / dV workspace only matters when the query rows take several passes; a caller built against the
                    // round-2 header leaves the field uninitialised: never trust it for single-pass shapes, refuse a misaligned one
                    if (args[i].a <= BQ) t.kv_acc = nullptr;
                    else if (t.kv_acc && (((uintptr_t)t.kv_acc) & 15) != 0) { mtn_set_error("mtn_attention_bwd: kv_acc must be 16-byte aligned"); return MTN_ERR_ARG; }
                    t.q0 = q0; t.qn = args[i].a - q0 < BQ ? args[i].a - q0 : BQ; t.kv_accum = q0 > 0;
                    t.kv_last = q0 + BQ >= args[i].a;
                    const int kt = (t.m + BK - 1) / BK;
                    nw = kt > nw ? kt : nw;
                    wgs += t.B * t.h;
                    px = t.B * t.h > px ? t.B * t.h : px;
                }
                // 2 waves up to 8 key tiles: a 226-register wave leaves room for 2 per SIMD, and a group's one-tile members
                // (one live wave each) then share the CU with the long member instead of waiting for a second round
                nw = nw <= 1 ? 1 : (nw <= 8 ? 2 : ((wgs <= 256) ? 8 : 4));
                const dim3 pgrid(px, 1, P.count);
                rc = (dtype == MTN_BF16) ? dispatch_bwd_mfma<bf16_t>(dk, P, pgrid, nw, s) : dispatch_bwd_mfma<float>(dk, P, pgrid, nw, s);
                if (rc != MTN_OK && q0 > 0) return rc;          // (a first-pass refusal falls through to the VALU kernel)
            }
            if (rc == MTN_OK) { MTN_CHECK_LAUNCH(); return MTN_OK; }
        }
    }
    for (int i = 0; i < count; ++i)
        MTN_CHECK_ARG(args[i].a <= AMAX_B, "attention backward (VALU fallback: d_k outside {16,32,64,128}) supports at most 64 query rows per sequence");
    if (dtype == MTN_BF16) {
        if (int rc = set_lds(attn_bwd_kernel<bf16_t>, lds)) return rc;
        hipLaunchKernelGGL((attn_bwd_kernel<bf16_t>), grid, block, lds, s, G);
    } else {
        if (int rc = set_lds(attn_bwd_kernel<float>, lds)) return rc;
        hipLaunchKernelGGL((attn_bwd_kernel<float>), grid, block, lds, s, G);
    }
    MTN_CHECK_LAUNCH();
    return MTN_OK;
}

extern "C" int mtn_attention_fwd(int dtype, const mtn_attn_args* A, void* stream) { return mtn_attention_fwd_group(dtype, 1, A, stream); }
extern "C" int mtn_attention_bwd(int dtype, const mtn_attn_args* A, void* stream) { return mtn_attention_bwd_group(dtype, 1, A, stream); }
