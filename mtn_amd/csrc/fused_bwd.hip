// fused_bwd.hip — backward twin of fused.hip for the attention members of a sublayer group: ONE launch for
//     dO = dy W_o (this head's 64 columns)            (gradient of MultiHeadedAttention.linears[-1], mtn.py:267)
//     dq, dk, dv = attention backward of the head      (autograd of attention(), mtn.py:221-231)
// instead of a grouped GEMM launch (dO for all heads, through HBM) followed by the attention-backward launch.  The gradient of
// the input projections (dq|dk|dv -> d LayerNorm-out, a contraction over ALL heads) stays a grouped GEMM, as does everything
// after it.  bf16, d_model = 512, d_k = 64, at most 32 query rows per sample (longer sequences keep the two-launch path).
//
// Same construction as the forward kernel: a 512-thread workgroup = (member, block of whole samples, head) issues
// everything it reads up front — dy rows (bf16, [row][512]) and the head's q, k, v, o rows (128 bytes each) by LDS-DMA, mask bytes,
// the 64 rows of W_o^T it needs as coalesced loads (8 waves = 4 column blocks x 2 halves of the contraction, fragments put in
// MFMA operand order with ds_bpermute) — then works on chip: dO on mfma_f32_16x16x32_bf16 (halves met through LDS) -> bf16 LDS
// image (dO never goes to HBM), then one wave per sample runs the attention backward over key tiles of 32:
//     S = Q K^T, dP = dO V^T (row-major images, A rows permuted so that P's C layout is the standard B-operand slot order),
//     P from the saved {row max, 1 / row sum}, dropout regenerated, dS = P (dP - D),
//     dV^T = dO^T P, dK^T = Q^T dS (A operands by ds_read_b64_tr_b16 from the row-major images), dQ^T += K^T dS^T (dS through LDS).
#include "fused_common.h"

#define FB_MAX_MEMBERS MTN_SUBLAYER_MAX_GROUP

struct FbMember {
    int rows, rows_per_wg;     // query rows (B * a), rows per workgroup (blk * a)
    int a, m, blk, mt;         // query rows / memory rows per sample, samples per workgroup, row tiles of the workgroup
    int hg, sg;                // XCD map (fused.hip)
    int self_attn;             // k, v rows are the query rows' own (packed qkv buffer)
    int ldq, ldkv;             // row strides (elements) of q / dq and of k, v / dk, dv
    const bf16_t* dyl;         // [rows, 512] gradient entering the dropped-out branch
    const bf16_t* wot;         // W_o^T [512, 512]
    const bf16_t* q;
    const bf16_t* k;
    const bf16_t* v;
    const bf16_t* o;           // [rows, 512]
    const float* lse;
    const uint8_t* mask;
    long mask_sb, mask_sq;
    mtn_dropout drop;
    bf16_t* dq;
    bf16_t* dk;
    bf16_t* dv;
};
struct FbGroup {
    int count;
    int wg_start[FB_MAX_MEMBERS + 1];
    FbMember m[FB_MAX_MEMBERS];
};

static constexpr int FB_DSROW = 80;        // bytes per row of a wave's dS tile image [32 queries][32 keys] (+ 16 pad)
static constexpr int FB_WAVE_SCRATCH = 32 * FB_DSROW + 32 * 4;

struct FbLds { int dy, qi, oi, doi, ki, vi, mask, scratch, total; };
__host__ __device__ inline FbLds fb_lds_map(int mt, int key_rows, int mask_bytes) {
    FbLds L;
    L.dy = 0;                                              // also the exchange area of the two contraction halves (4 * mt KiB)
    L.qi = mt * 16 * FH_ROWB;
    L.oi = L.qi + mt * 16 * FH_HROWB;
    L.doi = L.oi + mt * 16 * FH_HROWB;
    L.ki = L.doi + mt * 16 * FH_HROWB;
    const int krows = ((key_rows + 7) & ~7) + 32;          // + one key tile of finite padding
    L.vi = L.ki + krows * FH_HROWB;
    L.mask = L.vi + krows * FH_HROWB;
    L.total = L.mask + ((mask_bytes + 15) & ~15);
    L.scratch = L.dy;                                      // the waves' dS tile images reuse the dy image (dead once dO exists): 8 x 2.6 KiB
    return L;
}

// 128-byte head rows [nrows_valid of nrows_total] -> LDS image by LDS-DMA; rows past the end arrive as zeros
__device__ __forceinline__ void fb_dma_head_rows(const bf16_t* base, int ld_elems, int valid, int total, unsigned char* img, int wave, int lane) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (valid - 1) * ld_elems * 2 + FH_HROWB, 0x00020000);
    const int ninst = (total + 7) >> 3;
    for (int i = wave; i < ninst; i += 8) {
        const int row = i * 8 + (lane >> 3), slot = lane & 7;
        const unsigned vo = row < valid ? (unsigned)row * (unsigned)(ld_elems * 2) + (unsigned)((slot ^ (row & 7)) << 4) : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (fh_lds_void_t*)(img + i * 1024), 16, vo, 0, 0, 0);
    }
}
// A-operand fragment of X^T from a row-major [row][64] image swizzled with row & 7: head columns n_off + (lane & 15),
// rows row0 + 8*lg .. +7 (ds_read_b64_tr_b16)
__device__ __forceinline__ uint4 fb_tfrag(const unsigned char* img, int row0, int n_off, int l15, int lg) {
    uint4 f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int krow = row0 + 8 * lg + 4 * r + (l15 >> 2);
        const int col = n_off + 4 * (l15 & 3);
        const int slot = (col >> 3) ^ (krow & 7);
        const unsigned addr = (unsigned)(size_t)(img + krow * FH_HROWB + slot * 16 + (col & 7) * 2);
        unsigned long long v;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
        if (r == 0) { f.x = (unsigned)v; f.y = (unsigned)(v >> 32); }
        else { f.z = (unsigned)v; f.w = (unsigned)(v >> 32); }
    }
    return f;
}
__device__ __forceinline__ uint4 fb_frag_from_c(const f32x4_t& lo, const f32x4_t& hi) {
    return make_uint4(fh_pack2(lo[0], lo[1]), fh_pack2(lo[2], lo[3]), fh_pack2(hi[0], hi[1]), fh_pack2(hi[2], hi[3]));
}
__device__ __forceinline__ float fb_dot8(const uint4& x, const uint4& y) {
    float s = 0.f;
    const uint32_t xa[4] = {x.x, x.y, x.z, x.w}, ya[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
    for (int i = 0; i < 4; ++i)
        s += __uint_as_float(xa[i] << 16) * __uint_as_float(ya[i] << 16) + __uint_as_float(xa[i] & 0xffff0000u) * __uint_as_float(ya[i] & 0xffff0000u);
    return s;
}

template <int MT>
__device__ __forceinline__ void fb_body(const FbMember& M, const int slice, const int rb, unsigned char* smem) {
    const int tid = threadIdx.x;
    const int row0 = rb * M.rows_per_wg;
    const int R = (M.rows - row0) < M.rows_per_wg ? (M.rows - row0) : M.rows_per_wg;
    const int lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave & 3, kh = wave >> 2;
    const int a = M.a, nsamp = R / a, b0 = rb * M.blk;
    const int mk = M.self_attn ? a : M.m;                  // keys (= image rows) per sample
    const int Kr = nsamp * mk;                             // key rows of the block
    const size_t krow_g0 = M.self_attn ? (size_t)row0 : (size_t)b0 * M.m;     // first key row of the block in k / v / dk / dv
    const int qa = M.mask_sq ? a : 1;
    const int mask_bytes = M.mask ? nsamp * qa * mk : 0;
    const FbLds L = fb_lds_map(MT, M.self_attn ? MT * 16 : M.blk * M.m, M.mask ? M.blk * qa * mk : 0);
    unsigned char* dy_s = smem + L.dy;
    unsigned char* qi_s = smem + L.qi;
    unsigned char* oi_s = smem + L.oi;
    unsigned char* doi_s = smem + L.doi;
    unsigned char* ki_s = smem + L.ki;
    unsigned char* vi_s = smem + L.vi;
    unsigned char* mk_s = smem + L.mask;
    unsigned char* ds_s = smem + L.scratch + wave * FB_WAVE_SCRATCH;      // this wave's dS tile image
    float* Ds = (float*)(ds_s + 32 * FB_DSROW);                           // ... and D_q of its sample

    // ================================================================ everything this workgroup reads, issued now
    uint8_t mkb[FH_MASKB];
    const uint8_t* mask_g = M.mask ? M.mask + (size_t)b0 * M.mask_sb : nullptr;
#pragma unroll
    for (int i = 0; i < FH_MASKB; ++i) {
        const int idx = tid + FH_THREADS * i;
        mkb[i] = 1;
        if (idx < mask_bytes) mkb[i] = mask_g[M.mask_sb ? idx : idx % (qa * mk)];
    }
    {   // dy rows [R][512] (rows past R: zeros), 16-byte slots swizzled with row & 15
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(M.dyl + (size_t)row0 * FH_D), 0, R * FH_ROWB, 0x00020000);
        for (int r = wave; r < MT * 16; r += 8) {
            const unsigned vo = r < R ? (unsigned)r * FH_ROWB + (unsigned)((lane ^ (r & 15)) << 4) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (fh_lds_void_t*)(dy_s + r * FH_ROWB), 16, vo, 0, 0, 0);
        }
    }
    fb_dma_head_rows(M.q + (size_t)row0 * M.ldq + slice * FH_DK, M.ldq, R, MT * 16, qi_s, wave, lane);
    fb_dma_head_rows(M.o + (size_t)row0 * FH_D + slice * FH_DK, FH_D, R, MT * 16, oi_s, wave, lane);
    fb_dma_head_rows(M.k + krow_g0 * M.ldkv + slice * FH_DK, M.ldkv, Kr, ((Kr + 7) & ~7) + 32, ki_s, wave, lane);
    fb_dma_head_rows(M.v + krow_g0 * M.ldkv + slice * FH_DK, M.ldkv, Kr, ((Kr + 7) & ~7) + 32, vi_s, wave, lane);
    // W_o^T rows slice*64 + 16wc .. +15, this wave's half of the contraction; coalesced: lane 4r + c reads (row r, chunk c)
    uint4 wf[8];
    {
        const bf16_t* wrow = M.wot + (size_t)(slice * FH_DK + 16 * wc + (lane >> 2)) * FH_D + kh * 256 + (lane & 3) * 8;
#pragma unroll
        for (int s = 0; s < 8; ++s) wf[s] = *(const uint4*)(wrow + s * 32);
    }
    // softmax statistics {row max, 1 / row sum} of this wave's first sample (queries 8lg + 4qt + r): in flight with everything else
    float mxq[2][4], invq[2][4];
    {
        const int si0 = wave < nsamp ? wave : 0;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = 8 * lg + 4 * qt + r, qc = q < a ? q : a - 1;
                const float2 st = *(const float2*)(M.lse + 2 * ((size_t)((b0 + si0) * (FH_D / FH_DK) + slice) * a + qc));
                mxq[qt][r] = st.x;
                invq[qt][r] = st.y;
            }
    }
    const DropState ds = drop_init(M.drop);

    // ================================================================ on chip from here
#pragma unroll
    for (int i = 0; i < FH_MASKB; ++i) {
        const int idx = tid + FH_THREADS * i;
        if (idx < mask_bytes) mk_s[idx] = mkb[i];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // LDS-DMA images and weight fragments have landed
    __syncthreads();
    {
        const int src = (4 * l15 + lg) * 4;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            wf[s].x = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)wf[s].x);
            wf[s].y = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)wf[s].y);
            wf[s].z = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)wf[s].z);
            wf[s].w = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)wf[s].w);
        }
    }
    // ---- dO[row][16wc .. +15] over this wave's half of the contraction
    f32x4_t acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) mma16<bf16_t>(acc[mt], wf[s], fh_xfrag(dy_s, mt * 16 + l15, (kh * 8 + s) * 4 + lg));
    __syncthreads();                                            // everybody is past the dy image: it becomes the exchange area
    {
        float* ex = (float*)dy_s + (size_t)wc * (MT * 256) + lane * 4;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            if ((mt & 1) != kh) *(f32x4_t*)(ex + mt * 256) = acc[mt];
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            if ((mt & 1) == kh) {
                const f32x4_t o = *(const f32x4_t*)(ex + mt * 256);
                const int r = mt * 16 + l15;
                const uint2 u = make_uint2(fh_pack2(acc[mt][0] + o[0], acc[mt][1] + o[1]), fh_pack2(acc[mt][2] + o[2], acc[mt][3] + o[3]));
                *(uint2*)(doi_s + r * FH_HROWB + (((2 * wc + (lg >> 1)) ^ (r & 7)) << 4) + (lg & 1) * 8) = u;
            }
    }
    __syncthreads();                                            // dO image complete; the exchange area (in the dy image) is dead too

    // ---- attention backward, one wave per sample of the block
    const float scale = 0.125f;
    for (int si = wave; si < nsamp; si += 8) {
        const int b = b0 + si, qrow0 = si * a, krow0 = si * mk;
        // D_q = sum_c dO[q][c] O[q][c]: lanes 2q, 2q+1 each take half a row
        {
            const int q = lane >> 1, half = lane & 1, row = qrow0 + q;
            float sacc = 0.f;
            if (q < a) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    sacc += fb_dot8(fh_hfrag(doi_s, row, half * 4 + c), fh_hfrag(oi_s, row, half * 4 + c));
            }
            sacc += __shfl_xor(sacc, 1, 64);
            if (half == 0) Ds[q] = sacc;
        }
        // A fragments of Q and dO: accumulator row i of query tile qt <-> query 8(i/4) + 4qt + (i%4), so that a lane's C values of the
        // two tiles are queries 8lg + 0..7 — the B-operand slot order of the contractions over the query index
        uint4 qf[2][2], dof[2][2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            int q = 8 * (l15 >> 2) + 4 * qt + (l15 & 3);
            q = q < a ? q : a - 1;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                qf[qt][ks] = fh_hfrag(qi_s, qrow0 + q, ks * 4 + lg);
                dof[qt][ks] = fh_hfrag(doi_s, qrow0 + q, ks * 4 + lg);
            }
        }
        __builtin_amdgcn_wave_barrier();
        float Dq[2][4];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = 8 * lg + 4 * qt + r, qc = q < a ? q : a - 1;
                if (si != wave) {                                  // (blocks of more than 8 samples: later rounds load theirs here)
                    const float2 st = *(const float2*)(M.lse + 2 * ((size_t)(b * (FH_D / FH_DK) + slice) * a + qc));
                    mxq[qt][r] = st.x;
                    invq[qt][r] = st.y;
                }
                Dq[qt][r] = Ds[q < 32 ? q : 31];
            }
        f32x4_t dqt[4][2];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) dqt[nt][qt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

        for (int j0 = 0; j0 < mk; j0 += 32) {
            uint4 kf[2][2], vf[2][2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                int key = j0 + kt * 16 + l15;
                key = key < mk ? key : mk - 1;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    kf[kt][ks] = fh_hfrag(ki_s, krow0 + key, ks * 4 + lg);
                    vf[kt][ks] = fh_hfrag(vi_s, krow0 + key, ks * 4 + lg);
                }
            }
            // S = Q K^T, dP = dO V^T   (C layout: rows q = 8lg + 4qt + r, column key = j0 + 16kt + l15)
            f32x4_t sc[2][2], dp[2][2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    sc[qt][kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    dp[qt][kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        mma16<bf16_t>(sc[qt][kt], qf[qt][ks], kf[kt][ks]);
                        mma16<bf16_t>(dp[qt][kt], dof[qt][ks], vf[kt][ks]);
                    }
                }
            // P, dS in registers (sc <- dropped-out P, dp <- dS); dS also goes to LDS as [q][key]
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    const int key = j0 + kt * 16 + l15;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int q = 8 * lg + 4 * qt + r;
                        float pd = 0.f, dsv = 0.f;
                        if (key < mk && q < a) {
                            const bool keep_score = !M.mask || mk_s[(size_t)(si * qa + (M.mask_sq ? q : 0)) * mk + key] != 0;
                            const float sv = keep_score ? sc[qt][kt][r] * scale : -1e9f;
                            const float p = __expf(sv - mxq[qt][r]) * invq[qt][r];
                            float dpd = dp[qt][kt][r];
                            pd = p;
                            if (ds.on) {
                                const uint64_t idx = ((uint64_t)(b * (FH_D / FH_DK) + slice) * a + q) * (uint64_t)mk + key;
                                const bool kp = drop_keep(ds, idx);
                                pd = kp ? p * ds.scale : 0.f;
                                dpd = kp ? dpd * ds.scale : 0.f;
                            }
                            dsv = keep_score ? p * (dpd - Dq[qt][r]) : 0.f;
                        }
                        sc[qt][kt][r] = pd;
                        dp[qt][kt][r] = dsv;
                        *(bf16_t*)(ds_s + q * FB_DSROW + (kt * 16 + l15) * 2) = f32_to_bf16(dsv);
                    }
                }
            // dV^T = dO^T Pdrop, dK^T = Q^T dS (contraction over the 32 query rows), written per key tile
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const int key = j0 + kt * 16 + l15;
                const uint4 pf = fb_frag_from_c(sc[0][kt], sc[1][kt]);
                const uint4 sf = fb_frag_from_c(dp[0][kt], dp[1][kt]);
#pragma unroll
                for (int nh = 0; nh < 2; ++nh) {                  // head columns in two halves of 32 (register pressure)
                    uint4 dot_[2], qt_[2];
#pragma unroll
                    for (int n2 = 0; n2 < 2; ++n2) {
                        dot_[n2] = fb_tfrag(doi_s, qrow0, (2 * nh + n2) * 16, l15, lg);
                        qt_[n2] = fb_tfrag(qi_s, qrow0, (2 * nh + n2) * 16, l15, lg);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int n2 = 0; n2 < 2; ++n2) {
                        const int nt = 2 * nh + n2;
                        f32x4_t av = f32x4_t{0.f, 0.f, 0.f, 0.f}, ak = av;
                        mma16<bf16_t>(av, dot_[n2], pf);
                        mma16<bf16_t>(ak, qt_[n2], sf);
                        if (key < mk) {   // lane holds head columns nt*16 + 4lg + r of key column `key`
                            const size_t go = (krow_g0 + krow0 + key) * M.ldkv + slice * FH_DK + nt * 16 + 4 * lg;
                            *(uint2*)(M.dv + go) = make_uint2(fh_pack2(av[0], av[1]), fh_pack2(av[2], av[3]));
                            *(uint2*)(M.dk + go) = make_uint2(fh_pack2(ak[0] * scale, ak[1] * scale), fh_pack2(ak[2] * scale, ak[3] * scale));
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();                   // the dS tile image is complete (same wave, in-order LDS)
            // dQ^T += K^T dS^T (contraction over the 32 keys of the tile): B = dS rows (query qt*16 + l15), keys 8lg .. 8lg+7
            {
                uint4 sfq[2], kft[4];
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) sfq[qt] = *(const uint4*)(ds_s + (qt * 16 + l15) * FB_DSROW + lg * 16);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) kft[nt] = fb_tfrag(ki_s, krow0 + j0, nt * 16, l15, lg);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt) mma16<bf16_t>(dqt[nt][qt], kft[nt], sfq[qt]);
            }
            __builtin_amdgcn_wave_barrier();
        }
        // dQ: lane holds dQ^T[head column nt*16 + 4lg + r][query qt*16 + l15]
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int q = qt * 16 + l15;
            if (q < a) {
                bf16_t* dqg = M.dq + (size_t)(row0 + qrow0 + q) * M.ldq + slice * FH_DK + 4 * lg;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    *(uint2*)(dqg + nt * 16) = make_uint2(fh_pack2(dqt[nt][qt][0] * scale, dqt[nt][qt][1] * scale), fh_pack2(dqt[nt][qt][2] * scale, dqt[nt][qt][3] * scale));
            }
        }
    }
}

__global__ __launch_bounds__(FH_THREADS) void fused_head_bwd_kernel(const FbGroup G) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int g = 0;
    while (g + 1 < G.count && (int)blockIdx.x >= G.wg_start[g + 1]) ++g;
    const FbMember& M = G.m[g];
    const int t = (int)blockIdx.x - G.wg_start[g];
    const int xcd = t & 7, j = t >> 3;                          // XCD-aware map, as in fused.hip
    const int hpg = (FH_D / FH_DK) / M.hg;
    const int slice = (xcd % M.hg) * hpg + (j % hpg), rb = (j / hpg) * M.sg + (xcd / M.hg);
    if (rb * M.rows_per_wg >= M.rows) return;
    if (M.mt <= 2) fb_body<2>(M, slice, rb, smem);
    else if (M.mt == 3) fb_body<3>(M, slice, rb, smem);
    else fb_body<5>(M, slice, rb, smem);
}

// ------------------------------------------------------------------------------------------ host side
int fh_is_enabled();                                            // fused.hip
static constexpr int FB_LDS_MAX = 160 * 1024;

// workspace pointers of one member (sublayer.hip carves them)
struct FbIo { const void* dyl; void *dq, *dk, *dv; int ldq, ldkv; };

struct FbLaunch { FbGroup G; int wgs; size_t lds; };
static bool fb_plan(int n_mha, const mtn_mha_args* mha, const FbIo* io, FbLaunch& P) {
    if (n_mha < 1 || n_mha > FB_MAX_MEMBERS) return false;
    FbGroup& G = P.G;
    memset(&G, 0, sizeof(G));
    const int budget = 256 / n_mha > 8 ? 256 / n_mha : 8;
    const int mts[3] = {2, 3, 5};
    int wgs = 0;
    size_t lds = 0;
    for (int i = 0; i < n_mha; ++i) {
        const mtn_mha_args& A = mha[i];
        if (A.d != FH_D || A.h != FH_D / FH_DK || !A.w_o_t || A.a > 32 || A.a < 1) return false;
        const bool self = A.self_attn != 0;
        const int m = self ? A.a : A.m, qa = A.mask_sq ? A.a : 1;
        if (A.mask && A.mask_sb != 0 && A.mask_sb != (long)qa * m) return false;
        if (A.mask && A.mask_sq != 0 && A.mask_sq != m) return false;
        int blk = 0, mt = 0, l = 0;
        for (int b = 1; b <= A.B; ++b) {
            int t = 0;
            for (int c = 0; c < 3; ++c)
                if (mts[c] * 16 >= b * A.a) { t = mts[c]; break; }
            if (!t) break;
            if (A.mask && b * qa * m > FH_THREADS * FH_MASKB) break;
            const int need = fb_lds_map(t, self ? t * 16 : b * m, A.mask ? b * qa * m : 0).total;
            if (need > FB_LDS_MAX) break;
            blk = b; mt = t; l = need;
            if (((A.B + b - 1) / b) * (FH_D / FH_DK) <= budget) break;
        }
        if (!blk) return false;
        FbMember& M = G.m[i];
        M.rows = A.B * A.a; M.rows_per_wg = blk * A.a; M.a = A.a; M.m = m; M.blk = blk; M.mt = mt; M.self_attn = self;
        M.dyl = (const bf16_t*)io[i].dyl; M.wot = (const bf16_t*)A.w_o_t;
        M.q = (const bf16_t*)A.qkv; M.ldq = self ? 3 * FH_D : FH_D;
        M.k = self ? (const bf16_t*)A.qkv + FH_D : (const bf16_t*)A.kv;
        M.v = self ? (const bf16_t*)A.qkv + 2 * FH_D : (const bf16_t*)A.kv + FH_D;
        M.ldkv = self ? 3 * FH_D : 2 * FH_D;
        M.o = (const bf16_t*)A.o; M.lse = A.lse;
        M.mask = A.mask; M.mask_sb = A.mask_sb; M.mask_sq = A.mask_sq; M.drop = A.drop_attn;
        M.dq = (bf16_t*)io[i].dq; M.dk = (bf16_t*)io[i].dk; M.dv = (bf16_t*)io[i].dv;
        if (io[i].ldq != M.ldq || io[i].ldkv != M.ldkv) return false;
        const int nrb = (A.B + blk - 1) / blk;
        // XCD map: least bytes through the fabric, 8 x (row bytes / sg + weight bytes / hg)
        double best = 1e30;
        M.hg = 8; M.sg = 1;
        for (int sg = 1; sg <= 8; sg *= 2) {
            const int hg = 8 / sg;
            if (nrb % sg != 0) continue;
            const double c = ((double)M.rows * FH_D * 2 + 4.0 * M.rows * FH_DK * 2 * 8) / sg + (double)FH_D * FH_D * 2 / hg;
            if (c < best) { best = c; M.hg = hg; M.sg = sg; }
        }
        G.wg_start[i] = wgs;
        wgs += nrb * (FH_D / FH_DK);
        lds = (size_t)l > lds ? (size_t)l : lds;
    }
    G.count = n_mha;
    for (int i = n_mha; i <= FB_MAX_MEMBERS; ++i) G.wg_start[i] = wgs;
    P.wgs = wgs; P.lds = lds;
    return true;
}

int fb_group_eligible(int dtype, int n_mha, const mtn_mha_args* mha, const FbIo* io) {
    if (!fh_is_enabled() || dtype != MTN_BF16) return 0;
    FbLaunch P;
    return fb_plan(n_mha, mha, io, P) ? 1 : 0;
}

int fb_group_bwd_stage(int n_mha, const mtn_mha_args* mha, const FbIo* io, void* stream) {
    FbLaunch P;
    MTN_CHECK_ARG(fb_plan(n_mha, mha, io, P), "group outside the fused backward kernel's tiling");
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)fused_head_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS_MAX) != hipSuccess) {
            mtn_set_error("fused_head_bwd_kernel: cannot raise the dynamic LDS limit");
            return MTN_ERR_LAUNCH;
        }
        attr = true;
    }
    hipLaunchKernelGGL(fused_head_bwd_kernel, dim3(P.wgs), dim3(FH_THREADS), P.lds, (hipStream_t)stream, P.G);
    MTN_CHECK_LAUNCH();
    return MTN_OK;
}
