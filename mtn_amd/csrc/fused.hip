// fused.hip — first launch of a lockstep sublayer group (sublayer.hip), forward:
//     LayerNorm(x) -> head slice of the input projections -> softmax(QK^T/sqrt(dk) masked) V          (attention members)
//     LayerNorm(x) -> column slice of w_1 x + b_1 -> ReLU -> dropout                                   (feed-forward members)
// i.e. SublayerConnection.norm ∘ MultiHeadedAttention.linears[0..2] ∘ attention()  (mtn.py:127, 256-258, 221-231) and
// SublayerConnection.norm ∘ PositionwiseFeedForward.w_1/relu/dropout (mtn.py:127, 280) in ONE kernel; the output projection /
// w_2 (+ bias + dropout + residual) stays a grouped GEMM launch.  bf16, d_model = 512, d_k = 64 (other shapes keep the
// four-launch path of sublayer.hip).
//
// One 256-thread workgroup = (member, block of whole samples: R <= 80 rows, head h | 192-column slice of w_1):
//   * every lane first issues the loads of ITS x rows (a wave normalises rows wave, wave+4, ...; the row stays in registers)
//     and then all weight fragments of the slice — wave w owns output columns 16w..16w+15 of each 64-column block, as the MFMA
//     A operand straight from global memory (16 B per lane per 32-deep step, 16 steps, up to 3 blocks = 192 VGPRs).  A wave's
//     loads return in issue order, so the x rows land first and the LayerNorm runs while the 192 KiB weight slice streams in;
//   * the normalised rows go to LDS as a [row][512] bf16 image (16-byte slots XOR-swizzled with row & 15: conflict-free
//     B-operand fragment reads); the workgroup of head 0 / slice 0 also writes xn, mean, rstd for backward;
//   * projections: acc[block][row tile] on mfma_f32_16x16x32_bf16; + bias -> q | k | v (bf16) to the saved-for-backward
//     buffers, which the attention stage of the SAME workgroup reads back (L2 hits) with the stand-alone kernel's code
//     (attn_mfma.h): one wave per sample of the block, or the 4 waves splitting the keys of one sample.
// Workgroup id % 8 = head, and workgroups go to the 8 XCDs round-robin: each XCD streams only its head's weight slice.
#include <stdlib.h>

#include "attn_mfma.h"

static constexpr int FH_D = 512;          // d_model
static constexpr int FH_DK = 64;          // head width
static constexpr int FH_ROWB = FH_D * 2;  // bytes per LDS image row
enum { FH_SELF = 0, FH_CROSS_READY = 1, FH_CROSS_RAW = 2, FH_FFN = 3 };
#define FH_MAX_MEMBERS (2 * MTN_SUBLAYER_MAX_GROUP)

struct FhMember {
    int kind;
    int rows;          // rows of x (B * a, or the FFN's row count)
    int rows_per_wg;   // attention: blk * a (whole samples)
    int nslice;        // heads, or 192-column slices of w_1
    int a, m, blk;     // attention: query rows per sample, memory rows per sample, samples per workgroup
    int ncols;         // FFN: d_ff
    int ld_out;        // row stride of `out`
    float eps;
    const float* x;
    const float* ln_a;
    const float* ln_b;
    const bf16_t* w;   // [3d, d] (q | k | v) or w_1 [d_ff, d]
    const float* bias;
    const bf16_t* mem; // FH_CROSS_RAW: [B * m, d]
    bf16_t* xn;
    float* mean;
    float* rstd;
    bf16_t* out;       // qkv [rows, 3d] | q [rows, d] | hid [rows, d_ff]
    bf16_t* kv;        // FH_CROSS_RAW: [B * m, 2d]
    mtn_dropout drop;  // FFN hidden dropout
    mtn_attn_args attn;
};
struct FhGroup {
    int count;
    int wg_start[FH_MAX_MEMBERS + 1];
    FhMember m[FH_MAX_MEMBERS];
};

typedef __attribute__((address_space(3))) void fh_lds_void_t;

__device__ __forceinline__ uint4 fh_xfrag(const unsigned char* img, int row, int chunk) {
    return *(const uint4*)(img + row * FH_ROWB + ((chunk ^ (row & 15)) << 4));
}

// NP = 64-column weight blocks per workgroup (3: q|k|v or 192 FFN columns; 1: q only), MT = row tiles (16 rows) at most.
template <int NP, int MT>
__global__ __launch_bounds__(256, NP == 3 ? 1 : 2) void fused_head_fwd_kernel(const FhGroup G) {
    constexpr int RPW = MT * 4;            // rows a wave normalises at most
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int g = 0;
    while (g + 1 < G.count && (int)blockIdx.x >= G.wg_start[g + 1]) ++g;
    const FhMember& M = G.m[g];
    const int t = (int)blockIdx.x - G.wg_start[g];
    const int slice = t % M.nslice, rb = t / M.nslice;
    const int row0 = rb * M.rows_per_wg;
    const int R = (M.rows - row0) < M.rows_per_wg ? (M.rows - row0) : M.rows_per_wg;
    const int mt_n = (R + 15) >> 4;
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kind = M.kind;
    unsigned char* xn_s = smem;
    unsigned char* xm_s = smem + mt_n * 16 * FH_ROWB;

    // ---- memory rows of an un-projected memory (x attends an auto-encoder stream, mtn.py:215): bf16 rows -> LDS by LDS-DMA
    int Rm = 0, rm0 = 0;
    if (kind == FH_CROSS_RAW) {
        const int nsamp = R / M.a;
        rm0 = rb * M.blk * M.m;
        Rm = nsamp * M.m;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(M.mem + (size_t)rm0 * FH_D), 0, Rm * FH_ROWB, 0x00020000);
        for (int r = wave; r < Rm; r += 4) {          // one wave-instruction = one 1 KiB row; slot `lane` receives chunk lane ^ (r & 15)
            const unsigned voff = (unsigned)r * FH_ROWB + (unsigned)((lane ^ (r & 15)) << 4);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (fh_lds_void_t*)(xm_s + r * FH_ROWB), 16, voff, 0, 0, 0);
        }
    }
    const int mtm_n = (Rm + 15) >> 4;

    // ---- this wave's x rows (fp32), then gains, then the weight fragments: returns arrive in this order
    const float* __restrict__ xg = M.x + (size_t)row0 * FH_D;
    float4 xv[RPW][2];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int r = wave + 4 * i;
        if (r < R) {
            xv[i][0] = *(const float4*)(xg + (size_t)r * FH_D + lane * 4);
            xv[i][1] = *(const float4*)(xg + (size_t)r * FH_D + 256 + lane * 4);
        }
    }
    float4 ga[2], gb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        ga[j] = *(const float4*)(M.ln_a + lane * 4 + 256 * j);
        gb[j] = *(const float4*)(M.ln_b + lane * 4 + 256 * j);
    }
    // weight block p of this workgroup covers output columns ncol[p] .. +63 of the Linear; wave w takes 16w..16w+15 of them
    int ncol[NP];
    bool act[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if (kind == FH_FFN) { ncol[p] = slice * (64 * NP) + p * 64; act[p] = ncol[p] + 16 * wave < M.ncols; }
        else { ncol[p] = p * FH_D + slice * FH_DK; act[p] = (p == 0) || kind != FH_CROSS_READY; }
    }
    uint4 wf[NP][16];
#pragma unroll
    for (int p = 0; p < NP; ++p)
        if (act[p]) {
            const bf16_t* wrow = M.w + (size_t)(ncol[p] + 16 * wave + l15) * FH_D + lg * 8;
#pragma unroll
            for (int s = 0; s < 16; ++s) wf[p][s] = *(const uint4*)(wrow + s * 32);
        }
    const DropState ds = drop_init(M.drop);

    // ---- LayerNorm (mtn.py:111-114; same arithmetic as ln_fwd_small_kernel): row -> bf16 -> LDS image
    const bool save = slice == 0;
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int r = wave + 4 * i;
        if (r < R) {
            const float4 v0 = xv[i][0], v1 = xv[i][1];
            float s = (v0.x + v0.y) + (v0.z + v0.w);
            s += (v1.x + v1.y) + (v1.z + v1.w);
            const float mean = wave_sum(s) / (float)FH_D;
            float q;
            {
                const float e0 = v0.x - mean, e1 = v0.y - mean, e2 = v0.z - mean, e3 = v0.w - mean;
                q = (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
            }
            {
                const float e0 = v1.x - mean, e1 = v1.y - mean, e2 = v1.z - mean, e3 = v1.w - mean;
                q += (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
            }
            const float std_u = sqrtf(wave_sum(q) / (float)(FH_D - 1));
            const float rstd = 1.0f / (std_u + M.eps);
            if (save && lane == 0) { M.mean[row0 + r] = mean; M.rstd[row0 + r] = rstd; }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float4 v = j ? v1 : v0;
                float4 o;
                o.x = ga[j].x * (v.x - mean) * rstd + gb[j].x;
                o.y = ga[j].y * (v.y - mean) * rstd + gb[j].y;
                o.z = ga[j].z * (v.z - mean) * rstd + gb[j].z;
                o.w = ga[j].w * (v.w - mean) * rstd + gb[j].w;
                uint2 u;
                u.x = (uint32_t)f32_to_bf16(o.x) | ((uint32_t)f32_to_bf16(o.y) << 16);
                u.y = (uint32_t)f32_to_bf16(o.z) | ((uint32_t)f32_to_bf16(o.w) << 16);
                const int chunk = (lane >> 1) + 32 * j;
                *(uint2*)(xn_s + r * FH_ROWB + ((chunk ^ (r & 15)) << 4) + (lane & 1) * 8) = u;
                if (save) *(uint2*)(M.xn + (size_t)(row0 + r) * FH_D + lane * 4 + 256 * j) = u;
            }
        }
    }
    if (kind == FH_CROSS_RAW) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the memory image (LDS-DMA) has landed
    __syncthreads();

    // ---- projections: acc[p][mt] = W block p (A operand: 16 output columns) x rows of tile mt (B operand)
    f32x4_t acc[NP][MT];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[p][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const bool raw = kind == FH_CROSS_RAW;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        uint4 xf[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            if (mt < mt_n) xf[mt] = fh_xfrag(xn_s, mt * 16 + l15, s * 4 + lg);
#pragma unroll
        for (int p = 0; p < NP; ++p)
            if (act[p] && !(raw && p > 0)) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    if (mt < mt_n) mma16<bf16_t>(acc[p][mt], wf[p][s], xf[mt]);
            }
    }
    if constexpr (NP == 3) {
        if (raw) {                          // k | v of the memory rows
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                uint4 xf[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    if (mt < mtm_n) xf[mt] = fh_xfrag(xm_s, mt * 16 + l15, s * 4 + lg);
#pragma unroll
                for (int p = 1; p < NP; ++p)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        if (mt < mtm_n) mma16<bf16_t>(acc[p][mt], wf[p][s], xf[mt]);
            }
        }
    }

    // ---- epilogue: a lane holds output row (tile row l15) x four consecutive columns 16w + 4lg .. +3 of each block
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if (!act[p]) continue;
        const int col = ncol[p] + 16 * wave + 4 * lg;           // column of the Linear's output
        const float4 bv = *(const float4*)(M.bias + col);
        const bool to_kv = raw && p > 0;
        const int rows_p = to_kv ? Rm : R, r0_p = to_kv ? rm0 : row0, tiles = to_kv ? mtm_n : mt_n;
        bf16_t* dst;
        int ld;
        if (kind == FH_FFN) { dst = M.out + col; ld = M.ld_out; }
        else if (to_kv) { dst = M.kv + (p - 1) * FH_D + slice * FH_DK + 16 * wave + 4 * lg; ld = 2 * FH_D; }
        else if (kind == FH_SELF) { dst = M.out + col; ld = M.ld_out; }
        else { dst = M.out + slice * FH_DK + 16 * wave + 4 * lg; ld = M.ld_out; }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (mt >= tiles) continue;
            const int r = mt * 16 + l15;
            if (r >= rows_p) continue;
            float v[4] = {acc[p][mt][0] + bv.x, acc[p][mt][1] + bv.y, acc[p][mt][2] + bv.z, acc[p][mt][3] + bv.w};
            if (kind == FH_FFN) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
                if (ds.on) {
                    const uint64_t idx = (uint64_t)(r0_p + r) * (uint64_t)M.ncols + col;
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = drop_keep(ds, idx + k) ? v[k] * ds.scale : 0.f;
                }
            }
            store_lp4<bf16_t>(dst + (size_t)(r0_p + r) * ld, make_float4(v[0], v[1], v[2], v[3]));
        }
    }
    if (kind == FH_FFN) return;

    // ---- attention of this head over the samples of the block; q, k, v come back from L2 (written above by this workgroup)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int nsamp = R / M.a, b0 = rb * M.blk;
    if (nsamp == 1) {
        for (int q0 = 0; q0 < M.a; q0 += MQ) attn_fwd_mfma_body<bf16_t, FH_DK, 4>(M.attn, b0, slice, q0, wave, smem);
    } else {
        for (int i = wave; i < nsamp; i += 4)
            for (int q0 = 0; q0 < M.a; q0 += MQ)
                attn_fwd_mfma_body<bf16_t, FH_DK, 1>(M.attn, b0 + i, slice, q0, 0, smem + wave * (FH_DK * (MK * 2 + 16)));
    }
}

// ------------------------------------------------------------------------------------------ host side
static constexpr int FH_MT = 5;                   // row tiles of the big instantiation (80 rows)
static constexpr int FH_MT_SMALL = 3;             // q-only instantiation (two workgroups per CU): 48 rows

static int fh_enabled = -1;          // -1: not decided yet (environment MTN_FUSED=0 turns the fused launches off)
extern "C" int mtn_fused_enable(int on) {
    const int prev = fh_enabled;
    fh_enabled = on ? 1 : 0;
    return prev;
}
static bool fh_env_off() {
    if (fh_enabled < 0) { const char* e = getenv("MTN_FUSED"); fh_enabled = (e && e[0] == '0') ? 0 : 1; }
    return fh_enabled == 0;
}

// Samples per workgroup: whole samples, at most `rmax` rows; as few workgroups as it takes to stay within one round of the
// chip (256 CUs), but not fewer than that: these launches are bound by the bytes each CU pulls (weight slice + x rows).
static int fh_pick_blk(int B, int a, int rmax, int wg_per_block, int budget) {
    int best = 1;
    for (int blk = 1; blk * a <= rmax && blk <= B; ++blk) {
        best = blk;
        if (((B + blk - 1) / blk) * wg_per_block <= budget) break;
    }
    return best;
}

// Can this group take the fused first launch?  (bf16, d = 512, h = 8, shapes inside the kernel's tiling)
int fh_group_eligible(int dtype, int n_mha, const mtn_mha_args* mha, int n_ffn, const mtn_ffn_args* ffn) {
    if (fh_env_off() || dtype != MTN_BF16) return 0;
    for (int i = 0; i < n_mha; ++i) {
        const mtn_mha_args& a = mha[i];
        if (a.d != FH_D || a.h != FH_D / FH_DK) return 0;
        if (a.a > 64 || a.a < 1) return 0;
        const bool raw = !a.self_attn && !a.kv_ready;
        if (raw && a.m > 64) return 0;
        if (!a.self_attn && a.m > 256) return 0;          // long memories: the stand-alone kernel splits keys over 8 waves
    }
    for (int i = 0; i < n_ffn; ++i)
        if (ffn[i].d != FH_D || ffn[i].d_ff % 64 != 0) return 0;
    return 1;
}

void attn_args_of(const mtn_mha_args* a, int dtype, mtn_attn_args* t);   // sublayer.hip

int fh_group_fwd_stage1(int n_mha, const mtn_mha_args* mha, int n_ffn, const mtn_ffn_args* ffn, void* stream) {
    MTN_CHECK_ARG(n_mha + n_ffn >= 1 && n_mha + n_ffn <= FH_MAX_MEMBERS, "bad group size");
    FhGroup G;
    memset(&G, 0, sizeof(G));
    bool need3 = n_ffn > 0;
    for (int i = 0; i < n_mha; ++i)
        if (mha[i].self_attn || !mha[i].kv_ready) need3 = true;
    const int rmax = need3 ? FH_MT * 16 : FH_MT_SMALL * 16;
    // share of one round (256 workgroups) each member may take
    int n = 0, wgs = 0;
    const int members = n_mha + n_ffn;
    const int budget = 256 / members > 8 ? 256 / members : 8;
    int max_rows = 0;
    for (int i = 0; i < n_mha; ++i) {
        const mtn_mha_args& a = mha[i];
        FhMember& M = G.m[n];
        M.kind = a.self_attn ? FH_SELF : (a.kv_ready ? FH_CROSS_READY : FH_CROSS_RAW);
        M.a = a.a; M.m = a.self_attn ? a.a : a.m;
        int rm = rmax;
        int blk = fh_pick_blk(a.B, a.a, rm, FH_D / FH_DK, budget);
        if (M.kind == FH_CROSS_RAW)                       // the memory image shares the LDS with the x image
            while (blk > 1 && ((blk * a.a + 15) / 16 + (blk * a.m + 15) / 16) * 16 > 144) --blk;
        M.blk = blk;
        M.rows = a.B * a.a; M.rows_per_wg = blk * a.a; M.nslice = FH_D / FH_DK;
        M.eps = a.ln_eps; M.x = a.x; M.ln_a = a.ln_a; M.ln_b = a.ln_b;
        M.w = (const bf16_t*)a.w_qkv; M.bias = a.b_qkv; M.mem = (const bf16_t*)a.mem;
        M.xn = (bf16_t*)a.xn; M.mean = a.mean; M.rstd = a.rstd;
        M.out = (bf16_t*)a.qkv; M.ld_out = a.self_attn ? 3 * FH_D : FH_D; M.kv = (bf16_t*)a.kv;
        attn_args_of(&a, MTN_BF16, &M.attn);
        G.wg_start[n] = wgs;
        wgs += ((a.B + blk - 1) / blk) * M.nslice;
        const int rows_lds = M.kind == FH_CROSS_RAW ? ((blk * a.a + 15) / 16 + (blk * a.m + 15) / 16) * 16 : ((blk * a.a + 15) / 16) * 16;
        max_rows = rows_lds > max_rows ? rows_lds : max_rows;
        ++n;
    }
    for (int i = 0; i < n_ffn; ++i) {
        const mtn_ffn_args& a = ffn[i];
        FhMember& M = G.m[n];
        M.kind = FH_FFN;
        M.rows = a.rows; M.ncols = a.d_ff; M.nslice = (a.d_ff + 191) / 192;
        int rpw = rmax;                                   // rows per workgroup: a multiple of 16, within one round if possible
        while (rpw > 16 && ((a.rows + rpw - 16 - 1) / (rpw - 16)) * M.nslice <= budget) rpw -= 16;
        M.rows_per_wg = rpw; M.a = 1; M.blk = rpw;
        M.eps = a.ln_eps; M.x = a.x; M.ln_a = a.ln_a; M.ln_b = a.ln_b;
        M.w = (const bf16_t*)a.w1; M.bias = a.b1;
        M.xn = (bf16_t*)a.xn; M.mean = a.mean; M.rstd = a.rstd;
        M.out = (bf16_t*)a.hid; M.ld_out = a.d_ff; M.drop = a.drop_hidden;
        G.wg_start[n] = wgs;
        wgs += ((a.rows + rpw - 1) / rpw) * M.nslice;
        max_rows = rpw > max_rows ? rpw : max_rows;
        ++n;
    }
    G.count = n;
    for (int i = n; i <= FH_MAX_MEMBERS; ++i) G.wg_start[i] = wgs;
    size_t lds = (size_t)max_rows * FH_ROWB;
    const size_t attn_lds = sizeof(float) * (4 * FH_DK * 33 + 2 * 4 * 32);        // 4-wave combine area >= 4 V^T images
    const size_t vt_lds = 4 * (size_t)FH_DK * (MK * 2 + 16);
    if (n_mha && lds < attn_lds) lds = attn_lds;
    if (n_mha && lds < vt_lds) lds = vt_lds;
    hipStream_t s = (hipStream_t)stream;
    if (need3) {
        static bool attr = false;
        if (!attr) {
            if (hipFuncSetAttribute((const void*)fused_head_fwd_kernel<3, FH_MT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                mtn_set_error("fused_head_fwd_kernel: cannot raise the dynamic LDS limit");
                return MTN_ERR_LAUNCH;
            }
            attr = true;
        }
        hipLaunchKernelGGL((fused_head_fwd_kernel<3, FH_MT>), dim3(wgs), dim3(256), lds, s, G);
    } else {
        static bool attr = false;
        if (!attr) {
            if (hipFuncSetAttribute((const void*)fused_head_fwd_kernel<1, FH_MT_SMALL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                mtn_set_error("fused_head_fwd_kernel: cannot raise the dynamic LDS limit");
                return MTN_ERR_LAUNCH;
            }
            attr = true;
        }
        hipLaunchKernelGGL((fused_head_fwd_kernel<1, FH_MT_SMALL>), dim3(wgs), dim3(256), lds, s, G);
    }
    MTN_CHECK_LAUNCH();
    return MTN_OK;
}
