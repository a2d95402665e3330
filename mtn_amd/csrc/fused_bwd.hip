// fused_bwd.hip — backward twin of fused.hip for the attention members of a sublayer group: ONE launch for
//     dO = dy W_o (this head's 64 columns)            (gradient of MultiHeadedAttention.linears[-1], mtn.py:267)
//     dq, dk, dv = attention backward of the head      (autograd of attention(), mtn.py:221-231)
// instead of a grouped GEMM launch (dO for all heads, through HBM) followed by the attention-backward launch.  The gradient of
// the input projections (dq|dk|dv -> d LayerNorm-out, a contraction over ALL heads) stays a grouped GEMM, as does everything
// after it.  bf16, d_model = 512, d_k = 64; a sample's query rows are taken in blocks of 32 (NQB = 1 or 2 blocks: up to 64 query
// rows — AVSD targets reach 54 tokens, queries 42; longer streams keep the two-launch path).  A memory too long for the LDS
// (BASELINE configs[3]: 512 history tokens, 256 frames) streams its K / V head rows through a two-slot ring of 128 keys while
// both four-wave teams work on the workgroup's one sample.
//
// Same construction as the forward kernel: a 512-thread workgroup = (member, block of whole samples, head) issues
// everything it reads up front — dy rows (bf16, [row][512]) and the head's q, k, v, o rows (128 bytes each) by LDS-DMA, mask bytes,
// the 64 rows of W_o^T it needs as coalesced loads (8 waves = 4 column blocks x 2 halves of the contraction, fragments put in
// MFMA operand order with ds_bpermute) — then works on chip: dO on mfma_f32_16x16x32_bf16 (halves met through LDS) -> bf16 LDS
// image (dO never goes to HBM), then the attention backward over key tiles of 32, a team of FOUR waves per sample (a single wave
// issues one vector instruction per ~5 clocks, and the per-score work — exp, dropout hash, selects — is ~50 instructions):
//     phase 1, wave = 16 x 16 quadrant of the score tile: S = Q K^T, dP = dO V^T (A rows permuted so that a lane's four scores are
//              consecutive queries), P from the saved {row max, 1 / row sum}, dropout regenerated, dS = P (dP - D);
//              P^T and dS^T go to the team's [key][query] LDS images (8-byte writes; double-buffered, one barrier per tile);
//     phase 2, wave = 16 head columns: dV^T = dO^T P, dK^T = Q^T dS, dQ^T += K^T dS^T — A operands by ds_read_b64_tr_b16 from the
//              row-major q / dO / k images, B operands straight (or transposed-read) from the tile images; no cross-wave reduction.
#include "fused_common.h"

#define FB_MAX_MEMBERS MTN_SUBLAYER_MAX_GROUP

struct FbMember {
    int rows, rows_per_wg;     // query rows (B * a), rows per workgroup (blk * a)
    int a, m, blk, mt;         // query rows / memory rows per sample, samples per workgroup, row tiles of the workgroup
    int hg, sg;                // XCD map (fused.hip)
    int self_attn;             // k, v rows are the query rows' own (packed qkv buffer)
    int ring;                  // long memory (one sample per workgroup): K / V head rows stream through a two-slot ring of 128 keys,
                               // both four-wave teams work on the sample (alternate key tiles), dQ partials met through LDS
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
    // LayerNorm backward by linearity (mtn_ln_epilogue, include/mtn_hip.h): with the fold vectors u | c of w_qkv (lnf: u[lnK] then
    // c[lnK]; lnK = 1536 self / 512 cross) the workgroup also writes, per query row, the head's share of the two row sums the
    // dLN-out GEMM's epilogue needs: ln_part[row][head] = {sum dq u, sum dq (q - c)} over the head's 64 columns (+ the dk, dv
    // terms of a self-attention, whose key rows are the query rows).  NULL = off.
    const float* lnf;
    float* ln_part;
    int lnK;
};
// The one development switch of this file: -DFB_TIMELINE (tools/fb_timeline.py, tools/fb_rounds.py) — 16 wall-clock stamps (100 MHz) per
// workgroup of the launch selected with MTN_FB_TL_LAUNCH, slot 15 = where the workgroup ran (XCC_ID (hardware register 20) << 32 | HW_ID
// (register 4: wave / SIMD / CU / SH / SE ids)); read back with mtn_fb_timeline_read().
#ifdef FB_TIMELINE
__device__ unsigned long long fb_timeline[1024 * 16];
#define FB_STAMP(k) do { if (tl && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) == 0) fb_timeline[(size_t)blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#define FB_WHERE() do { if (tl && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) == 0) fb_timeline[(size_t)blockIdx.x * 16 + 15] = ((unsigned long long)__builtin_amdgcn_s_getreg(0xF814) << 32) | (unsigned)__builtin_amdgcn_s_getreg(0xF804); } while (0)
#define FB_SELECT_LAUNCH(P) do { \
        static int launch = 0; \
        static const int want = [] { const char* e = getenv("MTN_FB_TL_LAUNCH"); return e ? atoi(e) : 0; }(); \
        (P).G.tl = (launch++ == want); \
        if ((P).G.tl) { \
            fprintf(stderr, "fb timeline: launch %d, %d workgroups, %zu B LDS\n", want, (P).wgs, (P).lds); \
            for (int i = 0; i < (P).G.count; ++i) \
                fprintf(stderr, "  member %d: wgs %d..%d a=%d m=%d blk=%d mt=%d self=%d mask=%d\n", i, (P).G.wg_start[i], (P).G.wg_start[i + 1], (P).G.m[i].a, \
                        (P).G.m[i].m, (P).G.m[i].blk, (P).G.m[i].mt, (P).G.m[i].self_attn, (P).G.m[i].mask != nullptr); \
        } \
    } while (0)
extern "C" int mtn_fb_timeline_read(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(fb_timeline), sizeof(unsigned long long) * 1024 * 16) == hipSuccess ? MTN_OK : MTN_ERR_LAUNCH;
}
#else
#define FB_STAMP(k) do { } while (0)
#define FB_WHERE() do { } while (0)
#define FB_SELECT_LAUNCH(P) do { } while (0)
#endif
struct FbGroup {
    int count;
    int tl;                    // FB_TIMELINE builds: this launch records its stamps
    int stop;                  // development: leave after stage `stop` (1 loads landed, 2 dO image, 3 attention math; 0 = everything); MTN_FB_STOP
    int wg_start[FB_MAX_MEMBERS + 1];
    FbMember m[FB_MAX_MEMBERS];
};

static constexpr int FB_DSROW = 80;        // bytes per row of a tile image [32 keys][32 queries] bf16 (+ 16 pad)
static constexpr int FB_TILE_IMG = 32 * FB_DSROW;
static constexpr int FB_TEAM_SCRATCH = 4 * FB_TILE_IMG;   // a team's {P^T, dS^T} images, double-buffered over key tiles
static constexpr int FB_DS_BYTES = 2 * 128;                // D_q of a wave's sample: up to 2 query blocks of 32
static constexpr int FB_SCRATCH = 2 * FB_TEAM_SCRATCH + 8 * FB_DS_BYTES;      // ... two teams, then D_q[64] of each wave: 22.0 KiB
// behind them (the dy image is >= 32 KiB): the head's fold vectors [q | k | v][u | c][64] (1.5 KiB) and the row-sum slots
// [row][wave of the team][2] (32 B per row, <= 2.5 KiB) of the LayerNorm-by-linearity side output
static constexpr int FB_FOLD_FLOATS = 3 * 2 * 64;

static constexpr int FB_RING_KEYS = 128;   // keys per ring slot
struct FbLds { int dy, qi, oi, doi, ki, vi, mask, scratch, total; };
__host__ __device__ inline FbLds fb_lds_map(int mt, int key_rows, int mask_bytes, bool ring = false) {
    FbLds L;
    L.dy = 0;                                              // also the exchange area of the two contraction halves (4 * mt KiB)
    L.qi = mt * 16 * FH_ROWB;
    L.oi = L.qi + mt * 16 * FH_HROWB;
    L.doi = L.oi + mt * 16 * FH_HROWB;
    L.ki = L.doi + mt * 16 * FH_HROWB;
    const int krows = ring ? 2 * FB_RING_KEYS : ((key_rows + 7) & ~7) + 32;          // two ring slots | all keys + one key tile of finite padding
    L.vi = L.ki + krows * FH_HROWB;
    L.mask = L.vi + krows * FH_HROWB;
    L.total = L.mask + ((mask_bytes + 15) & ~15);
    L.scratch = L.dy;                                      // the teams' tile images reuse the dy image (dead once dO exists; >= 32 KiB)
    return L;
}

// 128-byte head rows [nrows_valid of nrows_total] -> LDS image by LDS-DMA; rows past the end arrive as zeros
__device__ __forceinline__ void fb_dma_head_rows(const bf16_t* base, int ld_elems, int valid, int total, unsigned char* img, int wave, int lane) {
    // (inline-asm LDS-DMA, fused_common.h: with the builtin the compiler drains every DMA in flight in front of the LDS reads it can
    //  see — the ring's refill never overlapped the tiles computed from the other slot)
    const fh_rsrc_t rs = fh_make_rsrc(base, (unsigned)((valid - 1) * ld_elems * 2 + FH_HROWB));
    const int ninst = (total + 7) >> 3;
    for (int i = wave; i < ninst; i += 8) {
        const int row = i * 8 + (lane >> 3), slot = lane & 7;
        const unsigned vo = row < valid ? (unsigned)row * (unsigned)(ld_elems * 2) + (unsigned)((slot ^ (row & 7)) << 4) : 0x80000000u;
        fh_dma16(rs, (unsigned)(size_t)(img + i * 1024), vo);
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
// the same transposing read on a tile image [32 rows][32 columns] with FB_DSROW-byte rows (no swizzle): columns n_off + (lane & 15),
// rows 8*lg .. +7
__device__ __forceinline__ uint4 fb_tfrag_img(const unsigned char* img, int n_off, int l15, int lg) {
    uint4 f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int krow = 8 * lg + 4 * r + (l15 >> 2);
        const int col = n_off + 4 * (l15 & 3);
        const unsigned addr = (unsigned)(size_t)(img + krow * FB_DSROW + col * 2);
        unsigned long long v;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
        if (r == 0) { f.x = (unsigned)v; f.y = (unsigned)(v >> 32); }
        else { f.z = (unsigned)v; f.w = (unsigned)(v >> 32); }
    }
    return f;
}
__device__ __forceinline__ float fb_dot8(const uint4& x, const uint4& y) {
    float s = 0.f;
    const uint32_t xa[4] = {x.x, x.y, x.z, x.w}, ya[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
    for (int i = 0; i < 4; ++i)
        s += __uint_as_float(xa[i] << 16) * __uint_as_float(ya[i] << 16) + __uint_as_float(xa[i] & 0xffff0000u) * __uint_as_float(ya[i] & 0xffff0000u);
    return s;
}

template <int MT, int NQB, bool RING>
__device__ __forceinline__ void fb_body(const FbMember& M, const int slice, const int rb, unsigned char* smem, const int stop, const int tl) {
    const int tid = threadIdx.x;
    FB_STAMP(0);
    FB_WHERE();
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
    const int mask_bytes = nsamp * qa * mk;                // the mask image always exists (all ones without a mask): no branch per score
    const FbLds L = fb_lds_map(MT, M.self_attn ? MT * 16 : M.blk * M.m, M.blk * qa * mk, RING);
    unsigned char* dy_s = smem + L.dy;
    unsigned char* qi_s = smem + L.qi;
    unsigned char* oi_s = smem + L.oi;
    unsigned char* doi_s = smem + L.doi;
    unsigned char* ki_s = smem + L.ki;
    unsigned char* vi_s = smem + L.vi;
    unsigned char* mk_s = smem + L.mask;
    // attention stage: a team of four waves per sample; wave w4 of the team = quadrant (query tile qt, key sub-tile kt) of a 32 x 32
    // score tile in the first phase, head columns 16 w4 .. +15 of dq, dk, dv in the second
    const int team = wave >> 2, w4 = wave & 3, qt = w4 & 1, kt = w4 >> 1;
    unsigned char* tm_s = smem + L.scratch + team * FB_TEAM_SCRATCH;
    float* Ds = (float*)(smem + L.scratch + 2 * FB_TEAM_SCRATCH + wave * FB_DS_BYTES);    // D_q of the wave's sample

    // ================================================================ everything this workgroup reads, issued now
    // Order of issue (round 4): what the FIRST stage (dO = dy W_o^T) reads goes out first — the LDS-DMA images, then the W_o^T
    // fragments — and the loads only later stages need (mask bytes, softmax statistics, fold values: 8 + 4 NQB + 1 per thread, ALL of
    // them unconditional so that the count is exact) behind them: the kernel waits for all but those, and the dO stage runs while they
    // are being accepted (a wave's memory instructions are accepted at the rate its earlier ones return: the whole issue phase took 3-4 us
    // with nothing running under it).
    {   // dy rows [R][512] (rows past R: zeros), 16-byte slots swizzled with row & 15
        const fh_rsrc_t rs = fh_make_rsrc(M.dyl + (size_t)row0 * FH_D, (unsigned)(R * FH_ROWB));
        for (int r = wave; r < MT * 16; r += 8) {
            const unsigned vo = r < R ? (unsigned)r * FH_ROWB + (unsigned)((lane ^ (r & 15)) << 4) : 0x80000000u;
            fh_dma16(rs, (unsigned)(size_t)(dy_s + r * FH_ROWB), vo);
        }
    }
    fb_dma_head_rows(M.q + (size_t)row0 * M.ldq + slice * FH_DK, M.ldq, R, MT * 16, qi_s, wave, lane);
    fb_dma_head_rows(M.o + (size_t)row0 * FH_D + slice * FH_DK, FH_D, R, MT * 16, oi_s, wave, lane);
    // K / V head rows of key block bk (FB_RING_KEYS keys) into ring slot bk & 1: 4 LDS-DMA instructions per wave
    auto ring_dma = [&](const int bk) {
        const int k0 = bk * FB_RING_KEYS, valid = Kr - k0 < FB_RING_KEYS ? Kr - k0 : FB_RING_KEYS;
        const size_t g0 = (krow_g0 + k0) * M.ldkv + slice * FH_DK;
        fb_dma_head_rows(M.k + g0, M.ldkv, valid, FB_RING_KEYS, ki_s + (bk & 1) * (FB_RING_KEYS * FH_HROWB), wave, lane);
        fb_dma_head_rows(M.v + g0, M.ldkv, valid, FB_RING_KEYS, vi_s + (bk & 1) * (FB_RING_KEYS * FH_HROWB), wave, lane);
    };
    if constexpr (RING) {
        ring_dma(0);
        if (FB_RING_KEYS < Kr) ring_dma(1);
    } else {
        fb_dma_head_rows(M.k + krow_g0 * M.ldkv + slice * FH_DK, M.ldkv, Kr, ((Kr + 7) & ~7) + 32, ki_s, wave, lane);
        fb_dma_head_rows(M.v + krow_g0 * M.ldkv + slice * FH_DK, M.ldkv, Kr, ((Kr + 7) & ~7) + 32, vi_s, wave, lane);
    }
    // W_o^T rows slice*64 + 16wc .. +15, this wave's half of the contraction; coalesced: lane 4r + c reads (row r, chunk c)
    uint4 wf[8];
    {
        const bf16_t* wrow = M.wot + (size_t)(slice * FH_DK + 16 * wc + (lane >> 2)) * FH_D + kh * 256 + (lane & 3) * 8;
#pragma unroll
        for (int s = 0; s < 8; ++s) wf[s] = *(const uint4*)(wrow + s * 32);
    }
    __builtin_amdgcn_sched_barrier(0);
    // mask bytes of the block's samples.  Wide form (the block's bytes start on a dword, a multiple of 8 of them, no sharing between
    // samples): thread t takes bytes 8t .. 8t+7 as TWO dword loads; otherwise FH_MASKB byte loads per thread (without a mask: of the
    // statistics buffer, ignored).  Either way an exact count per thread for the wait below.
    const uint8_t* mask_g = M.mask ? M.mask + (size_t)b0 * M.mask_sb : (const uint8_t*)M.lse;
    const bool mask_wide = M.mask != nullptr && M.mask_sb != 0 && (((size_t)mask_g | (size_t)mask_bytes) & 7) == 0 || M.mask == nullptr;
    uint8_t mkb[FH_MASKB];
    unsigned mkw0 = 0x01010101u, mkw1 = 0x01010101u;
    if (mask_wide) {
        const int o8 = (M.mask && 8 * tid < mask_bytes) ? 8 * tid : 0;
        mkw0 = *(const unsigned*)(mask_g + o8);
        mkw1 = *(const unsigned*)(mask_g + o8 + 4);
    } else {
        const bool bcast = M.mask_sb == 0;
#pragma unroll
        for (int i = 0; i < FH_MASKB; ++i) {
            const int idx = tid + FH_THREADS * i;
            int src = idx < mask_bytes ? idx : 0;
            if (bcast) src = src % (qa * mk);                      // (uniform branch: the division only for masks shared by the samples)
            mkb[i] = mask_g[src];                                  // (whether the byte counts is decided where it is stored: a select here would wait for the load)
        }
    }
    // softmax statistics {row max, 1 / row sum} of the team's first sample (queries 8lg + 4qt + r): in flight with everything else
    float mxq[NQB][4], invq[NQB][4];
    {
        const int si0 = (!RING && team < nsamp) ? team : 0;
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = 32 * qb + 8 * lg + 4 * qt + r, qc = q < a ? q : a - 1;
                const float2 st = *(const float2*)(M.lse + 2 * ((size_t)((b0 + si0) * (FH_D / FH_DK) + slice) * a + qc));
                mxq[qb][r] = st.x;
                invq[qb][r] = st.y;
            }
    }
    const bool lnon = M.lnf != nullptr;
    float fold_v;
    {   // thread -> (part q|k|v, u|c, head column); one load per thread whatever the member (statistics buffer when there is nothing to read)
        const bool fon = lnon && tid < (M.self_attn ? FB_FOLD_FLOATS : 2 * 64);
        const float* fp = fon ? M.lnf + (((tid >> 6) & 1) * M.lnK + (tid >> 7) * FH_D + slice * FH_DK + (tid & 63)) : M.lse;
        const float fv = *fp;
        fold_v = fon ? fv : 0.f;
    }
    float* fold_s = (float*)(smem + L.scratch + FB_SCRATCH);
    float* rs_s = fold_s + FB_FOLD_FLOATS;
    const DropState ds = drop_init(M.drop);
    FB_STAMP(1);

    // ================================================================ on chip from here
    // LDS-DMA images and weight fragments have landed once at most the FH_MASKB + 4 NQB + 1 loads issued behind them fly
    static_assert(FH_MASKB == 8 && (NQB == 1 || NQB == 2), "the counted waits below spell the number of younger loads out");
    // (MTN_SAFE_WAITS: the full-wait build the counted one is compared with bit for bit, tests/test_counted_waits_gpu.py)
    if (mask_wide) {                                            // 2 (mask dwords) + 4 NQB (statistics) + 1 (fold value)
        if constexpr (NQB == 1) FH_WAIT_VM(7);
        else FH_WAIT_VM(11);
    } else {                                                    // 8 mask bytes instead
        if constexpr (NQB == 1) FH_WAIT_VM(13);
        else FH_WAIT_VM(17);
    }
    __builtin_amdgcn_s_barrier();
    FB_STAMP(2);
    if (stop == 1) return;
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
    FB_STAMP(3);
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
    if (mask_wide) {                                            // (the loads were accepted under the dO stage)
        if (8 * tid < mask_bytes) *(uint2*)(mk_s + 8 * tid) = M.mask ? make_uint2(mkw0, mkw1) : make_uint2(0x01010101u, 0x01010101u);
    } else {
#pragma unroll
        for (int i = 0; i < FH_MASKB; ++i) {
            const int idx = tid + FH_THREADS * i;
            if (idx < mask_bytes) mk_s[idx] = mkb[i];
        }
    }
    __syncthreads();                                            // dO image complete; the exchange area (in the dy image) is dead too; mask image written
    FB_STAMP(4);
    if (stop == 2) return;
    // the lane's fold-vector values: parts q | k | v, head columns 16 w4 + 4 lg .. +3 (registers for the whole attention stage)
    float4 fu[3], fc[3];
    fu[0] = fu[1] = fu[2] = fc[0] = fc[1] = fc[2] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lnon) {
        if (tid < FB_FOLD_FLOATS) fold_s[tid] = fold_v;
        for (int i = tid; i < MT * 16 * 8; i += FH_THREADS) rs_s[i] = 0.f;
        __syncthreads();
        const int colh = 16 * w4 + 4 * lg;
#pragma unroll
        for (int part = 0; part < 3; ++part) {
            if (part > 0 && !M.self_attn) break;
            fu[part] = *(const float4*)(fold_s + part * 128 + colh);
            fc[part] = *(const float4*)(fold_s + part * 128 + 64 + colh);
        }
    }
    // the lane's share of the two LayerNorm row sums for gradient values val[0..3] (head columns 16 w4 + 4 lg .. +3) of the row whose
    // saved projection output sits in row `irow` of `img` (q, k or v image; part 0 / 1 / 2 of the fold vectors).  fp32 values: the
    // GEMM reads them rounded to bf16, a difference of rounding noise in two sums over 512-1536 terms
    auto ln_dot = [&](const float (&val)[4], const unsigned char* img, const int irow, const int part, float& p1, float& p2) {
        const int colh = 16 * w4 + 4 * lg;
        const float4 u4 = fu[part], c4 = fc[part];
        const uint2 sv = *(const uint2*)(img + irow * FH_HROWB + (((colh >> 3) ^ (irow & 7)) << 4) + (colh & 7) * 2);
        const float s0 = __uint_as_float(sv.x << 16), s1 = __uint_as_float(sv.x & 0xffff0000u), s2 = __uint_as_float(sv.y << 16), s3 = __uint_as_float(sv.y & 0xffff0000u);
        p1 += (val[0] * u4.x + val[1] * u4.y) + (val[2] * u4.z + val[3] * u4.w);
        p2 += (val[0] * (s0 - c4.x) + val[1] * (s1 - c4.y)) + (val[2] * (s2 - c4.z) + val[3] * (s3 - c4.w));
    };
    // ... summed over the four lane groups; lane group 0 adds it to this wave's slot of block row `brow` (nobody else touches it)
    auto ln_add = [&](float p1, float p2, const int brow, const bool ok) {
        p1 = fh_cross_sum(p1);
        p2 = fh_cross_sum(p2);
        if (ok && lg == 0) {
            float2* slot = (float2*)rs_s + brow * 4 + w4;
            const float2 o = *slot;
            *slot = make_float2(o.x + p1, o.y + p2);
        }
    };

    // ---- attention backward: two samples at a time, a team of four waves each; one workgroup barrier per key tile of 32.
    //      RING: ONE sample, both teams on it — team t takes the key tiles 64 i + 32 t — and the keys stream through the ring
    const float scale = 0.125f;
    const int nrounds = RING ? 1 : (nsamp + 1) >> 1;
    int buf = 0;
    for (int rd = 0; rd < nrounds; ++rd) {
        const int si = RING ? 0 : 2 * rd + team;
        const bool live = si < nsamp;                              // team-uniform; an idle team only keeps the barriers
        const int sic = live ? si : nsamp - 1;
        const int b = b0 + sic, qrow0 = sic * a, krow0 = sic * mk;
        const DropBase dbase = drop_base((uint64_t)(b * (FH_D / FH_DK) + slice) * (uint64_t)a * (uint64_t)mk);   // P-dropout index of (q, key) = base + q * mk + key
        uint4 qf[NQB][2], dof[NQB][2];
        float Dq[NQB][4];
        f32x4_t dqt[NQB][2];                                       // dQ^T[head column 16 w4 + 4lg + r][query 32 qb + 16 q2 + l15]
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) dqt[qb][0] = dqt[qb][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (live) {
            // D_q = sum_c dO[q][c] O[q][c]: lanes 2q, 2q+1 each take half a row (every wave of the team, for its own use)
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) {
                const int q = 32 * qb + (lane >> 1), half = lane & 1, row = qrow0 + q;
                float sacc = 0.f;
                if (q < a) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        sacc += fb_dot8(fh_hfrag(doi_s, row, half * 4 + c), fh_hfrag(oi_s, row, half * 4 + c));
                }
                sacc += __shfl_xor(sacc, 1, 64);
                if (half == 0) Ds[q] = sacc;
            }
            // A fragments of Q and dO: accumulator row i of query tile qt <-> query 8(i/4) + 4qt + (i%4), so that the C values of the
            // two query tiles at one lane group are queries 8lg + 0..7 — the B-operand slot order of the contractions over the query
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) {
                int q = 32 * qb + 8 * (l15 >> 2) + 4 * qt + (l15 & 3);
                q = q < a ? q : a - 1;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    qf[qb][ks] = fh_hfrag(qi_s, qrow0 + q, ks * 4 + lg);
                    dof[qb][ks] = fh_hfrag(doi_s, qrow0 + q, ks * 4 + lg);
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = 32 * qb + 8 * lg + 4 * qt + r, qc = q < a ? q : a - 1;
                    if (rd != 0) {                                 // (later rounds load their statistics here)
                        const float2 st = *(const float2*)(M.lse + 2 * ((size_t)(b * (FH_D / FH_DK) + slice) * a + qc));
                        mxq[qb][r] = st.x;
                        invq[qb][r] = st.y;
                    }
                    Dq[qb][r] = Ds[q];
                }
        }
        if (rd == 0) FB_STAMP(5);

        const int niter = RING ? (mk + 63) >> 6 : (mk + 31) >> 5;
        for (int it = 0; it < niter; ++it) {
            const int j0 = RING ? 64 * it + 32 * team : 32 * it;   // first key of this team's tile
            const bool tlive = live && j0 < mk;
            int jimg = krow0 + j0;                                 // image row of key j0
            if constexpr (RING) {
                if ((it & 1) == 0) {
                    // key block it / 2 has landed in its slot: the only younger vector-memory operations of this wave are the 4 LDS-DMA
                    // instructions of the next block (loads return in order), if there is one
                    if ((it / 2 + 1) * FB_RING_KEYS < mk) FH_WAIT_VM(4);
                    else FH_WAIT_VM(0);
                    __builtin_amdgcn_s_barrier();          // (raw barriers in this loop: __syncthreads() would drain the refill DMA in flight)
                }
                jimg = ((j0 / FB_RING_KEYS) & 1) * FB_RING_KEYS + (j0 & (FB_RING_KEYS - 1));
            }
            // dV^T, dK^T of this key tile: summed over the sample's query blocks before they are stored
            f32x4_t av[2], ak[2];
            av[0] = av[1] = ak[0] = ak[1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb, buf ^= 1) {
                unsigned char* pt_s = tm_s + buf * (2 * FB_TILE_IMG);  // P^T (dropped out) [key][query of the block]
                unsigned char* dst_s = pt_s + FB_TILE_IMG;             // dS^T [key][query of the block]
                if (tlive) {
                    // ---- phase 1: this wave's 16 x 16 quadrant of S = Q K^T and dP = dO V^T, then P and dS
                    const int key = j0 + kt * 16 + l15, kc = key < mk ? key : mk - 1;
                    uint4 kf[2], vf[2];
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        kf[ks] = fh_hfrag(ki_s, jimg + (kc - j0), ks * 4 + lg);
                        vf[ks] = fh_hfrag(vi_s, jimg + (kc - j0), ks * 4 + lg);
                    }
                    f32x4_t sc = f32x4_t{0.f, 0.f, 0.f, 0.f}, dp = sc;   // C layout: rows q = 32qb + 8lg + 4qt + r, column key
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        mma16<bf16_t>(sc, qf[qb][ks], kf[ks]);
                        mma16<bf16_t>(dp, dof[qb][ks], vf[ks]);
                    }
                    if (rd == 0 && j0 == 0 && qb == 0) FB_STAMP(6);
                    // straight-line code (clamped indices and selects), so that the four elements' dependent chains overlap
                    float pd[4], dsv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int q = 32 * qb + 8 * lg + 4 * qt + r, qc = q < a ? q : a - 1;
                        const bool valid = key < mk && q < a;
                        const bool keep_score = mk_s[(sic * qa + (M.mask_sq ? qc : 0)) * mk + kc] != 0;
                        const float sv = keep_score ? sc[r] * scale : -1e9f;
                        const float p = __expf(sv - mxq[qb][r]) * invq[qb][r];
                        const bool kp = drop_keep_at(ds, dbase, (uint32_t)(qc * mk + kc));     // dropout off: threshold 0, scale 1
                        const float pdr = kp ? p * ds.scale : 0.f;
                        const float dpd = kp ? dp[r] * ds.scale : 0.f;
                        pd[r] = valid ? pdr : 0.f;
                        dsv[r] = (valid && keep_score) ? p * (dpd - Dq[qb][r]) : 0.f;
                    }
                    const int off = (kt * 16 + l15) * FB_DSROW + (8 * lg + 4 * qt) * 2;      // the lane's four queries are consecutive
                    *(uint2*)(pt_s + off) = make_uint2(fh_pack2(pd[0], pd[1]), fh_pack2(pd[2], pd[3]));
                    *(uint2*)(dst_s + off) = make_uint2(fh_pack2(dsv[0], dsv[1]), fh_pack2(dsv[2], dsv[3]));
                    if (rd == 0 && j0 == 0 && qb == 0) FB_STAMP(7);
                }
                if constexpr (RING) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                } else __syncthreads();                                // the tile's images are complete (the other buffer may still be read)
                if (tlive) {
                    // ---- phase 2: head columns 16 w4 .. +15.  dV^T += dO^T P, dK^T += Q^T dS (contraction over the block's 32 queries,
                    // per key sub-tile), dQ^T += K^T dS^T (contraction over the tile's 32 keys); A operands by transposing reads of the
                    // row-major images
                    const uint4 dot_ = fb_tfrag(doi_s, qrow0 + 32 * qb, w4 * 16, l15, lg);
                    const uint4 qt_ = fb_tfrag(qi_s, qrow0 + 32 * qb, w4 * 16, l15, lg);
                    const uint4 kt_ = fb_tfrag(ki_s, jimg, w4 * 16, l15, lg);
                    uint4 pf[2], sf[2], sfq[2];
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2) {
                        pf[k2] = *(const uint4*)(pt_s + (k2 * 16 + l15) * FB_DSROW + lg * 16);      // keys 16 k2 + l15, queries 8lg .. +7
                        sf[k2] = *(const uint4*)(dst_s + (k2 * 16 + l15) * FB_DSROW + lg * 16);
                        sfq[k2] = fb_tfrag_img(dst_s, k2 * 16, l15, lg);                           // queries 16 k2 + l15, keys 8lg .. +7
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2) {
                        mma16<bf16_t>(av[k2], dot_, pf[k2]);
                        mma16<bf16_t>(ak[k2], qt_, sf[k2]);
                        mma16<bf16_t>(dqt[qb][k2], kt_, sfq[k2]);
                    }
                    if (rd == 0 && j0 == 0 && qb == 0) FB_STAMP(8);
                }
            }
            if (tlive && stop != 3) {
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    const int key = j0 + k2 * 16 + l15;
                    if (key < mk) {                    // lane holds head columns 16 w4 + 4lg + r of key column `key`
                        const size_t go = (krow_g0 + krow0 + key) * M.ldkv + slice * FH_DK + w4 * 16 + 4 * lg;
                        *(uint2*)(M.dv + go) = make_uint2(fh_pack2(av[k2][0], av[k2][1]), fh_pack2(av[k2][2], av[k2][3]));
                        *(uint2*)(M.dk + go) = make_uint2(fh_pack2(ak[k2][0] * scale, ak[k2][1] * scale), fh_pack2(ak[k2][2] * scale, ak[k2][3] * scale));
                    }
                }
                if (lnon && M.self_attn) {             // a self-attention's key rows are its query rows: dk, dv feed the same LayerNorm
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2) {
                        if (j0 + k2 * 16 >= mk) break;                 // (uniform) no key in this half of the tile
                        const int key = j0 + k2 * 16 + l15, kc = key < mk ? key : mk - 1;
                        const float dkv[4] = {ak[k2][0] * scale, ak[k2][1] * scale, ak[k2][2] * scale, ak[k2][3] * scale};
                        const float dvv[4] = {av[k2][0], av[k2][1], av[k2][2], av[k2][3]};
                        float p1 = 0.f, p2 = 0.f;
                        ln_dot(dkv, ki_s, jimg + (kc - j0), 1, p1, p2);
                        ln_dot(dvv, vi_s, jimg + (kc - j0), 2, p1, p2);
                        ln_add(p1, p2, krow0 + kc, key < mk);
                    }
                }
            }
            if constexpr (RING) {
                if ((it & 1) == 1) {                               // the slot of key block it / 2 is free once every wave is past it: refill it
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    if ((it / 2 + 2) * FB_RING_KEYS < mk) ring_dma(it / 2 + 2);
                }
            }
        }
        if (rd == 0) FB_STAMP(10);
        if constexpr (RING) {
            // the two teams hold partial dQ^T sums over their key tiles: team 1 hands its over through LDS (the dy / tile-image area is dead)
            __syncthreads();
            float* dx = (float*)dy_s + (size_t)(w4 * 64 + lane) * (8 * NQB);
            if (team == 1) {
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
                    for (int q2 = 0; q2 < 2; ++q2) *(f32x4_t*)(dx + (qb * 2 + q2) * 4) = dqt[qb][q2];
            }
            __syncthreads();
            if (team == 0) {
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
                    for (int q2 = 0; q2 < 2; ++q2) {
                        const f32x4_t o = *(const f32x4_t*)(dx + (qb * 2 + q2) * 4);
                        dqt[qb][q2][0] += o[0]; dqt[qb][q2][1] += o[1]; dqt[qb][q2][2] += o[2]; dqt[qb][q2][3] += o[3];
                    }
            }
        }
        if (live && (!RING || team == 0)) {
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    const int q = 32 * qb + q2 * 16 + l15;
                    if (q < a && stop != 3) {
                        bf16_t* dqg = M.dq + (size_t)(row0 + qrow0 + q) * M.ldq + slice * FH_DK + w4 * 16 + 4 * lg;
                        *(uint2*)dqg = make_uint2(fh_pack2(dqt[qb][q2][0] * scale, dqt[qb][q2][1] * scale), fh_pack2(dqt[qb][q2][2] * scale, dqt[qb][q2][3] * scale));
                    }
                    if (lnon && 32 * qb + q2 * 16 < a) {           // (wave-uniform: the tile holds at least one query)
                        const int qc = q < a ? q : a - 1;
                        const float dqv[4] = {dqt[qb][q2][0] * scale, dqt[qb][q2][1] * scale, dqt[qb][q2][2] * scale, dqt[qb][q2][3] * scale};
                        float p1 = 0.f, p2 = 0.f;
                        ln_dot(dqv, qi_s, qrow0 + qc, 0, p1, p2);
                        ln_add(p1, p2, qrow0 + qc, q < a);
                    }
                }
        }
        if (rd == 0) FB_STAMP(11);
    }
    if (lnon) {                                                 // the four waves' slots of every row -> this head's pair of the row
        __syncthreads();
        if (tid < R) {
            const float2* sl = (const float2*)rs_s + tid * 4;
            const float2 a0 = sl[0], a1 = sl[1], a2 = sl[2], a3 = sl[3];
            ((float2*)M.ln_part)[(size_t)(row0 + tid) * (FH_D / FH_DK) + slice] = make_float2((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y));
        }
    }
}

// WIDE = false: every member has at most 32 query rows per sample and keeps all its keys in LDS (the train step of BASELINE
// configs[1..2]); WIDE = true: also the bodies for query blocks of 32 (a <= 80) and for long memories through the key ring.
// Two kernels so that the common launch keeps its register allocation (the wide bodies need more live fragments).
template <bool WIDE>
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
    if constexpr (WIDE) {
        if (M.ring) {                                           // one sample per workgroup, a <= 64 (host)
            if (M.a <= 32) fb_body<2, 1, true>(M, slice, rb, smem, G.stop, G.tl);
            else if (M.mt == 3) fb_body<3, 2, true>(M, slice, rb, smem, G.stop, G.tl);
            else if (M.mt == 4) fb_body<4, 2, true>(M, slice, rb, smem, G.stop, G.tl);
            else fb_body<5, 2, true>(M, slice, rb, smem, G.stop, G.tl);
            return;
        }
        if (M.a > 32) {                                         // query blocks of 32: 2 (at least 3 row tiles)
            if (M.mt == 3) fb_body<3, 2, false>(M, slice, rb, smem, G.stop, G.tl);
            else if (M.mt == 4) fb_body<4, 2, false>(M, slice, rb, smem, G.stop, G.tl);
            else fb_body<5, 2, false>(M, slice, rb, smem, G.stop, G.tl);
            return;
        }
    }
    if (M.mt <= 2) fb_body<2, 1, false>(M, slice, rb, smem, G.stop, G.tl);
    else if (M.mt == 3) fb_body<3, 1, false>(M, slice, rb, smem, G.stop, G.tl);
    else fb_body<5, 1, false>(M, slice, rb, smem, G.stop, G.tl);
}

// ------------------------------------------------------------------------------------------ host side
int fh_is_enabled();                                            // fused.hip
static constexpr int FB_LDS_MAX = 160 * 1024;

// workspace pointers of one member (sublayer.hip carves them)
struct FbIo { const void* dyl; void *dq, *dk, *dv; int ldq, ldkv; const float* lnf; float* ln_part; };

struct FbLaunch { FbGroup G; int wgs; size_t lds; bool wide; };
static bool fb_plan(int n_mha, const mtn_mha_args* mha, const FbIo* io, FbLaunch& P) {
    if (n_mha < 1 || n_mha > FB_MAX_MEMBERS) return false;
    FbGroup& G = P.G;
    memset(&G, 0, sizeof(G));
    { static const int stop = [] { const char* e = getenv("MTN_FB_STOP"); return e ? atoi(e) : 0; }(); G.stop = stop; }
    const int budget = 256 / n_mha > 8 ? 256 / n_mha : 8;
    // row tiles (16 rows) per workgroup: 2, 3 or 5 — and, for members with more than 32 query rows per sample (the two-block bodies), 4: a
    // 49..64-row sample then takes a 64 KB dy image instead of 80, which is what lets an AVSD-length answer (56 tokens) attend a long history
    // (>= 160 keys: the key ring) inside the fused kernel at all (round 6: that member — and with it its whole lockstep group — fell back to
    // attn_bwd_mfma + a separate dO GEMM + LayerNorm backward launches, 8 % of the step at the ragged corpus' commonest shape:
    // profiles/r06_avsd32_one_step_breakdown_before.txt).  Members of <= 32 query rows keep {2, 3, 5}: the benchmark's launches are unchanged.
    const int mts_all[4] = {2, 3, 4, 5};
    int wgs = 0;
    size_t lds = 0;
    P.wide = false;
    for (int i = 0; i < n_mha; ++i) {
        const mtn_mha_args& A = mha[i];
        const int max_a = 64;                     // (round 2's limit was 32: +12 % target tokens/s on AVSD-length answers, profiles/r03_i_ragged_corpus_ab.txt)
        if (A.d != FH_D || A.h != FH_D / FH_DK || !A.w_o_t || A.a > max_a || A.a < 1) return false;
        const bool self = A.self_attn != 0;
        const int m = self ? A.a : A.m, qa = A.mask_sq ? A.a : 1;
        if (A.mask && A.mask_sb != 0 && A.mask_sb != (long)qa * m) return false;
        if (A.mask && A.mask_sq != 0 && A.mask_sq != m) return false;
        int blk = 0, mt = 0, l = 0, ring = 0;
        for (int b = 1; b <= A.B; ++b) {
            int t = 0;
            for (int c = 0; c < 4; ++c)
                if (mts_all[c] * 16 >= b * A.a && (mts_all[c] != 4 || A.a > 32)) { t = mts_all[c]; break; }
            if (!t) break;
            if (A.mask && b * qa * m > FH_THREADS * FH_MASKB) break;
            int need = fb_lds_map(t, self ? t * 16 : b * m, b * qa * m).total;
            if (need > FB_LDS_MAX && b == 1 && !self && A.a <= 64) {       // a long memory: its keys stream through the ring, one sample per workgroup
                need = fb_lds_map(t, m, qa * m, true).total;
                if (need <= FB_LDS_MAX) { blk = 1; mt = t; l = need; ring = 1; }
                break;
            }
            if (need > FB_LDS_MAX) break;
            blk = b; mt = t; l = need;
            if (((A.B + b - 1) / b) * (FH_D / FH_DK) <= budget) break;
        }
        if (!blk) return false;
        const int ring_min = 192;
        if (blk == 1 && !ring && !self && m >= ring_min && A.a <= 64) {
            // one sample per workgroup anyway (two would not fit): take the ring form, where BOTH four-wave teams work on the sample
            const int need = fb_lds_map(mt, m, qa * m, true).total;
            if (need <= FB_LDS_MAX) { ring = 1; l = need; }
        }
        FbMember& M = G.m[i];
        M.rows = A.B * A.a; M.rows_per_wg = blk * A.a; M.a = A.a; M.m = m; M.blk = blk; M.mt = mt; M.self_attn = self; M.ring = ring;
        P.wide = P.wide || ring || A.a > 32;
        M.dyl = (const bf16_t*)io[i].dyl; M.wot = (const bf16_t*)A.w_o_t;
        M.q = (const bf16_t*)A.qkv; M.ldq = self ? 3 * FH_D : FH_D;
        M.k = self ? (const bf16_t*)A.qkv + FH_D : (const bf16_t*)A.kv;
        M.v = self ? (const bf16_t*)A.qkv + 2 * FH_D : (const bf16_t*)A.kv + FH_D;
        M.ldkv = self ? 3 * FH_D : 2 * FH_D;
        M.o = (const bf16_t*)A.o; M.lse = A.lse;
        M.mask = A.mask; M.mask_sb = A.mask_sb; M.mask_sq = A.mask_sq; M.drop = A.drop_attn;
        M.dq = (bf16_t*)io[i].dq; M.dk = (bf16_t*)io[i].dk; M.dv = (bf16_t*)io[i].dv;
        M.lnf = io[i].ln_part ? io[i].lnf : nullptr; M.ln_part = io[i].ln_part; M.lnK = self ? 3 * FH_D : FH_D;
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
        if (hipFuncSetAttribute((const void*)fused_head_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS_MAX) != hipSuccess ||
            hipFuncSetAttribute((const void*)fused_head_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS_MAX) != hipSuccess) {
            mtn_set_error("fused_head_bwd_kernel: cannot raise the dynamic LDS limit");
            return MTN_ERR_LAUNCH;
        }
        attr = true;
    }
    FB_SELECT_LAUNCH(P);
    if (P.wide) hipLaunchKernelGGL(fused_head_bwd_kernel<true>, dim3(P.wgs), dim3(FH_THREADS), P.lds, (hipStream_t)stream, P.G);
    else hipLaunchKernelGGL(fused_head_bwd_kernel<false>, dim3(P.wgs), dim3(FH_THREADS), P.lds, (hipStream_t)stream, P.G);
    MTN_CHECK_LAUNCH();
    return MTN_OK;
}
