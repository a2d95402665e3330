// fused.hip — first launch of a lockstep sublayer group (sublayer.hip), forward:
//     LayerNorm(x) -> head slice of the input projections -> softmax(QK^T/sqrt(dk) masked) V          (attention members)
//     LayerNorm(x) -> column slice of w_1 x + b_1 -> ReLU -> dropout                                   (feed-forward members)
// i.e. SublayerConnection.norm ∘ MultiHeadedAttention.linears[0..2] ∘ attention()  (mtn.py:127, 256-258, 221-231) and
// SublayerConnection.norm ∘ PositionwiseFeedForward.w_1/relu/dropout (mtn.py:127, 280) in ONE kernel; the output projection /
// w_2 (+ bias + dropout + residual) stays a grouped GEMM launch.  bf16, d_model = 512, d_k = 64 (other shapes keep the
// four-launch path of sublayer.hip).
//
// These launches are chains of dependent memory round trips (~2.5 us each on a busy chip) around a microsecond of arithmetic,
// and a wave can keep at most 63 vector-memory instructions (1 KiB each) in flight.  So the kernel is built to make ONE round
// trip: a 512-thread workgroup = (member, block of whole samples: R <= 80 rows, head h | 192-column slice of w_1) issues
// EVERYTHING it will read before it computes anything, spread over 8 waves so that nobody queues behind the 63-deep counter —
//   mask bytes -> registers; K|V of a memory projected ahead of the layer loop -> LDS (LDS-DMA, 128-byte head rows); the rows of
//   an un-projected memory -> LDS (LDS-DMA); its x rows (fp32, 16 lanes per row) -> registers; LayerNorm gains -> registers;
//   the weight slice as MFMA A-operand fragments straight from global memory: wave w owns output columns 16(w&3)..+15 of each
//   64-column block and HALF of the contraction (k-steps 8(w>>2)..+7): 16 B per lane per 32-deep step, <= 24 loads per wave.
// A wave's loads return in issue order: the x rows land first and the LayerNorm (row sums by DPP inside 16-lane rows) runs while
// the weight slice streams in.  Then, on chip only:
//   * normalised rows -> LDS [row][512] bf16 image (16-byte slots XOR-swizzled with row & 15: conflict-free B-operand reads);
//   * projections on mfma_f32_16x16x32_bf16, acc[block][row tile] over the wave's half of k; the two halves meet through LDS
//     (each wave of a pair keeps every other tile); + bias -> bf16 -> the saved-for-backward buffers (stores the kernel never
//     waits for) AND row-major [row][64] LDS images of this head's Q, K, V;
//   * attention per (sample, 16 query rows) on one wave: S^T = K Q^T with the keys of a tile taken in the order that makes the
//     C layout of P^T the standard B-operand slot order, so V needs no transposition: O^T = V^T P^T reads its A operand from the
//     row-major V image with ds_read_b64_tr_b16; softmax max/sum across the four 16-lane rows by v_permlane16/32_swap.
// The workgroup of head 0 / slice 0 also writes xn, mean, rstd for backward.  Workgroup id % 8 = head and workgroups go to the
// 8 XCDs round-robin: each XCD streams only its head's weight slice.
#include "fused_common.h"

struct FhMember {
    int kind;
    int rows;          // rows of x (B * a, or the FFN's row count)
    int rows_per_wg;   // attention: blk * a (whole samples)
    int nslice;        // heads, or 192-column slices of w_1
    int a, m, blk;     // attention: query rows per sample, memory rows per sample, samples per workgroup
    int mt;            // row tiles (16 rows) of this member's workgroups: 2, 3 or 5
    int hg, sg;        // XCD-aware workgroup map: the 8 XCDs form hg slice groups x sg row-block groups (hg * sg = 8)
    int late_v;        // long memory projected ahead of the layer loop: the V image takes the place of the (dead) xn image, its DMA
                       // is issued once the projections are done (K + V of 512 keys x 64 columns do not fit beside the x rows)
    int nks;           // attention: key ranges per (sample, 16 query rows): > 1 = long memory and few samples per workgroup, every
                       // wave takes a range of 64-key chunks and the partial {max, sum, O} meet through LDS
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
    bf16_t* kv;        // cross attention: [B * m, 2d] (read when projected ahead, written for an un-projected memory)
    const uint8_t* mask;
    long mask_sb, mask_sq;
    mtn_dropout drop;  // FFN hidden dropout | attention-probability dropout
    bf16_t* o;         // attention output [rows, d]
    float* lse;        // {row max, 1 / row sum} per (b, head, query)
    // (LayerNorm FORWARD by linearity — pre-scaled bf16 rows + producer-side statistics — lived here in round 4: measured, no net gain,
    //  removed in round 5: profiles/r04_k_ln_forward_by_linearity.txt)
    // a member may be a PART of a sublayer (samples b_off .. of it; every pointer above already points at the part's first row):
    // dropout indices are those of the whole sublayer (the backward kernels regenerate the masks from them).  See fh_plan: two unit sizes.
    int b_off;
};
struct FhGroup {
    int count;
    int stop;          // development: leave the kernel after stage `stop` (0 = run everything); MTN_FH_STOP
#ifdef FH_TIMELINE
    unsigned long long* dbg;   // tools/fh_bench.hip: 16 time stamps (100 MHz wall clock) per workgroup
#endif
    int wg_start[FH_MAX_MEMBERS + 1];
    FhMember m[FH_MAX_MEMBERS];
};

// LDS map (bytes), the same arithmetic on host and device
struct FhLds { int gains, xn, xm, qi, ki, vi, mask, total; };
__host__ __device__ inline FhLds fh_lds_map(int mt, bool raw, int key_rows, int pad_rows, int mask_bytes, int ffn_blocks = 0, bool late_v = false) {
    FhLds L;
    L.gains = 0;                                            // a_2 | b_2 | biases of the workgroup's output columns
    L.xn = 4096 + 1024;                                     // also the exchange area of the two contraction halves (4*NP*mt KiB)
    const int krows = ((key_rows + pad_rows + 7) & ~7);     // keys + finite padding up to the end of the last 64-key chunk
    if (late_v) {                                           // V image over the xn image (dead after the projections), then Q, K, mask
        L.vi = L.xn;
        const int top = mt * 16 * FH_ROWB > krows * FH_HROWB ? mt * 16 * FH_ROWB : krows * FH_HROWB;
        L.xm = L.xn + top;
        L.qi = L.xm;
        L.ki = L.qi + mt * 16 * FH_HROWB;
        L.mask = L.ki + krows * FH_HROWB;
        L.total = L.mask + ((mask_bytes + 15) & ~15);
        return L;
    }
    L.xm = L.xn + mt * 16 * FH_ROWB;
    L.qi = L.xm + (raw ? mt * 16 * FH_ROWB : 0);
    L.ki = L.qi + mt * 16 * (ffn_blocks ? ffn_blocks * FH_HROWB : FH_HROWB);     // Q image, or the feed-forward output tile [row][64 * NP]
    L.vi = L.ki + krows * FH_HROWB;
    L.mask = L.vi + krows * FH_HROWB;
    L.total = L.mask + ((mask_bytes + 15) & ~15);
    return L;
}
static constexpr int FH_PART_FLOATS = 18;                   // a wave's attention partial per lane: O^T (16) + running max + running sum
// image rows behind the keys that a 64-key chunk of the last sample may touch
__host__ __device__ inline int fh_pad_rows(int kind, int mk) { return kind == FH_CROSS_READY ? ((mk + 63) & ~63) - mk : 64; }

// The one development switch of this file: -DFH_TIMELINE (tools/fh_bench.hip) records wall-clock / shader-clock stamps per workgroup.
#ifdef FH_TIMELINE
#define FH_STAMP(k) do { if (threadIdx.x == 0) G.dbg[(size_t)blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#define FH_STAMP_CLK(k) do { if (threadIdx.x == 0) G.dbg[(size_t)blockIdx.x * 16 + (k)] = clock64(); } while (0)
#else
#define FH_STAMP(k) do { } while (0)
#define FH_STAMP_CLK(k) do { } while (0)
#endif

// NP = 64-column weight blocks per workgroup (3: q|k|v or 192 FFN columns; 1: q only), MT = row tiles (16 rows).
template <int NP, int MT>
__device__ __forceinline__ void fh_body(const FhGroup& G, const FhMember& M, const int slice, const int rb, unsigned char* smem) {
    constexpr int NG = (MT * 4 + 7) / 8;           // row groups (4 rows) a wave normalises at most
    constexpr int NT = NP * MT;                    // accumulator tiles per wave
    const int tid = threadIdx.x;
    const int row0 = rb * M.rows_per_wg;
    const int R = (M.rows - row0) < M.rows_per_wg ? (M.rows - row0) : M.rows_per_wg;
    const int lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave & 3, kh = wave >> 2;       // column block of 16, half of the contraction
    const int kind = M.kind;
    const bool raw = kind == FH_CROSS_RAW, ffn = kind == FH_FFN;
    const int a = M.a, m = M.m;
    const int nsamp = ffn ? 0 : R / a, b0 = rb * M.blk;
    const int Rm = raw ? nsamp * m : 0, rm0 = b0 * m;
    const int mk = kind == FH_SELF ? a : m;        // keys (= image rows) per sample
    const int key_rows = ffn ? 0 : ((kind == FH_SELF || raw) ? MT * 16 : nsamp * m);
    const int qa = M.mask_sq ? a : 1;
    const int mask_bytes = ffn ? 0 : nsamp * qa * m;   // the mask image always exists (all ones without a mask): no branch per score
    const bool late_v = kind == FH_CROSS_READY && M.late_v != 0;
    const FhLds L = fh_lds_map(MT, raw, key_rows, fh_pad_rows(kind, mk), mask_bytes, ffn ? NP : 0, late_v);
    unsigned char* xn_s = smem + L.xn;
    unsigned char* xm_s = smem + L.xm;
    unsigned char* qi_s = smem + L.qi;
    unsigned char* ki_s = smem + L.ki;
    unsigned char* vi_s = smem + L.vi;
    unsigned char* mk_s = smem + L.mask;

    // ================================================================ everything this workgroup reads
    // Order of issue (round 3): the small things the LayerNorm needs from LDS first (gains, biases, masks), then the x rows, then
    // the LDS-DMA images, then the weight fragments.  The x rows have landed at 3.5 us, but a wave is busy issuing its weight loads
    // until 7.5 us and only then normalises (3.4 us: profiles/r03_fh_xrows_timeline.txt).  Tried: waves 4-7 normalise first and ask
    // for their weights afterwards (FH_SPLIT_ORDER) — no gain (profiles/r03_fh_order.txt): a wave gets one 1 KiB load through every
    // ~180 ns whatever the other waves do (about a dozen loads in flight per wave), so the memory phase is bounded by the EIGHT waves'
    // memory-level parallelism, not by the CU's pipe, and every wave still does "its loads, then its rows" one after the other.
    // (4) LayerNorm gains: threads 0..127 a_2, 128..255 b_2 (one float4 each) -> LDS
    float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 256) gv = *(const float4*)((tid < 128 ? M.ln_a : M.ln_b - FH_D) + tid * 4);
    // weight blocks: block p covers output columns ncol[p] .. +63 of the Linear; this wave: columns 16*wc .. +15, contraction
    // steps 8*kh .. +7
    int ncol[NP];
    bool act[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if (ffn) { ncol[p] = slice * (64 * NP) + p * 64; act[p] = ncol[p] + 16 * wc < M.ncols; }
        else { ncol[p] = p * FH_D + slice * FH_DK; act[p] = (p == 0) || (p < 3 && kind != FH_CROSS_READY); }
    }
    // biases of the workgroup's 64 * NP output columns -> LDS (behind the gains)
    float bias_v = 0.f;
    if (tid < 64 * NP) {
        const int p_ = tid >> 6;
        bool on = false;
        int nc = 0;
#pragma unroll
        for (int p = 0; p < NP; ++p) if (p == p_) { on = act[p] || (ffn && ncol[p] + (tid & 63) < M.ncols); nc = ncol[p]; }
        if (on) bias_v = M.bias[nc + (tid & 63)];
    }
    // (1) mask bytes of the block's samples
    // (wide form — the block's bytes start on a dword, a multiple of 8 of them, not shared between samples: thread t takes bytes
    //  8t .. 8t+7 as two dword loads instead of eight byte loads: these loads sit in front of the x rows in the wave's queue)
    uint8_t mkb[FH_MASKB];
    const uint8_t* mask_g = M.mask ? M.mask + (size_t)b0 * M.mask_sb : nullptr;   // contiguous: mask_sb is 0 or qa * m
    const bool mask_wide = M.mask == nullptr || (M.mask_sb != 0 && (((size_t)mask_g | (size_t)mask_bytes) & 7) == 0);
    unsigned mkw0 = 0x01010101u, mkw1 = 0x01010101u;
    if (mask_wide) {
        if (M.mask && 8 * tid < mask_bytes) { mkw0 = *(const unsigned*)(mask_g + 8 * tid); mkw1 = *(const unsigned*)(mask_g + 8 * tid + 4); }
    } else {
#pragma unroll
        for (int i = 0; i < FH_MASKB; ++i) {
            const int idx = tid + FH_THREADS * i;
            mkb[i] = 1;
            if (idx < mask_bytes) mkb[i] = mask_g[M.mask_sb ? idx : idx % (qa * m)];
        }
    }
    // (3) x rows: row group rg = wave + 8i holds rows 4rg .. 4rg+3, one per 16-lane row; lane l15 reads columns 64j + 4*l15
    const float* __restrict__ xg = M.x + (size_t)row0 * FH_D;
    float4 xv[NG][8];
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const int r = 4 * (wave + 8 * i) + lg;
        const bool ok = r < R;                         // (groups past the tile range have r >= MT*16 >= R)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            xv[i][j] = ok ? *(const float4*)(xg + (size_t)r * FH_D + 64 * j + 4 * l15) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    FH_STAMP(14);                                  // x rows issued
    // (2) LDS-DMA (inline asm: fused_common.h): rows of an un-projected memory (x attends an auto-encoder stream, mtn.py:215), or
    //     this head's K and V rows of a memory projected ahead of the layer loop.  Rows past the end arrive as zeros (buffer bound).
    if (raw) {
        const fh_rsrc_t rsrc = fh_make_rsrc(M.mem + (size_t)rm0 * FH_D, (unsigned)(Rm * FH_ROWB));
        for (int r = wave; r < MT * 16; r += 8) {     // one wave-instruction = one 1 KiB row; slot `lane` receives chunk lane ^ (r & 15)
            const unsigned voff = r < Rm ? (unsigned)r * FH_ROWB + (unsigned)((lane ^ (r & 15)) << 4) : 0x80000000u;
            fh_dma16(rsrc, (unsigned)(size_t)(xm_s + r * FH_ROWB), voff);
        }
    }
    // ... K and V head rows of a memory projected ahead of the layer loop: only the attention stage reads them, so (round 4) they are
    // asked for BETWEEN the LayerNorm row groups, behind the first weight block — in front of the x rows' consumers they only made
    // the wave's issue phase longer (a 128-key memory: 16 instructions per wave) before its first row group could start
    auto issue_kv_dma = [&]() {
        const int Kr = nsamp * m;
        const bf16_t* kbase = M.kv + (size_t)rm0 * (2 * FH_D) + slice * FH_DK;
        const fh_rsrc_t rk = fh_make_rsrc(kbase, (unsigned)((Kr - 1) * (4 * FH_D) + FH_HROWB));
        const fh_rsrc_t rv = fh_make_rsrc(kbase + FH_D, (unsigned)((Kr - 1) * (4 * FH_D) + FH_HROWB));
        const int ninst = ((Kr + fh_pad_rows(kind, mk) + 7) & ~7) >> 3;        // 8 head rows (128 B each) per wave-instruction
        for (int i = wave; i < ninst; i += 8) {
            const int row = i * 8 + (lane >> 3), slot = lane & 7;
            const unsigned vk = row < Kr ? (unsigned)row * (4 * FH_D) + (unsigned)((slot ^ (row & 7)) << 4) : 0x80000000u;
            const unsigned vv = row < Kr ? (unsigned)row * (4 * FH_D) + (unsigned)((slot ^ ((row >> 1) & 7)) << 4) : 0x80000000u;
            fh_dma16(rk, (unsigned)(size_t)(ki_s + i * 1024), vk);
            if (!late_v) fh_dma16(rv, (unsigned)(size_t)(vi_s + i * 1024), vv);
        }
    };
    // (5) weight fragments.  The MFMA A-operand layout (lane 16c + r <-> row r, 16-byte chunk c of the 64-byte step) would make every
    // quad of adjacent lanes touch four different weight rows: the texture addresser then takes 64 cycles per wave-instruction
    // instead of 16.  So the loads are issued COALESCED — lane 4r + c reads (row r, chunk c): a quad = 64 contiguous bytes — and the
    // fragments are put in operand order on chip, once they have landed (4 ds_bpermute per fragment).
    uint4 wf[NP][8];
    auto issue_weights = [&]() {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (act[p]) {
                const bf16_t* wrow = M.w + (size_t)(ncol[p] + 16 * wc + (lane >> 2)) * FH_D + kh * 256 + (lane & 3) * 8;
#pragma unroll
                for (int s = 0; s < 8; ++s) wf[p][s] = *(const uint4*)(wrow + s * 32);
            } else {                                    // inactive block (q-only member in a 3-block launch, attention member in a 4-block one)
#pragma unroll
                for (int s = 0; s < 8; ++s) wf[p][s] = make_uint4(0, 0, 0, 0);
            }
        }
    };
    // One block WITHOUT a condition (an inactive block re-reads block 0's rows — L1 / L2 hits — and its products are never kept):
    // with every load of the kernel unconditional the compiler knows exactly how many are younger than the x rows a LayerNorm row
    // group waits for, and the blocks can be issued BETWEEN the row groups (below).
    auto issue_block = [&](const int p_) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (p != p_) continue;
            const int nc = act[p] ? ncol[p] : ncol[0];
            const bf16_t* wrow = M.w + (size_t)(nc + 16 * wc + (lane >> 2)) * FH_D + kh * 256 + (lane & 3) * 8;
#pragma unroll
            for (int s = 0; s < 8; ++s) wf[p][s] = *(const uint4*)(wrow + s * 32);
        }
    };
    // masks, gains and biases -> LDS (they were asked for first: the x rows, the images and the weights may still fly)
    if (mask_wide) {
        if (8 * tid < mask_bytes) *(uint2*)(mk_s + 8 * tid) = make_uint2(mkw0, mkw1);
    } else {
#pragma unroll
        for (int i = 0; i < FH_MASKB; ++i) {
            const int idx = tid + FH_THREADS * i;
            if (idx < mask_bytes) mk_s[idx] = mkb[i];
        }
    }
    if (tid < 256) *(float4*)(smem + L.gains + tid * 16) = gv;
    if (tid < 64 * NP) *(float*)(smem + L.gains + 4096 + tid * 4) = bias_v;
    FH_STAMP(2);                                   // masks, gains and biases have landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                  // (raw: __syncthreads would wait for every load in flight)
    FH_STAMP(3);
    // A wave's memory instructions are accepted at the rate its earlier ones return (about a dozen in flight): issuing the 24 weight
    // loads takes ~4 us during which the wave does nothing else, and the LayerNorm — 3.4 us of arithmetic on rows that landed long
    // before — only started behind them (profiles/r03_fh_order.txt).  Interleaved (round 4): block 0, row group 0, block 1, row
    // group 1, ...: the arithmetic runs while the queue drains.  -DFH_NO_INTERLEAVE: all blocks first, as before.
    issue_block(0);
    __builtin_amdgcn_sched_barrier(0);
    FH_STAMP(13);                                  // first half of the waves: all loads issued
    const DropState ds = drop_init(M.drop);
    FH_STAMP(1);                                   // ... and the dropout key derived (one more scalar round trip)

    // ---- LayerNorm (mtn.py:111-114): 16 lanes per row, 32 elements per lane; row -> bf16 -> LDS image
    const bool save = slice == 0;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        if (i > 0 && i < NP) { __builtin_amdgcn_sched_barrier(0); issue_block(i); __builtin_amdgcn_sched_barrier(0); }
        if (i == 1 && kind == FH_CROSS_READY) issue_kv_dma();
        if (wave + 8 * i >= MT * 4) continue;
        const int r = 4 * (wave + 8 * i) + lg;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += (xv[i][j].x + xv[i][j].y) + (xv[i][j].z + xv[i][j].w);
        const float mean = fh_row16_sum(s) * (1.0f / (float)FH_D);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float e0 = xv[i][j].x - mean, e1 = xv[i][j].y - mean, e2 = xv[i][j].z - mean, e3 = xv[i][j].w - mean;
            q += (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
        }
        const float std_u = sqrtf(fh_row16_sum(q) * (1.0f / (float)(FH_D - 1)));
        const float rstd = 1.0f / (std_u + M.eps);
        if (save && l15 == 0 && r < R) { M.mean[row0 + r] = mean; M.rstd[row0 + r] = rstd; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 ga = *(const float4*)(smem + L.gains + (64 * j + 4 * l15) * 4);
            const float4 gb = *(const float4*)(smem + L.gains + 2048 + (64 * j + 4 * l15) * 4);
            const float4 v = xv[i][j];
            uint2 u;
            u.x = fh_pack2(ga.x * (v.x - mean) * rstd + gb.x, ga.y * (v.y - mean) * rstd + gb.y);
            u.y = fh_pack2(ga.z * (v.z - mean) * rstd + gb.z, ga.w * (v.w - mean) * rstd + gb.w);
            const int chunk = 8 * j + (l15 >> 1);
            *(uint2*)(xn_s + r * FH_ROWB + ((chunk ^ (r & 15)) << 4) + (l15 & 1) * 8) = u;
            if (save && r < R) *(uint2*)(M.xn + (size_t)(row0 + r) * FH_D + 64 * j + 4 * l15) = u;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = NG > 1 ? NG : 1; p < NP; ++p) issue_block(p);          // the blocks that found no row group to precede
    if (NG == 1 && kind == FH_CROSS_READY) issue_kv_dma();
    __builtin_amdgcn_sched_barrier(0);
    // zero padding behind the key images (a key chunk may run past the last key: its V rows must be finite)
    if (!ffn && kind != FH_CROSS_READY) {
        for (int i = tid; i < 64 * FH_HROWB / 16; i += FH_THREADS) {
            *(uint4*)(ki_s + MT * 16 * FH_HROWB + i * 16) = make_uint4(0, 0, 0, 0);
            *(uint4*)(vi_s + MT * 16 * FH_HROWB + i * 16) = make_uint4(0, 0, 0, 0);
        }
    }
    FH_STAMP(4);                                   // LayerNorm done
    // Everything OLDER than the weight blocks has landed — the x rows, the LDS-DMA images — once at most the 8 NP weight loads (every
    // wave issues exactly that many, and they are its youngest loads) are still in flight; the normalised rows are in LDS once the
    // LDS counter is down.  No vmcnt(0) here (__syncthreads has one): the last weight block has only just been asked for, and the
    // projections below start on the fragments that have landed (counted waits by the compiler, fragment by fragment).
    // (The count assumes what the compiler emits today.  MTN_SAFE_WAITS builds the same kernels with full waits —
    // libmtn_hip_safewaits.so, mtn_amd/build.py — and tests/test_counted_waits_gpu.py compares the two libraries bit for bit.)
    static_assert(NP == 1 || NP == 3 || NP == 4, "the counted wait below spells 8 * NP out");
    if constexpr (NP == 1) FH_WAIT_VM_LGKM0(8);
    else if constexpr (NP == 3) FH_WAIT_VM_LGKM0(24);
    else FH_WAIT_VM_LGKM0(32);
    __builtin_amdgcn_s_barrier();
    FH_STAMP(5);
    if (G.stop == 1) return;

    // weight fragments into operand order (lane 16c + r takes the 16 bytes lane 4r + c loaded): step by step INSIDE the projection loops,
    // so that a step's MFMAs only wait for that step's fragments
    const int wsrc = (4 * l15 + lg) * 4;
    auto permute_step = [&](const int s) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            wf[p][s].x = (unsigned)__builtin_amdgcn_ds_bpermute(wsrc, (int)wf[p][s].x);
            wf[p][s].y = (unsigned)__builtin_amdgcn_ds_bpermute(wsrc, (int)wf[p][s].y);
            wf[p][s].z = (unsigned)__builtin_amdgcn_ds_bpermute(wsrc, (int)wf[p][s].z);
            wf[p][s].w = (unsigned)__builtin_amdgcn_ds_bpermute(wsrc, (int)wf[p][s].w);
        }
    };
    // ---- projections: acc[p][mt] = W block p (A operand: 16 output columns) x rows of tile mt (B operand), this wave's half of k
    f32x4_t acc[NP][MT];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[p][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (!raw) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            permute_step(s);
            uint4 xf[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xf[mt] = fh_xfrag(xn_s, mt * 16 + l15, (kh * 8 + s) * 4 + lg);
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) mma16<bf16_t>(acc[p][mt], wf[p][s], xf[mt]);
        }
    } else if constexpr (NP >= 3) {           // q from the normalised rows, k | v from the memory rows
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            permute_step(s);
            uint4 xf[MT], mf[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                xf[mt] = fh_xfrag(xn_s, mt * 16 + l15, (kh * 8 + s) * 4 + lg);
                mf[mt] = fh_xfrag(xm_s, mt * 16 + l15, (kh * 8 + s) * 4 + lg);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                mma16<bf16_t>(acc[0][mt], wf[0][s], xf[mt]);
                mma16<bf16_t>(acc[1][mt], wf[1][s], mf[mt]);
                mma16<bf16_t>(acc[2][mt], wf[2][s], mf[mt]);
            }
        }
    }
    FH_STAMP(6);                                   // this wave's half of the projections done (all weight fragments landed)
    // ---- the two halves of the contraction meet: of each pair of waves (same columns), wave kh keeps the tiles with
    //      (tile index & 1) == kh and hands the others over through LDS (the xn image is dead: everybody is past it)
    __syncthreads();
    {
        float* ex = (float*)xn_s + (size_t)wc * (NT * 256) + lane * 4;
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                if (((p * MT + mt) & 1) != kh) *(f32x4_t*)(ex + (p * MT + mt) * 256) = acc[p][mt];
        __syncthreads();
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                if (((p * MT + mt) & 1) == kh) {
                    const f32x4_t o = *(const f32x4_t*)(ex + (p * MT + mt) * 256);
                    acc[p][mt][0] += o[0]; acc[p][mt][1] += o[1]; acc[p][mt][2] += o[2]; acc[p][mt][3] += o[3];
                }
    }
    if (late_v) {
        // the xn image and the exchange area inside it are dead once every wave has read its partner's tiles: the V head rows of
        // the block's samples go there now (second memory generation of this workgroup: it lands under the epilogue and the
        // Q-row stores, and is waited for just before the attention)
        __syncthreads();
        const int Kr = nsamp * m;
        const bf16_t* vbase = M.kv + (size_t)rm0 * (2 * FH_D) + slice * FH_DK + FH_D;
        const fh_rsrc_t rv = fh_make_rsrc(vbase, (unsigned)((Kr - 1) * (4 * FH_D) + FH_HROWB));
        const int ninst = ((Kr + fh_pad_rows(kind, mk) + 7) & ~7) >> 3;
        for (int i = wave; i < ninst; i += 8) {
            const int row = i * 8 + (lane >> 3), slot = lane & 7;
            const unsigned vv = row < Kr ? (unsigned)row * (4 * FH_D) + (unsigned)((slot ^ ((row >> 1) & 7)) << 4) : 0x80000000u;
            fh_dma16(rv, (unsigned)(size_t)(vi_s + i * 1024), vv);
        }
    }
    if (G.stop == 2) {
        float t_ = 0.f;
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) t_ += acc[p][mt][0] + acc[p][mt][1] + acc[p][mt][2] + acc[p][mt][3];
        if (t_ == 123.456f) M.mean[0] = t_;
        return;
    }

    // ---- epilogue (the tiles this wave kept): a lane holds output row (tile row l15) x four consecutive columns 16wc + 4lg .. +3.
    //      + bias (ReLU, dropout) -> bf16 -> LDS: this head's Q / K / V images ([row][64], 16-byte slots swizzled), or the
    //      feed-forward output tile [row][64 * NP]; global memory gets them afterwards as whole rows (a lane's 8 bytes of 16
    //      different rows per store would cost the texture addresser 64 cycles per wave-instruction)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if (!act[p]) continue;
        const int col = ncol[p] + 16 * wc + 4 * lg;             // column of the Linear's output
        unsigned char* img = p == 0 ? qi_s : (p == 1 ? ki_s : vi_s);          // (p == 3 exists for FFN members only)
        const float4 bvp = *(const float4*)(smem + L.gains + 4096 + (p * 64 + 16 * wc + 4 * lg) * 4);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (((p * MT + mt) & 1) != kh) continue;
            const int r = mt * 16 + l15;
            float v[4] = {acc[p][mt][0] + bvp.x, acc[p][mt][1] + bvp.y, acc[p][mt][2] + bvp.z, acc[p][mt][3] + bvp.w};
            if (ffn) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
                if (ds.on) {
                    const DropBase db = drop_base((uint64_t)(M.b_off + row0 + r) * (uint64_t)M.ncols + col);     // (FFN parts: b_off counts rows)
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = drop_keep_at(ds, db, k) ? v[k] * ds.scale : 0.f;
                }
            }
            const uint2 u = make_uint2(fh_pack2(v[0], v[1]), fh_pack2(v[2], v[3]));
            if (ffn) {
                const int chunk = p * 8 + 2 * wc + (lg >> 1);       // 16-byte chunk of the (128 * NP)-byte row
                *(uint2*)(qi_s + r * (NP * FH_HROWB) + ((chunk ^ (r & 7)) << 4) + (lg & 1) * 8) = u;
            } else {
                const int chunk = 2 * wc + (lg >> 1);
                const int sw = p == 2 ? ((r >> 1) & 7) : (r & 7);
                *(uint2*)(img + r * FH_HROWB + ((chunk ^ sw) << 4) + (lg & 1) * 8) = u;
            }
        }
    }
    FH_STAMP(7);                                   // images / output tile written
    __syncthreads();
    // ---- LDS -> global, whole rows: 8 lanes per 128-byte head row (8 rows per wave-instruction), 8 * NP lanes per FFN row
    if (ffn) {
        constexpr int LPR = 8 * NP;                            // lanes per row (16 bytes each)
        const int ncols_here = (M.ncols - slice * 64 * NP) < 64 * NP ? (M.ncols - slice * 64 * NP) : 64 * NP;
        for (int r = tid / LPR; r < R; r += FH_THREADS / LPR) {
            const int c = tid % LPR;
            if (c * 8 < ncols_here)
                *(uint4*)(M.out + (size_t)(row0 + r) * M.ld_out + slice * 64 * NP + c * 8) = *(const uint4*)(qi_s + r * (NP * FH_HROWB) + ((c ^ (r & 7)) << 4));
        }
        return;
    }
    {
        const int c = tid & 7;
        bf16_t* qdst = M.out + (size_t)row0 * M.ld_out + (kind == FH_SELF ? slice * FH_DK : slice * FH_DK) + c * 8;
        for (int r = tid >> 3; r < R; r += FH_THREADS / 8)
            *(uint4*)(qdst + (size_t)r * M.ld_out) = *(const uint4*)(qi_s + r * FH_HROWB + ((c ^ (r & 7)) << 4));
        if (kind != FH_CROSS_READY) {
            const int rows_kv = raw ? Rm : R;
            bf16_t* kdst = raw ? M.kv + (size_t)rm0 * (2 * FH_D) + slice * FH_DK + c * 8 : M.out + (size_t)row0 * M.ld_out + FH_D + slice * FH_DK + c * 8;
            const int ldkv = raw ? 2 * FH_D : M.ld_out;
            for (int r = tid >> 3; r < rows_kv; r += FH_THREADS / 8) {
                *(uint4*)(kdst + (size_t)r * ldkv) = *(const uint4*)(ki_s + r * FH_HROWB + ((c ^ (r & 7)) << 4));
                *(uint4*)(kdst + (size_t)r * ldkv + FH_D) = *(const uint4*)(vi_s + r * FH_HROWB + ((c ^ ((r >> 1) & 7)) << 4));
            }
        }
    }
    if (G.stop == 3) return;

    // ---- attention of this head, on chip.  Item = (sample, 16 query rows[, key range]), one wave each.
    FH_STAMP(8);
    if (late_v || kind == FH_CROSS_READY) {        // ... and the K | V images asked for between the LayerNorm row groups: EXPLICITLY (the
                                                   // LDS-DMA is inline asm: the compiler may drop the vmcnt(0) of a __syncthreads as redundant)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const float scale = 0.125f;                    // 1 / sqrt(64)
    const int nqt = (a + 15) >> 4;
    // one item over the 64-key chunks [cbeg, cend): S^T = K Q^T, masked online softmax, O^T += V^T P^T.  Leaves the un-normalised
    // O^T[head column 16nt + 4lg + r][query l15], the running max and the running sum of the range.
    auto attend = [&](const int si, const int qt, const int cbeg, const int cend, f32x4_t (&ot)[4], float& mrun, float& lrun) {
        const int b = b0 + si;
        const DropBase dbase = drop_base((uint64_t)((b + M.b_off) * (FH_D / FH_DK) + slice) * (uint64_t)a * (uint64_t)mk);   // P-dropout index of (q, key) = base + q * mk + key
        const int q = qt * 16 + l15, qc = q < a ? q : a - 1;
        const int qrow = si * a + qc, krow0 = si * mk;
        uint4 qf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[ks] = fh_hfrag(qi_s, qrow, ks * 4 + lg);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) ot[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        mrun = -INFINITY; lrun = 0.f;
        const unsigned char* mrow = mk_s + (size_t)(si * qa + (M.mask_sq ? qc : 0)) * mk;
        for (int c0 = cbeg; c0 < cend; c0 += 64) {
            // S^T tile kt, accumulator row i <-> key c0 + 32(kt/2) + 8(i/4) + 4(kt%2) + (i%4): a lane's values of tiles 2u, 2u+1
            // are keys c0 + 32u + 8lg + 0..7 — the B-operand slot order of the V^T P^T contraction
            const bool wide = mk - c0 > 32;                        // keys 32..63 of the chunk exist (uniform): short memories skip that half
            f32x4_t st[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                st[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                if (kt >= 2 && !wide) continue;
                int key = c0 + 32 * (kt >> 1) + 8 * (l15 >> 2) + 4 * (kt & 1) + (l15 & 3);
                key = key < mk ? key : mk - 1;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) mma16<bf16_t>(st[kt], fh_hfrag(ki_s, krow0 + key, ks * 4 + lg), qf[ks]);
            }
            // straight-line per score (clamped indices, selects): the lane's dependent chains have to overlap each other
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                if (kt >= 2 && !wide) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = c0 + 32 * (kt >> 1) + 8 * lg + 4 * (kt & 1) + r, kc = key < mk ? key : mk - 1;
                    float sv = mrow[kc] != 0 ? st[kt][r] * scale : -1e9f;
                    sv = key < mk ? sv : -INFINITY;
                    st[kt][r] = sv;
                    mx = fmaxf(mx, sv);
                }
            }
            mx = fh_cross_max(mx);
            const float mn = fmaxf(mrun, mx);
            float psum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                if (kt >= 2 && !wide) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = c0 + 32 * (kt >> 1) + 8 * lg + 4 * (kt & 1) + r, kc = key < mk ? key : mk - 1;
                    const float pv = __expf(st[kt][r] - mn);       // exp(-inf) = 0 for chunk padding
                    psum += pv;
                    st[kt][r] = drop_keep_at(ds, dbase, (uint32_t)(qc * mk + kc)) ? pv * ds.scale : 0.f;     // dropout off: threshold 0, scale 1
                }
            }
            psum = fh_cross_sum(psum);
            const float alpha = (mrun == -INFINITY) ? 0.f : __expf(mrun - mn);
            lrun = lrun * alpha + psum;
            mrun = mn;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) { ot[nt][0] *= alpha; ot[nt][1] *= alpha; ot[nt][2] *= alpha; ot[nt][3] *= alpha; }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u == 1 && !wide) continue;
                const uint4 pf = make_uint4(fh_pack2(st[2 * u][0], st[2 * u][1]), fh_pack2(st[2 * u][2], st[2 * u][3]),
                                            fh_pack2(st[2 * u + 1][0], st[2 * u + 1][1]), fh_pack2(st[2 * u + 1][2], st[2 * u + 1][3]));
                uint4 vf[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) vf[nt] = fh_vfrag(vi_s, krow0 + c0 + 32 * u, nt * 16, l15, lg);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) mma16<bf16_t>(ot[nt], vf[nt], pf);
            }
        }
    };
    // lane holds O^T[head column 16nt + 4lg + r][query l15]: normalise, store the head's output columns and {max, 1 / sum}
    auto finish = [&](const int si, const int qt, const f32x4_t (&ot)[4], const float mrun, const float lrun) {
        const int b = b0 + si, q = qt * 16 + l15;
        if (q < a) {
            const float inv = 1.0f / lrun;
            bf16_t* og = M.o + (size_t)(b * a + q) * FH_D + slice * FH_DK + 4 * lg;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                *(uint2*)(og + nt * 16) = make_uint2(fh_pack2(ot[nt][0] * inv, ot[nt][1] * inv), fh_pack2(ot[nt][2] * inv, ot[nt][3] * inv));
            if (M.lse && lg == 0) {
                float* stp = M.lse + 2 * ((size_t)(b * (FH_D / FH_DK) + slice) * a + q);
                stp[0] = mrun;
                stp[1] = inv;
            }
        }
    };
    const int nks = M.nks;
    if (nks <= 1) {
        for (int it = wave; it < nsamp * nqt; it += 8) {
            const int si = it / nqt, qt = it - si * nqt;
            f32x4_t ot[4];
            float mrun, lrun;
            attend(si, qt, 0, mk, ot, mrun, lrun);
            finish(si, qt, ot, mrun, lrun);
        }
    } else {
        // Few (sample, query tile) pairs and a long memory: nks waves share a pair, each over its own range of 64-key chunks;
        // the partial {O^T, max, sum} of a pair's waves meet in LDS (over the K image, dead by then) and the pair's first wave
        // combines them:  m* = max m_k,  l* = sum l_k e^(m_k - m*),  O* = sum O_k e^(m_k - m*).
        const int npairs = nsamp * nqt;                        // host: npairs * nks <= 8
        const int pair = wave / nks, kr = wave - pair * nks;
        const bool live = pair < npairs;
        const int si = live ? pair / nqt : 0, qt = live ? pair - si * nqt : 0;
        const int chunks = (mk + 63) >> 6, per = (chunks + nks - 1) / nks;
        int cbeg = kr * per * 64, cend = (kr + 1) * per * 64;
        cend = cend < mk ? cend : mk;
        f32x4_t ot[4];
        float mrun = -INFINITY, lrun = 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) ot[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (live && cbeg < cend) attend(si, qt, cbeg, cend, ot, mrun, lrun);
        __syncthreads();                                       // every wave is done with the K and V images
        // slot of a non-first wave of a pair: pair * (nks - 1) + kr - 1  (at most 7 slots of FH_PART_FLOATS x 64 floats)
        float* part = (float*)ki_s + (size_t)(pair * (nks - 1) + kr - 1) * (FH_PART_FLOATS * 64) + lane;
        if (live && kr != 0) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) part[(nt * 4 + r) * 64] = ot[nt][r];
            part[16 * 64] = mrun;
            part[17 * 64] = lrun;
        }
        __syncthreads();
        if (live && kr == 0) {
            for (int k = 1; k < nks; ++k) {
                const float* pk = (const float*)ki_s + (size_t)(pair * (nks - 1) + k - 1) * (FH_PART_FLOATS * 64) + lane;
                const float mo = pk[16 * 64], lo = pk[17 * 64];
                const float mn = fmaxf(mrun, mo);
                const float wa = (mrun == -INFINITY) ? 0.f : __expf(mrun - mn), wb = (mo == -INFINITY) ? 0.f : __expf(mo - mn);
                lrun = lrun * wa + lo * wb;
                mrun = mn;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ot[nt][r] = ot[nt][r] * wa + pk[(nt * 4 + r) * 64] * wb;
            }
            finish(si, qt, ot, mrun, lrun);
        }
    }
    FH_STAMP(9);
    FH_STAMP_CLK(11);
}

template <int NP>
__global__ __launch_bounds__(FH_THREADS) void fused_head_fwd_kernel(const FhGroup G) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    FH_STAMP(0);
    FH_STAMP_CLK(10);
    int g = 0;
    while (g + 1 < G.count && (int)blockIdx.x >= G.wg_start[g + 1]) ++g;
    const FhMember& M = G.m[g];
    const int t = (int)blockIdx.x - G.wg_start[g];
    // Workgroups go to the 8 XCDs round-robin (id % 8; every member starts at a multiple of 8) and each XCD has its own L2: the
    // launch is bound by what crosses the fabric into the L2s, i.e. by how often an x row or a weight slice is pulled by
    // DIFFERENT XCDs.  XCD (hgi, sgi) takes the slices of group hgi and the row blocks congruent to sgi mod sg: an x row is
    // then pulled by hg XCDs instead of 8, a weight slice by sg instead of 8 (same-XCD re-reads are L2 hits).
    const int xcd = t & 7, j = t >> 3;
    const int hpg = M.nslice / M.hg;
    const int slice = (xcd % M.hg) * hpg + (j % hpg), rb = (j / hpg) * M.sg + (xcd / M.hg);
    FH_STAMP(12);                                  // member found, first fields read
    if (rb * M.rows_per_wg >= M.rows) return;                  // padding workgroup (row-block count rounded up to the map's grid)
    if constexpr (NP == 4) {                       // 4 weight blocks: 128 VGPRs of fragments -> at most 64 rows (two row groups per wave)
        if (M.mt <= 2) fh_body<NP, 2>(G, M, slice, rb, smem);
        else fh_body<NP, 4>(G, M, slice, rb, smem);
    } else {
        if (M.mt <= 2) fh_body<NP, 2>(G, M, slice, rb, smem);
        else if (M.mt == 3) fh_body<NP, 3>(G, M, slice, rb, smem);
        else if (M.mt == 4) fh_body<NP, 4>(G, M, slice, rb, smem);       // (round 6: samples of 49..64 rows — AVSD-length answers; host: fh_plan_mha_at)
        else fh_body<NP, 5>(G, M, slice, rb, smem);
    }
}

// ------------------------------------------------------------------------------------------ host side
static int fh_enabled = -1;          // -1: not decided yet (environment MTN_FUSED=0 turns the fused launches off)
extern "C" int mtn_fused_enable(int on) {
    const int prev = fh_enabled;
    fh_enabled = on ? 1 : 0;
    return prev;
}
static bool fh_env_off() {
    if (fh_enabled < 0) fh_enabled = 1;          // (mtn_fused_enable() switches the fused launches off / on: A/B measurements, tests)
    return fh_enabled == 0;
}

int fh_is_enabled() { return fh_env_off() ? 0 : 1; }

static constexpr int FH_LDS_MAX = 160 * 1024;
// row tiles a workgroup may have: {2, 3, 5} in the 1- and 3-block kernels — plus 4 for members of more than 32 rows per sample (round 6: a 56-row
// answer took an 80-row tile; members of <= 32 rows keep {2, 3, 5}, so the benchmark's launches are unchanged) —, {2, 4} in the 4-block kernel
static const int fh_mt_sets[2][4] = {{2, 3, 4, 5}, {2, 4, 4, 4}};

struct FhPlan { int blk, mt, lds, late_v; };
static int fh_member_lds(const mtn_mha_args& A, int blk, int mt, bool late_v = false) {
    const bool self = A.self_attn != 0, raw = !self && !A.kv_ready;
    const int m = self ? A.a : A.m, qa = A.mask_sq ? A.a : 1;
    const int kind = self ? FH_SELF : (raw ? FH_CROSS_RAW : FH_CROSS_READY);
    return fh_lds_map(mt, raw, (self || raw) ? mt * 16 : blk * m, fh_pad_rows(kind, m), blk * qa * m, 0, late_v).total;
}
// Rows per workgroup for an attention member: whole samples, inside the row tiles (<= 80 rows) and the LDS; as few workgroups as
// it takes to stay within the member's share of one round of the chip (these launches are bound by the bytes each CU pulls:
// weight slice + x rows), but not fewer.  Returns blk = 0 when even one sample does not fit.
// the plan for exactly `blk` samples per workgroup (blk = 0 in the result: does not fit the row tiles, the mask staging or the LDS)
static FhPlan fh_plan_mha_at(const mtn_mha_args& A, int np, int blk) {
    const int* fh_mt_choices = fh_mt_sets[np == 4];
    const bool self = A.self_attn != 0, raw = !self && !A.kv_ready;
    const int m = self ? A.a : A.m, qa = A.mask_sq ? A.a : 1;
    FhPlan none = {0, 0, 0, 0};
    const int R = blk * A.a, Rm = blk * m;
    int mt = 0;
    for (int c = 0; c < 4; ++c)
        if (fh_mt_choices[c] * 16 >= R && (!raw || fh_mt_choices[c] * 16 >= Rm) && (np == 4 || fh_mt_choices[c] != 4 || A.a > 32)) { mt = fh_mt_choices[c]; break; }
    if (!mt) return none;
    if (A.mask && blk * qa * m > FH_THREADS * FH_MASKB) return none;
    int lds = fh_member_lds(A, blk, mt), late = 0;
    if (lds > FH_LDS_MAX && !self && !raw) {       // K + V images of a long projected memory: V over the (dead) xn image
        lds = fh_member_lds(A, blk, mt, true);
        late = 1;
    }
    if (lds > FH_LDS_MAX) return none;
    return FhPlan{blk, mt, lds, late};
}
static FhPlan fh_plan_mha(const mtn_mha_args& A, int budget, int np) {
    const bool self = A.self_attn != 0;
    const int m = self ? A.a : A.m, qa = A.mask_sq ? A.a : 1;
    FhPlan best = {0, 0, 0, 0};
    if (A.mask && A.mask_sb != 0 && A.mask_sb != (long)qa * m) return best;      // the block's mask bytes must be contiguous
    if (A.mask && A.mask_sq != 0 && A.mask_sq != m) return best;
    for (int blk = 1; blk <= A.B; ++blk) {
        const FhPlan pl = fh_plan_mha_at(A, np, blk);
        if (pl.blk == 0) break;
        best = pl;
        if (((A.B + blk - 1) / blk) * (FH_D / FH_DK) <= budget) break;
    }
    return best;
}

// XCD map of a member: (hg, sg) with hg * sg = 8, sg | row-block count, hg | slice count, least bytes crossing the fabric:
// 8 XCDs x (x bytes / sg + weight bytes / hg).
static void fh_pick_xcd_map(FhMember& M, int nrb, double x_bytes, double w_bytes) {
    double best = 1e30;
    M.hg = 8; M.sg = 1;
    for (int sg = 1; sg <= 8; sg *= 2) {
        const int hg = 8 / sg;
        if (nrb % sg != 0 || M.nslice % hg != 0) continue;
        const double c = x_bytes / sg + w_bytes / hg;
        if (c < best) { best = c; M.hg = hg; M.sg = sg; }
    }
}

struct FhLaunch { FhGroup G; int wgs, np; size_t lds; };
// The whole launch plan; false = this group keeps the four-launch path.
static bool fh_plan(int n_mha, const mtn_mha_args* mha, int n_ffn, const mtn_ffn_args* ffn, FhLaunch& P) {
    if (n_mha + n_ffn < 1 || n_mha + n_ffn > FH_MAX_MEMBERS) return false;
    FhGroup& G = P.G;
    memset(&G, 0, sizeof(G));
    // weight blocks (64 output columns) per workgroup: 1 = q only; 3 = q | k | v; 4 = launches with feed-forward members
    // (256 columns of w_1 per workgroup: d_ff / 256 slices, 8 at d_ff = 2048 — one per XCD group like the heads)
    int np = n_ffn > 0 ? 4 : 1;
    for (int i = 0; i < n_mha; ++i) {
        if (mha[i].d != FH_D || mha[i].h != FH_D / FH_DK) return false;
        if ((mha[i].self_attn || !mha[i].kv_ready) && np < 3) np = 3;
    }
    for (int i = 0; i < n_ffn; ++i)
        if (ffn[i].d != FH_D || ffn[i].d_ff % 256 != 0) return false;
    const int members = n_mha + n_ffn;
    const int budget = 256 / members > 8 ? 256 / members : 8;       // share of one round (256 workgroups) per member
    int n = 0, wgs = 0;
    size_t lds = 0;
    // one member = samples [s0, s0 + Bp) of attention sublayer `a`, in workgroups of pl.blk samples
    auto emit_mha = [&](const mtn_mha_args& a, const FhPlan& pl, const int s0, const int Bp) {
        FhMember& M = G.m[n];
        const int mm = a.self_attn ? a.a : a.m;
        const size_t r0 = (size_t)s0 * a.a, rm0 = (size_t)s0 * mm;
        M.kind = a.self_attn ? FH_SELF : (a.kv_ready ? FH_CROSS_READY : FH_CROSS_RAW);
        M.a = a.a; M.m = mm; M.blk = pl.blk; M.mt = pl.mt; M.late_v = pl.late_v;
        {   // key ranges per (sample, 16 query rows): all 8 waves of a workgroup that holds few pairs and a long memory
            const bool ksplit = true;
            const int npairs = pl.blk * ((a.a + 15) / 16), chunks = (M.m + 63) / 64;
            int nks = 1;
            if (ksplit && npairs <= 4 && chunks >= 4 && !a.self_attn && a.kv_ready) {     // (long memories projected ahead of the layer loop)
                nks = npairs == 1 ? 8 : (npairs == 2 ? 4 : 2);
                while (nks > chunks) nks >>= 1;
                // the partials of a pair's non-first waves (FH_PART_FLOATS x 64 floats each) meet over the K image
                const int kimg = ((pl.blk * M.m + fh_pad_rows(FH_CROSS_READY, M.m) + 7) & ~7) * FH_HROWB;
                if (npairs * (nks - 1) * FH_PART_FLOATS * 64 * 4 > kimg) nks = 1;
            }
            M.nks = nks;
        }
        const int ldo = a.self_attn ? 3 * FH_D : FH_D;
        M.rows = Bp * a.a; M.rows_per_wg = pl.blk * a.a; M.nslice = FH_D / FH_DK; M.b_off = s0;
        M.eps = a.ln_eps; M.x = a.x + r0 * FH_D; M.ln_a = a.ln_a; M.ln_b = a.ln_b;
        M.w = (const bf16_t*)a.w_qkv; M.bias = a.b_qkv; M.mem = a.mem ? (const bf16_t*)a.mem + rm0 * FH_D : nullptr;
        M.xn = (bf16_t*)a.xn + r0 * FH_D; M.mean = a.mean + r0; M.rstd = a.rstd + r0;
        M.out = (bf16_t*)a.qkv + r0 * ldo; M.ld_out = ldo; M.kv = a.kv ? (bf16_t*)a.kv + rm0 * 2 * FH_D : nullptr;
        M.mask = a.mask ? a.mask + (size_t)s0 * a.mask_sb : nullptr; M.mask_sb = a.mask_sb; M.mask_sq = a.mask_sq; M.drop = a.drop_attn;
        M.o = (bf16_t*)a.o + r0 * FH_D; M.lse = a.lse ? a.lse + 2 * (size_t)s0 * (FH_D / FH_DK) * a.a : nullptr;
        G.wg_start[n] = wgs;
        const int nrb = (Bp + pl.blk - 1) / pl.blk;
        fh_pick_xcd_map(M, nrb, (double)M.rows * FH_D * 4, (M.kind == FH_CROSS_READY ? 1.0 : 3.0) * FH_D * FH_D * 2);
        wgs += nrb * M.nslice;                            // a multiple of 8: the next member starts on XCD 0 again
        lds = (size_t)pl.lds > lds ? (size_t)pl.lds : lds;
        ++n;
    };
    FhPlan plan[MTN_SUBLAYER_MAX_GROUP];
    for (int i = 0; i < n_mha; ++i) {
        plan[i] = fh_plan_mha(mha[i], budget, np);
        if (plan[i].blk == 0) return false;
    }
    // (Two unit sizes for launches that need a round and a half of the chip — batch 64 — were measured in round 4 and LOSE: a unit's time is
    // mostly fixed, profiles/r04_l_two_unit_sizes_batch64.txt.  Every member keeps its uniform units.)
    for (int i = 0; i < n_mha; ++i) emit_mha(mha[i], plan[i], 0, mha[i].B);
    for (int i = 0; i < n_ffn; ++i) {
        const mtn_ffn_args& a = ffn[i];
        FhMember& M = G.m[n];
        M.kind = FH_FFN;
        M.rows = a.rows; M.ncols = a.d_ff; M.nslice = a.d_ff / 256;
        int mt = 4;                                       // rows per workgroup: 32 or 64; 32 if that stays within the member's share
        if (((a.rows + 31) / 32) * M.nslice <= budget) mt = 2;
        M.mt = mt; M.rows_per_wg = mt * 16; M.a = 1; M.m = 1; M.blk = mt * 16;
        M.eps = a.ln_eps; M.x = a.x; M.ln_a = a.ln_a; M.ln_b = a.ln_b;
        M.w = (const bf16_t*)a.w1; M.bias = a.b1;
        M.xn = (bf16_t*)a.xn; M.mean = a.mean; M.rstd = a.rstd;
        M.out = (bf16_t*)a.hid; M.ld_out = a.d_ff; M.drop = a.drop_hidden;
        G.wg_start[n] = wgs;
        int nrb = (a.rows + M.rows_per_wg - 1) / M.rows_per_wg;
        while ((nrb * M.nslice) % 8 != 0) ++nrb;          // padding row blocks (their workgroups leave at once): members start on XCD 0
        fh_pick_xcd_map(M, nrb, (double)a.rows * FH_D * 4, (double)a.d_ff * FH_D * 2);
        wgs += nrb * M.nslice;
        const size_t l = (size_t)fh_lds_map(mt, false, 0, 0, 0, 4).total;
        lds = l > lds ? l : lds;
        ++n;
    }
    G.count = n;
    for (int i = n; i <= FH_MAX_MEMBERS; ++i) G.wg_start[i] = wgs;
    { static const int stop = [] { const char* e = getenv("MTN_FH_STOP"); return e ? atoi(e) : 0; }(); G.stop = stop; }
    P.wgs = wgs; P.np = np; P.lds = lds;
    return true;
}

// Can this group take the fused first launch?  (bf16, d = 512, h = 8, shapes inside the kernel's tiling)
int fh_group_eligible(int dtype, int n_mha, const mtn_mha_args* mha, int n_ffn, const mtn_ffn_args* ffn) {
    if (fh_env_off() || dtype != MTN_BF16) return 0;
    FhLaunch P;
    return fh_plan(n_mha, mha, n_ffn, ffn, P) ? 1 : 0;
}

template <int NP> static int fh_launch(const FhGroup& G, int wgs, size_t lds, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)fused_head_fwd_kernel<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, FH_LDS_MAX) != hipSuccess) {
            mtn_set_error("fused_head_fwd_kernel: cannot raise the dynamic LDS limit");
            return MTN_ERR_LAUNCH;
        }
        attr = true;
    }
    hipLaunchKernelGGL((fused_head_fwd_kernel<NP>), dim3(wgs), dim3(FH_THREADS), lds, s, G);
    return MTN_OK;
}

int fh_group_fwd_stage1(int n_mha, const mtn_mha_args* mha, int n_ffn, const mtn_ffn_args* ffn, void* stream) {
    FhLaunch P;
    MTN_CHECK_ARG(fh_plan(n_mha, mha, n_ffn, ffn, P), "group outside the fused kernel's tiling");
    hipStream_t s = (hipStream_t)stream;
    const int rc = P.np == 4 ? fh_launch<4>(P.G, P.wgs, P.lds, s) : (P.np == 3 ? fh_launch<3>(P.G, P.wgs, P.lds, s) : fh_launch<1>(P.G, P.wgs, P.lds, s));
    if (rc != MTN_OK) return rc;
    MTN_CHECK_LAUNCH();
    return MTN_OK;
}
