// gemm_k512.hip — y = x W^T + b for launches of many large problems with a SHORT contraction (K = d_model = 512): the K|V
// projections of the encoder-side memories for all decoder layers (ops.project_memories; mtn.py:257-258 for 3 + F memories x N
// layers), bf16 in, bf16 out.  The staged GEMM kernels get 8 dependent k-steps out of K = 512 — every step a memory latency —
// and the 128 x 128 variant of gemm.hip measured 179 us for the two launches against 153 us for the 64 x 64 register-staged
// kernel.  Here a 512-thread workgroup owns a 128 x 128 output tile and has ALL of its operands in flight at once, the
// construction of the fused kernels: the x tile [128][512] by LDS-DMA into a swizzled row image (128 KiB), the
// W tile [128][512] as MFMA fragments in registers (wave w = output columns 16w .. 16w+15; coalesced loads — lane 4r + c reads row r,
// 16-byte chunk c — put in operand order with ds_bpermute), one wait, 1024 MFMAs, bias, bf16, and the tile leaves through LDS as
// whole 256-byte row segments.
#include "fused_common.h"

static constexpr int GK_THREADS = 512;
static constexpr int GK_K = 512;
static constexpr int GK_LDS = 128 * FH_ROWB;              // 128 KiB: the x image; reused as the output staging area
static constexpr int GK_CPITCH = 272;                     // bytes per staged output row (256 + 16)

struct GkProblem { const bf16_t* A; const bf16_t* B; const float* bias; bf16_t* out; int lda, ldb, ldc, M, N, tiles_m; };
struct GkGroup {
    int count;
    int tile_start[MTN_GEMM_MAX_GROUP + 1];
    GkProblem p[MTN_GEMM_MAX_GROUP];
};

__global__ __launch_bounds__(GK_THREADS) void gemm_k512_kernel(const GkGroup G) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int g = 0;
    while (g + 1 < G.count && (int)blockIdx.x >= G.tile_start[g + 1]) ++g;
    const GkProblem& P = G.p[g];
    const int t = (int)blockIdx.x - G.tile_start[g];
    // row tiles vary fastest: consecutive workgroups (= different XCDs) share the W tile and read different x rows, so an XCD's
    // L2 sees 1/8 of the x rows (for every column tile) and W once
    const int tn = t / P.tiles_m, tm = t - tn * P.tiles_m;
    const int row0 = tm * 128, col0 = tn * 128;
    const int R = (P.M - row0) < 128 ? (P.M - row0) : 128;
    // ---- everything issued now
    {
        const unsigned ldab = (unsigned)P.lda * 2u;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(P.A + (size_t)row0 * P.lda), 0, (R - 1) * ldab + FH_ROWB, 0x00020000);
        for (int r = wave; r < 128; r += 8) {
            const unsigned vo = r < R ? (unsigned)r * ldab + (unsigned)((lane ^ (r & 15)) << 4) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (fh_lds_void_t*)(smem + r * FH_ROWB), 16, vo, 0, 0, 0);
        }
    }
    uint4 wf[16];
    {
        int n = col0 + 16 * wave + (lane >> 2);
        n = n < P.N ? n : P.N - 1;
        const bf16_t* wrow = P.B + (size_t)n * P.ldb + (lane & 3) * 8;
#pragma unroll
        for (int s = 0; s < 16; ++s) wf[s] = *(const uint4*)(wrow + s * 32);
    }
    const int colq = col0 + 16 * wave + 4 * lg;                  // this lane's four output columns
    float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (P.bias && colq < P.N) bq = *(const float4*)(P.bias + colq);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
        const int src = (4 * l15 + lg) * 4;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            wf[s].x = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)wf[s].x);
            wf[s].y = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)wf[s].y);
            wf[s].z = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)wf[s].z);
            wf[s].w = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)wf[s].w);
        }
    }
    // ---- C[16 mt + l15][16 wave + 4 lg + j]  (wave = 16 output columns, all 128 rows: every wave reads the whole x image from LDS;
    // 64 rows x 32 columns per wave halves those reads but doubles the W loads — measured 158 vs 136 us for the two launches: the
    // load phase is what bounds a tile)
    f32x4_t acc[8];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) acc[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) mma16<bf16_t>(acc[mt], wf[s], fh_xfrag(smem, mt * 16 + l15, s * 4 + lg));
    __syncthreads();                                             // the x image is dead: it becomes the output staging area
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        const int r = mt * 16 + l15;
        *(uint2*)(smem + r * GK_CPITCH + (16 * wave + 4 * lg) * 2) =
            make_uint2(fh_pack2(acc[mt][0] + bq.x, acc[mt][1] + bq.y), fh_pack2(acc[mt][2] + bq.z, acc[mt][3] + bq.w));
    }
    __syncthreads();
    // whole 256-byte row segments: 16 lanes per row, 32 rows per pass
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int r = ps * 32 + (tid >> 4), c = tid & 15;
        const int col = col0 + c * 8;
        if (r < R && col < P.N)                                  // N % 8 == 0: the 8 columns are in or out together
            *(uint4*)(P.out + (size_t)(row0 + r) * P.ldc + col) = *(const uint4*)(smem + r * GK_CPITCH + c * 16);
    }
}

// ====================================================================================================================
// Wide form (round 4): the same one-shot construction on a 128 x 256 tile.  An ablation of the kernel above
// (profiles/r04_ae_k512_ablation.txt: 76 us per launch; without the x image 65, without the W loads 51, without the stores 69,
// with 1/16 of the MFMAs and of their LDS reads 32) showed that its COMPUTE phase is the largest part: every MFMA (16 K FLOP)
// reads a 1 KiB x fragment from LDS — 16 FLOP per LDS byte against the 32 the matrix cores need at 128 B/clk — so the phase runs
// at LDS speed.  Here a wave owns two 16-column blocks (32 of the tile's 256 columns): an x fragment feeds two MFMAs, the LDS
// reads per FLOP halve, and an x row tile is re-read for 4 column tiles instead of 8.  W fragments: 128 VGPRs.
// ====================================================================================================================
static constexpr int GW_CPITCH = 528;                     // bytes per staged output row (512 + 16)
__global__ __launch_bounds__(GK_THREADS) void gemm_k512w_kernel(const GkGroup G) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int g = 0;
    while (g + 1 < G.count && (int)blockIdx.x >= G.tile_start[g + 1]) ++g;
    const GkProblem& P = G.p[g];
    const int t = (int)blockIdx.x - G.tile_start[g];
    const int tn = t / P.tiles_m, tm = t - tn * P.tiles_m;      // row tiles vary fastest (see gemm_k512_kernel)
    const int row0 = tm * 128, col0 = tn * 256;
    const int R = (P.M - row0) < 128 ? (P.M - row0) : 128;
    {
        const unsigned ldab = (unsigned)P.lda * 2u;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(P.A + (size_t)row0 * P.lda), 0, (R - 1) * ldab + FH_ROWB, 0x00020000);
        for (int r = wave; r < 128; r += 8) {
            const unsigned vo = r < R ? (unsigned)r * ldab + (unsigned)((lane ^ (r & 15)) << 4) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (fh_lds_void_t*)(smem + r * FH_ROWB), 16, vo, 0, 0, 0);
        }
    }
    float4 bq[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int colq = col0 + 128 * cb + 16 * wave + 4 * lg;
        bq[cb] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (P.bias && colq < P.N) bq[cb] = *(const float4*)(P.bias + colq);
    }
    __builtin_amdgcn_sched_barrier(0);
    // W fragments in CONTRACTION order (step s of both column blocks, then step s + 1, ...): the MFMAs of step s start as soon as its
    // two fragments have landed, while the later ones are still being accepted — a wave's 32 loads take ~6 us to issue, and the kernel
    // used to wait for all of them before its first MFMA.
    uint4 wf[2][16];                                             // column block cb: columns col0 + 128 cb + 16 wave .. + 15
    {
        const bf16_t* wrow[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            int n = col0 + 128 * cb + 16 * wave + (lane >> 2);
            n = n < P.N ? n : P.N - 1;
            wrow[cb] = P.B + (size_t)n * P.ldb + (lane & 3) * 8;
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) { wf[0][s] = *(const uint4*)(wrow[0] + s * 32); wf[1][s] = *(const uint4*)(wrow[1] + s * 32); }
    }
    __builtin_amdgcn_sched_barrier(0);
    // the x image (and the bias) has landed once at most the 32 W loads behind it fly: s_waitcnt vmcnt(32) lgkmcnt(15) expcnt(7) as a
    // BUILTIN, so that the compiler's own counter knows that no LDS-DMA is outstanding any more (behind an inline-asm wait it would put
    // vmcnt(0) in front of the first LDS read it sees)
    __builtin_amdgcn_s_waitcnt(0x8F70);
    __builtin_amdgcn_s_barrier();
    f32x4_t acc[2][8];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) acc[cb][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int src = (4 * l15 + lg) * 4;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {                         // coalesced load order -> MFMA operand order
            wf[cb][s].x = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)wf[cb][s].x);
            wf[cb][s].y = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)wf[cb][s].y);
            wf[cb][s].z = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)wf[cb][s].z);
            wf[cb][s].w = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)wf[cb][s].w);
        }
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const uint4 xf = fh_xfrag(smem, mt * 16 + l15, s * 4 + lg);
            mma16<bf16_t>(acc[0][mt], wf[0][s], xf);
            mma16<bf16_t>(acc[1][mt], wf[1][s], xf);
        }
    }
    __syncthreads();                                             // the x image is dead: it becomes the output staging area
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const int r = mt * 16 + l15;
            *(uint2*)(smem + r * GW_CPITCH + (128 * cb + 16 * wave + 4 * lg) * 2) =
                make_uint2(fh_pack2(acc[cb][mt][0] + bq[cb].x, acc[cb][mt][1] + bq[cb].y), fh_pack2(acc[cb][mt][2] + bq[cb].z, acc[cb][mt][3] + bq[cb].w));
        }
    __syncthreads();
    // whole 512-byte row segments: 32 lanes per row, 16 rows per pass
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
        const int r = ps * 16 + (tid >> 5), c = tid & 31;
        const int col = col0 + c * 8;
        if (r < R && col < P.N)
            *(uint4*)(P.out + (size_t)(row0 + r) * P.ldc + col) = *(const uint4*)(smem + r * GW_CPITCH + c * 16);
    }
}

// ====================================================================================================================
// Persistent form (round 3).  The kernel above runs one workgroup per CU through load-everything -> wait -> compute -> store:
// the CU's MFMAs idle while its 256 KiB arrive and its memory path idles while they run (6.9 rounds x 11 us for 1 768 tiles).
// Here 256 workgroups stay resident (one per CU) and each walks a list of UNITS = 64 rows x 128 columns of one problem:
//   * the W tile [128][512] is loaded ONCE per (problem, column tile) and stays in registers across the unit's row tiles;
//   * the x tile [64][512] arrives by LDS-DMA into one of two 64 KiB images, TWO units ahead of the MFMAs (counted vmcnt: the
//     8 instructions of the newest image may still fly), so the loads of unit i+1 run under the MFMAs, staging and stores of unit i;
//   * the output tile leaves through a separate staging area as whole 256-byte row segments.
// Unit -> workgroup map: XCD x (= blockIdx % 8: workgroups are dealt to the XCDs round-robin) owns the row tiles tm = x (mod 8)
// of every problem, for ALL column tiles — its L2 sees an eighth of the x rows, once — and inside an XCD workgroup j of 32 takes
// units u = j, j + 32, ... of the (row tile, column tile) list with the column tile fastest: with 8 (or 4, 2, 16) column tiles a
// workgroup keeps ONE column tile, and the 8 workgroups that share a row tile walk their lists side by side.
// ====================================================================================================================
static constexpr int GP_ROWS = 64;
static constexpr int GP_IMG = GP_ROWS * FH_ROWB;           // 64 KiB
static constexpr int GP_STG = 2 * GP_IMG;                  // staging area behind the two images
static constexpr int GP_LDS = GP_STG + GP_ROWS * GK_CPITCH;   // 148 480 B
static constexpr int GP_GRID = 256;

struct GpUnit { int p, tn, tm; };                          // p < 0: none

__device__ __forceinline__ GpUnit gp_first_from(const GkGroup& G, int p, int u, int xcd, int j) {
    // first unit at or after (problem p, list position u) of this workgroup; u < 0: start of problem p
    for (; p < G.count; ++p, u = -1) {
        const GkProblem& P = G.p[p];
        const int ntn = (P.N + 127) >> 7, tm64 = (P.M + GP_ROWS - 1) / GP_ROWS;
        const int nr = tm64 > xcd ? (tm64 - xcd + 7) >> 3 : 0;
        if (u < 0) u = (j + 5 * p) & 31;                   // rotate the workgroups per problem: short lists do not always feed the same ones
        if (u < nr * ntn) return GpUnit{p, u % ntn, xcd + 8 * (u / ntn)};
    }
    return GpUnit{-1, 0, 0};
}

__global__ __launch_bounds__(GK_THREADS, 2) void gemm_k512p_kernel(const GkGroup G) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;

    auto unit_u = [&](const GpUnit& c) {                   // list position of a unit (inverse of gp_first_from's map)
        const int ntn = (G.p[c.p].N + 127) >> 7;
        return ((c.tm - xcd) >> 3) * ntn + c.tn;
    };
    auto next_of = [&](const GpUnit& c) { return gp_first_from(G, c.p, unit_u(c) + 32, xcd, j); };
    // Roles: waves 0-3 bring the x images (16 LDS-DMA instructions each per image), waves 4-7 store the output tiles.  vmcnt is
    // per wave and counts loads and stores alike, which complete out of order with each other: a wave that did both could not
    // let the newest image fly without first draining its stores (measured: 6-7 us per unit instead of 1.3).
    const bool loader = wave < 4;
    auto issue_x = [&](const GpUnit& c, unsigned char* img) {
        if (!loader || c.p < 0) return;
        const GkProblem& P = G.p[c.p];
        const int row0 = c.tm * GP_ROWS;
        const int R = (P.M - row0) < GP_ROWS ? (P.M - row0) : GP_ROWS;
        const unsigned ldab = (unsigned)P.lda * 2u;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(P.A + (size_t)row0 * P.lda), 0, (R - 1) * ldab + FH_ROWB, 0x00020000);
#pragma unroll
        for (int i = 0; i < GP_ROWS / 4; ++i) {
            const int r = wave + 4 * i;
            const unsigned vo = r < R ? (unsigned)r * ldab + (unsigned)((lane ^ (r & 15)) << 4) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (fh_lds_void_t*)(img + r * FH_ROWB), 16, vo, 0, 0, 0);
        }
    };
    uint4 wf[16];
    float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
    auto issue_w = [&](const GpUnit& c) {                  // coalesced: lane 4r + c reads (row r, 16-byte chunk c); operand order by ds_bpermute
        const GkProblem& P = G.p[c.p];
        int n = c.tn * 128 + 16 * wave + (lane >> 2);
        n = n < P.N ? n : P.N - 1;
        const bf16_t* wrow = P.B + (size_t)n * P.ldb + (lane & 3) * 8;
#pragma unroll
        for (int s = 0; s < 16; ++s) wf[s] = *(const uint4*)(wrow + s * 32);
        const int colq = c.tn * 128 + 16 * wave + 4 * lg;
        bq = make_float4(0.f, 0.f, 0.f, 0.f);
        if (P.bias && colq < P.N) bq = *(const float4*)(P.bias + colq);
    };

    // W fragments into operand order as soon as they land — inside the block that loaded them, so that no pending register load
    // crosses the loop's back edge: the compiler would otherwise wait vmcnt(0) in front of every unit's first MFMA (it cannot know
    // how many image instructions follow the W loads on every path) and the x prefetch would never overlap anything.
    auto load_w = [&](const GpUnit& c) {
        issue_w(c);
        const int src = (4 * l15 + lg) * 4;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            wf[s].x = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)wf[s].x);
            wf[s].y = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)wf[s].y);
            wf[s].z = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)wf[s].z);
            wf[s].w = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)wf[s].w);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
#ifdef GP_STAMPS
    // timing build: shader-clock stamps of workgroup GP_STAMPS's lane 0 of wave 0 and wave 4, written behind problem 0's output
    long long* dbg_ = (long long*)(G.p[0].out + (size_t)G.p[0].M * G.p[0].ldc) + (wave >= 4 ? 512 : 0);
    int dbg_n_ = 0;
#define GP_STAMP() do { if ((int)blockIdx.x == GP_STAMPS && (tid == 0 || tid == 256) && dbg_n_ < 500) dbg_[dbg_n_++] = (long long)wall_clock64(); } while (0)
#else
#define GP_STAMP() do { } while (0)
#endif
    GpUnit cur = gp_first_from(G, 0, -1, xcd, j);
    if (cur.p < 0) return;
    GpUnit nxt = next_of(cur);
    GP_STAMP();
    issue_x(cur, smem);
    if (nxt.p >= 0) issue_x(nxt, smem + GP_IMG);
    load_w(cur);                                                       // (waits for the two images too: once per workgroup)
    GP_STAMP();
    for (int i = 0;; ++i) {
        // image i (and a W tile issued before the newest image) has landed once at most the newest image's 16 instructions fly:
        // loads return in order among loads (the loader waves issue nothing else)
        if (loader) {
            if (nxt.p >= 0) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        GP_STAMP();                                                    // [5i+2] image wait over
        __builtin_amdgcn_s_barrier();
        GP_STAMP();                                                    // [5i+3] barrier A passed
        const unsigned char* img = smem + (i & 1) * GP_IMG;
        f32x4_t acc[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        {   // x fragments of step s+1 in flight under the MFMAs of step s (two register sets).  The reads are INLINE ASM on purpose:
            // the compiler puts s_waitcnt vmcnt(0) in front of any LDS read it can see while LDS-DMA instructions are outstanding
            // (it cannot tell the image being read from the one being filled) — which would drain the prefetched image before every
            // unit.  Ordering is by hand: the counted vmcnt + barrier above, one lgkmcnt(0) per step here.
            u32x4_t xa[4], xb[4];
            const unsigned ibase = (unsigned)(size_t)img;
            auto xload = [&](u32x4_t* x, int s) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int row = mt * 16 + l15;
                    const unsigned addr = ibase + row * FH_ROWB + (((s * 4 + lg) ^ (row & 15)) << 4);
                    asm volatile("ds_read_b128 %0, %1" : "=v"(x[mt]) : "v"(addr));
                }
            };
            auto landed = [&](u32x4_t* x) {                            // (MTN_LANDED: common.h)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) MTN_LANDED(x[mt]);
                __builtin_amdgcn_sched_barrier(0);
            };
            xload(xa, 0);
#pragma unroll
            for (int s = 0; s < 16; s += 2) {
                landed(xa);
                xload(xb, s + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) mma16<bf16_t>(acc[mt], wf[s], as_uint4(xa[mt]));
                __builtin_amdgcn_sched_barrier(0);                     // (the MFMAs are not memory operations: without this fence the scheduler
                landed(xb);                                            //  moves the wait in front of them and nothing overlaps)
                if (s + 2 < 16) xload(xa, s + 2);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) mma16<bf16_t>(acc[mt], wf[s + 1], as_uint4(xb[mt]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        GP_STAMP();                                                    // [5i+4] MFMAs issued
        // C[16 mt + l15][16 wave + 4 lg + k] -> staging rows (the storing waves took the previous tile out of it before this unit's first barrier)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int r = mt * 16 + l15;
            *(uint2*)(smem + GP_STG + r * GK_CPITCH + (16 * wave + 4 * lg) * 2) =
                make_uint2(fh_pack2(acc[mt][0] + bq.x, acc[mt][1] + bq.y), fh_pack2(acc[mt][2] + bq.z, acc[mt][3] + bq.w));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        GP_STAMP();                                                    // [5i+5] staged
        __builtin_amdgcn_s_barrier();                                  // staging complete; every wave is done with image i
        GP_STAMP();                                                    // [5i+6] barrier B passed
        const GpUnit nn = nxt.p >= 0 ? next_of(nxt) : GpUnit{-1, 0, 0};
        const bool new_w = nxt.p >= 0 && (nxt.p != cur.p || nxt.tn != cur.tn);
        if (new_w) load_w(nxt);                                        // (every wave is done with the old fragments; waits for image i+1 as well)
        // the tile's stores go into the CU's memory pipeline BEFORE the next image's 64 LDS-DMA instructions: behind them they sat
        // until the image had landed (the storing waves took 4-5 us per unit, timeline in profiles/r03_k512_persistent.txt)
        if (loader) { __builtin_amdgcn_s_barrier(); issue_x(nn.p >= 0 ? nn : GpUnit{-1, 0, 0}, smem + (i & 1) * GP_IMG); }
        if (!loader) {                                                 // whole 256-byte row segments: 16 lanes per row, 16 rows per pass
            const GkProblem& P = G.p[cur.p];
            const int row0 = cur.tm * GP_ROWS, col0 = cur.tn * 128;
            const int R = (P.M - row0) < GP_ROWS ? (P.M - row0) : GP_ROWS;
            const int t4 = tid - 256;
            u32x4_t o[4];
            // (inline asm for the same reason as the fragment reads: a visible LDS read would be given s_waitcnt vmcnt(0) — here that
            //  is this wave's previous stores, one full store latency per read)
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const unsigned addr = (unsigned)(size_t)(smem + GP_STG + (ps * 16 + (t4 >> 4)) * GK_CPITCH + (t4 & 15) * 16);
                asm volatile("ds_read_b128 %0, %1" : "=v"(o[ps]) : "v"(addr));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the tile is in registers: it has left the staging area before this wave reaches the next barrier
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) MTN_LANDED(o[ps]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int r = ps * 16 + (t4 >> 4), c8 = t4 & 15;
                const int col = col0 + c8 * 8;
                if (r < R && col < P.N)
                    *(uint4*)(P.out + (size_t)(row0 + r) * P.ldc + col) = as_uint4(o[ps]);
            }
            __builtin_amdgcn_s_barrier();
        }
        if (nxt.p < 0) break;
        cur = nxt; nxt = nn;
    }
}

// -> 1 when the launch was taken (every problem: bf16, row-major x row-major, K = 512, plain bias epilogue into out_lp), 0 when the
// caller should use the general kernels, < 0 on a launch error
int gemm_k512_try(int count, const mtn_gemm_problem* p, int min_tiles, hipStream_t s, int* tiles_out) {
    if (count < 1 || count > MTN_GEMM_MAX_GROUP) return 0;
    GkGroup G;
    memset(&G, 0, sizeof(G));
    int tiles = 0;
    for (int i = 0; i < count; ++i) {
        const mtn_gemm_problem& q = p[i];
        if (q.a_trans || q.b_trans || q.K != GK_K || q.relu || q.gate || q.residual || q.out_f32 || q.rowsum_out || q.adam || !q.out_lp) return 0;
        if (q.drop.p > 0.f && q.drop.seed) return 0;
        if (q.M < 1 || q.N < 8 || (q.N & 7) || (q.lda & 7) || (q.ldb & 7) || (q.ldc & 7) || q.lda < GK_K || q.ldb < GK_K) return 0;
        if ((long)q.M * q.lda * 2 >= (1L << 31)) return 0;
        if (((size_t)q.A | (size_t)q.B | (size_t)q.out_lp) & 15) return 0;
        if (q.bias && ((size_t)q.bias & 15)) return 0;
        GkProblem& P = G.p[i];
        P.A = (const bf16_t*)q.A; P.B = (const bf16_t*)q.B; P.bias = q.bias; P.out = (bf16_t*)q.out_lp;
        P.lda = q.lda; P.ldb = q.ldb; P.ldc = q.ldc; P.M = q.M; P.N = q.N; P.tiles_m = (q.M + 127) / 128;
        G.tile_start[i] = tiles;
        tiles += P.tiles_m * ((q.N + 127) / 128);
    }
    if (tiles < min_tiles) return 0;
    G.count = count;
    for (int i = count; i <= MTN_GEMM_MAX_GROUP; ++i) G.tile_start[i] = tiles;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)gemm_k512_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GK_LDS) != hipSuccess) return 0;
        if (hipFuncSetAttribute((const void*)gemm_k512p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GP_LDS) != hipSuccess) return 0;
        attr = true;
    }
    // Persistent form: measured per launch in the step, same box (profiles/r03_u_k512_persistent_ab.txt): 69.4 vs 69.0 us at 14 units
    // per workgroup (cfg2, batch 32), 104 vs 123 us at 28 (batch 64) — the resident workgroups pay ~5 us of start-up per W tile and
    // their x images land in 4-5 us however few are in flight, so it wins only when the lists are long.  MTN_K512_PERSIST=1 / 0 forces.
    long units = 0;
    for (int i = 0; i < count; ++i) units += (long)((p[i].M + GP_ROWS - 1) / GP_ROWS) * ((p[i].N + 127) / 128);
    const char* pe = MTN_ENV("MTN_K512_PERSIST");
    if (pe ? pe[0] != '0' : units >= 20L * GP_GRID) {
        hipLaunchKernelGGL(gemm_k512p_kernel, dim3(GP_GRID), dim3(GK_THREADS), GP_LDS, s, G);
        if (hipGetLastError() != hipSuccess) return -1;
        if (tiles_out) *tiles_out = GP_GRID;
        return 1;
    }
    // wide tiles (128 x 256) when every problem's N fills them: half the LDS reads per FLOP (MTN_K512_WIDE=0: 128 x 128 tiles)
    bool wide = true;
    for (int i = 0; i < count; ++i) wide = wide && (p[i].N % 256) == 0;
    if (wide) {
        static bool attr_w = false;
        if (!attr_w) {
            if (hipFuncSetAttribute((const void*)gemm_k512w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GK_LDS) != hipSuccess) return 0;
            attr_w = true;
        }
        int tw = 0;
        for (int i = 0; i < count; ++i) { G.tile_start[i] = tw; tw += G.p[i].tiles_m * (p[i].N / 256); }
        for (int i = count; i <= MTN_GEMM_MAX_GROUP; ++i) G.tile_start[i] = tw;
        hipLaunchKernelGGL(gemm_k512w_kernel, dim3(tw), dim3(GK_THREADS), GK_LDS, s, G);
        if (hipGetLastError() != hipSuccess) return -1;
        if (tiles_out) *tiles_out = tw;
        return 1;
    }
    hipLaunchKernelGGL(gemm_k512_kernel, dim3(tiles), dim3(GK_THREADS), GK_LDS, s, G);
    if (hipGetLastError() != hipSuccess) return -1;
    if (tiles_out) *tiles_out = tiles;
    return 1;
}
