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
        attr = true;
    }
    hipLaunchKernelGGL(gemm_k512_kernel, dim3(tiles), dim3(GK_THREADS), GK_LDS, s, G);
    if (hipGetLastError() != hipSuccess) return -1;
    if (tiles_out) *tiles_out = tiles;
    return 1;
}
