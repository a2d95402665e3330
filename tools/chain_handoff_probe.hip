// chain_handoff_probe.hip — VERDICT r5 "Next round" item 1(a): would decode.hip's execution model (classes of workgroups that have
// already asked for their weights and sit in a poll, values handed over inside ONE launch) make the TRAINING forward chain faster?
//
// The chain at batch 32 is [fused LN + projection + attention launch] -> [grouped output-projection GEMM launch] -> next sublayer ...
// on 640 rows: ~170 dependent launches of 7-25 us.  This probe runs a faithful miniature of that alternation, both ways, on the same
// operands, and checks that both give the same bits:
//
//   stage P ("fused head" stand-in), unit = (member g, row block of 80 rows, head): 64 KB weight slice + its 80 x 512 bf16 x rows
//            -> LDS by LDS-DMA, [80 x 64 x 512] on MFMA, publishes its 80 x 64 bf16 block of o           (64 units per member)
//   stage C ("output projection" stand-in), unit = (member g, 64-row tile, 64-column tile): 64 KB weight tile + 64 x 512 rows of o
//            -> LDS, [64 x 64 x 512] on MFMA, + residual, writes its 64 x 64 block of x (read by the next P)  (80 units per member)
//   chain  = S stages alternating P, C (S = 12: six sublayers); every stage has its own weights, and consecutive iterations rotate
//            through NSETS weight sets (> 256 MiB in all: the Infinity Cache cannot hold them, as in the real step)
//
//   variant L (today): one launch per stage, every workgroup issues ALL its loads (weights + rows) at entry; S launches per chain,
//            replayed from a hipGraph.
//   variant H (decode.hip's model): ONE launch per chain; P-class and C-class workgroups; a workgroup asks for its NEXT unit's weight
//            slice before it polls the flags of the units it depends on (R1 of the guide: 16-byte sc1 write-through stores, drained,
//            then ONE sc1 flag per unit; the consumer polls flags from one wave and reads the rows with sc1 LDS-DMA).
//            A P unit (row block rb) depends on the C units of the 64-row tiles that overlap it (16 flags), a C unit (tile rt) on the P
//            units of the row blocks that overlap it (8-16 flags) — the real all-heads / whole-row dependencies of the chain.
//   G = members per stage (lockstep group size): G = 1 -> 64 + 80 units per stage pair (everything resident, one unit per workgroup);
//            G = 3 -> 192 + 240 units (today's launches at batch 32): the 256 resident workgroups of H are split 112 P + 144 C and walk
//            two units per stage each.
//
// Reported: us per stage (HIP events over whole chains), and from wall-clock stamps the seam itself: producer's first store ->
// consumer's rows usable in LDS (H) against last producer store -> consumer's operands usable (L: boundary + cold prologue).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/chain_handoff_probe.hip -o tools/chain_handoff_probe.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

typedef unsigned short bf16_t;
typedef unsigned long long u64;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned v2u_t;
typedef __attribute__((address_space(3))) void lds_void_t;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)

static constexpr int D = 512, ROWS = 640, RB = 80, NRB = 8, NH = 8, RT = 64, NRT = 10, NCT = 8;
static constexpr int PU = NRB * NH, CU_ = NRT * NCT;          // units per member: 64 P, 80 C
static constexpr int WSLICE = 64 * D;                         // elements of one weight slice / tile (64 KB)
static constexpr int NT = 512;                                // threads per workgroup (8 waves)
static constexpr int SC1 = 16;                                // buffer cache-policy bit: agent scope (bypass L1, write through)
static constexpr int LDS_W = 0, LDS_X = 65536, LDS_BYTES = 65536 + 81920;      // weight image | rows image (80 rows of 1 KiB)
static constexpr int MAXS = 16, NSTAMP = 6;

static bf16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }
__device__ __forceinline__ bf16_t d_f2bf(float f) { uint32_t u = __float_as_uint(f); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }
__device__ __forceinline__ float d_bf2f(bf16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

struct Params {
    const bf16_t* Wp;      // [S/2][G][8 heads][64][512]   (this iteration's set)
    const bf16_t* Wc;      // [S/2][G][8 column tiles][64][512]
    bf16_t* x;             // [G][640][512]
    bf16_t* o;             // [G][640][512]
    unsigned* flagP;       // [G][64]   (H) value = base + stage + 1 when the unit's block of o is published
    unsigned* flagC;       // [G][80]
    u64* stamps;           // [S][units of the stage <= G * 80][NSTAMP]
    unsigned base;         // launch generation * (MAXS + 1)
    int S, G, stage;       // (L: `stage` = the one stage this launch runs)
    int nP, nC;            // (H) class sizes
    int* err;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000); }

// `rows` rows of 1 KiB (512 bf16) starting at element row `row0` of a [.][512] bf16 matrix -> LDS image [row][64 chunks of 16 B], chunk c of
// row r stored in slot c ^ (r & 15) (fragment reads of 16 rows then touch 16 different slots).  One wave-instruction per row.
template <int AUX>
__device__ __forceinline__ void dma_rows(const __amdgpu_buffer_rsrc_t rs, unsigned char* img, const int rows, const int row0, const int wave, const int lane) {
    for (int r = wave; r < rows; r += NT / 64) {
        const unsigned voff = (unsigned)(row0 + r) * 1024u + (unsigned)((lane ^ (r & 15)) * 16);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(img + r * 1024), 16, voff, 0, 0, AUX);
    }
}
__device__ __forceinline__ bf16x8_t frag(const unsigned char* img, const int row, const int chunk) {
    return *(const bf16x8_t*)(img + row * 1024 + ((chunk ^ (row & 15)) << 4));
}
__device__ __forceinline__ void st8_sc1(const __amdgpu_buffer_rsrc_t rs, const unsigned off, const unsigned a, const unsigned b) {
    __builtin_amdgcn_raw_buffer_store_b64(v2u_t{a, b}, rs, (int)off, 0, SC1);
}
__device__ __forceinline__ unsigned ld_flag(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stamp(u64* st, const int i) { if (threadIdx.x == 0) st[i] = wall_clock64(); }

// wave 0: lanes < n poll flags[idx(lane)] until all equal `want`; bounded.  Returns false on timeout.
template <typename IDX>
__device__ __forceinline__ bool poll_flags(const unsigned* flags, const int n, const unsigned want, IDX idx, int* err) {
    __shared__ int ok_s;
    const int tid = threadIdx.x;
    if (tid < 64) {
        bool ok = false;
        for (unsigned spins = 0; spins < (1u << 22); ++spins) {
            const bool mine = tid >= n || ld_flag(flags + idx(tid)) == want;
            if (__builtin_amdgcn_ballot_w64(!mine) == 0) { ok = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (tid == 0) { ok_s = ok; if (!ok) atomicExch(err, 1); }
    }
    __syncthreads();
    return ok_s != 0;
}

// ---- one P unit: o[g][rb rows][head cols] = x[g][rb rows][:] . Wp[stage/2][g][head]^T     (weights image must be in LDS_W .. or on its way)
template <bool HANDOFF>
__device__ __forceinline__ void p_compute_store(const Params& P, unsigned char* smem, const int g, const int rb, const int h, const int wave, const int lane) {
    const int l15 = lane & 15, lg = lane >> 4;
    const int ct = wave & 3;                                   // column tile (16 of the head's 64 columns)
    const __amdgpu_buffer_rsrc_t rO = rsrc_of(P.o + (size_t)g * ROWS * D, ROWS * D * 2);
    for (int rt = wave >> 2; rt < RB / 16; rt += 2) {           // row tiles of 16: 5 per block
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < D / 32; ++ks) {
            const bf16x8_t a = frag(smem + LDS_X, rt * 16 + l15, ks * 4 + lg);
            const bf16x8_t b = frag(smem + LDS_W, ct * 16 + l15, ks * 4 + lg);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc, 0, 0, 0);          // swapped: lane (l15 = row, lg): columns 4 lg ..+3
        }
        const unsigned lo = (unsigned)d_f2bf(acc[0]) | ((unsigned)d_f2bf(acc[1]) << 16), hi = (unsigned)d_f2bf(acc[2]) | ((unsigned)d_f2bf(acc[3]) << 16);
        const unsigned off = ((unsigned)(rb * RB + rt * 16 + l15) * D + h * 64 + ct * 16 + lg * 4) * 2u;
        if (HANDOFF) st8_sc1(rO, off, lo, hi);
        else *(uint2*)((unsigned char*)(P.o + (size_t)g * ROWS * D) + off) = make_uint2(lo, hi);
    }
}
// ---- one C unit: x[g][rt rows][ct cols] = 0.5 x + o[g][rt rows][:] . Wc[stage/2][g][ct]^T
template <bool HANDOFF>
__device__ __forceinline__ void c_compute_store(const Params& P, unsigned char* smem, const int g, const int rt, const int ct, const uint2 res[2], const int wave, const int lane) {
    const int l15 = lane & 15, lg = lane >> 4;
    const int rh = wave >> 2, cq = wave & 3;                    // 32-row half, 16-column quarter
    const __amdgpu_buffer_rsrc_t rX = rsrc_of(P.x + (size_t)g * ROWS * D, ROWS * D * 2);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < D / 32; ++ks) {
            const bf16x8_t a = frag(smem + LDS_X, rh * 32 + t * 16 + l15, ks * 4 + lg);
            const bf16x8_t b = frag(smem + LDS_W, cq * 16 + l15, ks * 4 + lg);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc, 0, 0, 0);
        }
        const float r0 = d_bf2f((bf16_t)res[t].x), r1 = d_bf2f((bf16_t)(res[t].x >> 16)), r2 = d_bf2f((bf16_t)res[t].y), r3 = d_bf2f((bf16_t)(res[t].y >> 16));
        const unsigned lo = (unsigned)d_f2bf(0.5f * r0 + acc[0]) | ((unsigned)d_f2bf(0.5f * r1 + acc[1]) << 16);
        const unsigned hi = (unsigned)d_f2bf(0.5f * r2 + acc[2]) | ((unsigned)d_f2bf(0.5f * r3 + acc[3]) << 16);
        const unsigned off = ((unsigned)(rt * RT + rh * 32 + t * 16 + l15) * D + ct * 64 + cq * 16 + lg * 4) * 2u;
        if (HANDOFF) st8_sc1(rX, off, lo, hi);
        else *(uint2*)((unsigned char*)(P.x + (size_t)g * ROWS * D) + off) = make_uint2(lo, hi);
    }
}
__device__ __forceinline__ unsigned c_res_off(const int rt, const int ct, const int t, const int wave, const int lane) {
    return ((unsigned)(rt * RT + (wave >> 2) * 32 + t * 16 + (lane & 15)) * D + ct * 64 + (wave & 3) * 16 + (lane >> 4) * 4) * 2u;
}

// =============================================================== variant L: one launch per stage
__global__ __launch_bounds__(NT) void stage_p_kernel(const Params P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int u = blockIdx.x, g = u / PU, rb = (u % PU) / NH, h = u % NH;
    u64* st = P.stamps + ((size_t)P.stage * (P.G * CU_) + u) * NSTAMP;
    stamp(st, 0);
    const bf16_t* w = P.Wp + ((size_t)(P.stage / 2) * P.G + g) * NH * WSLICE + (size_t)h * WSLICE;
    dma_rows<0>(rsrc_of(w, WSLICE * 2), smem + LDS_W, 64, 0, wave, lane);
    dma_rows<0>(rsrc_of(P.x + (size_t)g * ROWS * D, ROWS * D * 2), smem + LDS_X, RB, rb * RB, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    stamp(st, 3);                                              // operands usable
    p_compute_store<false>(P, smem, g, rb, h, wave, lane);
    stamp(st, 4);                                              // stores issued
}
__global__ __launch_bounds__(NT) void stage_c_kernel(const Params P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int u = blockIdx.x, g = u / CU_, rt = (u % CU_) / NCT, ct = u % NCT;
    u64* st = P.stamps + ((size_t)P.stage * (P.G * CU_) + u) * NSTAMP;
    stamp(st, 0);
    const bf16_t* w = P.Wc + ((size_t)(P.stage / 2) * P.G + g) * NCT * WSLICE + (size_t)ct * WSLICE;
    dma_rows<0>(rsrc_of(w, WSLICE * 2), smem + LDS_W, 64, 0, wave, lane);
    dma_rows<0>(rsrc_of(P.o + (size_t)g * ROWS * D, ROWS * D * 2), smem + LDS_X, RT, rt * RT, wave, lane);
    uint2 res[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) res[t] = *(const uint2*)((const unsigned char*)(P.x + (size_t)g * ROWS * D) + c_res_off(rt, ct, t, wave, lane));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    stamp(st, 3);
    c_compute_store<false>(P, smem, g, rt, ct, res, wave, lane);
    stamp(st, 4);
}

// =============================================================== variant H: one launch per chain, two classes, flags
__global__ __launch_bounds__(NT) void chain_kernel(const Params P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = blockIdx.x;
    const bool isP = wg < P.nP;
    const int idx = isP ? wg : wg - P.nP, csize = isP ? P.nP : P.nC;
    const int units = P.G * (isP ? PU : CU_);
    // prefetch of a unit's weight image (everything a unit can ask for before its operands exist)
    auto weights = [&](const int s, const int u) {
        if (isP) {
            const int g = u / PU, h = u % NH;
            const bf16_t* w = P.Wp + ((size_t)(s / 2) * P.G + g) * NH * WSLICE + (size_t)h * WSLICE;
            dma_rows<0>(rsrc_of(w, WSLICE * 2), smem + LDS_W, 64, 0, wave, lane);
        } else {
            const int g = u / CU_, ct = u % NCT;
            const bf16_t* w = P.Wc + ((size_t)(s / 2) * P.G + g) * NCT * WSLICE + (size_t)ct * WSLICE;
            dma_rows<0>(rsrc_of(w, WSLICE * 2), smem + LDS_W, 64, 0, wave, lane);
        }
    };
    bool alive = true;
    int s = isP ? 0 : 1, u = idx;
    if (u < units) weights(s, u);
    for (; s < P.S && alive; s += 2) {
        for (u = idx; u < units && alive; u += csize) {
            u64* st = P.stamps + ((size_t)s * (P.G * CU_) + u) * NSTAMP;
            stamp(st, 0);
            if (isP) {
                const int g = u / PU, rb = (u % PU) / NH, h = u % NH;
                if (s > 0) {       // x rows of this block: the C units of the previous stage whose 64-row tiles overlap it, all 8 column tiles
                    const int t0 = (rb * RB) / RT, t1 = (rb * RB + RB - 1) / RT;
                    alive = poll_flags(P.flagC + g * CU_, (t1 - t0 + 1) * NCT, P.base + (unsigned)s, [&](int i) { return (t0 + i / NCT) * NCT + i % NCT; }, P.err);
                }
                stamp(st, 1);                                    // flags seen
                dma_rows<SC1>(rsrc_of(P.x + (size_t)g * ROWS * D, ROWS * D * 2), smem + LDS_X, RB, rb * RB, wave, lane);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                stamp(st, 3);                                    // operands usable
                p_compute_store<true>(P, smem, g, rb, h, wave, lane);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // write-through stores drained ...
                __syncthreads();
                stamp(st, 4);
                if (tid == 0) __hip_atomic_store(P.flagP + g * PU + rb * NH + h, P.base + (unsigned)s + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... then the flag
            } else {
                const int g = u / CU_, rt = (u % CU_) / NCT, ct = u % NCT;
                uint2 res[2];
                const __amdgpu_buffer_rsrc_t rX = rsrc_of(P.x + (size_t)g * ROWS * D, ROWS * D * 2);
#pragma unroll
                for (int t = 0; t < 2; ++t) {       // the residual: this unit's own block, written by the same unit two stages ago (sc1 load: L2)
                    const v2u_t v = __builtin_amdgcn_raw_buffer_load_b64(rX, (int)c_res_off(rt, ct, t, wave, lane), 0, SC1);
                    res[t] = make_uint2(v.x, v.y);
                }
                const int b0 = (rt * RT) / RB, b1 = (rt * RT + RT - 1) / RB;
                alive = poll_flags(P.flagP + g * PU, (b1 - b0 + 1) * NH, P.base + (unsigned)s, [&](int i) { return (b0 + i / NH) * NH + i % NH; }, P.err);
                stamp(st, 1);
                dma_rows<SC1>(rsrc_of(P.o + (size_t)g * ROWS * D, ROWS * D * 2), smem + LDS_X, RT, rt * RT, wave, lane);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                stamp(st, 3);
                c_compute_store<true>(P, smem, g, rt, ct, res, wave, lane);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                stamp(st, 4);
                if (tid == 0) __hip_atomic_store(P.flagC + g * CU_ + rt * NCT + ct, P.base + (unsigned)s + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // the next unit's weights go out before its poll (same stage's second unit, or the first unit of this class's next stage)
            const int un = u + csize;
            if (un < units) weights(s, un);
            else if (s + 2 < P.S && idx < units) weights(s + 2, idx);
        }
    }
}

// =============================================================== host
struct Stats { double med, p90, mx; };
static Stats stats_of(std::vector<double> v) {
    std::sort(v.begin(), v.end());
    Stats s{0, 0, 0};
    if (v.empty()) return s;
    s.med = v[v.size() / 2]; s.p90 = v[(size_t)(v.size() * 0.9)]; s.mx = v.back();
    return s;
}

int main(int argc, char** argv) {
    const int S = 12;
    const int iters = argc > 1 ? atoi(argv[1]) : 40;
    int n_cu = 0;
    CK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0));
    printf("# chain hand-off probe: %d stages per chain (P, C alternating), 640 rows x 512, %d compute units, %d chains timed per variant\n", S, n_cu, iters);
    CK(hipFuncSetAttribute((const void*)stage_p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)stage_c_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (int G : {1, 3}) {
        for (int cold = 1; cold >= 0; --cold) {
            const size_t set_elems = (size_t)(S / 2) * G * 8 * WSLICE;                 // per class
            const int nsets = cold ? (int)((300ull << 20) / (2 * set_elems * 2) + 1) : 1;
            bf16_t *Wp, *Wc, *x, *o, *x0;
            unsigned *flagP, *flagC;
            u64* stamps;
            int* err;
            CK(hipMalloc(&Wp, set_elems * 2 * nsets)); CK(hipMalloc(&Wc, set_elems * 2 * nsets));
            CK(hipMalloc(&x, (size_t)G * ROWS * D * 2)); CK(hipMalloc(&o, (size_t)G * ROWS * D * 2)); CK(hipMalloc(&x0, (size_t)G * ROWS * D * 2));
            CK(hipMalloc(&flagP, G * PU * 4)); CK(hipMalloc(&flagC, G * CU_ * 4));
            const size_t nst = (size_t)S * G * CU_ * NSTAMP;
            CK(hipMalloc(&stamps, nst * 8)); CK(hipMalloc(&err, 4));
            CK(hipMemset(flagP, 0, G * PU * 4)); CK(hipMemset(flagC, 0, G * CU_ * 4)); CK(hipMemset(err, 0, 4));
            {
                std::vector<bf16_t> hw(set_elems * nsets), hx((size_t)G * ROWS * D);
                uint32_t r = 12345u + G;
                auto rnd = [&]() { r = r * 1664525u + 1013904223u; return ((r >> 8) & 0xffff) / 65536.0f - 0.5f; };
                for (auto& v : hw) v = f2bf(rnd() * 0.12f);
                CK(hipMemcpy(Wp, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
                for (auto& v : hw) v = f2bf(rnd() * 0.12f);
                CK(hipMemcpy(Wc, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
                for (auto& v : hx) v = f2bf(rnd() * 2.0f);
                CK(hipMemcpy(x0, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
            }
            Params base{};
            base.x = x; base.o = o; base.flagP = flagP; base.flagC = flagC; base.stamps = stamps; base.S = S; base.G = G; base.err = err;
            // H classes: everything resident when the units fit (G = 1: one unit per workgroup), else 256 workgroups split by the classes' work
            base.nP = G * PU; base.nC = G * CU_;
            if (base.nP + base.nC > n_cu) { base.nP = (int)((long)n_cu * PU / (PU + CU_)) & ~7; base.nC = n_cu - base.nP; }
            unsigned gen = 0;
            auto chain_L = [&](int set) {
                Params p = base; p.Wp = Wp + (size_t)set * set_elems; p.Wc = Wc + (size_t)set * set_elems;
                for (int s = 0; s < S; ++s) {
                    p.stage = s;
                    if (s % 2 == 0) hipLaunchKernelGGL(stage_p_kernel, dim3(G * PU), dim3(NT), LDS_BYTES, st, p);
                    else hipLaunchKernelGGL(stage_c_kernel, dim3(G * CU_), dim3(NT), LDS_BYTES, st, p);
                }
            };
            auto chain_H = [&](int set) {
                Params p = base; p.Wp = Wp + (size_t)set * set_elems; p.Wc = Wc + (size_t)set * set_elems;
                p.base = (++gen) * (MAXS + 1);
                hipLaunchKernelGGL(chain_kernel, dim3(p.nP + p.nC), dim3(NT), LDS_BYTES, st, p);
            };
            // graphs of `nsets` chains each (the weight set rotates inside the graph; kernel arguments are frozen at capture)
            hipGraph_t gL, gH; hipGraphExec_t eL, eH;
            const int per_graph = cold ? nsets : 4;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
            for (int i = 0; i < per_graph; ++i) chain_L(i % nsets);
            CK(hipStreamEndCapture(st, &gL)); CK(hipGraphInstantiate(&eL, gL, nullptr, nullptr, 0));
            // (H graphs are re-captured per replay below: the generation is a kernel argument)
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            // ---- equality of the two variants: one chain each from the same x
            std::vector<bf16_t> xl((size_t)G * ROWS * D), xh((size_t)G * ROWS * D);
            CK(hipMemcpyAsync(x, x0, xl.size() * 2, hipMemcpyDeviceToDevice, st));
            chain_L(0); CK(hipStreamSynchronize(st));
            CK(hipMemcpy(xl.data(), x, xl.size() * 2, hipMemcpyDeviceToHost));
            CK(hipMemcpyAsync(x, x0, xl.size() * 2, hipMemcpyDeviceToDevice, st));
            chain_H(0); CK(hipStreamSynchronize(st));
            CK(hipMemcpy(xh.data(), x, xh.size() * 2, hipMemcpyDeviceToHost));
            int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            size_t diff = 0; double asum = 0;
            for (size_t i = 0; i < xl.size(); ++i) { diff += xl[i] != xh[i]; uint32_t u = (uint32_t)xl[i] << 16; float f; memcpy(&f, &u, 4); asum += f < 0 ? -f : f; }
            // ---- timing
            auto time_L = [&]() {
                const int reps = (iters + per_graph - 1) / per_graph;
                CK(hipGraphLaunch(eL, st)); CK(hipStreamSynchronize(st));
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(eL, st));
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                return (double)ms * 1e3 / (reps * per_graph) / S;
            };
            auto time_H = [&]() {
                const int reps = (iters + per_graph - 1) / per_graph;
                for (int i = 0; i < per_graph; ++i) chain_H(i % nsets);
                CK(hipStreamSynchronize(st));
                CK(hipEventRecord(e0, st));
                for (int r = 0; r < reps; ++r) for (int i = 0; i < per_graph; ++i) chain_H(i % nsets);      // eager: ONE launch per 12 stages, the host is far ahead
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                return (double)ms * 1e3 / (reps * per_graph) / S;
            };
            double tl = 1e9, th = 1e9;
            for (int k = 0; k < 3; ++k) { tl = std::min(tl, time_L()); th = std::min(th, time_H()); }
            CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            // ---- the seam, from the stamps of one more chain of each variant (cold set)
            std::vector<u64> sl(nst), sh(nst);
            CK(hipMemset(stamps, 0, nst * 8)); chain_L(nsets - 1); CK(hipStreamSynchronize(st)); CK(hipMemcpy(sl.data(), stamps, nst * 8, hipMemcpyDeviceToHost));
            CK(hipMemset(stamps, 0, nst * 8)); chain_H(nsets - 1); CK(hipStreamSynchronize(st)); CK(hipMemcpy(sh.data(), stamps, nst * 8, hipMemcpyDeviceToHost));
            auto at = [&](const std::vector<u64>& v, int s, int u, int k) { return (double)v[((size_t)s * (G * CU_) + u) * NSTAMP + k] / 100.0; };   // us
            // per consumer unit of stages >= 1: (its operands usable) - (latest "compute done / stores begin" of the producers it depends on)
            auto seams = [&](const std::vector<u64>& v, bool H, std::vector<double>& seam, std::vector<double>& wait, std::vector<double>& fill, std::vector<double>& stage_len) {
                for (int s = 1; s < S; ++s) {
                    const bool cons_is_C = s % 2 == 1;
                    const int nu = G * (cons_is_C ? CU_ : PU);
                    double first_in = 1e300, last_out = 0;
                    for (int u = 0; u < nu; ++u) {
                        const int g = u / (cons_is_C ? CU_ : PU), lu = u % (cons_is_C ? CU_ : PU);
                        double prod_ready = 0;                     // producers' stamp 3 (operands usable = just before compute + stores)... use 4 = stores issued/drained
                        if (cons_is_C) { const int rt = lu / NCT; for (int b = (rt * RT) / RB; b <= (rt * RT + RT - 1) / RB; ++b) for (int h = 0; h < NH; ++h) prod_ready = std::max(prod_ready, at(v, s - 1, g * PU + b * NH + h, 3)); }
                        else { const int rb = lu / NH; for (int t = (rb * RB) / RT; t <= (rb * RB + RB - 1) / RT; ++t) for (int c = 0; c < NCT; ++c) prod_ready = std::max(prod_ready, at(v, s - 1, g * CU_ + t * NCT + c, 3)); }
                        seam.push_back(at(v, s, u, 3) - prod_ready);
                        if (H) { wait.push_back(at(v, s, u, 1) - prod_ready); fill.push_back(at(v, s, u, 3) - at(v, s, u, 1)); }
                        else fill.push_back(at(v, s, u, 3) - at(v, s, u, 0));
                        first_in = std::min(first_in, at(v, s, u, 3)); last_out = std::max(last_out, at(v, s, u, 4));
                    }
                    stage_len.push_back(last_out - first_in);
                }
            };
            std::vector<double> seamL, seamH, waitL, waitH, fillL, fillH, lenL, lenH;
            seams(sl, false, seamL, waitL, fillL, lenL); seams(sh, true, seamH, waitH, fillH, lenH);
            const Stats a = stats_of(seamL), b = stats_of(seamH), fl = stats_of(fillL), fh = stats_of(fillH), wh = stats_of(waitH);
            printf("G=%d members (%3d P + %3d C units per stage pair), weights %s (%d sets, %.0f MiB): results %s (%zu of %zu values differ, mean |x| %.3f%s)\n",
                   G, G * PU, G * CU_, cold ? "COLD" : "warm", nsets, 2.0 * set_elems * 2 * nsets / 1048576.0, diff == 0 && !herr ? "EQUAL" : "DIFFER", diff, xl.size(), asum / xl.size(), herr ? ", POLL TIMEOUT" : "");
            printf("   L  one launch per stage (hipGraph)        : %6.2f us per stage | producer computes -> consumer operands usable: median %5.2f  p90 %5.2f  max %5.2f us | consumer entry -> operands usable (weights + rows, cold prologue): median %5.2f us\n",
                   tl, a.med, a.p90, a.mx, fl.med);
            printf("   H  one launch per chain, %3d P + %3d C WGs : %6.2f us per stage | producer computes -> consumer operands usable: median %5.2f  p90 %5.2f  max %5.2f us | of it: until the flags are seen %5.2f, rows flags -> LDS %5.2f us (weights were prefetched)\n",
                   base.nP, base.nC, th, b.med, b.p90, b.mx, wh.med, fh.med);
            printf("   H / L = %.3f\n", th / tl);
            fflush(stdout);
            CK(hipGraphExecDestroy(eL)); CK(hipGraphDestroy(gL));
            (void)gH; (void)eH;
            CK(hipFree(Wp)); CK(hipFree(Wc)); CK(hipFree(x)); CK(hipFree(o)); CK(hipFree(x0)); CK(hipFree(flagP)); CK(hipFree(flagC)); CK(hipFree(stamps)); CK(hipFree(err));
        }
    }
    return 0;
}
