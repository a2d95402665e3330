// fh_bench.hip — stand-alone timing of the fused first launch of a sublayer group (mtn_amd/csrc/fused.hip), with an
// in-kernel timeline (FH_TIMELINE: ten 100 MHz wall-clock stamps per workgroup).  Weights rotate through a pool larger than
// the Infinity Cache so that every launch streams them from HBM, as in the train step.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DFH_TIMELINE tools/fh_bench.hip -o tools/fh_bench.bin && tools/fh_bench.bin [B]
#include <stdarg.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

#include "../mtn_amd/csrc/fused.hip"
void mtn_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <typename T> static T* dalloc(size_t n, int fill = 0) {
    T* p; CK(hipMalloc(&p, n * sizeof(T)));
    CK(hipMemset(p, fill, n * sizeof(T)));
    return p;
}

struct Case { const char* name; int n_mha, n_ffn; int self[3], ready[3], m[3]; };

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32, a = 20, d = 512, ff = 2048;
    const int NSET = argc > 2 ? atoi(argv[2]) : 96;        // weight sets: 96 x (1.5 + 2) MiB x 3 members > 256 MiB (1 = warm weights)
    const Case cases[] = {
        {"g0 self x3", 3, 0, {1, 1, 1}, {0, 0, 0}, {20, 20, 20}},
        {"g1 cross-ready x3 (128,32,32)", 3, 0, {0, 0, 0}, {1, 1, 1}, {128, 32, 32}},
        {"g2 cross-ready(40) + 2 ffn", 1, 2, {0, 0, 0}, {1, 0, 0}, {40, 0, 0}},
        {"g3 cross-ready(20)", 1, 0, {0, 0, 0}, {1, 0, 0}, {20, 0, 0}},
        {"g4 raw(20)", 1, 0, {0, 0, 0}, {0, 0, 0}, {20, 0, 0}},
        {"g6 ffn", 0, 1, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}},
    };
    // shared buffers
    std::vector<bf16_t*> wq(NSET * 3), w1(NSET * 3);
    for (auto& p : wq) p = dalloc<bf16_t>((size_t)3 * d * d, 0x11);
    for (auto& p : w1) p = dalloc<bf16_t>((size_t)ff * d, 0x11);
    float* bias = dalloc<float>(3 * ff);
    float* lna = dalloc<float>(d); float* lnb = dalloc<float>(d);
    unsigned long long* dbg = dalloc<unsigned long long>(2048 * 16);
    uint8_t* mask = dalloc<uint8_t>((size_t)B * 256 * 256, 1);
    hipStream_t st; CK(hipStreamCreate(&st));
    for (const Case& c : cases) {
        mtn_mha_args mha[3]; mtn_ffn_args ffn[3];
        memset(mha, 0, sizeof(mha)); memset(ffn, 0, sizeof(ffn));
        for (int i = 0; i < c.n_mha; ++i) {
            mtn_mha_args& A = mha[i];
            A.B = B; A.a = a; A.m = c.m[i]; A.d = d; A.h = 8; A.self_attn = c.self[i]; A.kv_ready = c.ready[i]; A.ln_eps = 1e-6f;
            A.x = dalloc<float>((size_t)B * a * d, 0x3c);
            A.mem = dalloc<bf16_t>((size_t)B * A.m * d, 0x3c);
            A.mask = mask; A.mask_sb = c.self[i] && i == 0 ? a * a : A.m; A.mask_sq = c.self[i] && i == 0 ? a : 0;
            if (c.self[i]) { A.mask_sb = (i == 0) ? a * a : a; A.m = a; }
            A.ln_a = lna; A.ln_b = lnb; A.b_qkv = bias; A.b_o = bias;
            A.xn = dalloc<bf16_t>((size_t)B * a * d); A.mean = dalloc<float>(B * a); A.rstd = dalloc<float>(B * a);
            A.qkv = dalloc<bf16_t>((size_t)B * a * 3 * d); A.kv = dalloc<bf16_t>((size_t)B * 256 * 2 * d, 0x3c);
            A.o = dalloc<bf16_t>((size_t)B * a * d); A.lse = dalloc<float>((size_t)2 * B * 8 * a);
        }
        for (int i = 0; i < c.n_ffn; ++i) {
            mtn_ffn_args& F = ffn[i];
            F.rows = B * a; F.d = d; F.d_ff = ff; F.ln_eps = 1e-6f;
            F.x = dalloc<float>((size_t)B * a * d, 0x3c); F.ln_a = lna; F.ln_b = lnb; F.b1 = bias; F.b2 = bias;
            F.xn = dalloc<bf16_t>((size_t)B * a * d); F.mean = dalloc<float>(B * a); F.rstd = dalloc<float>(B * a);
            F.hid = dalloc<bf16_t>((size_t)B * a * ff);
        }
        FhLaunch P;
        auto set_w = [&](int it) {
            for (int i = 0; i < c.n_mha; ++i) mha[i].w_qkv = wq[(it % NSET) * 3 + i];
            for (int i = 0; i < c.n_ffn; ++i) ffn[i].w1 = w1[(it % NSET) * 3 + i];
        };
        set_w(0);
        if (!fh_plan(c.n_mha, mha, c.n_ffn, ffn, P)) { printf("%s: not eligible\n", c.name); continue; }
        const int iters = NSET > 8 ? 2 * NSET : 64;
        const int skip = iters / 2;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e9f, tot = 0.f;
        for (int it = 0; it < iters; ++it) {
            set_w(it);
            fh_plan(c.n_mha, mha, c.n_ffn, ffn, P);
            P.G.dbg = dbg;
            CK(hipEventRecord(e0, st));
            const int rc = P.np == 4 ? fh_launch<4>(P.G, P.wgs, P.lds, st) : (P.np == 3 ? fh_launch<3>(P.G, P.wgs, P.lds, st) : fh_launch<1>(P.G, P.wgs, P.lds, st));
            CK(hipGetLastError());
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (it >= skip) { tot += ms; best = std::min(best, ms); }
            (void)rc;
        }
        std::vector<unsigned long long> h((size_t)P.wgs * 16);
        CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull;
        for (int w = 0; w < P.wgs; ++w) t0 = std::min(t0, h[(size_t)w * 16]);
        const int nst = 10;
        printf("%-34s wgs %4d np %d lds %6zu B blk/mt", c.name, P.wgs, P.np, P.lds);
        for (int i = 0; i < P.G.count; ++i) printf(" %d/%d(%dx%d)", P.G.m[i].blk, P.G.m[i].mt, P.G.m[i].hg, P.G.m[i].sg);
        printf(": avg %.2f us  best %.2f us (event pair, incl. launch)\n", tot / (iters - skip) * 1e3f, best * 1e3f);
        // median / max over workgroups of each stamp, relative to the first workgroup's start (us)
        printf("    stamp:   start  issued  landed   bar1   LNend   bar2   projd   epil    bar3    end\n");
        for (int pass = 0; pass < 3; ++pass) {
            printf("    %-6s", pass == 0 ? "min" : (pass == 1 ? "median" : "max"));
            for (int k = 0; k < nst; ++k) {
                std::vector<double> v;
                for (int w = 0; w < P.wgs; ++w) { const unsigned long long s = h[(size_t)w * 16 + k]; if (s >= t0 && s != 0) v.push_back((s - t0) * 0.01); }
                std::sort(v.begin(), v.end());
                if (v.empty()) { printf("      - "); continue; }
                printf(" %7.2f", pass == 0 ? v.front() : (pass == 1 ? v[v.size() / 2] : v.back()));
            }
            printf("\n");
        }
        {   // entry -> member found -> loads issued -> dropout key (medians, us)
            double v12 = 0, v13 = 0, v1 = 0; int nn = 0;
            for (int w = 0; w < P.wgs; ++w) {
                const unsigned long long* q = &h[(size_t)w * 16];
                if (q[12] >= q[0] && q[13] >= q[12]) { v12 += (q[12] - q[0]) * 0.01; v13 += (q[13] - q[0]) * 0.01; v1 += (q[1] - q[0]) * 0.01; ++nn; }
            }
            if (nn) printf("    mean since workgroup entry: member found %.2f us, loads issued %.2f us, dropout key %.2f us\n", v12 / nn, v13 / nn, v1 / nn);
            double v14 = 0, v15 = 0; int n14 = 0, n15 = 0;
            for (int w = 0; w < P.wgs; ++w) {
                const unsigned long long* q = &h[(size_t)w * 16];
                if (q[14] >= q[0] && q[14] > 0) { v14 += (q[14] - q[0]) * 0.01; ++n14; }
                if (q[15] >= q[0] && q[15] > 0) { v15 += (q[15] - q[0]) * 0.01; ++n15; }
            }
            if (n14) printf("    mean since workgroup entry: x rows issued %.2f us%s\n", v14 / n14, n15 ? "" : "");
            if (n15) printf("    (experiment: wait for the x rows before asking for anything else) x rows landed %.2f us\n", v15 / n15);
        }
        {   // shader clock: cycles between the first and the last stamp of a workgroup that ran to the end / wall time between them
            double mhz = 0; int nn = 0;
            for (int w = 0; w < P.wgs; ++w) {
                const unsigned long long* q = &h[(size_t)w * 16];
                if (q[9] > q[0] && q[11] > q[10]) { mhz += (double)(q[11] - q[10]) / ((q[9] - q[0]) * 0.01); ++nn; }
            }
            if (nn) printf("    shader clock while the kernel ran: %.0f MHz\n", mhz / nn);
        }
        CK(hipMemset(dbg, 0, 2048 * 16 * 8));
    }
    return 0;
}
