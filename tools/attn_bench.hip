// Standalone micro-benchmark of the attention kernels (hipEvent timing).
#include "../mtn_amd/csrc/attention.hip"
#include "../mtn_amd/csrc/elementwise.hip"
#include <vector>
static void fillr(void* d, size_t bytes) {
    std::vector<unsigned short> h(bytes / 2);
    for (auto& x : h) x = (unsigned short)(0x3c00 + (rand() & 0x1ff) + ((rand() & 1) ? 0x8000 : 0));
    (void)hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
}
int main() {
    const int B = 32, h = 8, a = 20, dk = 64, d = h * dk;
    int ms[] = {20, 32, 40, 128};
    hipStream_t st; (void)hipStreamCreate(&st);
    const int NR = getenv("NR") ? atoi(getenv("NR")) : 24;
    for (int m : ms) {
        std::vector<void*> q(NR), kv(NR), o(NR), dO(NR), dq(NR), dkv(NR); std::vector<float*> lse(NR);
        for (int i = 0; i < NR; ++i) {
            (void)hipMalloc(&q[i], (size_t)B * a * d * 2); fillr(q[i], (size_t)B * a * d * 2);
            (void)hipMalloc(&kv[i], (size_t)B * m * 2 * d * 2); fillr(kv[i], (size_t)B * m * 2 * d * 2);
            (void)hipMalloc(&o[i], (size_t)B * a * d * 2); (void)hipMalloc(&dO[i], (size_t)B * a * d * 2); fillr(dO[i], (size_t)B * a * d * 2);
            (void)hipMalloc(&dq[i], (size_t)B * a * d * 2); (void)hipMalloc(&dkv[i], (size_t)B * m * 2 * d * 2);
            (void)hipMalloc(&lse[i], (size_t)2 * B * h * a * 4);
        }
        unsigned char* mask; (void)hipMalloc(&mask, (size_t)B * m); (void)hipMemset(mask, 1, (size_t)B * m);
        for (int valu = 0; valu < 2; ++valu) {
            if (valu) setenv("MTN_ATTN_VALU", "1", 1); else unsetenv("MTN_ATTN_VALU");
            for (int bwd = 0; bwd < 2; ++bwd) {
                hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
                const int iters = 200;
                auto run = [&](int i) {
                    mtn_attn_args A; memset(&A, 0, sizeof(A));
                    const int r = i % NR;
                    A.B = B; A.h = h; A.a = a; A.m = m; A.dk = dk; A.q = q[r]; A.k = kv[r]; A.v = (char*)kv[r] + d * 2; A.ldq = d; A.ldkv = 2 * d;
                    A.mask = mask; A.mask_sb = m; A.mask_sq = 0; A.o = o[r]; A.ldo = d; A.lse = lse[r];
                    A.d_o = dO[r]; A.dq = dq[r]; A.dk_out = dkv[r]; A.dv_out = (char*)dkv[r] + d * 2;
                    int rc = bwd ? mtn_attention_bwd(MTN_BF16, &A, st) : mtn_attention_fwd(MTN_BF16, &A, st);
                    if (rc) { printf("ERR %s\n", mtn_last_error()); exit(1); }
                };
                if (bwd) for (int i = 0; i < NR; ++i) { bool v = getenv("MTN_ATTN_VALU"); unsetenv("MTN_ATTN_VALU"); mtn_attn_args A; (void)A; if (v) setenv("MTN_ATTN_VALU", "1", 1); }
                for (int i = 0; i < NR; ++i) run(i);
                (void)hipEventRecord(e0, st);
                for (int i = 0; i < iters; ++i) run(i);
                (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
                float t; (void)hipEventElapsedTime(&t, e0, e1);
                printf("m=%3d %s %s: %6.2f us/launch\n", m, valu ? "valu" : "mfma", bwd ? "bwd" : "fwd", t * 1e3 / iters);
            }
        }
    }
    // ---- the step's his-attention group: x attends history (m = 128) + two auto-encoder video attentions (m = 32), dropout on
    {
        unsetenv("MTN_ATTN_VALU");
        const int msz[3] = {128, 32, 32};
        void *q[3], *kv[3], *o[3], *dO[3], *dq[3], *dkv[3]; float* lse[3]; unsigned char* mk[3];
        for (int i = 0; i < 3; ++i) {
            const int m = msz[i];
            (void)hipMalloc(&q[i], (size_t)B * a * d * 2); fillr(q[i], (size_t)B * a * d * 2);
            (void)hipMalloc(&kv[i], (size_t)B * m * 2 * d * 2); fillr(kv[i], (size_t)B * m * 2 * d * 2);
            (void)hipMalloc(&o[i], (size_t)B * a * d * 2); (void)hipMalloc(&dO[i], (size_t)B * a * d * 2); fillr(dO[i], (size_t)B * a * d * 2);
            (void)hipMalloc(&dq[i], (size_t)B * a * d * 2); (void)hipMalloc(&dkv[i], (size_t)B * m * 2 * d * 2);
            (void)hipMalloc(&lse[i], (size_t)2 * B * h * a * 4);
            (void)hipMalloc(&mk[i], (size_t)B * m); (void)hipMemset(mk[i], 1, (size_t)B * m);
        }
        uint64_t* seed; (void)hipMalloc(&seed, 8); (void)hipMemset(seed, 7, 8);
        for (int drop = 0; drop < 2; ++drop)
            for (int cnt = 1; cnt <= 3; cnt += 2)
                for (int bwd = 0; bwd < 2; ++bwd) {
                    mtn_attn_args A[3];
                    for (int i = 0; i < 3; ++i) {
                        memset(&A[i], 0, sizeof(A[i]));
                        const int m = msz[i];
                        A[i].B = B; A[i].h = h; A[i].a = a; A[i].m = m; A[i].dk = dk; A[i].q = q[i]; A[i].k = kv[i]; A[i].v = (char*)kv[i] + d * 2;
                        A[i].ldq = d; A[i].ldkv = 2 * d; A[i].mask = mk[i]; A[i].mask_sb = m; A[i].mask_sq = 0; A[i].o = o[i]; A[i].ldo = d; A[i].lse = lse[i];
                        A[i].d_o = dO[i]; A[i].dq = dq[i]; A[i].dk_out = dkv[i]; A[i].dv_out = (char*)dkv[i] + d * 2;
                        if (drop) { A[i].drop.p = 0.1f; A[i].drop.salt = 5 + i; A[i].drop.seed = seed; }
                    }
                    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
                    const int iters = 200;
                    for (int i = 0; i < 10; ++i) (void)(bwd ? mtn_attention_bwd_group(MTN_BF16, cnt, A, st) : mtn_attention_fwd_group(MTN_BF16, cnt, A, st));
                    (void)hipEventRecord(e0, st);
                    for (int i = 0; i < iters; ++i) if (bwd ? mtn_attention_bwd_group(MTN_BF16, cnt, A, st) : mtn_attention_fwd_group(MTN_BF16, cnt, A, st)) { printf("ERR %s\n", mtn_last_error()); return 1; }
                    (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
                    float t; (void)hipEventElapsedTime(&t, e0, e1);
                    printf("group of %d (m=128,32,32) dropout %d %s: %6.2f us/launch\n", cnt, drop, bwd ? "bwd" : "fwd", t * 1e3 / iters);
                }
    }
    return 0;
}
