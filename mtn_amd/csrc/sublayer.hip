// sublayer.hip — the two fused residual sublayers of MTN's DecoderLayer, forward and backward:
//   y = x + dropout(MHA(LayerNorm(x), mem, mem, mask))    SublayerConnection ∘ MultiHeadedAttention
//                                                          (mtn.py:125-127, 248-267, 221-231)
//   y = x + dropout(W2 dropout(relu(W1 LayerNorm(x))))     SublayerConnection ∘ PositionwiseFeedForward
//                                                          (mtn.py:125-127, 279-280)
// executed as LOCKSTEP GROUPS: sublayers that do not depend on each other (x's text attention and the two
// auto-encoder chains of a DecoderLayer, mtn.py:183-215) share every launch —
//   forward : grouped LayerNorm -> grouped GEMM (QKV | Q+KV | FFN-1) -> grouped attention -> grouped GEMM
//             (output projection | FFN-2, + bias + dropout + residual)
//   backward: grouped dropout-backward cast -> grouped GEMM (dO | dh) -> grouped attention backward ->
//             grouped GEMM (dLN-out, + dmem) -> grouped LayerNorm backward (+ residual gradient)
// so a group costs 4 (5) launches whatever its size, and every launch carries 2-3x the workgroups of a single
// sublayer (these kernels are latency-bound at M = B*L = 640 rows).  Nothing synchronises, nothing allocates; parameter
// gradients are written straight into the caller's buffers, immediately or deferred (mtn_*_param_grad_work).
#include "common.h"

// csrc/fused.hip
int fh_group_eligible(int dtype, int n_mha, const mtn_mha_args* mha, int n_ffn, const mtn_ffn_args* ffn);
int fh_group_fwd_stage1(int n_mha, const mtn_mha_args* mha, int n_ffn, const mtn_ffn_args* ffn, void* stream);
int fh_is_enabled();
// csrc/fused_bwd.hip
struct FbIo { const void* dyl; void *dq, *dk, *dv; int ldq, ldkv; const float* lnf; float* ln_part; };
int fb_group_eligible(int dtype, int n_mha, const mtn_mha_args* mha, const FbIo* io);
int fb_group_bwd_stage(int n_mha, const mtn_mha_args* mha, const FbIo* io, void* stream);

static constexpr int LN_PART_HEADS = 8;     // heads of the fused head backward (d = 512, d_k = 64): pairs per row of its LayerNorm row-sum partials

static inline const char* lp_off(const void* p, long elems, int dtype) {
    return (const char*)p + elems * (dtype == MTN_BF16 ? 2 : 4);
}
static inline char* lp_off(void* p, long elems, int dtype) { return (char*)p + elems * (dtype == MTN_BF16 ? 2 : 4); }

static mtn_gemm_problem gemm_init(const void* A, int lda, const void* B, int ldb, int M, int N, int K, int at, int bt) {
    mtn_gemm_problem p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.M = M; p.N = N; p.K = K; p.a_trans = at; p.b_trans = bt;
    p.gate_scale = 1.f;
    return p;
}

// how often a group took the fused launches (tests assert that the fused kernels really ran): {fwd fused, fwd per-stage, bwd fused, bwd per-stage}
static long g_fused_counts[4] = {0, 0, 0, 0};
static long g_ln_epi_groups = 0;               // backward groups whose LayerNorm backward rode in the stage-4 GEMM's epilogue
extern "C" long mtn_ln_epilogue_groups(void) { return g_ln_epi_groups; }
extern "C" int mtn_fused_counters(long* out4) {
    MTN_CHECK_ARG(out4, "null output");
    for (int i = 0; i < 4; ++i) out4[i] = g_fused_counts[i];
    return MTN_OK;
}

#define RUN(expr)                        \
    do {                                 \
        int rc__ = (expr);               \
        if (rc__ != MTN_OK) return rc__; \
    } while (0)

// Problems of one stage may mix operand layouts (members with and without transposed weight copies): one launch per layout.
static int run_gemms(int dtype, int n, const mtn_gemm_problem* p, void* stream) {
    mtn_gemm_problem buf[MTN_GEMM_MAX_GROUP];
    for (int bt = 0; bt < 2; ++bt) {
        int k = 0;
        for (int i = 0; i < n; ++i)
            if (p[i].b_trans == bt) buf[k++] = p[i];
        if (k) RUN(mtn_gemm(dtype, k, buf, stream));
    }
    return MTN_OK;
}

static int check_mha(const mtn_mha_args* a, bool bwd) {
    MTN_CHECK_ARG(a, "null args");
    MTN_CHECK_ARG(a->B > 0 && a->a > 0 && a->d > 0 && a->h > 0 && a->d % a->h == 0, "bad shape");
    MTN_CHECK_ARG(a->d % 8 == 0 && (a->d / a->h) % 4 == 0, "d_model must be a multiple of 8 and d_k of 4");
    MTN_CHECK_ARG(a->self_attn || (a->m > 0 && a->mem && a->kv), "cross attention needs mem/kv");
    MTN_CHECK_ARG(a->x && a->ln_a && a->ln_b && a->w_qkv && a->b_qkv && a->w_o && a->b_o, "null parameter");
    MTN_CHECK_ARG(a->xn && a->mean && a->rstd && a->qkv && a->o && a->lse, "null saved buffer");
    if (!bwd) MTN_CHECK_ARG(a->y, "null output");
    else {
        MTN_CHECK_ARG(a->dy && a->dx && a->d_ln_a && a->d_ln_b && a->d_w_qkv && a->d_b_qkv && a->d_w_o && a->d_b_o, "null gradient buffer");
        MTN_CHECK_ARG(a->ws_lp && a->ws_f32, "null workspace");
    }
    return MTN_OK;
}

static int check_ffn(const mtn_ffn_args* a, bool bwd) {
    MTN_CHECK_ARG(a, "null args");
    MTN_CHECK_ARG(a->rows > 0 && a->d > 0 && a->d_ff > 0 && a->d % 8 == 0 && a->d_ff % 8 == 0, "bad shape");
    MTN_CHECK_ARG(a->x && a->ln_a && a->ln_b && a->w1 && a->b1 && a->w2 && a->b2, "null parameter");
    MTN_CHECK_ARG(a->xn && a->mean && a->rstd && a->hid, "null saved buffer");
    if (!bwd) MTN_CHECK_ARG(a->y, "null output");
    else {
        MTN_CHECK_ARG(a->dy && a->dx && a->d_ln_a && a->d_ln_b && a->d_w1 && a->d_b1 && a->d_w2 && a->d_b2, "null gradient buffer");
        MTN_CHECK_ARG(a->ws_lp && a->ws_f32, "null workspace");
    }
    return MTN_OK;
}

void attn_args_of(const mtn_mha_args* a, int dtype, mtn_attn_args* t) {
    const int d = a->d, m = a->self_attn ? a->a : a->m;
    memset(t, 0, sizeof(*t));
    t->B = a->B; t->h = a->h; t->a = a->a; t->m = m; t->dk = d / a->h;
    if (a->self_attn) {
        t->q = a->qkv; t->k = lp_off(a->qkv, d, dtype); t->v = lp_off(a->qkv, 2 * d, dtype); t->ldq = t->ldkv = 3 * d;
    } else {
        t->q = a->qkv; t->ldq = d; t->k = a->kv; t->v = lp_off(a->kv, d, dtype); t->ldkv = 2 * d;
    }
    t->mask = a->mask; t->mask_sb = a->mask_sb; t->mask_sq = a->mask_sq; t->drop = a->drop_attn;
    t->o = a->o; t->ldo = d; t->lse = a->lse;
}

// ------------------------------------------------------------------------------------------ forward
extern "C" int mtn_sublayer_group_fwd(int dtype, int n_mha, const mtn_mha_args* mha, int n_ffn, const mtn_ffn_args* ffn, void* stream) {
    MTN_CHECK_ARG(dtype == MTN_F32 || dtype == MTN_BF16, "bad dtype");
    MTN_CHECK_ARG(n_mha >= 0 && n_mha <= MTN_SUBLAYER_MAX_GROUP && n_ffn >= 0 && n_ffn <= MTN_SUBLAYER_MAX_GROUP && n_mha + n_ffn > 0, "bad group size");
    for (int i = 0; i < n_mha; ++i) RUN(check_mha(&mha[i], false));
    for (int i = 0; i < n_ffn; ++i) RUN(check_ffn(&ffn[i], false));
    // Fused first launch (csrc/fused.hip): stages 1-3 in one kernel per (sample block, head | w_1 column slice)
    bool fused = fh_group_eligible(dtype, n_mha, mha, n_ffn, ffn) != 0;
    mtn_mha_args pre[MTN_SUBLAYER_MAX_GROUP];
    if (!fused && n_mha > 0 && dtype == MTN_BF16 && fh_is_enabled()) {
        // A member that attends an un-projected memory (x attends an auto-encoder output, mtn.py:215) keeps that memory's rows in LDS
        // beside its own: with long query sequences (AVSD targets of 40-54 tokens) that does not fit.  Project K | V of those
        // memories with one grouped GEMM first and hand them to the fused kernel as memories projected ahead (kv_ready): three
        // launches for the group instead of five.
        bool any = false;
        for (int i = 0; i < n_mha; ++i) {
            pre[i] = mha[i];
            if (!mha[i].self_attn && !mha[i].kv_ready) { pre[i].kv_ready = 1; any = true; }
        }
        if (any && fh_group_eligible(dtype, n_mha, pre, n_ffn, ffn) != 0) {
            mtn_gemm_problem p[MTN_SUBLAYER_MAX_GROUP];
            int n = 0;
            for (int i = 0; i < n_mha; ++i) {
                const mtn_mha_args* a = &mha[i];
                if (a->self_attn || a->kv_ready) continue;
                const int d = a->d;
                p[n] = gemm_init(a->mem, d, lp_off(a->w_qkv, (long)d * d, dtype), d, a->B * a->m, 2 * d, d, 0, 0);
                p[n].bias = a->b_qkv + d; p[n].out_lp = a->kv; p[n].ldc = 2 * d; ++n;
            }
            RUN(mtn_gemm(dtype, n, p, stream));
            mha = pre;
            fused = true;
        }
    }
    ++g_fused_counts[fused ? 0 : 1];
    if (fused) RUN(fh_group_fwd_stage1(n_mha, mha, n_ffn, ffn, stream));
    // 1. LayerNorm(x) -> xn (compute dtype); row statistics saved for backward
    if (!fused) {
        mtn_ln_fwd_desc L[2 * MTN_SUBLAYER_MAX_GROUP];
        int n = 0;
        for (int i = 0; i < n_mha; ++i) {
            const mtn_mha_args* a = &mha[i];
            memset(&L[n], 0, sizeof(L[n]));
            L[n].rows = a->B * a->a; L[n].d = a->d; L[n].eps = a->ln_eps; L[n].x = a->x; L[n].a2 = a->ln_a; L[n].b2 = a->ln_b;
            L[n].y_lp = a->xn; L[n].mean = a->mean; L[n].rstd = a->rstd; ++n;
        }
        for (int i = 0; i < n_ffn; ++i) {
            const mtn_ffn_args* a = &ffn[i];
            memset(&L[n], 0, sizeof(L[n]));
            L[n].rows = a->rows; L[n].d = a->d; L[n].eps = a->ln_eps; L[n].x = a->x; L[n].a2 = a->ln_a; L[n].b2 = a->ln_b;
            L[n].y_lp = a->xn; L[n].mean = a->mean; L[n].rstd = a->rstd; ++n;
        }
        RUN(mtn_layernorm_fwd_group(dtype, n, L, stream));
    }
    // 2. input projections (mtn.py:256-258) and FFN first Linear + ReLU + dropout (mtn.py:280)
    if (!fused) {
        mtn_gemm_problem p[3 * MTN_SUBLAYER_MAX_GROUP];
        int n = 0;
        for (int i = 0; i < n_mha; ++i) {
            const mtn_mha_args* a = &mha[i];
            const int d = a->d, rows = a->B * a->a;
            if (a->self_attn) {
                p[n] = gemm_init(a->xn, d, a->w_qkv, d, rows, 3 * d, d, 0, 0);
                p[n].bias = a->b_qkv; p[n].out_lp = a->qkv; p[n].ldc = 3 * d; ++n;
            } else {
                p[n] = gemm_init(a->xn, d, a->w_qkv, d, rows, d, d, 0, 0);
                p[n].bias = a->b_qkv; p[n].out_lp = a->qkv; p[n].ldc = d; ++n;
                if (!a->kv_ready) {
                    p[n] = gemm_init(a->mem, d, lp_off(a->w_qkv, (long)d * d, dtype), d, a->B * a->m, 2 * d, d, 0, 0);
                    p[n].bias = a->b_qkv + d; p[n].out_lp = a->kv; p[n].ldc = 2 * d; ++n;
                }
            }
        }
        for (int i = 0; i < n_ffn; ++i) {
            const mtn_ffn_args* a = &ffn[i];
            p[n] = gemm_init(a->xn, a->d, a->w1, a->d, a->rows, a->d_ff, a->d, 0, 0);
            p[n].bias = a->b1; p[n].relu = 1; p[n].drop = a->drop_hidden; p[n].out_lp = a->hid; p[n].ldc = a->d_ff; ++n;
        }
        RUN(mtn_gemm(dtype, n, p, stream));
    }
    // 3. softmax(QK^T/sqrt(dk) masked) V per head (mtn.py:221-231)
    if (n_mha && !fused) {
        mtn_attn_args t[MTN_SUBLAYER_MAX_GROUP];
        for (int i = 0; i < n_mha; ++i) attn_args_of(&mha[i], dtype, &t[i]);
        RUN(mtn_attention_fwd_group(dtype, n_mha, t, stream));
    }
    // 4. output projection / FFN second Linear, + bias + dropout + residual (mtn.py:267, 280, 127)
    {
        mtn_gemm_problem p[2 * MTN_SUBLAYER_MAX_GROUP];
        int n = 0;
        for (int i = 0; i < n_mha; ++i) {
            const mtn_mha_args* a = &mha[i];
            const int d = a->d, rows = a->B * a->a;
            p[n] = gemm_init(a->o, d, a->w_o, d, rows, d, d, 0, 0);
            p[n].bias = a->b_o; p[n].drop = a->drop_out; p[n].residual = a->x; p[n].ldr = d; p[n].out_f32 = a->y; p[n].ldc = d;
            ++n;
        }
        for (int i = 0; i < n_ffn; ++i) {
            const mtn_ffn_args* a = &ffn[i];
            p[n] = gemm_init(a->hid, a->d_ff, a->w2, a->d_ff, a->rows, a->d, a->d_ff, 0, 0);
            p[n].bias = a->b2; p[n].drop = a->drop_out; p[n].residual = a->x; p[n].ldr = a->d; p[n].out_f32 = a->y; p[n].out_lp = a->y_lp; p[n].ldc = a->d;
            ++n;
        }
        RUN(mtn_gemm(dtype, n, p, stream));
    }
    return MTN_OK;
}

extern "C" int mtn_mha_sublayer_fwd(int dtype, const mtn_mha_args* a, void* stream) { return mtn_sublayer_group_fwd(dtype, 1, a, 0, nullptr, stream); }
extern "C" int mtn_ffn_sublayer_fwd(int dtype, const mtn_ffn_args* a, void* stream) { return mtn_sublayer_group_fwd(dtype, 0, nullptr, 1, a, stream); }

// ------------------------------------------------------------------------------------------ workspaces
extern "C" long mtn_mha_bwd_ws_lp_elems(int B, int a, int m, int d, int self_attn) {
    long rows = (long)B * a, rows_m = (long)B * (self_attn ? a : m);
    return 2 * rows * d + (self_attn ? 3 * rows * d : rows * d + 2 * rows_m * d);
}
extern "C" long mtn_mha_bwd_ws_f32_floats(int B, int a, int m, int d) {
    long rows = (long)B * a;
    // + the fp32 dK / dV sums of a multi-pass attention backward (more than 32 query rows on the per-stage path)
    const long kvacc = a > 32 ? 2L * B * (m > a ? m : a) * d : 0;
    // + the LayerNorm row-sum partials of the fused head backward (mtn_ln_epilogue): {s1, s2} per row and head
    return rows * d + mtn_layernorm_bwd_partial_floats((int)rows, d) + kvacc + rows * 2 * LN_PART_HEADS;
}
extern "C" long mtn_ffn_bwd_ws_f32_floats(int rows, int d, int d_ff) {
    // ... and of the dh GEMM: {s1, s2} per row and 64-column block of the hidden layer
    return (long)rows * d + mtn_layernorm_bwd_partial_floats(rows, d) + (long)rows * 2 * ((d_ff + 63) / 64);
}

// Workspace carving (same for the backward kernels and for the deferred parameter-gradient problems).
struct MhaWs { void *dyl, *dO, *dqkv, *dkv; float *dxn, *ln_partial, *kv_acc, *ln_part; };
static MhaWs mha_ws(const mtn_mha_args* a, int dtype) {
    const long rows = (long)a->B * a->a, d = a->d;
    MhaWs w;
    w.dyl = a->dyl_ready ? (void*)a->dyl_ready : a->ws_lp;   // [rows,d]  dropout-backward of dy (or the hand-off copy)
    w.dO = lp_off(a->ws_lp, rows * d, dtype);           // [rows,d]
    w.dqkv = lp_off(a->ws_lp, 2 * rows * d, dtype);     // self: [rows,3d]; cross: dq [rows,d] then dkv [rows_m,2d]
    w.dkv = lp_off(w.dqkv, rows * d, dtype);
    w.dxn = a->ws_f32;                                  // [rows,d]
    w.ln_partial = a->ws_f32 + rows * d;
    w.kv_acc = a->a > 32 ? w.ln_partial + mtn_layernorm_bwd_partial_floats((int)rows, (int)d) : nullptr;   // [2][rows_m, d] fp32
    const long rows_kv = (long)a->B * (a->m > a->a ? a->m : a->a);
    w.ln_part = w.ln_partial + mtn_layernorm_bwd_partial_floats((int)rows, (int)d) + (a->a > 32 ? 2L * rows_kv * d : 0);   // [rows][heads][2]
    return w;
}
struct FfnWs { void *dyl, *dh; float *dxn, *ln_partial, *ln_part; };
static FfnWs ffn_ws(const mtn_ffn_args* a, int dtype) {
    FfnWs w;
    w.dyl = a->dyl_ready ? (void*)a->dyl_ready : a->ws_lp;          // [rows,d]
    w.dh = lp_off(a->ws_lp, (long)a->rows * a->d, dtype);           // [rows,d_ff]
    w.dxn = a->ws_f32;
    w.ln_partial = a->ws_f32 + (long)a->rows * a->d;
    w.ln_part = w.ln_partial + mtn_layernorm_bwd_partial_floats(a->rows, a->d);      // [rows][d_ff / 64][2]
    return w;
}

// ------------------------------------------------------------------------------------------ backward
extern "C" int mtn_sublayer_group_bwd(int dtype, int n_mha, const mtn_mha_args* mha, int n_ffn, const mtn_ffn_args* ffn, void* stream) {
    MTN_CHECK_ARG(dtype == MTN_F32 || dtype == MTN_BF16, "bad dtype");
    MTN_CHECK_ARG(n_mha >= 0 && n_mha <= MTN_SUBLAYER_MAX_GROUP && n_ffn >= 0 && n_ffn <= MTN_SUBLAYER_MAX_GROUP && n_mha + n_ffn > 0, "bad group size");
    for (int i = 0; i < n_mha; ++i) RUN(check_mha(&mha[i], true));
    for (int i = 0; i < n_ffn; ++i) RUN(check_ffn(&ffn[i], true));
    // 1. gradient entering the dropped-out branch, in the compute dtype
    {
        mtn_cast_desc c[2 * MTN_SUBLAYER_MAX_GROUP];
        int n = 0;
        for (int i = 0; i < n_mha; ++i)
            if (!mha[i].dyl_ready) c[n++] = mtn_cast_desc{(long)mha[i].B * mha[i].a * mha[i].d, mha[i].dy, mha_ws(&mha[i], dtype).dyl, mha[i].drop_out};
        for (int i = 0; i < n_ffn; ++i)
            if (!ffn[i].dyl_ready) c[n++] = mtn_cast_desc{(long)ffn[i].rows * ffn[i].d, ffn[i].dy, ffn_ws(&ffn[i], dtype).dyl, ffn[i].drop_out};
        if (n) RUN(mtn_cast_group(dtype, n, c, stream));
    }
    // LayerNorm backward in the epilogue of the stage-4 GEMM (mtn_ln_epilogue): every member brings the fold vectors of the Linear
    // behind its LayerNorm, and the kernels that produce dq | dh can emit the row-sum partials (the fused head backward; the dh GEMM)
    // The decision is per MEMBER (a member alone and the same member inside a lockstep group take the same path: the two
    // schedules stay bitwise equal): a feed-forward member whenever its shapes allow, an attention member when the fused head backward
    // runs for the group's attention members.
    const bool epi_on = dtype == MTN_BF16 && !(MTN_ENV("MTN_LN_EPI") && MTN_ENV("MTN_LN_EPI")[0] == '0');
    bool epi_m[MTN_SUBLAYER_MAX_GROUP], epi_f[MTN_SUBLAYER_MAX_GROUP];
    for (int i = 0; i < n_mha; ++i) epi_m[i] = epi_on && mha[i].ln_fold && mha[i].d == 512 && mha[i].h == LN_PART_HEADS;
    for (int i = 0; i < n_ffn; ++i) epi_f[i] = epi_on && ffn[i].ln_fold && ffn[i].d_ff % 64 == 0 && ffn[i].d % 16 == 0 && ffn[i].d <= 512;
    // Fused stage 2 + 3 of the attention members (csrc/fused_bwd.hip): dO of a head and the head's attention backward in one
    // kernel per (sample block, head); dO never goes to HBM
    FbIo io[MTN_SUBLAYER_MAX_GROUP];
    for (int i = 0; i < n_mha; ++i) {
        const mtn_mha_args* a = &mha[i];
        const MhaWs w = mha_ws(a, dtype);
        const int d = a->d;
        io[i].dyl = w.dyl; io[i].dq = w.dqkv;
        io[i].lnf = a->ln_fold; io[i].ln_part = epi_m[i] ? w.ln_part : nullptr;
        if (a->self_attn) { io[i].dk = lp_off(w.dqkv, d, dtype); io[i].dv = lp_off(w.dqkv, 2 * d, dtype); io[i].ldq = io[i].ldkv = 3 * d; }
        else { io[i].dk = w.dkv; io[i].dv = lp_off(w.dkv, d, dtype); io[i].ldq = d; io[i].ldkv = 2 * d; }
    }
    const bool fb = n_mha > 0 && fb_group_eligible(dtype, n_mha, mha, io) != 0;
    if (n_mha > 0 && !fb)                         // the per-stage attention backward emits no partials: those members keep the LayerNorm launch
        for (int i = 0; i < n_mha; ++i) { io[i].ln_part = nullptr; epi_m[i] = false; }
    {
        bool any = false;
        for (int i = 0; i < n_mha; ++i) any = any || epi_m[i];
        for (int i = 0; i < n_ffn; ++i) any = any || epi_f[i];
        if (any) ++g_ln_epi_groups;
    }
    if (n_mha > 0) ++g_fused_counts[fb ? 2 : 3];
    if (fb) RUN(fb_group_bwd_stage(n_mha, mha, io, stream));
    // 2. dO = dyl Wo ;  dh = (dyl W2) * relu'(h) * hidden-dropout mask (both recovered from the saved hidden: hid > 0)
    mtn_ln_epilogue lne[2 * MTN_SUBLAYER_MAX_GROUP];
    if (!fb || n_ffn > 0) {
        mtn_gemm_problem p[2 * MTN_SUBLAYER_MAX_GROUP];
        int n = 0;
        // (one operand layout per launch: when a feed-forward member reads its weight as it lies (b_trans = 1: no w2_t), the attention
        //  members' dO problems do too, although their W_o^T exists for the fused head backward — a mixed stage would split in two launches)
        bool any_bt = false;
        for (int i = 0; i < n_ffn; ++i) any_bt = any_bt || !ffn[i].w2_t;
        for (int i = 0; i < n_mha && !fb; ++i) {
            const mtn_mha_args* a = &mha[i];
            const MhaWs w = mha_ws(a, dtype);
            const int d = a->d, rows = a->B * a->a;
            p[n] = (a->w_o_t && !any_bt) ? gemm_init(w.dyl, d, a->w_o_t, d, rows, d, d, 0, 0) : gemm_init(w.dyl, d, a->w_o, d, rows, d, d, 0, 1);
            p[n].out_lp = w.dO; p[n].ldc = d; ++n;
        }
        for (int i = 0; i < n_ffn; ++i) {
            const mtn_ffn_args* a = &ffn[i];
            const FfnWs w = ffn_ws(a, dtype);
            const int d = a->d, ff = a->d_ff;
            p[n] = a->w2_t ? gemm_init(w.dyl, d, a->w2_t, d, a->rows, ff, d, 0, 0) : gemm_init(w.dyl, d, a->w2, ff, a->rows, ff, d, 0, 1);
            p[n].gate = a->hid;
            p[n].gate_scale = (a->drop_hidden.p > 0.f && a->drop_hidden.seed) ? 1.0f / (1.0f - a->drop_hidden.p) : 1.0f;
            p[n].out_lp = w.dh; p[n].ldc = ff;
            if (epi_f[i]) {                       // dh . u and dh . (pre-activation - c) per row and 64-column block, on the way out
                mtn_ln_epilogue& e = lne[n];
                memset(&e, 0, sizeof(e));
                e.mode = MTN_LN_EMIT; e.fold = a->ln_fold; e.gate_inv_scale = 1.0f / p[n].gate_scale; e.part = w.ln_part;
                p[n].ln = &e;
            }
            ++n;
        }
        RUN(run_gemms(dtype, n, p, stream));
    }
    // 3. attention core backward -> dq, dk, dv
    if (n_mha && !fb) {
        mtn_attn_args t[MTN_SUBLAYER_MAX_GROUP];
        for (int i = 0; i < n_mha; ++i) {
            const mtn_mha_args* a = &mha[i];
            const MhaWs w = mha_ws(a, dtype);
            attn_args_of(a, dtype, &t[i]);
            t[i].d_o = w.dO;
            t[i].kv_acc = w.kv_acc;
            if (a->self_attn) { t[i].dq = w.dqkv; t[i].dk_out = lp_off(w.dqkv, a->d, dtype); t[i].dv_out = lp_off(w.dqkv, 2 * a->d, dtype); }
            else { t[i].dq = w.dqkv; t[i].dk_out = w.dkv; t[i].dv_out = lp_off(w.dkv, a->d, dtype); }
        }
        RUN(mtn_attention_bwd_group(dtype, n_mha, t, stream));
    }
    // 4. gradients w.r.t. the LayerNorm output (and the attention memory): w_qkv^T is [d,3d] — columns 0..d-1 are
    //    Wq^T, columns d..3d-1 are Wkv^T (row stride 3d)
    {
        mtn_gemm_problem p[3 * MTN_SUBLAYER_MAX_GROUP];
        mtn_ln_epilogue lc[2 * MTN_SUBLAYER_MAX_GROUP];
        int n = 0, nlc = 0;
        // LayerNorm backward of a member in the epilogue of its g = dq W problem (what stage 5 does otherwise)
        auto consume = [&](mtn_gemm_problem& q, const float* x, const float* a2, const float* mean, const float* rstd, float eps, const float* dy, float* dx,
                           void* next_dyl, const mtn_dropout& next_drop, float* part, int np, float* colpart) {
            mtn_ln_epilogue& e = lc[nlc++];
            memset(&e, 0, sizeof(e));
            e.mode = MTN_LN_CONSUME; e.part = part; e.np = np; e.x = x; e.a2 = a2; e.mean = mean; e.rstd = rstd; e.dres = dy; e.eps = eps;
            e.dx = dx; e.dx_lp = next_dyl; e.dx_lp_drop = next_drop; e.colpart = colpart;
            q.ln = &e; q.out_f32 = nullptr;
        };
        for (int i = 0; i < n_mha; ++i) {
            const mtn_mha_args* a = &mha[i];
            const MhaWs w = mha_ws(a, dtype);
            const int d = a->d, rows = a->B * a->a, rows_m = a->B * a->m;
            if (a->self_attn) {
                p[n] = (a->w_qkv_t && !epi_m[i]) ? gemm_init(w.dqkv, 3 * d, a->w_qkv_t, 3 * d, rows, d, 3 * d, 0, 0) : gemm_init(w.dqkv, 3 * d, a->w_qkv, d, rows, d, 3 * d, 0, 1);
                p[n].out_f32 = w.dxn; p[n].ldc = d;
                if (epi_m[i]) consume(p[n], a->x, a->ln_a, a->mean, a->rstd, a->ln_eps, a->dy, a->dx, a->next_dyl, a->next_drop, w.ln_part, LN_PART_HEADS, w.ln_partial);
                ++n;
            } else {
                p[n] = (a->w_qkv_t && !epi_m[i]) ? gemm_init(w.dqkv, d, a->w_qkv_t, 3 * d, rows, d, d, 0, 0) : gemm_init(w.dqkv, d, a->w_qkv, d, rows, d, d, 0, 1);
                p[n].out_f32 = w.dxn; p[n].ldc = d;
                if (epi_m[i]) consume(p[n], a->x, a->ln_a, a->mean, a->rstd, a->ln_eps, a->dy, a->dx, a->next_dyl, a->next_drop, w.ln_part, LN_PART_HEADS, w.ln_partial);
                ++n;
                if (a->dmem) {
                    p[n] = (a->w_qkv_t && !epi_m[i]) ? gemm_init(w.dkv, 2 * d, lp_off(a->w_qkv_t, d, dtype), 3 * d, rows_m, d, 2 * d, 0, 0)
                                                  : gemm_init(w.dkv, 2 * d, lp_off(a->w_qkv, (long)d * d, dtype), d, rows_m, d, 2 * d, 0, 1);
                    if (a->dmem_accumulate) { p[n].residual = a->dmem; p[n].ldr = d; }
                    if (a->dmem_lp) { p[n].out_lp = a->dmem_lp; p[n].drop = a->dmem_lp_drop; p[n].lp_drop_after_residual = 1; }
                    p[n].out_f32 = a->dmem; p[n].ldc = d; ++n;
                }
            }
        }
        for (int i = 0; i < n_ffn; ++i) {
            const mtn_ffn_args* a = &ffn[i];
            const FfnWs w = ffn_ws(a, dtype);
            const int d = a->d, ff = a->d_ff;
            p[n] = (a->w1_t && !epi_f[i]) ? gemm_init(w.dh, ff, a->w1_t, ff, a->rows, d, ff, 0, 0) : gemm_init(w.dh, ff, a->w1, d, a->rows, d, ff, 0, 1);
            p[n].out_f32 = w.dxn; p[n].ldc = d;
            if (epi_f[i]) consume(p[n], a->x, a->ln_a, a->mean, a->rstd, a->ln_eps, a->dy, a->dx, a->next_dyl, a->next_drop, w.ln_part, ff / 64, w.ln_partial);
            ++n;
        }
        RUN(run_gemms(dtype, n, p, stream));      // (one launch: the training step keeps no transposed copies of these weights, b_trans = 1 throughout)
    }
    // 5. LayerNorm backward fused with the residual-branch gradient (critical path: dx only)
    {
        mtn_ln_bwd_desc L[2 * MTN_SUBLAYER_MAX_GROUP];
        int n = 0;
        for (int i = 0; i < n_mha; ++i) {
            if (epi_m[i]) continue;               // done in the epilogue of the member's stage-4 problem
            const mtn_mha_args* a = &mha[i];
            const MhaWs w = mha_ws(a, dtype);
            L[n++] = mtn_ln_bwd_desc{a->B * a->a, a->d, a->ln_eps, a->x, a->ln_a, a->mean, a->rstd, w.dxn, a->dy, a->dx, w.ln_partial,
                                     a->next_dyl, dtype, a->next_drop};
        }
        for (int i = 0; i < n_ffn; ++i) {
            if (epi_f[i]) continue;
            const mtn_ffn_args* a = &ffn[i];
            const FfnWs w = ffn_ws(a, dtype);
            L[n++] = mtn_ln_bwd_desc{a->rows, a->d, a->ln_eps, a->x, a->ln_a, a->mean, a->rstd, w.dxn, a->dy, a->dx, w.ln_partial,
                                     a->next_dyl, dtype, a->next_drop};
        }
        if (n) RUN(mtn_layernorm_bwd_group(n, L, stream));
    }
    // 6. parameter gradients (off the critical path): now, unless the caller batches them across sublayers
    {
        mtn_gemm_problem p[5 * MTN_SUBLAYER_MAX_GROUP];
        mtn_ln_finalize_desc ln[2 * MTN_SUBLAYER_MAX_GROUP];
        int n = 0, nl = 0;
        for (int i = 0; i < n_mha; ++i)
            if (!mha[i].defer_param_grads) n += mtn_mha_param_grad_work(dtype, &mha[i], p + n, &ln[nl++]);
        for (int i = 0; i < n_ffn; ++i)
            if (!ffn[i].defer_param_grads) n += mtn_ffn_param_grad_work(dtype, &ffn[i], p + n, &ln[nl++]);
        for (int i = 0; i < n; i += MTN_GEMM_MAX_GROUP) RUN(mtn_gemm(dtype, n - i < MTN_GEMM_MAX_GROUP ? n - i : MTN_GEMM_MAX_GROUP, p + i, stream));
        if (nl) RUN(mtn_layernorm_bwd_finalize(nl, ln, stream));
    }
    return MTN_OK;
}

extern "C" int mtn_mha_sublayer_bwd(int dtype, const mtn_mha_args* a, void* stream) { return mtn_sublayer_group_bwd(dtype, 1, a, 0, nullptr, stream); }
extern "C" int mtn_ffn_sublayer_bwd(int dtype, const mtn_ffn_args* a, void* stream) { return mtn_sublayer_group_bwd(dtype, 0, nullptr, 1, a, stream); }

// dWo = dyl^T O, dWqkv = dqkv^T xn (+ dkv^T mem for cross attention); bias gradients ride as row sums of the A operand.
extern "C" int mtn_mha_param_grad_work(int dtype, const mtn_mha_args* a, mtn_gemm_problem* p, mtn_ln_finalize_desc* ln) {
    const int d = a->d, rows = a->B * a->a, m = a->self_attn ? a->a : a->m, rows_m = a->B * m;
    const MhaWs w = mha_ws(a, dtype);
    int n = 0;
    p[n] = gemm_init(w.dyl, d, a->o, d, d, d, rows, 1, 1);
    p[n].out_f32 = a->d_w_o; p[n].ldc = d; p[n].rowsum_out = a->d_b_o; ++n;
    if (a->self_attn) {
        p[n] = gemm_init(w.dqkv, 3 * d, a->xn, d, 3 * d, d, rows, 1, 1);
        p[n].out_f32 = a->d_w_qkv; p[n].ldc = d; p[n].rowsum_out = a->d_b_qkv; ++n;
    } else {
        p[n] = gemm_init(w.dqkv, d, a->xn, d, d, d, rows, 1, 1);
        p[n].out_f32 = a->d_w_qkv; p[n].ldc = d; p[n].rowsum_out = a->d_b_qkv; ++n;
        p[n] = gemm_init(w.dkv, 2 * d, a->mem, d, 2 * d, d, rows_m, 1, 1);
        p[n].out_f32 = a->d_w_qkv + (long)d * d; p[n].ldc = d; p[n].rowsum_out = a->d_b_qkv + d; ++n;
    }
    ln->partial = w.ln_partial;
    ln->nparts = mtn_layernorm_bwd_nparts(rows);
    ln->d = d; ln->da2 = a->d_ln_a; ln->db2 = a->d_ln_b;
    return n;
}

// dW2 = dyl^T hid, dW1 = dh^T xn; bias gradients as row sums of the A operand.
extern "C" int mtn_ffn_param_grad_work(int dtype, const mtn_ffn_args* a, mtn_gemm_problem* p, mtn_ln_finalize_desc* ln) {
    const int d = a->d, ff = a->d_ff, rows = a->rows;
    const FfnWs w = ffn_ws(a, dtype);
    p[0] = gemm_init(w.dyl, d, a->hid, ff, d, ff, rows, 1, 1);
    p[0].out_f32 = a->d_w2; p[0].ldc = ff; p[0].rowsum_out = a->d_b2;
    p[1] = gemm_init(w.dh, ff, a->xn, d, ff, d, rows, 1, 1);
    p[1].out_f32 = a->d_w1; p[1].ldc = d; p[1].rowsum_out = a->d_b1;
    ln->partial = w.ln_partial;
    ln->nparts = mtn_layernorm_bwd_nparts(rows);
    ln->d = d; ln->da2 = a->d_ln_a; ln->db2 = a->d_ln_b;
    return 2;
}
