// sublayer.hip — the two fused residual sublayers of MTN's DecoderLayer, forward and backward:
//   y = x + dropout(MHA(LayerNorm(x), mem, mem, mask))    SublayerConnection ∘ MultiHeadedAttention
//                                                          (mtn.py:125-127, 248-267, 221-231)
//   y = x + dropout(W2 dropout(relu(W1 LayerNorm(x))))     SublayerConnection ∘ PositionwiseFeedForward
//                                                          (mtn.py:125-127, 279-280)
// Each entry point enqueues a short fixed chain of kernels on the caller's stream (no syncs, no
// allocation): LayerNorm -> grouped MFMA GEMM(s) -> attention core -> GEMM with bias/dropout/residual
// epilogue.  Backward mirrors it and writes parameter gradients straight into the caller's buffers.
#include "common.h"

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

#define RUN(expr)                     \
    do {                              \
        int rc__ = (expr);            \
        if (rc__ != MTN_OK) return rc__; \
    } while (0)

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

extern "C" int mtn_mha_sublayer_fwd(int dtype, const mtn_mha_args* a, void* stream) {
    MTN_CHECK_ARG(dtype == MTN_F32 || dtype == MTN_BF16, "bad dtype");
    RUN(check_mha(a, false));
    const int d = a->d, rows = a->B * a->a, m = a->self_attn ? a->a : a->m, rows_m = a->B * m;
    // 1. LayerNorm(x) -> xn (lowp), row statistics saved for backward
    RUN(mtn_layernorm_fwd(dtype, rows, d, a->ln_eps, a->x, a->ln_a, a->ln_b, nullptr, a->xn, a->mean, a->rstd, stream));
    // 2. input projections (mtn.py:256-258): one launch
    if (a->self_attn) {
        mtn_gemm_problem p = gemm_init(a->xn, d, a->w_qkv, d, rows, 3 * d, d, 0, 0);
        p.bias = a->b_qkv; p.out_lp = a->qkv; p.ldc = 3 * d;
        RUN(mtn_gemm(dtype, 1, &p, stream));
    } else {
        mtn_gemm_problem p[2];
        p[0] = gemm_init(a->xn, d, a->w_qkv, d, rows, d, d, 0, 0);
        p[0].bias = a->b_qkv; p[0].out_lp = a->qkv; p[0].ldc = d;
        p[1] = gemm_init(a->mem, d, lp_off(a->w_qkv, (long)d * d, dtype), d, rows_m, 2 * d, d, 0, 0);
        p[1].bias = a->b_qkv + d; p[1].out_lp = a->kv; p[1].ldc = 2 * d;
        RUN(mtn_gemm(dtype, 2, p, stream));
    }
    // 3. softmax(QK^T/sqrt(dk) masked) V per head (mtn.py:221-231)
    mtn_attn_args t;
    memset(&t, 0, sizeof(t));
    t.B = a->B; t.h = a->h; t.a = a->a; t.m = m; t.dk = d / a->h;
    if (a->self_attn) {
        t.q = a->qkv; t.k = lp_off(a->qkv, d, dtype); t.v = lp_off(a->qkv, 2 * d, dtype); t.ldq = t.ldkv = 3 * d;
    } else {
        t.q = a->qkv; t.ldq = d; t.k = a->kv; t.v = lp_off(a->kv, d, dtype); t.ldkv = 2 * d;
    }
    t.mask = a->mask; t.mask_sb = a->mask_sb; t.mask_sq = a->mask_sq; t.drop = a->drop_attn;
    t.o = a->o; t.ldo = d; t.lse = a->lse;
    RUN(mtn_attention_fwd(dtype, &t, stream));
    // 4. output projection + dropout + residual (mtn.py:267, 127)
    mtn_gemm_problem po = gemm_init(a->o, d, a->w_o, d, rows, d, d, 0, 0);
    po.bias = a->b_o; po.drop = a->drop_out; po.residual = a->x; po.ldr = d; po.out_f32 = a->y; po.ldc = d;
    RUN(mtn_gemm(dtype, 1, &po, stream));
    return MTN_OK;
}

extern "C" long mtn_mha_bwd_ws_lp_elems(int B, int a, int m, int d, int self_attn) {
    long rows = (long)B * a, rows_m = (long)B * (self_attn ? a : m);
    return 2 * rows * d + (self_attn ? 3 * rows * d : rows * d + 2 * rows_m * d);
}
extern "C" long mtn_mha_bwd_ws_f32_floats(int B, int a, int m, int d) {
    (void)m;
    long rows = (long)B * a;
    return rows * d + mtn_layernorm_bwd_partial_floats((int)rows, d);
}

extern "C" int mtn_mha_sublayer_bwd(int dtype, const mtn_mha_args* a, void* stream) {
    MTN_CHECK_ARG(dtype == MTN_F32 || dtype == MTN_BF16, "bad dtype");
    RUN(check_mha(a, true));
    const int d = a->d, rows = a->B * a->a, m = a->self_attn ? a->a : a->m, rows_m = a->B * m;
    void* dyl = a->ws_lp;                                   // [rows,d]   dropout-backward of dy, lowp
    void* dO = lp_off(a->ws_lp, (long)rows * d, dtype);     // [rows,d]
    void* dqkv = lp_off(a->ws_lp, 2L * rows * d, dtype);    // self: [rows,3d]; cross: dq [rows,d] then dkv [rows_m,2d]
    void* dkv = lp_off(dqkv, (long)rows * d, dtype);
    float* dxn = a->ws_f32;                                 // [rows,d]
    float* ln_partial = a->ws_f32 + (long)rows * d;

    // 1. gradient entering the dropped-out branch
    RUN(mtn_dropout_bwd_to_lp(dtype, (long)rows * d, a->dy, a->drop_out, dyl, stream));
    // 2. dO = dyl @ Wo
    {
        mtn_gemm_problem p = gemm_init(dyl, d, a->w_o, d, rows, d, d, 0, 1);
        p.out_lp = dO; p.ldc = d;
        RUN(mtn_gemm(dtype, 1, &p, stream));
    }
    // 3. attention core backward -> dq, dk, dv
    mtn_attn_args t;
    memset(&t, 0, sizeof(t));
    t.B = a->B; t.h = a->h; t.a = a->a; t.m = m; t.dk = d / a->h;
    if (a->self_attn) {
        t.q = a->qkv; t.k = lp_off(a->qkv, d, dtype); t.v = lp_off(a->qkv, 2 * d, dtype); t.ldq = t.ldkv = 3 * d;
        t.dq = dqkv; t.dk_out = lp_off(dqkv, d, dtype); t.dv_out = lp_off(dqkv, 2 * d, dtype);
    } else {
        t.q = a->qkv; t.ldq = d; t.k = a->kv; t.v = lp_off(a->kv, d, dtype); t.ldkv = 2 * d;
        t.dq = dqkv; t.dk_out = dkv; t.dv_out = lp_off(dkv, d, dtype);
    }
    t.mask = a->mask; t.mask_sb = a->mask_sb; t.mask_sq = a->mask_sq; t.drop = a->drop_attn;
    t.o = a->o; t.ldo = d; t.lse = a->lse; t.d_o = dO;
    RUN(mtn_attention_bwd(dtype, &t, stream));
    // 4. gradients w.r.t. the projection inputs (one launch)
    if (a->self_attn) {
        mtn_gemm_problem p = gemm_init(dqkv, 3 * d, a->w_qkv, d, rows, d, 3 * d, 0, 1);
        p.out_f32 = dxn; p.ldc = d;
        RUN(mtn_gemm(dtype, 1, &p, stream));
    } else {
        mtn_gemm_problem p[2];
        p[0] = gemm_init(dqkv, d, a->w_qkv, d, rows, d, d, 0, 1);
        p[0].out_f32 = dxn; p[0].ldc = d;
        int n = 1;
        if (a->dmem) {
            p[1] = gemm_init(dkv, 2 * d, lp_off(a->w_qkv, (long)d * d, dtype), d, rows_m, d, 2 * d, 0, 1);
            if (a->dmem_accumulate) { p[1].residual = a->dmem; p[1].ldr = d; }
            p[1].out_f32 = a->dmem; p[1].ldc = d;
            n = 2;
        }
        RUN(mtn_gemm(dtype, n, p, stream));
    }
    // 5. parameter gradients (one launch): dWo = dyl^T O, dWqkv = dqkv^T xn (+ dkv^T mem); biases ride as row sums
    {
        mtn_gemm_problem p[3];
        int n = 0;
        p[n] = gemm_init(dyl, d, a->o, d, d, d, rows, 1, 1);
        p[n].out_f32 = a->d_w_o; p[n].ldc = d; p[n].rowsum_out = a->d_b_o; ++n;
        if (a->self_attn) {
            p[n] = gemm_init(dqkv, 3 * d, a->xn, d, 3 * d, d, rows, 1, 1);
            p[n].out_f32 = a->d_w_qkv; p[n].ldc = d; p[n].rowsum_out = a->d_b_qkv; ++n;
        } else {
            p[n] = gemm_init(dqkv, d, a->xn, d, d, d, rows, 1, 1);
            p[n].out_f32 = a->d_w_qkv; p[n].ldc = d; p[n].rowsum_out = a->d_b_qkv; ++n;
            p[n] = gemm_init(dkv, 2 * d, a->mem, d, 2 * d, d, rows_m, 1, 1);
            p[n].out_f32 = a->d_w_qkv + (long)d * d; p[n].ldc = d; p[n].rowsum_out = a->d_b_qkv + d; ++n;
        }
        RUN(mtn_gemm(dtype, n, p, stream));
    }
    // 6. LayerNorm backward, fused with the residual-branch gradient
    RUN(mtn_layernorm_bwd(rows, d, a->ln_eps, a->x, a->ln_a, a->mean, a->rstd, dxn, a->dy, a->dx, a->d_ln_a, a->d_ln_b, ln_partial, stream));
    return MTN_OK;
}

// ------------------------------------------------------------------------------------------ FFN
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

extern "C" int mtn_ffn_sublayer_fwd(int dtype, const mtn_ffn_args* a, void* stream) {
    MTN_CHECK_ARG(dtype == MTN_F32 || dtype == MTN_BF16, "bad dtype");
    RUN(check_ffn(a, false));
    const int d = a->d, ff = a->d_ff, rows = a->rows;
    RUN(mtn_layernorm_fwd(dtype, rows, d, a->ln_eps, a->x, a->ln_a, a->ln_b, nullptr, a->xn, a->mean, a->rstd, stream));
    mtn_gemm_problem p1 = gemm_init(a->xn, d, a->w1, d, rows, ff, d, 0, 0);
    p1.bias = a->b1; p1.relu = 1; p1.drop = a->drop_hidden; p1.out_lp = a->hid; p1.ldc = ff;
    RUN(mtn_gemm(dtype, 1, &p1, stream));
    mtn_gemm_problem p2 = gemm_init(a->hid, ff, a->w2, ff, rows, d, ff, 0, 0);
    p2.bias = a->b2; p2.drop = a->drop_out; p2.residual = a->x; p2.ldr = d; p2.out_f32 = a->y; p2.ldc = d;
    RUN(mtn_gemm(dtype, 1, &p2, stream));
    return MTN_OK;
}

extern "C" long mtn_ffn_bwd_ws_f32_floats(int rows, int d, int d_ff) {
    (void)d_ff;
    return (long)rows * d + mtn_layernorm_bwd_partial_floats(rows, d);
}

extern "C" int mtn_ffn_sublayer_bwd(int dtype, const mtn_ffn_args* a, void* stream) {
    MTN_CHECK_ARG(dtype == MTN_F32 || dtype == MTN_BF16, "bad dtype");
    RUN(check_ffn(a, true));
    const int d = a->d, ff = a->d_ff, rows = a->rows;
    void* dyl = a->ws_lp;                                 // [rows,d]
    void* dh = lp_off(a->ws_lp, (long)rows * d, dtype);   // [rows,d_ff]
    float* dxn = a->ws_f32;
    float* ln_partial = a->ws_f32 + (long)rows * d;
    RUN(mtn_dropout_bwd_to_lp(dtype, (long)rows * d, a->dy, a->drop_out, dyl, stream));
    {   // dh = (dyl @ W2) * relu'(h) * hidden-dropout mask  — both recovered from the saved hidden (hid > 0)
        mtn_gemm_problem p = gemm_init(dyl, d, a->w2, ff, rows, ff, d, 0, 1);
        p.gate = a->hid;
        p.gate_scale = (a->drop_hidden.p > 0.f && a->drop_hidden.seed) ? 1.0f / (1.0f - a->drop_hidden.p) : 1.0f;
        p.out_lp = dh; p.ldc = ff;
        RUN(mtn_gemm(dtype, 1, &p, stream));
    }
    {   // dxn = dh @ W1
        mtn_gemm_problem p = gemm_init(dh, ff, a->w1, d, rows, d, ff, 0, 1);
        p.out_f32 = dxn; p.ldc = d;
        RUN(mtn_gemm(dtype, 1, &p, stream));
    }
    {   // dW2 = dyl^T hid, dW1 = dh^T xn; bias gradients as row sums
        mtn_gemm_problem p[2];
        p[0] = gemm_init(dyl, d, a->hid, ff, d, ff, rows, 1, 1);
        p[0].out_f32 = a->d_w2; p[0].ldc = ff; p[0].rowsum_out = a->d_b2;
        p[1] = gemm_init(dh, ff, a->xn, d, ff, d, rows, 1, 1);
        p[1].out_f32 = a->d_w1; p[1].ldc = d; p[1].rowsum_out = a->d_b1;
        RUN(mtn_gemm(dtype, 2, p, stream));
    }
    RUN(mtn_layernorm_bwd(rows, d, a->ln_eps, a->x, a->ln_a, a->mean, a->rstd, dxn, a->dy, a->dx, a->d_ln_a, a->d_ln_b, ln_partial, stream));
    return MTN_OK;
}
