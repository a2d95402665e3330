/*
 * mtn_hip.h — C ABI of libmtn_hip.so: MI355X (gfx950) kernels for the MTN transformer hot path.
 *
 * The reference (henryhungle/MTN) has no FFI: its operator boundary is the Python nn.Module call
 * surface in mtn.py.  Each entry point below replaces the op sequence of one reference operator
 * (file:line cited per function) and is what a ctypes binding in the reference would call
 * (INTEGRATION.md shows the stub).  Conventions:
 *   - extern "C", plain pointers and sizes, no exceptions cross the boundary;
 *   - every function returns 0 on success, nonzero on error; mtn_last_error() gives the text;
 *   - all tensors are BORROWED device pointers, row-major, innermost dimension contiguous;
 *     the caller allocates outputs / saved-for-backward buffers and owns their lifetime;
 *   - kernels are enqueued on the hipStream_t passed in (void*), never synchronise, keep no
 *     global mutable state: re-entrant per stream, capturable into a hipGraph;
 *   - "lowp" buffers hold the compute element type selected by `dtype`:
 *       MTN_F32  : float   (exact-fp32 MFMA path, parity mode)
 *       MTN_BF16 : bfloat16 (fp32 accumulate; throughput mode)
 *     residual streams, LayerNorm / softmax statistics, biases, LN gains and all gradients of
 *     parameters are always float.
 */
#ifndef MTN_HIP_H
#define MTN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTN_F32 0
#define MTN_BF16 1

#define MTN_OK 0
#define MTN_ERR_ARG 1
#define MTN_ERR_LAUNCH 2

/* Library / error reporting. */
const char* mtn_last_error(void);
/* 100: rounds 1-3.  110 (round 4): mtn_gemm_problem gained `ln`, mtn_mha_args / mtn_ffn_args gained `ln_fold` (callers must zero
 * the structs or set them), mtn_attn_args gained kv_acc / kv_last in round 3, and mtn_mha_bwd_ws_f32_floats() /
 * mtn_ffn_bwd_ws_f32_floats() return larger workspaces (multi-pass dK / dV sums; LayerNorm row-sum partials).
 * 111: mtn_transpose_desc gained `dst_off` (the transposed copies have a compact buffer of their own).
 * 112 (round 5): new entry points only (mtn_measure_mfma_peak_shapes, ...: see INTEGRATION.md); mtn_measure_mfma_peak now reports
 * the better of two MFMA shapes.
 * 113 (round 6): mtn_decode_args gained the trailing field `max_m`; mtn_decode_step takes W <= 16 rows, clamps `grid` to the device's
 * compute-unit count and bounds its polls in time (see there). */
int mtn_version(void);

/* ------------------------------------------------------------------------------------------
 * Dropout stream: keep-mask bit for element `idx` of site `salt` is a pure function of
 * (*seed, salt, idx), so backward regenerates it.  `seed` is a DEVICE pointer (a replayed
 * hipGraph sees a fresh value each step).  p == 0 or seed == NULL disables dropout.
 * (reference: nn.Dropout at mtn.py:123,230,246,277)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    float p;
    uint32_t salt;
    const uint64_t* seed;
} mtn_dropout;

/* ------------------------------------------------------------------------------------------
 * GEMM (building block of every Linear on the path: mtn.py:243-244,256-258,267,273-280).
 *   C[i][j] = epilogue( sum_k opA(i,k) * opB(j,k) )          i<M, j<N, k<K
 *   a_trans == 0 : opA(i,k) = A[i*lda + k]      a_trans == 1 : opA(i,k) = A[k*lda + i]
 *   b_trans == 0 : opB(j,k) = B[j*ldb + k]      b_trans == 1 : opB(j,k) = B[k*ldb + j]
 *   epilogue(v) : v += bias[j]; relu; dropout; gate (v = gate[i][j] > 0 ? v*gate_scale : 0);
 *                 v += residual[i*ldr + j]; then stored to out_f32 and/or out_lp (row stride ldc).
 *   rowsum_out (optional): rowsum_out[i] = sum_k opA(i,k)   (bias gradients ride on the dW GEMM).
 * Forward Linear: y = x W^T  -> A=x, B=W (both non-trans).  dX = dY W -> A=dY, B=W with b_trans.
 * dW = dY^T X -> A=dY with a_trans, B=X with b_trans.
 * Constraints: A,B are lowp of `dtype`; K, lda, ldb multiples of 8; 16-byte aligned bases.
 * Up to MTN_GEMM_MAX_GROUP independent problems are executed by ONE launch.
 * ------------------------------------------------------------------------------------------ */
/* Optional optimiser epilogue of a parameter-gradient GEMM (dW = dY^T X, a_trans = b_trans = 1): C is not stored as a
 * gradient; it IS the gradient of the parameter block p[M,N] (row stride ldc, the same element offsets as out_f32), and the
 * Adam update of mtn_adam_step is applied to it tile by tile while the accumulators are still in registers:
 *   g = C * *grad_scale;  m,v updated;  p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps);
 *   p_lp[i*ldc+j] = lowp(p)   (optional; the compute-dtype weight copy)
 *   p_lpT[j*ldT+i] = lowp(p)  (optional; the transposed copy the dX GEMMs read)
 * This removes the gradient's HBM round trip (8 B/param), the separate transpose pass, and hides the optimiser's 26 B/param
 * behind the contraction.  Only valid when this GEMM is the ONLY contribution to that gradient in the step (no residual /
 * accumulation, no gradient exchange between ranks).  write_grad != 0 also stores C to out_f32. */
typedef struct {
    float *p, *m, *v;
    void* p_lp;
    void* p_lpT;
    int ldT;
    int write_grad;
    const float* state;      /* device [8]: as mtn_adam_step */
    const float* grad_scale; /* optional device scalar */
    float beta1, beta2, eps;
} mtn_adam_fuse;

/* LayerNorm backward by linearity (round 4; bf16, row-major A, B as the weight lies: dX = dY W).
 * A sublayer computes q = W LN(x) + b with LN(x) = a2 (x - mean) rstd + b2 (mtn.py:111-114, 127).  The two row sums LayerNorm
 * backward needs from g = dq W —  s1 = sum_c a2_c g_c  and  s2 = sum_c a2_c (x_c - mean) g_c  — are linear in dq:
 *     s1 = sum_k dq_k u_k,                 u_k = sum_c W_kc a2_c
 *     s2 = 1/rstd sum_k dq_k (q_k - c_k),  c_k = b_k + sum_c W_kc b2_c        (q = the SAVED projection output)
 * so the kernels that PRODUCE dq emit them as per-row partial sums (one pair per row and column block, plain stores, no
 * cross-workgroup synchronisation) and the GEMM g = dq W finishes LayerNorm backward in its epilogue:
 *     dx = rstd a2 g - rstd s1/d - s2 rstd^2/(std (d-1)) (x - mean) + dres
 * instead of writing g for a separate LayerNorm-backward launch.  u | c ("fold vectors", float [2K]) come from mtn_ln_fold().
 *   mode MTN_LN_EMIT    (the GEMM that produces dq itself: dh = (dy W2) gated, FFN sublayers; N % 64 == 0, 64-column tiles):
 *       part[row][N/64][2] += {sum v u, sum v (gate * gate_inv_scale - c)} over each 64-column block of the row; v = the stored value
 *   mode MTN_LN_CONSUME (g = dq W, N = d_model <= 512):  out_f32 / out_lp of the problem are ignored (may be NULL);
 *       part[row][np][2] are summed in index order; dx, dx_lp (through dx_lp_drop, as mtn_ln_bwd_desc) are written;
 *       colpart [ceil(M/8)][2N] receives the da2 | db2 partial rows of mtn_layernorm_bwd (same layout: finalize unchanged).
 *   (mode 3, round 4's forward twin — LayerNorm FORWARD by linearity — was measured without net gain and removed in version 112.) */
#define MTN_LN_EMIT 1
#define MTN_LN_CONSUME 2
typedef struct mtn_ln_epilogue_s {
    int mode;
    const float* fold;     /* EMIT: u[N] then c[N] */
    float gate_inv_scale;  /* EMIT: 1 / (the gate's dropout scale) */
    float* part;           /* EMIT: written; CONSUME: read */
    int np;                /* CONSUME: partial pairs per row */
    const float *x, *a2, *mean, *rstd, *dres; /* CONSUME: the fields of mtn_ln_bwd_desc; x, mean, rstd saved by forward, dres optional */
    float eps;
    float* dx;
    void* dx_lp;           /* optional, compute dtype */
    mtn_dropout dx_lp_drop;
    float* colpart;        /* optional */
} mtn_ln_epilogue;

typedef struct {
    const void* A;
    const void* B;
    int lda, ldb;
    int M, N, K;
    int a_trans, b_trans;
    const float* bias;
    int relu;
    mtn_dropout drop;
    const void* gate; /* lowp [M,N], row stride ldc */
    float gate_scale;
    const float* residual;
    int ldr;
    float* out_f32;
    void* out_lp;
    int ldc;
    int lp_drop_after_residual; /* != 0: `drop` is NOT applied to v; it is applied to the out_lp copy only, after the residual
                                   (an accumulated gradient leaving also as the masked compute-dtype operand of its consumer) */
    float* rowsum_out;
    const mtn_adam_fuse* adam; /* HOST pointer, read during the call only; NULL = plain GEMM */
    const struct mtn_ln_epilogue_s* ln; /* HOST pointer, read during the call only; NULL = none (LayerNorm-backward epilogue, above) */
} mtn_gemm_problem;

#define MTN_GEMM_MAX_GROUP 16
int mtn_gemm(int dtype, int count, const mtn_gemm_problem* problems /* host array */, void* stream);
/* Table form for parameter-gradient problems (a_trans = b_trans = 1, bf16, plain C or the optimiser epilogue): up to 1024
 * problems in ONE launch of 128 x 128 tiles, in the order given (put contractions of different length next to each other:
 * resident workgroups then drift out of phase and the epilogue's HBM streaming overlaps other tiles' contractions).  The
 * problem list is staged to device memory on `stream`; safe under hipGraph capture. */
int mtn_gemm_tt_table(int dtype, int count, const mtn_gemm_problem* problems /* host array */, void* stream);
/* The same launch also carrying the REST of the optimiser step (round 5, version 112) — `opt.step()` of data_utils.py:154 for the
 * parameters no dW epilogue covers, so that nothing runs behind the launch:
 *   ln[]        LayerNorm gains / biases: gradient = the sum of the partial rows left by the LayerNorm-backward kernels (what
 *               mtn_layernorm_bwd_finalize computes, same summation order, same bits; stored at g + a_off / g + b_off) + Adam;
 *   chunks      ranges of the flat buffers whose gradient is COMPLETE before the launch (embedding tables): Adam as
 *               mtn_adam_step_chunks;
 *   bias_adam   every problem whose rowsum_out lies inside [g, g + n_flat): that bias is updated by the tile that sums its gradient.
 * All with the arithmetic of mtn_adam_step (bit-identical results).  The caller leaves exactly these ranges out of any later pass.
 * ln / chunk_off / chunk_len are HOST arrays read during the call; offsets and n_flat are in elements of the flat buffers. */
typedef struct { const float* partial; int nparts, d; long a_off, b_off; } mtn_tt_ln_unit;
typedef struct {
    float *p, *g, *m, *v; void* lp;          /* flat fp32 parameters, gradients, both moments; compute-dtype copy (NULL = none) */
    long n_flat;
    int n_ln; const mtn_tt_ln_unit* ln;
    int n_chunks; const long* chunk_off; const int* chunk_len;   /* <= 4096 elements each, multiples of 4 */
    int bias_adam;
    const float* state; const float* grad_scale; float beta1, beta2, eps;     /* as mtn_adam_fuse */
} mtn_tt_aux;
int mtn_gemm_tt_table_aux(int dtype, int count, const mtn_gemm_problem* problems /* host array */, const mtn_tt_aux* aux /* host, NULL = none */, void* stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm, MTN variant (mtn.py:103-114): y = a2 * (x-mean) / (std_unbiased + eps) + b2.
 * fwd: x [rows,d] float.  Optional outputs: y_f32, y_lp (lowp), mean[rows], rstd[rows]
 *      with rstd = 1/(std+eps).
 * bwd: g = dL/dy [rows,d] float; dres (optional) = gradient arriving on the residual branch,
 *      added to dx.  dx [rows,d] float (may alias dres).  da2/db2 [d] float are WRITTEN
 *      (not accumulated); `partial` is caller scratch of mtn_layernorm_bwd_partial_floats().
 * ------------------------------------------------------------------------------------------ */
#define MTN_LN_MAX_GROUP 8
typedef struct {
    int rows, d;
    float eps;
    const float *x, *a2, *b2;
    float* y_f32;
    void* y_lp;
    float *mean, *rstd;
    /* Optional fused source (Embeddings + PositionalEncoding, mtn.py:282-309): when `tokens` is set the row is built as
     * x[row] = dropout(lut[tokens[row]] * emb_scale + pe[row % seq_len]) instead of being read from `x`; when only `pe` is
     * set, x[row] = dropout(x[row] + pe[row % seq_len]) (feature streams, mtn.py:378).  `x_out` (optional) receives that
     * row (saved for backward).  no_ln != 0: no normalisation, y = the row (target embedding, mtn.py:59). */
    const long* tokens;
    const float* lut;
    float emb_scale;
    const float* pe;
    int seq_len;
    mtn_dropout drop;
    float* x_out;
    int no_ln;
} mtn_ln_fwd_desc;
/* Embedding backward: dlut[tokens[row]] += dx[row] * emb_scale * keep(row*d+c)/(1-p).
 * Default (lut_rows > 0 on every descriptor: the vocabulary size of the table behind dlut): bitwise reproducible — each
 * vocabulary entry is owned by one wave (frequent entries by one workgroup) that scans the token lists of all streams sharing
 * the table and adds the matching rows in list order, no atomics; measured 31 vs 13 us (uniform tokens) and 175 vs 69 us
 * (ragged, 25 % pads) at cfg2 batch 32 against the alternative.  MTN_EMBED_DETERMINISTIC=0 in the environment (or lut_rows == 0)
 * selects that alternative: float atomic adds, whose rounding depends on arrival order — then the only run-to-run noise on the
 * whole path. */
typedef struct {
    int rows, d;
    const long* tokens;
    const float* dx;
    float emb_scale;
    mtn_dropout drop;
    float* dlut;
    int lut_rows;
} mtn_embed_bwd_desc;
int mtn_embed_bwd_group(int count, const mtn_embed_bwd_desc* descs /* host array, <= MTN_LN_MAX_GROUP */, void* stream);
typedef struct {
    int rows, d;
    float eps;
    const float *x, *a2, *mean, *rstd, *g, *dres;
    float* dx;
    float* partial; /* NULL: dx only */
    /* optional second output: dx as the NEXT backward stage wants it — through that sublayer's output-dropout mask and in
       the compute dtype (saves its cast launch).  dx_lp NULL = off; dx_lp_dtype MTN_F32 | MTN_BF16. */
    void* dx_lp;
    int dx_lp_dtype;
    mtn_dropout dx_lp_drop;
} mtn_ln_bwd_desc;
/* Grouped forms: up to MTN_LN_MAX_GROUP independent row streams per launch. */
int mtn_layernorm_fwd_group(int dtype, int count, const mtn_ln_fwd_desc* descs /* host array */, void* stream);
int mtn_layernorm_bwd_group(int count, const mtn_ln_bwd_desc* descs /* host array */, void* stream);
int mtn_layernorm_fwd(int dtype, int rows, int d, float eps, const float* x, const float* a2, const float* b2,
                      float* y_f32, void* y_lp, float* mean, float* rstd, void* stream);
/* Fold vectors of the Linears that follow a LayerNorm (see mtn_ln_epilogue): for each descriptor, with W [K, d] in the compute
 * dtype (bf16), out[k] = sum_c W[k][c] a2[c] and out[K + k] = bias[k] + sum_c W[k][c] b2[c].  One launch for the whole model:
 * `descs_device` is a DEVICE array of `count` descriptors, `block_desc` a DEVICE int32 array giving the descriptor of each
 * 64-row block (block_start = the descriptor's first block), total_blocks = sum ceil(K / 64).  d % 8 == 0, d <= 2048. */
typedef struct {
    const void* w;
    const float *bias, *a2, *b2;
    float* out;
    int K, block_start;
} mtn_ln_fold_desc;
int mtn_ln_fold(int dtype, const mtn_ln_fold_desc* descs_device, const int* block_desc, int total_blocks, int d, void* stream);
long mtn_layernorm_bwd_partial_floats(int rows, int d);
int mtn_layernorm_bwd_nparts(int rows);
/* da2/db2 == NULL: only dx is produced now and `partial` is left for a later grouped mtn_layernorm_bwd_finalize()
 * (many LayerNorms per launch, off the critical path of backward).  partial == NULL: dx only, no parameter gradients. */
typedef struct {
    const float* partial; /* [nparts][2*d] as written by mtn_layernorm_bwd */
    int nparts, d;
    float *da2, *db2;
} mtn_ln_finalize_desc;
#define MTN_LN_FINALIZE_MAX_GROUP 112 /* descriptors per launch (3.6 KB of kernel arguments): all LayerNorms of a cfg2 step in one */
int mtn_layernorm_bwd_finalize(int count, const mtn_ln_finalize_desc* descs /* host array */, void* stream);
int mtn_layernorm_bwd(int rows, int d, float eps, const float* x, const float* a2, const float* mean,
                      const float* rstd, const float* g, const float* dres, float* dx, float* da2, float* db2,
                      float* partial, void* stream);

/* ------------------------------------------------------------------------------------------
 * Scaled-dot-product attention core (mtn.py:221-231) for all heads of all batch rows.
 *   q  : lowp, element (b,i,h,c) at q[(b*a+i)*ldq + h*dk + c]      i<a
 *   k,v: lowp, element (b,j,h,c) at k[(b*m+j)*ldkv + h*dk + c]     j<m
 *   mask: uint8, element (b,i,j) at mask[b*mask_sb + i*mask_sq + j]; 0 => score := -1e9
 *         (mask_sq == 0 broadcasts over query rows; mask == NULL => nothing masked)
 *   o  : lowp [(b*a+i)*ldo + h*dk + c];  lse: float2 per (b,head,i) at lse[2*((b*h+hh)*a+i)] = {row max, 1/row sum}
 *        (kept apart: a fully masked row has max = -1e9, where max + log(sum) would lose log(sum))
 * bwd: d_o lowp (same layout as o) -> dq (ldq layout), dk/dv (ldkv layout), all lowp.
 *      Gradient does not flow through masked scores (masked_fill).  Any number of query rows (the PE table of the
 *      reference goes to 5000, mtn.py:293): the MFMA kernel walks them in passes of 32.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int B, h, a, m, dk;
    const void *q, *k, *v;
    int ldq, ldkv;
    const uint8_t* mask;
    long mask_sb, mask_sq;
    mtn_dropout drop;
    void* o;
    int ldo;
    float* lse;
    /* backward only */
    const void* d_o;
    void *dq, *dk_out, *dv_out;
    /* backward, set by the library (callers leave them 0): one launch covers query rows q0 .. q0+qn-1 of every sequence;
       sequences longer than 32 rows are walked in several passes, kv_accum != 0 = add dk/dv to the earlier passes' sums,
       kv_last != 0 = this pass writes the final dk/dv */
    int q0, qn, kv_accum;
    /* backward, optional (caller): fp32 workspace of 2 * B * m * h * dk floats.  With it a multi-pass backward keeps the dK / dV
       sums of the passes in fp32 ([B*m][h*dk] for dK, then the same for dV) and rounds to the output type once, in the last
       pass; without it the passes accumulate in the (low-precision) output buffers. */
    float* kv_acc;
    int kv_last;
} mtn_attn_args;
#define MTN_ATTN_MAX_GROUP 4
/* Grouped forms: up to MTN_ATTN_MAX_GROUP independent attention problems (different shapes allowed) per launch. */
int mtn_attention_fwd_group(int dtype, int count, const mtn_attn_args* args /* host array */, void* stream);
int mtn_attention_bwd_group(int dtype, int count, const mtn_attn_args* args /* host array */, void* stream);
int mtn_attention_fwd(int dtype, const mtn_attn_args* args, void* stream);
int mtn_attention_bwd(int dtype, const mtn_attn_args* args, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused sublayers: SublayerConnection.forward (mtn.py:125-127) around MultiHeadedAttention
 * (mtn.py:248-267) or PositionwiseFeedForward (mtn.py:279-280):
 *     y = x + dropout( f( LayerNorm(x) ) )
 * Weight layout: w_qkv is the three input Linears stacked [3d,d] (q,k,v order = linears[0..2]),
 * b_qkv [3d]; w_o/b_o = linears[3].  Self-attention (kv source = LayerNorm(x)) runs one packed
 * QKV projection; cross-attention projects q from LayerNorm(x) and k,v from `mem` [B,m,d] lowp.
 * Saved-for-backward buffers are caller-allocated (sizes in comments).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int B, a, m, d, h;
    int self_attn;
    float ln_eps;
    mtn_dropout drop_attn; /* on softmax probabilities (mtn.py:230) */
    mtn_dropout drop_out;  /* on the sublayer output (mtn.py:127)   */
    const float* x;        /* [B,a,d] */
    const void* mem;       /* lowp [B,m,d]; unused when self_attn */
    const uint8_t* mask;
    long mask_sb, mask_sq;
    const float *ln_a, *ln_b;
    const void* w_qkv; /* lowp [3d,d] */
    const float* b_qkv;
    const void* w_o; /* lowp [d,d] */
    const float* b_o;
    const void* w_qkv_t; /* lowp [d,3d] = w_qkv^T, optional (backward: NULL -> dX = dY W reads w_qkv as it lies, b_trans = 1 on the LDS-DMA GEMM: what the training step does since round 3) */
    const void* w_o_t;   /* lowp [d,d]  = w_o^T,  optional; REQUIRED for the fused head backward (it reads w_o^T rows), NULL -> staged path */
    float* y;    /* [B,a,d] */
    void* xn;    /* lowp [B*a,d]  saved */
    float* mean; /* [B*a]         saved */
    float* rstd; /* [B*a]         saved */
    void* qkv;   /* lowp: self [B*a,3d]; cross [B*a,d] (q only)   saved */
    void* kv;    /* lowp cross only [B*m,2d]                      saved */
    void* o;     /* lowp [B*a,d]  saved */
    float* lse;  /* [2*B*h*a]     saved */
    /* ---- backward (mtn_mha_sublayer_bwd) ---- */
    const float* dy; /* [B,a,d] */
    float* dx;       /* [B,a,d] written */
    float* dmem;     /* [B,m,d] float gradient of mem (cross only; NULL = not needed) */
    int dmem_accumulate; /* 0: dmem is written; 1: dmem += (memory shared by several sublayers) */
    float *d_ln_a, *d_ln_b;                   /* [d] written */
    float *d_w_qkv, *d_b_qkv, *d_w_o, *d_b_o; /* written */
    void* ws_lp;   /* lowp scratch: mtn_mha_bwd_ws_lp_elems() elements */
    float* ws_f32; /* float scratch: mtn_mha_bwd_ws_f32_floats() */
    int defer_param_grads; /* 1: backward skips the dW/db GEMMs and the LayerNorm finalize; the caller batches them
                              later from mtn_mha_param_grad_work() (ws_lp/ws_f32 and saved buffers must stay alive) */
    /* ---- gradient hand-off between consecutive sublayers of a chain (all optional) ----
       dyl_ready: dropout-masked compute-dtype image of dy already produced by the sublayer that ran before this one in
                  backward (its next_dyl): the cast stage is skipped and the dW GEMMs read it instead of ws_lp's copy.
       next_dyl / next_drop: ask this backward's LayerNorm stage to also write dx through `next_drop` (the output dropout of
                  the sublayer that will consume dx as its dy) in the compute dtype, [rows, d]. */
    const void* dyl_ready;
    void* next_dyl;
    mtn_dropout next_drop;
    /* forward, cross attention: 1 = `kv` already holds K|V of the memory (projected ahead of the layer loop for every layer
       that attends the same constant memory, one grouped GEMM): the sublayer skips that projection.  Backward is unchanged. */
    int kv_ready;
    /* backward, optional: the memory gradient after THIS sublayer's contribution is final (the caller knows: last user of a
       shared gradient buffer) and its producer wants it once more through its output dropout in the compute dtype
       [B*m, d] — written by the dmem GEMM's epilogue instead of a cast launch */
    void* dmem_lp;
    mtn_dropout dmem_lp_drop;
    /* optional: fold vectors of w_qkv behind this sublayer's LayerNorm (mtn_ln_fold: float [2K], K = 3d self / d cross: only the
       q block sees LN(x)).  With them (bf16, fused head backward) LayerNorm backward rides in the dLN-out GEMM's epilogue
       (mtn_ln_epilogue) and the group's LayerNorm-backward launch disappears; NULL = the separate launch. */
    const float* ln_fold;
    /* (version 112: the five trailing fields of round 4's LayerNorm FORWARD by linearity — xa, x_stats, ya, y_stats, next_ln_a — are gone) */
} mtn_mha_args;
int mtn_mha_sublayer_fwd(int dtype, const mtn_mha_args* args, void* stream);
int mtn_mha_sublayer_bwd(int dtype, const mtn_mha_args* args, void* stream);
/* Deferred parameter-gradient work of one sublayer backward: fills up to 3 GEMM problems (host structs, nothing is
 * launched) and one LayerNorm finalize descriptor; returns the number of GEMM problems.  Feed them to mtn_gemm()
 * (any grouping) and mtn_layernorm_bwd_finalize() once the sublayer's backward has been enqueued. */
int mtn_mha_param_grad_work(int dtype, const mtn_mha_args* args, mtn_gemm_problem* out3, mtn_ln_finalize_desc* out_ln);
long mtn_mha_bwd_ws_lp_elems(int B, int a, int m, int d, int self_attn);
long mtn_mha_bwd_ws_f32_floats(int B, int a, int m, int d);

typedef struct {
    int rows, d, d_ff;
    float ln_eps;
    mtn_dropout drop_hidden; /* mtn.py:280 */
    mtn_dropout drop_out;    /* mtn.py:127 */
    const float* x;          /* [rows,d] */
    const float *ln_a, *ln_b;
    const void* w1; /* lowp [d_ff,d] */
    const float* b1;
    const void* w2; /* lowp [d,d_ff] */
    const float* b2;
    const void* w1_t; /* lowp [d,d_ff] = w1^T, optional (backward; NULL -> w1 as it lies, b_trans = 1) */
    const void* w2_t; /* lowp [d_ff,d] = w2^T, optional (backward; NULL -> w2 as it lies, b_trans = 1) */
    float* y;    /* [rows,d] */
    void* xn;    /* lowp [rows,d]    saved */
    float* mean; /* saved */
    float* rstd; /* saved */
    void* hid;   /* lowp [rows,d_ff] saved (post-ReLU, post-dropout) */
    /* ---- backward ---- */
    const float* dy;
    float* dx;
    float *d_ln_a, *d_ln_b, *d_w1, *d_b1, *d_w2, *d_b2;
    void* ws_lp;   /* lowp scratch: rows*d + rows*d_ff elements */
    float* ws_f32; /* float scratch: mtn_ffn_bwd_ws_f32_floats() */
    int defer_param_grads;
    /* ---- gradient hand-off between consecutive sublayers of a chain (all optional) ----
       dyl_ready: dropout-masked compute-dtype image of dy already produced by the sublayer that ran before this one in
                  backward (its next_dyl): the cast stage is skipped and the dW GEMMs read it instead of ws_lp's copy.
       next_dyl / next_drop: ask this backward's LayerNorm stage to also write dx through `next_drop` (the output dropout of
                  the sublayer that will consume dx as its dy) in the compute dtype, [rows, d]. */
    const void* dyl_ready;
    void* next_dyl;
    mtn_dropout next_drop;
    /* forward, optional: compute-dtype copy of y [rows,d], written by the same epilogue — for an output that a later sublayer
       attends as un-projected memory (an auto-encoder stream, mtn.py:215), which otherwise costs a cast launch */
    void* y_lp;
    const float* ln_fold; /* optional: fold vectors of w1 (float [2 d_ff]), as mtn_mha_args.ln_fold */
} mtn_ffn_args;
int mtn_ffn_sublayer_fwd(int dtype, const mtn_ffn_args* args, void* stream);
int mtn_ffn_sublayer_bwd(int dtype, const mtn_ffn_args* args, void* stream);
long mtn_ffn_bwd_ws_f32_floats(int rows, int d, int d_ff);
int mtn_ffn_param_grad_work(int dtype, const mtn_ffn_args* args, mtn_gemm_problem* out2, mtn_ln_finalize_desc* out_ln);

/* ------------------------------------------------------------------------------------------
 * Lockstep sublayer groups.  The sublayers of one DecoderLayer that do not depend on each other (x's text attention,
 * the two auto-encoder chains) share every kernel launch: stage 1 one grouped LayerNorm, stage 2 one grouped GEMM
 * (QKV / Q+KV / FFN-1), stage 3 one grouped attention (attention members only), stage 4 one grouped GEMM (output
 * projection / FFN-2 with bias+dropout+residual).  Backward mirrors it in 5 grouped stages.  Launch count per group is
 * 4 (5) whatever the number of members; up to MTN_SUBLAYER_MAX_GROUP attention + as many FFN members.
 * Backward always defers parameter gradients (see mtn_*_param_grad_work).
 * ------------------------------------------------------------------------------------------ */
#define MTN_SUBLAYER_MAX_GROUP 4
/* Forward of a group whose members are bf16 with d = 512, h = 8 (d_k = 64), a <= 64 query rows per sample and memories of
 * <= 256 rows runs stages 1-3 as ONE launch (csrc/fused.hip: LayerNorm -> head slice of the input projections -> attention,
 * or LayerNorm -> column slice of w_1 + ReLU + dropout, per (block of samples, head | column slice) workgroup); results and
 * saved buffers are those of the four-launch path.  mtn_fused_enable(0) turns that off for the process (A/B measurements and
 * the tests that compare the two paths); returns the previous setting (-1 = never set: environment MTN_FUSED=0 disables). */
int mtn_fused_enable(int on);
/* How many sublayer groups took which path since the library was loaded: out4 = {forward fused, forward per-stage, backward
 * fused (groups with attention members), backward per-stage}.  The parity tests use it to assert that the fused kernels ran. */
int mtn_fused_counters(long* out4);
/* Backward groups (since the library was loaded) whose LayerNorm backward rode in the epilogue of the dLN-out GEMM (mtn_ln_epilogue)
 * instead of its own launch: the tests use it to assert which path ran. */
long mtn_ln_epilogue_groups(void);
int mtn_sublayer_group_fwd(int dtype, int n_mha, const mtn_mha_args* mha, int n_ffn, const mtn_ffn_args* ffn, void* stream);
int mtn_sublayer_group_bwd(int dtype, int n_mha, const mtn_mha_args* mha, int n_ffn, const mtn_ffn_args* ffn, void* stream);

/* ------------------------------------------------------------------------------------------
 * Elementwise helpers on the path.
 * ------------------------------------------------------------------------------------------ */
/* dst(lowp) = cast(src float); n elements. */
int mtn_cast_f32_to_lp(int dtype, long n, const float* src, void* dst, void* stream);
typedef struct {
    long n;
    const float* src;
    void* dst;
    mtn_dropout drop;
    const float* gate; /* optional: elements with gate[i] <= 0 are zeroed (ReLU backward from the saved activation) */
} mtn_cast_desc;
#define MTN_CAST_MAX_GROUP 8
/* Grouped cast / dropout-backward: dst_g(lowp)[i] = src_g[i] * keep_g(i)/(1-p_g) [* (gate_g[i] > 0)]. */
int mtn_cast_group(int dtype, int count, const mtn_cast_desc* descs /* host array */, void* stream);
/* dst(lowp)[i] = src[i] * keep(i)/(1-p): gradient entering a dropped-out branch (mtn.py:127). */
int mtn_dropout_bwd_to_lp(int dtype, long n, const float* src, mtn_dropout drop, void* dst, void* stream);

/* ------------------------------------------------------------------------------------------
 * Loss head: Generator log-softmax (mtn.py:62-69) + LabelSmoothing KLDivLoss(sum) (label_smoothing.py:20-32) +
 * the weighted sum of SimpleLossCompute (data_utils.py:133-144), per row, without materialising log-probabilities or
 * target distributions.  Rows of up to MTN_LOSSHEAD_MAX_SEG streams (main decoder output, auto-encoder outputs) are
 * stacked: rows[s] rows of stream s with targets target[s] (int64), loss weight coef[s] / *norm[s] (norm = device scalar).
 *   fwd: logits float [sum rows, ldz] -> lse[row], rowloss[row] (already weighted; the loss is their sum)
 *   bwd: dlogits lowp [sum rows, ldd] = *gloss * weight * (softmax * sum(td) - td); columns V..ldd-1 are zero-filled
 * ------------------------------------------------------------------------------------------ */
#define MTN_LOSSHEAD_MAX_SEG 4
typedef struct {
    int n_seg;
    int rows[MTN_LOSSHEAD_MAX_SEG];
    const long* target[MTN_LOSSHEAD_MAX_SEG];
    const float* norm[MTN_LOSSHEAD_MAX_SEG];
    float coef[MTN_LOSSHEAD_MAX_SEG];
    int V, ldz, pad;
    float smoothing;
    const float* logits;
    float* lse;
    float* rowloss;
    const float* gloss; /* backward: device scalar dL/dloss */
    void* dlogits;
    int ldd;
} mtn_losshead_args;
int mtn_losshead_fwd(const mtn_losshead_args* args, void* stream);
int mtn_losshead_bwd(int dtype, const mtn_losshead_args* args, void* stream);

/* Transposed compute-dtype weight copies (operand of dX = dY W on the LDS-DMA GEMM path): for each descriptor the
 * [rows, cols] matrix at src+off is written as [cols, rows] at dst+dst_off (v111: the copies have a compact buffer of
 * their own — by default only the W_o^T matrices are kept).  `descs_device` is a DEVICE array (built once),
 * tile_start = running sum of ceil(rows/64)*ceil(cols/64), total_tiles = the final sum. */
typedef struct {
    long off;       /* element offset of the matrix in the source buffer */
    int rows, cols;
    int tile_start;
    int reserved;
    long dst_off;   /* element offset of the transposed copy in dst */
} mtn_transpose_desc;
int mtn_transpose_group(int dtype, const void* src, void* dst, const mtn_transpose_desc* descs_device, int count,
                        int total_tiles, void* stream);

/* ------------------------------------------------------------------------------------------
 * Optimiser: Adam(betas=(0.9,0.98), eps=1e-9) under the Noam schedule
 * (train.py:190, data_utils.py:92-117), fused over ONE flat parameter buffer.
 *   state (device, 8 floats): [0]=step [1]=lr [2]=1-b1^t [3]=1-b2^t  (updated by mtn_noam_tick)
 *   mtn_adam_step: g *= grad_scale (optional float device scalar, e.g. 1/global_ntokens);
 *     m,v updated, p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps); p_lp (optional) = cast(p).
 * ------------------------------------------------------------------------------------------ */
int mtn_noam_tick(float* state, float factor, int model_size, int warmup, float beta1, float beta2, void* stream);
int mtn_adam_step(int dtype, long n, float* p, const float* g, float* m, float* v, void* p_lp, const float* state,
                  const float* grad_scale, float beta1, float beta2, float eps, void* stream);
/* The same update over chunks [off[c], off[c]+len[c]) of the flat buffers (off, len: DEVICE arrays; multiples of 4, len <=
 * 4096): the parameters that the GEMM optimiser epilogue (mtn_adam_fuse) does not cover. */
int mtn_adam_step_chunks(int dtype, int n_chunks, const long* off, const int* len, float* p, const float* g, float* m, float* v,
                         void* p_lp, const float* state, const float* grad_scale, float beta1, float beta2, float eps, void* stream);

/* ------------------------------------------------------------------------------------------
 * Device-side batch assembly (replaces the host padding of data_handler.py:206-274 `make_batch` and the mask passes of
 * data_utils.py:23-54 `Batch`).  The corpus stays resident in HBM: each token field as one flat int64 buffer with per-item
 * start/len tables, each feature type as one flat [frames, F] float buffer with per-video start/len tables.  A batch is
 * the list of item ids (`ids`, device int32[B]; NULL = items 0..B-1).
 *   tokens  : out[b,l] = l < len ? flat[start+l] : pad; mask[b,l] = (out != pad); std_mask[b,i,j] = (out[b,j] != pad) &
 *             (j <= i) (optional, data_utils.py:48-54); *n_nonpad += #non-pad (optional, data_utils.py:45; zero it first).
 *   features: frames start, start+skip, ...; out[b,v,:] = frame if it exists and has any element != 1, else 0;
 *             mask[b,v] = that validity (the reference pads with ones, detects all-ones frames, then zeroes them).
 * ------------------------------------------------------------------------------------------ */
#define MTN_ASSEMBLE_MAX_GROUP 8
typedef struct {
    const int64_t* flat;  /* all items of this field, concatenated */
    const int64_t* start; /* [n_items] offset of each item in flat */
    const int32_t* len;   /* [n_items] */
    const int32_t* ids;   /* [B] items of this batch, or NULL */
    int B, L;             /* L = padded length (>= longest item of the batch; longer items are cut) */
    int64_t pad;
    int64_t* out;         /* [B, L] */
    uint8_t* mask;        /* [B, L] or NULL */
    uint8_t* std_mask;    /* [B, L, L] or NULL */
    int64_t* n_nonpad;    /* scalar accumulator or NULL */
} mtn_assemble_tokens_desc;
int mtn_assemble_tokens(int count, const mtn_assemble_tokens_desc* descs /* host array */, void* stream);
typedef struct {
    const float* flat;    /* [total_frames, F] */
    const int64_t* start; /* [n_videos] first frame of each video */
    const int32_t* len;   /* [n_videos] frames per video */
    const int32_t* ids;   /* [B] videos of this batch, or NULL */
    int B, V, F, skip;    /* V = padded frame count; skip >= 1 (every skip-th frame, data_handler.py:233) */
    float* out;           /* [B, V, F] */
    uint8_t* mask;        /* [B, V] or NULL */
} mtn_assemble_features_desc;
int mtn_assemble_features(int count, const mtn_assemble_features_desc* descs /* host array */, void* stream);

/* ------------------------------------------------------------------------------------------
 * Candidate selection of the beam search (data_utils.py:219 argsorts each hypothesis' vocabulary row on the host).
 * Per row of x [rows, V] (row stride ldx): the k largest entries in descending order (equal values: ascending column),
 * packed as out[row] = { k values, k column indices as floats, x[row][extra_col] } (2k+1 floats; extra_col < 0: 0).
 * 1 <= k <= 16, V < 2^24.
 * ------------------------------------------------------------------------------------------ */
int mtn_topk_rows(const float* x, int rows, int V, long ldx, int k, int extra_col, float* out, void* stream);
/* One step of the beam search's hypothesis bookkeeping on the device (round 5, version 112; data_utils.py:209-240 — the loop body behind
 * `model.decode`): from the rows' heads left by mtn_topk_rows (k_top values, k_top columns, the <eos> column's value per live hypothesis)
 * build the new beam exactly as the reference does (hypotheses in order, candidates in descending log-probability, <unk> / <eos>
 * skipped, the worst member replaced while a candidate beats it), WRITE what the next mtn_decode_step reads (newest tokens, ancestor
 * table, position) and LOG the step for the host: log_* are [L][dialogues * width] (log_n_*: [L][dialogues]); log_done holds
 * lp + logp[<eos>] + penalty * (length + 1) of every hypothesis alive at a step >= min_len.  flags[0] is raised when a row's head holds
 * an exact tie (the reference's order then comes from the full row: re-run that search on the host path).  width <= 16.
 * State (zeroed / initialised by the caller per search): lp [dialogues * width] doubles, n_live / step [dialogues]. */
typedef struct {
    int dialogues, width, L, k_top, k, beam, unk, eos, pad, min_len;
    double penalty;
    const float* top;           /* [dialogues * width][2 * k_top + 1] */
    long* tokens; int* pos; int* anc;             /* as mtn_decode_args */
    double* lp; int* n_live; int* step; int* flags;
    int* log_parent; int* log_tok; double* log_score; double* log_done; int* log_n_old; int* log_n_new;
} mtn_beam_args;
int mtn_beam_advance(const mtn_beam_args* args /* host */, void* stream);
/* Generator (mtn.py:62-69) at inference: out[row][c] = x[row][c] - logsumexp(x[row][0..V-1]) over logit rows x [rows, V] (row
 * strides ldx / ldo; out may be x).  The logits themselves are one mtn_gemm (x W^T + b, fp32 out). */
int mtn_log_softmax_rows(const float* x, int rows, int V, long ldx, float* out, long ldo, void* stream);

/* ------------------------------------------------------------------------------------------
 * Measurement support (bench.py `roofline`): a census of the GEMM launches of one step.  Between mtn_census_begin() and
 * mtn_census_end() (returns the number of launches) every mtn_gemm call is recorded on the host (no device work, no
 * change to the launch); mtn_census_info() describes launch i — which kernel the dispatch picked, its workgroup count,
 * algorithmic FLOPs (2·M·N·K summed over the group) and algorithmic bytes (operands read once + outputs written once) —
 * and mtn_census_replay() re-issues it `reps` times on `stream` so the caller can bracket it with HIP events.  The replay
 * reuses the recorded device pointers: call it only while the model's buffers are alive (outputs are overwritten).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int dtype, count, variant, workgroups;
    double flops, bytes;
    int M[4], N[4], K[4];               /* shapes of the first four problems of the group */
} mtn_census_launch;
int mtn_census_begin(void);
int mtn_census_end(void);
int mtn_census_info(int i, mtn_census_launch* out);
int mtn_census_replay(int i, int reps, void* stream);
const char* mtn_census_variant_name(int variant);
/* Achievable dense bf16 MFMA rate of this box (register-only MFMA issue on independent accumulator chains, synchronises on its
 * own events): the measured denominator bench.py reports beside the 2.5 PFLOP/s spec figure.  scratch: >= 2048*256 floats.
 * mtn_measure_mfma_peak: the better of the two instruction shapes; mtn_measure_mfma_peak_shapes (112): both — v_mfma_f32_16x16x32_bf16
 * (the shape the path's kernels issue) and v_mfma_f32_32x32x16_bf16 (half the operand bytes per FLOP: the shape the guide's
 * 2 495 TFLOP/s figure is measured with). */
int mtn_measure_mfma_peak(int iters, float* scratch, void* stream, double* tflops);
int mtn_measure_mfma_peak_shapes(int iters, float* scratch, void* stream, double* tflops_16x16x32, double* tflops_32x32x16);
/* Achievable HBM rate of this box: a 16-byte-per-lane streaming copy src -> dst of `bytes` bytes (both buffers >= bytes, well
 * beyond the 256 MiB Infinity Cache; synchronises on its own events), (read + written bytes) / time in GB/s — the measured
 * denominator bench.py reports beside the 8 TB/s spec for the HBM-bound parameter-gradient + optimiser launch. */
int mtn_measure_hbm_peak(const void* src, void* dst, long bytes, void* stream, double* gbps);
/* ------------------------------------------------------------------------------------------
 * One decode step of the target stream as ONE persistent launch (round 5, version 112; W <= 16 and `max_m` since 113; csrc/decode.hip).  Replaces, for W <= 16 live
 * hypotheses, the per-token pass of `beam_search_decode` / `greedy_decode` (data_utils.py:197-208: `model.decode` + final LayerNorm)
 * in its cached form: the newest position of every hypothesis through N layers x (self-attention over the prefix cache, cross-attentions
 * over hoisted K|V, feed-forward), walking `stages` (a DEVICE array, built once per dialogue shape; stage 0 = MTN_DEC_EMBED, last = MTN_DEC_FINAL).
 * bf16 weights.  `grid` (<= 256: one per CU, all resident; clamped to the device's compute-unit count, and the call is refused when the three
 * classes do not fit) is the most workgroups the launch may use: W x h (<= 128) attention units + d / 16
 * writers of the residual stream + up to 128 for the wide projections — three classes, so that consecutive stages run on different
 * workgroups and each class prefetches its next stage while the others work.  Stages hand values over as 8-byte {data, tag} granules
 * (no grid barrier): xg / qg / og / hg are the granule buffers (zeroed ONCE by the caller); out_lp [W][d] bf16 = final LayerNorm
 * output (the generator's operand); sync: 4 unsigned, zeroed once: sync[0] = launch generation (advanced by the kernel), sync[2] = its check-in counter, sync[1] != 0 =
 * a poll timed out (50 ms, e.g. because another kernel held compute units and not every workgroup was resident) and the results are invalid:
 * every later launch returns at once until the caller zeroes sync[1] and sync[2] and ADVANCES sync[0] by one (stale granules of the failed
 * step must not match); mtn_amd/decode.py then re-runs the search on the launch-per-sublayer pass.
 * W x d <= 8192 and W x (2 max(d, d_ff) + 16) <= 66048 (16 rows up to d_ff = 2048, 8 at d_ff = 4096).
 * ------------------------------------------------------------------------------------------ */
#define MTN_DEC_EMBED 0     /* x = lut[token] * emb_scale + pe[pos] */
#define MTN_DEC_SELF_QKV 1  /* q | k | v = LayerNorm(x) W^T + b (N = 3d): q -> scratch, k | v -> cache row (hypothesis, pos) */
#define MTN_DEC_SELF_ATT 2  /* softmax(q K^T / sqrt(dk)) V over prefix positions 0..pos; position t of hypothesis j lives in cache slot anc[j][t] */
#define MTN_DEC_OUT 3       /* x += o W^T + b  (K = d) */
#define MTN_DEC_CROSS 4     /* q = LayerNorm(x) W_q^T + b_q per head, attention over the memory's hoisted K|V (kv, m keys, mask) */
#define MTN_DEC_FFN1 5      /* hid = relu(LayerNorm(x) W1^T + b1)  (N = d_ff) */
#define MTN_DEC_FFN2 6      /* x += hid W2^T + b2  (K = d_ff) */
#define MTN_DEC_FINAL 7     /* out_lp = LayerNorm(x) */
typedef struct {
    int kind, N, K;
    const void* w;              /* [N][K] bf16 (nn.Linear layout); MTN_DEC_CROSS: the q block of the packed q|k|v weight ([d][d]) */
    const float* bias;          /* [N] */
    const float* ln_a; const float* ln_b; float ln_eps;
    const void* kv;             /* MTN_DEC_CROSS: [W * m][2d] bf16, k | v of hypothesis j's memory rows j*m .. */
    int m;
    const unsigned char* mask; long mask_stride;     /* MTN_DEC_CROSS: key mask [W][mask_stride] uint8 (0 = masked: score -1e9); NULL = none */
    void* cache;                /* MTN_DEC_SELF_QKV / _ATT: this layer's prefix cache [W][L][2d] bf16 */
} mtn_decode_stage;
typedef struct {
    int W, d, h, L, n_stages, d_ff;
    void* xg; void* qg; void* og; void* hg;   /* granule buffers (8 bytes {data, tag} each), zeroed once: [W][d], [W][3d/2], [W][d/2], [W][d_ff/2] */
    void* out_lp;
    const long* tokens;         /* [W] newest token of every hypothesis */
    const float* lut; float emb_scale; const float* pe;     /* target embedding table [V][d], sqrt(d), positional encodings [>= L][d] */
    const int* pos;             /* device scalar: the position being decoded (0-based) */
    const int* anc;             /* [W][L] cache slot that holds position t of hypothesis j's prefix (anc[j][pos] = j) */
    unsigned* sync;
    void* dbg;                  /* NULL, or 4 x n_stages uint64: 100 MHz wall-clock stamps per stage of the first workgroup of the stage's class
                                   (entered / operands arrived / computed / stores issued) — tools/decode_timeline.py */
    int max_m;                  /* (113) the longest memory (`m`) of any MTN_DEC_CROSS stage, <= 1024: the stage list lives in device memory,
                                   so the host-side argument check needs it here */
} mtn_decode_args;
int mtn_decode_step(const mtn_decode_args* args /* host */, const mtn_decode_stage* stages_device, int grid, void* stream);
/* Test support (113): `n_wg` one-wave workgroups that hold `lds_bytes` of LDS each for `usec` microseconds and do nothing — compute units
 * taken away from whatever runs beside it.  tests/test_decode_gpu.py uses it to make mtn_decode_step's residency assumption fail. */
int mtn_debug_hold_cus(int n_wg, int lds_bytes, int usec, void* stream);
/* The library's development / test switches (MTN_GEMM_*, MTN_ATTN_*, MTN_LN_*, MTN_EMBED_DETERMINISTIC, ...) are read from the
 * environment once per call site and cached: a process that changes one after the library has used it calls this to make the
 * next launches re-read them.  Returns the new generation number. */
int mtn_reload_env(void);
/* Measurement support: a stream restricted to `n_cus` compute units (hipExtStreamCreateWithCUMask, bits dealt round-robin to the
 * 8 XCDs), for overlap experiments (tools/overlap_cu_mask_probe.py).  The caller destroys it with mtn_stream_destroy(). */
int mtn_stream_create_cu_masked(int n_cus, int low_priority, void** stream_out);
int mtn_stream_destroy(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MTN_HIP_H */
