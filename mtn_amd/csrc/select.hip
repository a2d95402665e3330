// select.hip — candidate selection of the beam search on the device (data_utils.py:219: the reference argsorts every hypothesis'
// full vocabulary row on the host and walks it from the top).  A hypothesis can place at most `beam` candidates and skips at
// most <unk> and <eos>, so only the beam + 2 largest log-probabilities of a row (plus its <eos> column) ever matter: one
// workgroup per row finds them and packs them as [k values | k indices (as floats) | x[row][extra_col]] — one small buffer, one
// device-to-host copy per generated token.  Ties: equal values come out in ascending index order; the caller detects a tie inside
// a row's head and takes the reference's order from the full row then (decode.py).
#include "common.h"

static constexpr int SEL_THREADS = 256;
static constexpr int SEL_MAX_K = 16;

__global__ __launch_bounds__(SEL_THREADS) void topk_rows_kernel(const float* __restrict__ x, int V, long ldx, int k, int extra_col, float* __restrict__ out) {
    __shared__ float s_val[SEL_THREADS / 64];
    __shared__ int s_idx[SEL_THREADS / 64];
    __shared__ int s_win;
    __shared__ int s_sel[SEL_MAX_K];                          // columns selected so far (rows longer than the register path)
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xr = x + (size_t)row * ldx;
    float* o = out + (size_t)row * (2 * k + 1);
    // this thread's candidates: columns tid, tid + 256, ... (ascending index inside a thread, so the first maximum is the lowest index)
    constexpr int PER = 16;                                   // up to 4096 columns in registers; longer rows re-read memory
    float v[PER];
    const bool in_regs = V <= PER * SEL_THREADS;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = tid + i * SEL_THREADS;
        v[i] = (in_regs && c < V) ? xr[c] : -INFINITY;
    }
    int taken_lo = 0;                                         // bit i: column tid + i * 256 already selected (register path)
    for (int j = 0; j < k; ++j) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
        if (in_regs) {
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const bool free = !((taken_lo >> i) & 1);
                if (free && v[i] > best) { best = v[i]; bi = tid + i * SEL_THREADS; }
            }
        } else {
            for (int c = tid; c < V; c += SEL_THREADS) {
                bool used = false;
                for (int t = 0; t < j; ++t) used = used || (s_sel[t] == c);
                const float xv = xr[c];
                if (!used && xv > best) { best = xv; bi = c; }
            }
        }
        // workgroup arg-max, ties to the lower column
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane == 0) { s_val[wave] = best; s_idx[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            float b = s_val[0];
            int w = s_idx[0];
            for (int t = 1; t < SEL_THREADS / 64; ++t)
                if (s_val[t] > b || (s_val[t] == b && s_idx[t] < w)) { b = s_val[t]; w = s_idx[t]; }
            if (w == 0x7fffffff) { w = 0; b = -INFINITY; }   // fewer than k finite columns
            o[j] = b;
            o[k + j] = (float)w;
            s_win = w;
            s_sel[j] = w;
        }
        __syncthreads();
        const int w = s_win;
        if (in_regs && (w & (SEL_THREADS - 1)) == tid) taken_lo |= 1 << (w / SEL_THREADS);
        __syncthreads();                                     // s_val / s_idx / s_win are rewritten next round
    }
    if (tid == 0) o[2 * k] = (extra_col >= 0 && extra_col < V) ? xr[extra_col] : 0.f;
}

extern "C" int mtn_topk_rows(const float* x, int rows, int V, long ldx, int k, int extra_col, float* out, void* stream) {
    MTN_CHECK_ARG(x && out && rows > 0 && V > 0 && ldx >= V, "bad row matrix");
    MTN_CHECK_ARG(k >= 1 && k <= SEL_MAX_K && k <= V, "k must be in [1, 16] and <= V");
    MTN_CHECK_ARG(V < (1 << 24), "column indices travel as floats: V must be below 2^24");
    hipLaunchKernelGGL(topk_rows_kernel, dim3(rows), dim3(SEL_THREADS), 0, (hipStream_t)stream, x, V, ldx, k, extra_col, out);
    MTN_CHECK_LAUNCH();
    return MTN_OK;
}

// ---------------------------------------------------------------- the beam search's hypothesis bookkeeping, one step (data_utils.py:209-240)
// What the reference does on the host after every `model.decode`: for every live hypothesis h (in order) — record the finished
// hypothesis `out_h + <eos>` with score lp_h + logp[eos] + penalty * (len + 1) once the length allows it, then walk h's candidates in
// descending log-probability, skip <unk> / <eos>, fill the new beam, and once it is full replace its WORST member (first minimum) while a
// candidate beats it, stopping at the first that does not.  Here as one tiny launch per step (one workgroup per dialogue, thread 0 walks
// the <= beam x k candidates), reading the rows' heads left by topk_rows_kernel, so that a whole search runs as ONE captured graph with
// no host in the loop: it writes what the next decode step needs (newest tokens, ancestor table, position) and LOGS the step (parents,
// tokens, scores, finished scores) for the host to rebuild the n-best lists from afterwards.  Arithmetic as on the host: scores are float32
// values held as doubles — sums in double, the candidate's score rounded to float32 (numpy's astype), the finished score kept in double.
// A tie inside a row's head raises `flags[0]`: the visiting order would then depend on the selection algorithm, and the caller re-runs the
// search on the host path, which takes the reference's order from the full row.
struct BeamArgs { mtn_beam_args a; };
__global__ __launch_bounds__(64) void beam_advance_kernel(const BeamArgs BA) {
    const mtn_beam_args& A = BA.a;
    __shared__ int s_anc[16 * 64];                             // the parents' ancestor rows (<= 16 hypotheses x <= 1024 positions in chunks of 64)
    __shared__ int s_parent[16], s_n;
    const int d = blockIdx.x, tid = threadIdx.x, Wd = A.width, base = d * Wd, k1 = A.k_top, cols = 2 * k1 + 1;
    const int l = A.step[d];                                     // tokens generated so far = index of this step
    if (tid == 0) {
        const int n = A.n_live[d];
        int np[16], nt[16]; double ns[16];
        int cnt = 0, argmin = 0, tie = 0;
        A.log_n_old[l * gridDim.x + d] = n;
        for (int h = 0; h < n; ++h) {
            const float* row = A.top + (size_t)(base + h) * cols;
            const double lp = A.lp[base + h];
            for (int i = 0; i + 1 < k1; ++i) tie |= row[i] == row[i + 1];
            if (l >= A.min_len) A.log_done[(size_t)l * gridDim.x * Wd + base + h] = (double)(float)((double)row[2 * k1] + lp) + A.penalty * (double)(l + 1);
            for (int i = 0; i < A.k; ++i) {
                const int o = (int)row[k1 + i];
                if (o == A.unk || o == A.eos) continue;
                const double sc = (double)(float)((double)row[i] + lp);
                if (cnt == A.beam) {
                    if (ns[argmin] < sc) {
                        np[argmin] = h; nt[argmin] = o; ns[argmin] = sc;
                        argmin = 0;
                        for (int q = 1; q < cnt; ++q) if (ns[q] < ns[argmin]) argmin = q;
                    } else break;
                } else {
                    np[cnt] = h; nt[cnt] = o; ns[cnt] = sc; ++cnt;
                    if (cnt == A.beam) { argmin = 0; for (int q = 1; q < cnt; ++q) if (ns[q] < ns[argmin]) argmin = q; }
                }
            }
        }
        for (int i = 0; i < Wd; ++i) {
            const size_t at = (size_t)l * gridDim.x * Wd + base + i;
            A.tokens[base + i] = i < cnt ? (long)nt[i] : (long)A.pad;
            if (i < cnt) { A.lp[base + i] = ns[i]; A.log_parent[at] = np[i]; A.log_tok[at] = nt[i]; A.log_score[at] = ns[i]; s_parent[i] = np[i]; }
        }
        A.log_n_new[l * gridDim.x + d] = cnt;
        A.n_live[d] = cnt;
        A.step[d] = l + 1;
        if (tie) A.flags[0] = 1;
        if (d == 0) *A.pos = l + 1;                             // (read by the NEXT decode step only)
        s_n = cnt;
    }
    __syncthreads();
    // ancestor table: row i takes its parent's slots for positions 0..l and its own slot for position l + 1 (all reads before any write)
    const int cnt = s_n, npos = l + 1;
    for (int c0 = 0; c0 < npos; c0 += 64) {
        const int t = c0 + tid;
        for (int i = 0; i < cnt; ++i) if (t < npos) s_anc[i * 64 + tid] = A.anc[(size_t)(base + s_parent[i]) * A.L + t];
        __syncthreads();
        for (int i = 0; i < cnt; ++i) if (t < npos) A.anc[(size_t)(base + i) * A.L + t] = s_anc[i * 64 + tid];
        __syncthreads();
    }
    if (tid < cnt && npos < A.L) A.anc[(size_t)(base + tid) * A.L + npos] = base + tid;
}

extern "C" int mtn_beam_advance(const mtn_beam_args* a, void* stream) {
    MTN_CHECK_ARG(a && a->top && a->tokens && a->pos && a->anc && a->lp && a->n_live && a->step && a->flags, "null buffer");
    MTN_CHECK_ARG(a->log_parent && a->log_tok && a->log_score && a->log_done && a->log_n_old && a->log_n_new, "null log buffer");
    MTN_CHECK_ARG(a->dialogues >= 1 && a->width >= 1 && a->width <= 16 && a->beam >= 1 && a->beam <= a->width, "1 <= beam <= width <= 16");
    MTN_CHECK_ARG(a->k >= 1 && a->k <= a->k_top && a->k_top <= SEL_MAX_K && a->L >= 1, "1 <= k <= k_top <= 16");
    BeamArgs BA; BA.a = *a;
    hipLaunchKernelGGL(beam_advance_kernel, dim3(a->dialogues), dim3(64), 0, (hipStream_t)stream, BA);
    MTN_CHECK_LAUNCH();
    return MTN_OK;
}

// ---------------------------------------------------------------- Generator: log-softmax of the logit rows (mtn.py:68-69)
// out[row][c] = x[row][c] - (max + log sum exp(x - max)); one 256-thread workgroup per row, the row read twice (it sits in L2: a
// decode step has beam x dialogues rows of |V| floats).  In place when out == x.
// (x and out may be the same buffer — the generator calls it in place —, so neither pointer is __restrict__: a thread reads the
// elements it later writes, in that order)
__global__ __launch_bounds__(256) void log_softmax_rows_kernel(const float* x, int V, long ldx, float* out, long ldo) {
    __shared__ float red[4];
    const float* xr = x + (size_t)blockIdx.x * ldx;
    float* orow = out + (size_t)blockIdx.x * ldo;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float mx = -INFINITY;
    for (int c = tid; c < V; c += 256) mx = fmaxf(mx, xr[c]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int c = tid; c < V; c += 256) sum += expf(xr[c] - mx);
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    const float lse = mx + logf((red[0] + red[1]) + (red[2] + red[3]));
    for (int c = tid; c < V; c += 256) orow[c] = xr[c] - lse;
}
extern "C" int mtn_log_softmax_rows(const float* x, int rows, int V, long ldx, float* out, long ldo, void* stream) {
    MTN_CHECK_ARG(x && out && rows > 0 && V > 0 && ldx >= V && ldo >= V, "bad row matrix");
    hipLaunchKernelGGL(log_softmax_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, V, ldx, out, ldo);
    MTN_CHECK_LAUNCH();
    return MTN_OK;
}
