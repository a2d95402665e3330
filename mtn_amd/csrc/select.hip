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
