// assemble.hip — device-side batch assembly (data_handler.py:206-274 make_batch + data_utils.py:23-54 Batch).
// The reference pads every ragged field on the host with numpy, uploads it, and derives the masks with several torch
// passes.  Here the whole corpus (token fields as flat int64 buffers, feature frames as flat [frames, F] float buffers, with
// per-item start/length tables) lives in HBM; a batch is described by the ids of its items, and two grouped launches
// produce the padded tensors AND their masks:
//   tokens  : out[b, l] = l < len ? flat[start + l] : pad ; mask[b, l] = out != pad      (data_utils.py:33-37)
//             optional std_mask[b, i, j] = (out[b, j] != pad) & (j <= i)                    (data_utils.py:48-54)
//             optional count of non-pad tokens (ntokens, data_utils.py:45)
//   features: frames every `skip`-th, rows past the end padded with ones, a frame is valid iff any element != 1, and
//             invalid frames are zeroed                                                      (data_utils.py:27-30)
// Integer / copy work: bit-exact against the numpy restatement in oracle/.
#include "common.h"

struct TokGroup { int count; int block_start[MTN_ASSEMBLE_MAX_GROUP + 1]; mtn_assemble_tokens_desc d[MTN_ASSEMBLE_MAX_GROUP]; };
struct FeatGroup { int count; int block_start[MTN_ASSEMBLE_MAX_GROUP + 1]; mtn_assemble_features_desc d[MTN_ASSEMBLE_MAX_GROUP]; };

__global__ __launch_bounds__(256) void assemble_tokens_kernel(const TokGroup grp) {
    int g = 0;
    while (g + 1 < grp.count && (int)blockIdx.x >= grp.block_start[g + 1]) ++g;
    const mtn_assemble_tokens_desc& D = grp.d[g];
    const int b = (int)blockIdx.x - grp.block_start[g];          // one workgroup per sample
    const int item = D.ids ? D.ids[b] : b;
    const long start = D.start[item];
    const int len = D.len[item] < D.L ? D.len[item] : D.L;
    int nonpad = 0;
    for (int l = threadIdx.x; l < D.L; l += 256) {
        const long v = l < len ? D.flat[start + l] : D.pad;
        D.out[(size_t)b * D.L + l] = v;
        const int keep = v != D.pad;
        nonpad += keep;
        if (D.mask) D.mask[(size_t)b * D.L + l] = (uint8_t)keep;
    }
    if (D.n_nonpad) {
        nonpad = (int)wave_sum((float)nonpad);
        if ((threadIdx.x & 63) == 0 && nonpad) atomicAdd((unsigned long long*)D.n_nonpad, (unsigned long long)nonpad);
    }
    if (D.std_mask) {
        __syncthreads();                                          // out[b, :] written by this workgroup
        const int n = D.L * D.L;
        for (int e = threadIdx.x; e < n; e += 256) {
            const int i = e / D.L, j = e % D.L;
            const long v = j < len ? D.flat[start + j] : D.pad;
            D.std_mask[(size_t)b * n + e] = (uint8_t)((v != D.pad) && (j <= i));
        }
    }
}

__global__ __launch_bounds__(256) void assemble_features_kernel(const FeatGroup grp) {
    int g = 0;
    while (g + 1 < grp.count && (int)blockIdx.x >= grp.block_start[g + 1]) ++g;
    const mtn_assemble_features_desc& D = grp.d[g];
    const int lane = threadIdx.x & 63;
    const int row = ((int)blockIdx.x - grp.block_start[g]) * 4 + (threadIdx.x >> 6);   // one wave per (sample, frame)
    if (row >= D.B * D.V) return;
    const int b = row / D.V, v = row % D.V;
    const int item = D.ids ? D.ids[b] : b;
    const int skip = D.skip > 0 ? D.skip : 1;
    const int n_frames = (D.len[item] + skip - 1) / skip;         // frames[::skip]
    float* out = D.out + (size_t)row * D.F;
    const bool real = v < n_frames;
    const float* src = D.flat + ((size_t)D.start[item] + (size_t)v * skip) * D.F;
    int any = 0;
    if (real)
        for (int c = lane; c < D.F; c += 64) any |= (src[c] != 1.0f);
    const bool valid = real && __any(any);
    for (int c = lane; c < D.F; c += 64) out[c] = valid ? src[c] : 0.0f;
    if (lane == 0 && D.mask) D.mask[row] = (uint8_t)valid;
}

extern "C" int mtn_assemble_tokens(int count, const mtn_assemble_tokens_desc* descs, void* stream) {
    MTN_CHECK_ARG(count >= 1 && count <= MTN_ASSEMBLE_MAX_GROUP && descs, "bad group");
    TokGroup grp;
    memset(&grp, 0, sizeof(grp));
    grp.count = count;
    int blocks = 0;
    for (int i = 0; i < count; ++i) {
        const mtn_assemble_tokens_desc& D = descs[i];
        MTN_CHECK_ARG(D.flat && D.start && D.len && D.out && D.B > 0 && D.L > 0, "bad descriptor");
        grp.block_start[i] = blocks;
        blocks += D.B;
        grp.d[i] = D;
    }
    for (int i = count; i <= MTN_ASSEMBLE_MAX_GROUP; ++i) grp.block_start[i] = blocks;
    hipLaunchKernelGGL(assemble_tokens_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grp);
    MTN_CHECK_LAUNCH();
    return MTN_OK;
}

extern "C" int mtn_assemble_features(int count, const mtn_assemble_features_desc* descs, void* stream) {
    MTN_CHECK_ARG(count >= 1 && count <= MTN_ASSEMBLE_MAX_GROUP && descs, "bad group");
    FeatGroup grp;
    memset(&grp, 0, sizeof(grp));
    grp.count = count;
    int blocks = 0;
    for (int i = 0; i < count; ++i) {
        const mtn_assemble_features_desc& D = descs[i];
        MTN_CHECK_ARG(D.flat && D.start && D.len && D.out && D.B > 0 && D.V > 0 && D.F > 0, "bad descriptor");
        grp.block_start[i] = blocks;
        blocks += (D.B * D.V + 3) / 4;
        grp.d[i] = D;
    }
    for (int i = count; i <= MTN_ASSEMBLE_MAX_GROUP; ++i) grp.block_start[i] = blocks;
    hipLaunchKernelGGL(assemble_features_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grp);
    MTN_CHECK_LAUNCH();
    return MTN_OK;
}
