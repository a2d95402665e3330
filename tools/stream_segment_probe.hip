// stream_segment_probe.hip — does the SHAPE of the optimiser epilogue's tiles cost HBM bandwidth?
// The parameter-gradient + optimiser launch (gemm_tt_dma128_table_kernel) streams p, m, v (read + written) per 128 x 128 output tile:
// 128 rows x 512-byte segments of matrices whose rows are 2 KB (N = 512) or 8 KB (N = 2048) long, four column tiles of a row block
// arriving at different times.  A flat pass (adam_kernel) moves the same bytes as long contiguous runs and reaches 5.4-5.9 TB/s;
// the launch reaches ~4.25 TB/s of traffic.  This probe moves three fp32 streams (read + write, the optimiser's 24 B/param) over a
// [M x N] matrix per "tile" of 16 384 elements in three shapes — 128 rows x 512 B, 64 x 1 KB, 32 x 2 KB — two workgroups of 256
// threads per CU's worth of grid, tiles taken in the table's order (column tiles of a row block adjacent), and as one flat pass.
//   hipcc --offload-arch=gfx950 -O3 tools/stream_segment_probe.hip -o tools/stream_segment_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// tile = TR rows x TC floats of an [M x N] matrix; 256 threads, float4 per thread per access, all loads of a pass in flight first
template <int TR, int TC>
__global__ __launch_bounds__(256, 2) void tile_stream(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, int N, int tiles_n) {
    const int t = blockIdx.x, tm = t / tiles_n, tn = t - tm * tiles_n;
    constexpr int TPR = TC / 4;                    // threads per row
    constexpr int RPP = 256 / TPR;                 // rows per pass
    const int c4 = (threadIdx.x % TPR) * 4, r0 = threadIdx.x / TPR;
    constexpr int PASSES = TR / RPP, U = PASSES < 8 ? PASSES : 8;
    for (int pass = 0; pass < PASSES; pass += U) {
        float4 a[U], b[U], c[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t o = (size_t)(tm * TR + (pass + u) * RPP + r0) * N + tn * TC + c4;
            a[u] = *(const float4*)(p + o); b[u] = *(const float4*)(m + o); c[u] = *(const float4*)(v + o);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t o = (size_t)(tm * TR + (pass + u) * RPP + r0) * N + tn * TC + c4;
            a[u].x += 1.f; b[u].y += 1.f; c[u].z += 1.f;
            *(float4*)(p + o) = a[u]; *(float4*)(m + o) = b[u]; *(float4*)(v + o) = c[u];
        }
    }
}

__global__ __launch_bounds__(256) void flat_stream(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 a = ((float4*)p)[i], b = ((float4*)m)[i], c = ((float4*)v)[i];
        a.x += 1.f; b.y += 1.f; c.z += 1.f;
        ((float4*)p)[i] = a; ((float4*)m)[i] = b; ((float4*)v)[i] = c;
    }
}

template <typename F> static float timeit(F launch, hipStream_t st) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 6; ++r) {
        CK(hipEventRecord(e0, st));
        launch();
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r && ms < best) best = ms;
    }
    CK(hipGetLastError());
    return best;
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int N : {512, 2048}) {
        const int M = (N == 512 ? 163840 : 40960);        // 83.9 M elements = 1 GB per stream: the three streams are far beyond the Infinity Cache
        const size_t n = (size_t)M * N;
        float *p, *m, *v;
        CK(hipMalloc(&p, n * 4)); CK(hipMalloc(&m, n * 4)); CK(hipMalloc(&v, n * 4));
        CK(hipMemset(p, 0, n * 4)); CK(hipMemset(m, 0, n * 4)); CK(hipMemset(v, 0, n * 4));
        const double gb = 24.0 * n / 1e9;
        const float t_flat = timeit([&] { hipLaunchKernelGGL(flat_stream, dim3(2048), dim3(256), 0, st, p, m, v, (long)(n / 4)); }, st);
        const float t128 = timeit([&] { hipLaunchKernelGGL((tile_stream<128, 128>), dim3((M / 128) * (N / 128)), dim3(256), 0, st, p, m, v, N, N / 128); }, st);
        const float t64 = timeit([&] { hipLaunchKernelGGL((tile_stream<64, 256>), dim3((M / 64) * (N / 256)), dim3(256), 0, st, p, m, v, N, N / 256); }, st);
        const float t32 = timeit([&] { hipLaunchKernelGGL((tile_stream<32, 512>), dim3((M / 32) * (N / 512)), dim3(256), 0, st, p, m, v, N, N / 512); }, st);
        printf("rows of %4d floats (%d KB), %.2f GB moved: flat pass %6.3f ms = %5.2f TB/s | 128 x 512 B tiles %6.3f ms = %5.2f TB/s | 64 x 1 KB %6.3f ms = %5.2f TB/s | 32 x 2 KB %6.3f ms = %5.2f TB/s\n",
               N, N * 4 / 1024, gb, t_flat, gb / t_flat, t128, gb / t128, t64, gb / t64, t32, gb / t32);
        CK(hipFree(p)); CK(hipFree(m)); CK(hipFree(v));
    }
    return 0;
}
