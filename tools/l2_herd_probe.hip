// Does the per-XCD L2 merge concurrent misses to the same line?  256 workgroups each read the SAME `bytes` buffer once
// (16 B/lane, coalesced); mode 0: all in the same order at the same time; mode 1: workgroup b starts at a rotated offset.
// Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv`: merged misses -> ~8 x bytes (once per XCD),
// unmerged -> ~256 x bytes.  Also prints the time per launch.
//   hipcc --offload-arch=gfx950 -O3 -o tools/l2_herd_probe.bin tools/l2_herd_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>
__global__ __launch_bounds__(256) void herd_kernel(const uint4* __restrict__ buf, long n16, uint4* sink, int nwg) {
    const long per = 256;                                  // uint4 per workgroup pass
    const long passes = n16 / per;
    long start = 0;
    if (MODE == 1) start = (long)blockIdx.x * passes / nwg;
    if (MODE == 2) start = (long)(blockIdx.x / 8) * passes / (nwg / 8);     // rotate only among workgroups of one XCD (b % 8 fixed)
    uint4 acc = {0, 0, 0, 0};
    for (long p = 0; p < passes; ++p) {
        long q = p + start; if (q >= passes) q -= passes;
        uint4 v = buf[q * per + threadIdx.x];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

typedef __attribute__((address_space(3))) void lds_void_t;
// mode 3: the same shared read, issued as LDS-DMA (buffer_load_dwordx4 ... lds), 4 KiB per workgroup pass
__global__ __launch_bounds__(256) void herd_dma_kernel(const uint4* __restrict__ buf, long n16, uint4* sink, int nwg) {
    __shared__ __attribute__((aligned(16))) unsigned char sm[8 * 4096];
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, (int)(n16 * 16), 0x00020000);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long passes = n16 / 256;
    for (long p = 0; p < passes; ++p) {
        unsigned voff = (unsigned)((p * 256 + wave * 64 + lane) * 16);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)(sm + (p & 7) * 4096 + wave * 1024), 16, voff, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    uint4 v = ((const uint4*)sm)[threadIdx.x];
    if (v.x == 0x12345678u && v.y == 0x9abcdef0u) sink[blockIdx.x * 256 + threadIdx.x] = v;
}

int main(int argc, char** argv) {
    const long bytes = (argc > 1 ? atol(argv[1]) : 1024) * 1024L;
    const int nwg = argc > 2 ? atoi(argv[2]) : 256;
    uint4 *buf, *sink;
    hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes);
    hipMalloc(&sink, nwg * 256 * sizeof(uint4));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            if (mode == 0) hipLaunchKernelGGL(herd_kernel<0>, dim3(nwg), dim3(256), 0, 0, buf, bytes / 16, sink, nwg);
            if (mode == 1) hipLaunchKernelGGL(herd_kernel<1>, dim3(nwg), dim3(256), 0, 0, buf, bytes / 16, sink, nwg);
            if (mode == 2) hipLaunchKernelGGL(herd_kernel<2>, dim3(nwg), dim3(256), 0, 0, buf, bytes / 16, sink, nwg);
            if (mode == 3) hipLaunchKernelGGL(herd_dma_kernel, dim3(nwg), dim3(256), 0, 0, buf, bytes / 16, sink, nwg);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("mode %d rep %d: %.2f us, per-WG pull %.1f GB/s, aggregate %.2f TB/s\n", mode, rep, ms * 1e3, bytes / (ms * 1e-3) / 1e9,
                   (double)bytes * nwg / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
