// Does a workgroup pull its operand slice faster when the bytes already sit in its XCD's L2?
// 256 workgroups (one per CU, 512 threads) each pull THEIR OWN slice (128 / 192 KiB) by LDS-DMA, everything in flight at once — the
// load phase of the fused kernels and of a K = 512 GEMM tile — and stamp the wall clock (100 MHz) around it.  Three states of the slice:
//   cold      : 1 GiB of other data written since it was last touched (HBM, and beyond the 256 MB MALL)
//   prefetched: a tiny kernel launched just before (workgroup b -> slice b: same XCD, NOT the same CU) read it with plain loads
//   prefetched by ANOTHER XCD: the same, slices rotated by one XCD (the bytes are in a different XCD's L2: MALL / fabric at best)
// and the prefetch kernel itself timed alone (what it costs when nothing hides it).
//   hipcc --offload-arch=gfx950 -O3 -o tools/l2_prefetch_probe.bin tools/l2_prefetch_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void_t;
__device__ __forceinline__ unsigned long long wall() { return __builtin_readcyclecounter() * 0 + wall_clock64(); }

template <int AUX>    // cache-policy bits of the LDS-DMA loads: 0 plain, 1 sc0, 16 sc1, 17 sc0 sc1, 2 nt
__global__ __launch_bounds__(512) void pull_kernel(const unsigned char* __restrict__ buf, int slice_bytes, unsigned long long* stamps, uint4* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned char* src = buf + (size_t)blockIdx.x * slice_bytes;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, slice_bytes, 0x00020000);
    const unsigned long long t0 = wall_clock64();
    const int ninst = slice_bytes / 1024;                  // 1 KiB per wave-instruction
    for (int i = wave; i < ninst; i += 8)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)(sm + i * 1024), 16, (unsigned)(i * 1024 + lane * 16), 0, 0, AUX);
    const unsigned long long t1 = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t2 = wall_clock64();
    if (threadIdx.x == 0) { stamps[blockIdx.x * 4 + 0] = t0; stamps[blockIdx.x * 4 + 1] = t1; stamps[blockIdx.x * 4 + 2] = t2; }
    uint4 v = ((const uint4*)sm)[threadIdx.x];
    if (v.x == 0x12345678u && v.y == 0x9abcdef0u) sink[blockIdx.x * 512 + threadIdx.x] = v;
}
// workgroup b touches slice (b + rot) % nwg: one 16-byte load per 128-byte line is enough to bring the line into the L2
__global__ __launch_bounds__(256) void touch_kernel(const unsigned char* __restrict__ buf, int slice_bytes, int rot, int nwg, uint4* sink) {
    const unsigned char* src = buf + (size_t)(((int)blockIdx.x + rot) % nwg) * slice_bytes;
    uint4 acc = {0, 0, 0, 0};
    for (int o = threadIdx.x * 128; o < slice_bytes; o += 256 * 128) {
        const uint4 v = *(const uint4*)(src + o);
        acc.x ^= v.x; acc.y ^= v.y;
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ void flush_kernel(uint4* p, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = make_uint4((unsigned)i, 1, 2, 3);
}

int main() {
    const int nwg = 256;
    hipStream_t st; hipStreamCreate(&st);
    unsigned char* buf; uint4 *sink, *junk; unsigned long long* stamps;
    const long junk_n = (1L << 30) / 16;
    hipMalloc(&buf, (size_t)nwg * 192 * 1024); hipMemset(buf, 1, (size_t)nwg * 192 * 1024);
    hipMalloc(&sink, nwg * 512 * 16); hipMalloc(&junk, junk_n * 16); hipMalloc(&stamps, nwg * 4 * 8);
    hipFuncSetAttribute((const void*)pull_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)pull_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)pull_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)pull_kernel<17>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)pull_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    std::vector<unsigned long long> h(nwg * 4);
    for (int slice_kb : {128, 192}) {
        const int sb = slice_kb * 1024;
        for (int mode = 0; mode < 4; ++mode) {           // 0 cold, 1 prefetched same XCD, 2 prefetched by the next XCD, 3 run twice (slice in L2 from the previous pull: same CU mostly)
            std::vector<double> issue, land, span;
            float touch_ms = 0;
            for (int rep = 0; rep < 12; ++rep) {
                if (mode == 3) hipLaunchKernelGGL(pull_kernel<0>, dim3(nwg), dim3(512), sb > 160 * 1024 ? 160 * 1024 : sb, st, buf, sb > 160 * 1024 ? 160 * 1024 : sb, stamps, sink);
                else hipLaunchKernelGGL(flush_kernel, dim3(4096), dim3(256), 0, st, junk, junk_n);
                if (mode == 1 || mode == 2) {
                    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                    hipEventRecord(e0, st);
                    hipLaunchKernelGGL(touch_kernel, dim3(nwg), dim3(256), 0, st, buf, sb, mode == 2 ? 1 : 0, nwg, sink);
                    hipEventRecord(e1, st); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1); touch_ms += ms;
                }
                const int lds = sb > 160 * 1024 ? 160 * 1024 : sb;
                hipLaunchKernelGGL(pull_kernel<0>, dim3(nwg), dim3(512), lds, st, buf, lds, stamps, sink);
                hipStreamSynchronize(st);
                hipMemcpy(h.data(), stamps, nwg * 4 * 8, hipMemcpyDeviceToHost);
                if (rep < 2) continue;
                unsigned long long first = ~0ull, last = 0;
                for (int b = 0; b < nwg; ++b) {
                    issue.push_back((h[b * 4 + 1] - h[b * 4 + 0]) / 100.0);
                    land.push_back((h[b * 4 + 2] - h[b * 4 + 0]) / 100.0);
                    first = std::min(first, h[b * 4 + 0]); last = std::max(last, h[b * 4 + 2]);
                }
                span.push_back((last - first) / 100.0);
            }
            std::sort(issue.begin(), issue.end()); std::sort(land.begin(), land.end()); std::sort(span.begin(), span.end());
            const char* nm[] = {"cold (1 GiB written since)", "prefetched on the SAME XCD", "prefetched by the NEXT XCD", "pulled a second time"};
            const int kb = slice_kb > 160 ? 160 : slice_kb;
            printf("slice %3d KiB  %-28s: issue median %5.2f us | landed median %5.2f  p90 %5.2f us = %5.1f GB/s per CU | launch span median %5.2f us", kb, nm[mode],
                   issue[issue.size() / 2], land[land.size() / 2], land[land.size() * 9 / 10], kb * 1.024 / land[land.size() / 2], span[span.size() / 2]);
            if (mode == 1 || mode == 2) printf(" | the touch kernel alone (event pair) %5.2f us", touch_ms / 12 * 1e3);
            printf("\n");
        }
    }
    // cache-policy bits on the pull itself: slices prefetched on the same XCD (L2 hits) and from the MALL (prefetched by the next XCD)
    for (int mode = 1; mode <= 2; ++mode)
        for (int aux : {0, 1, 16, 17, 2}) {
            std::vector<double> land;
            const int sb = 128 * 1024;
            for (int rep = 0; rep < 10; ++rep) {
                hipLaunchKernelGGL(flush_kernel, dim3(4096), dim3(256), 0, st, junk, junk_n);
                hipLaunchKernelGGL(touch_kernel, dim3(nwg), dim3(256), 0, st, buf, sb, mode == 2 ? 1 : 0, nwg, sink);
                if (aux == 0) hipLaunchKernelGGL(pull_kernel<0>, dim3(nwg), dim3(512), sb, st, buf, sb, stamps, sink);
                if (aux == 1) hipLaunchKernelGGL(pull_kernel<1>, dim3(nwg), dim3(512), sb, st, buf, sb, stamps, sink);
                if (aux == 16) hipLaunchKernelGGL(pull_kernel<16>, dim3(nwg), dim3(512), sb, st, buf, sb, stamps, sink);
                if (aux == 17) hipLaunchKernelGGL(pull_kernel<17>, dim3(nwg), dim3(512), sb, st, buf, sb, stamps, sink);
                if (aux == 2) hipLaunchKernelGGL(pull_kernel<2>, dim3(nwg), dim3(512), sb, st, buf, sb, stamps, sink);
                hipStreamSynchronize(st);
                hipMemcpy(h.data(), stamps, nwg * 4 * 8, hipMemcpyDeviceToHost);
                if (rep < 2) continue;
                for (int b = 0; b < nwg; ++b) land.push_back((h[b * 4 + 2] - h[b * 4 + 0]) / 100.0);
            }
            std::sort(land.begin(), land.end());
            printf("policy aux=%2d  128 KiB %s: landed median %5.2f us = %5.1f GB/s per CU\n", aux, mode == 1 ? "prefetched on the SAME XCD" : "prefetched by the NEXT XCD", land[land.size() / 2], 128 * 1.024 / land[land.size() / 2]);
        }
    return 0;
}
