// The premise of XCD-affine row ownership: do bytes WRITTEN by a producer kernel stay in the writing XCD's L2 across the
// kernel boundary, so that a consumer workgroup of the NEXT kernel placed on the same XCD pulls them at the own-L2 rate (94 GB/s per CU in
// l2_prefetch_probe, read-only lines) instead of the fabric / MALL rate (28 GB/s)?
// 256 workgroups (one per CU, 512 threads) each pull THEIR OWN slice by LDS-DMA (or by plain 16-byte loads: the fused kernels' x rows), everything in
// flight at once, and stamp the 100 MHz wall clock around it.  States of the slice before the pull:
//   cold                       : 1 GiB of other data written since it was last touched
//   written by the SAME XCD    : a producer kernel launched just before, workgroup b writes slice b (plain 16-byte stores, whole lines) — same XCD, NOT the same CU (rot 8)
//   written by the NEXT XCD    : the same, slices rotated by one workgroup (= one XCD)
//   written, then weights pass : as "same XCD", but a second kernel streams 1.5 MB of other read-only data through every XCD in between (what the real chain does)
//   written sc1 (write-through): the producer's stores carry sc1 (the guide: such lines are DROPPED from the L2)
// Slices 16 … 128 KiB per workgroup (4 … 32 MiB in all: the chain's activations are 0.2-8 MB per launch).
//   hipcc --offload-arch=gfx950 -O3 -o tools/l2_written_probe.bin tools/l2_written_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <bool PLAIN>    // PLAIN: 16-byte global loads into registers (then to LDS) instead of LDS-DMA
__global__ __launch_bounds__(512) void pull_kernel(const unsigned char* __restrict__ buf, int slice_bytes, unsigned long long* stamps, uint4* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned char* src = buf + (size_t)blockIdx.x * slice_bytes;
    const unsigned long long t0 = wall_clock64();
    unsigned long long t1;
    if (PLAIN) {
        uint4 acc = {0, 0, 0, 0};
        const int n16 = slice_bytes / 16;
        uint4 v[8];
        for (int base = 0; base < n16; base += 512 * 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int i = base + j * 512 + threadIdx.x; v[j] = i < n16 ? ((const uint4*)src)[i] : make_uint4(0, 0, 0, 0); }
#pragma unroll
            for (int j = 0; j < 8; ++j) { acc.x ^= v[j].x; acc.y ^= v[j].y; acc.z ^= v[j].z; acc.w ^= v[j].w; }
        }
        t1 = wall_clock64();
        ((uint4*)sm)[threadIdx.x] = acc;
    } else {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, slice_bytes, 0x00020000);
        const int ninst = slice_bytes / 1024;                  // 1 KiB per wave-instruction
        for (int i = wave; i < ninst; i += 8)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)(sm + i * 1024), 16, (unsigned)(i * 1024 + lane * 16), 0, 0, 0);
        t1 = wall_clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const unsigned long long t2 = wall_clock64();
    if (threadIdx.x == 0) { stamps[blockIdx.x * 4 + 0] = t0; stamps[blockIdx.x * 4 + 1] = t1; stamps[blockIdx.x * 4 + 2] = t2; }
    uint4 v = ((const uint4*)sm)[threadIdx.x];
    if (v.x == 0x12345678u && v.y == 0x9abcdef0u) sink[blockIdx.x * 512 + threadIdx.x] = v;
}
// workgroup b WRITES slice (b + rot) % nwg with whole-line 16-byte stores
// PV: 0 plain stores, 1 sc1 (write-through) stores, 2 plain + an agent-scope RELEASE fence by every wave behind its stores (buffer_wbl2 sc1: the lines are
// written back INSIDE the kernel and, per the guide, stay in the L2 as clean lines), 3 = 2 + the workgroup re-reads its lines afterwards, 4 nt stores
template <int PV>
__global__ __launch_bounds__(256) void write_kernel(unsigned char* __restrict__ buf, int slice_bytes, int rot, int nwg, unsigned seed, uint4* sink) {
    uint4* dst = (uint4*)(buf + (size_t)(((int)blockIdx.x + rot) % nwg) * slice_bytes);
    const int n16 = slice_bytes / 16;
    for (int i = threadIdx.x; i < n16; i += 256) {
        const uint4 v = make_uint4(seed + i, blockIdx.x, 3, 4);
        const v4u vv = {v.x, v.y, v.z, v.w};
        if (PV == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(&dst[i]), "v"(vv) : "memory");
        else if (PV == 4) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(&dst[i]), "v"(vv) : "memory");
        else dst[i] = v;
    }
    if (PV == 2 || PV == 3) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (PV == 3) {
        __syncthreads();
        uint4 acc = {0, 0, 0, 0};
        for (int i = threadIdx.x * 8; i < n16; i += 256 * 8) { v4u r; asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(r) : "v"(&dst[i]) : "memory"); acc.x ^= r.x; }
        if (acc.x == 0x12345678u) sink[blockIdx.x * 256 + threadIdx.x] = acc;
    }
}
// workgroup b touches (READS) slice (b + rot) % nwg: one 16-byte load per 128-byte line brings the line into the L2 (the r04 probe's control)
__global__ __launch_bounds__(256) void touch_kernel(const unsigned char* __restrict__ buf, int slice_bytes, int rot, int nwg, uint4* sink) {
    const unsigned char* src = buf + (size_t)(((int)blockIdx.x + rot) % nwg) * slice_bytes;
    uint4 acc = {0, 0, 0, 0};
    for (int o = threadIdx.x * 128; o < slice_bytes; o += 256 * 128) { const uint4 v = *(const uint4*)(src + o); acc.x ^= v.x; acc.y ^= v.y; }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[blockIdx.x * 256 + threadIdx.x] = acc;
}
// every XCD streams `bytes` of read-only data (the weights of a launch): workgroup b reads bytes [0, bytes) strided over its XCD's 32 workgroups
__global__ __launch_bounds__(256) void weights_kernel(const unsigned char* __restrict__ w, int bytes, uint4* sink) {
    const int local = blockIdx.x / 8;        // 0..31 inside the XCD
    uint4 acc = {0, 0, 0, 0};
    for (int o = (local * 256 + threadIdx.x) * 16; o < bytes; o += 32 * 256 * 16) { const uint4 v = *(const uint4*)(w + o); acc.x ^= v.x; acc.y ^= v.y; }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ void flush_kernel(uint4* p, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = make_uint4((unsigned)i, 1, 2, 3);
}

int main() {
    const int nwg = 256;
    hipStream_t st; hipStreamCreate(&st);
    unsigned char *buf, *wts; uint4 *sink, *junk; unsigned long long* stamps;
    const long junk_n = (1L << 30) / 16;
    hipMalloc(&buf, (size_t)nwg * 128 * 1024); hipMemset(buf, 1, (size_t)nwg * 128 * 1024);
    hipMalloc(&wts, 4 << 20); hipMemset(wts, 2, 4 << 20);
    hipMalloc(&sink, nwg * 512 * 16); hipMalloc(&junk, junk_n * 16); hipMalloc(&stamps, nwg * 4 * 8);
    hipFuncSetAttribute((const void*)pull_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)pull_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    std::vector<unsigned long long> h(nwg * 4);
    const char* nm[] = {"cold (1 GiB written since)", "written by the SAME XCD (other CU)", "written by the NEXT XCD", "written same XCD, then 1.5 MB weights pass",
                        "written sc1 by the SAME XCD", "written by the SAME CU", "READ (touched) by the SAME XCD [control]", "READ (touched) by the NEXT XCD [control]",
                        "written + in-kernel agent release, SAME XCD", "written + release + re-read, SAME XCD", "written nt by the SAME XCD",
                        "written SAME XCD, then a touch kernel reads it", "written + in-kernel agent release, NEXT XCD"};
    for (int plain = 0; plain < 1; ++plain)
        for (int slice_kb : {32, 128}) {
            const int sb = slice_kb * 1024;
            for (int mode = 0; mode < 13; ++mode) {
                std::vector<double> issue, land, span;
                for (int rep = 0; rep < 12; ++rep) {
                    hipLaunchKernelGGL(flush_kernel, dim3(4096), dim3(256), 0, st, junk, junk_n);
                    if (mode == 1 || mode == 3 || mode == 11) hipLaunchKernelGGL(write_kernel<0>, dim3(nwg), dim3(256), 0, st, buf, sb, 8, nwg, (unsigned)rep, sink);
                    if (mode == 2) hipLaunchKernelGGL(write_kernel<0>, dim3(nwg), dim3(256), 0, st, buf, sb, 1, nwg, (unsigned)rep, sink);
                    if (mode == 4) hipLaunchKernelGGL(write_kernel<1>, dim3(nwg), dim3(256), 0, st, buf, sb, 8, nwg, (unsigned)rep, sink);
                    if (mode == 5) hipLaunchKernelGGL(write_kernel<0>, dim3(nwg), dim3(256), 0, st, buf, sb, 0, nwg, (unsigned)rep, sink);
                    if (mode == 6 || mode == 11) hipLaunchKernelGGL(touch_kernel, dim3(nwg), dim3(256), 0, st, buf, sb, 8, nwg, sink);
                    if (mode == 7) hipLaunchKernelGGL(touch_kernel, dim3(nwg), dim3(256), 0, st, buf, sb, 1, nwg, sink);
                    if (mode == 8) hipLaunchKernelGGL(write_kernel<2>, dim3(nwg), dim3(256), 0, st, buf, sb, 8, nwg, (unsigned)rep, sink);
                    if (mode == 9) hipLaunchKernelGGL(write_kernel<3>, dim3(nwg), dim3(256), 0, st, buf, sb, 8, nwg, (unsigned)rep, sink);
                    if (mode == 10) hipLaunchKernelGGL(write_kernel<4>, dim3(nwg), dim3(256), 0, st, buf, sb, 8, nwg, (unsigned)rep, sink);
                    if (mode == 12) hipLaunchKernelGGL(write_kernel<2>, dim3(nwg), dim3(256), 0, st, buf, sb, 1, nwg, (unsigned)rep, sink);
                    if (mode == 3) hipLaunchKernelGGL(weights_kernel, dim3(nwg), dim3(256), 0, st, wts, 1536 * 1024, sink);
                    if (plain) hipLaunchKernelGGL(pull_kernel<true>, dim3(nwg), dim3(512), 8192, st, buf, sb, stamps, sink);
                    else hipLaunchKernelGGL(pull_kernel<false>, dim3(nwg), dim3(512), sb, st, buf, sb, stamps, sink);
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
                printf("%s slice %3d KiB  %-44s: issue median %5.2f us | landed median %5.2f  p90 %5.2f us = %5.1f GB/s per CU | launch span median %5.2f us\n",
                       plain ? "plain loads" : "LDS-DMA    ", slice_kb, nm[mode], issue[issue.size() / 2], land[land.size() / 2], land[land.size() * 9 / 10],
                       slice_kb * 1.024 / land[land.size() / 2], span[span.size() / 2]);
            }
        }
    return 0;
}
