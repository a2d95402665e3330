// Probe the semantics of ds_read_b64_tr_b16 on gfx950: LDS holds u16 values = their own index; every lane passes an address.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned* out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x;
    unsigned addr;
    if (mode == 0) addr = (unsigned)(size_t)lds + lane * 8;                       // lane-linear 8-byte chunks
    else addr = (unsigned)(size_t)lds + (lane & 15) * 64 + (lane >> 4) * 8;       // row = lane&15 (stride 64 B = 32 elems), chunk = lane>>4
    unsigned long long v;
    unsigned a, b;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
    a = (unsigned)v; b = (unsigned)(v >> 32);
    out[lane * 4 + 0] = a & 0xffff; out[lane * 4 + 1] = a >> 16; out[lane * 4 + 2] = b & 0xffff; out[lane * 4 + 3] = b >> 16;
}
int main() {
    unsigned* d; (void)hipMalloc(&d, 64 * 4 * 4);
    unsigned h[256];
    for (int mode = 0; mode < 2; ++mode) {
        probe<<<1, 64>>>(d, mode); (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4u %4u %4u %4u%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l % 4 == 3) ? "\n" : "   ");
    }
    return 0;
}
