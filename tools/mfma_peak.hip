// On-box achievable MFMA peak (SURVEY §8d asks for a measured denominator next to the 2.5 PFLOP/s spec figure):
// every wave issues independent v_mfma_f32_16x16x32_bf16 (4 accumulator chains) from registers, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_peak.bin tools/mfma_peak.hip && tools/mfma_peak.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__global__ __launch_bounds__(256) void mfma_spin(float* out, int iters) {
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
    f32x4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4, 0, 0, 0);
        c5 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c5, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c6, 0, 0, 0);
        c7 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c7, 0, 0, 0);
    }
    f32x4_t s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    if (s[0] == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = s[1];
}

int main() {
    float* out; if (hipMalloc(&out, 4096 * 256 * 4) != hipSuccess) return 1;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000, wgs = 256 * 8;                 // 8 workgroups x 4 waves per CU
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mfma_spin, dim3(wgs), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)wgs * 4 * iters * 8 * (2.0 * 16 * 16 * 32);
        printf("mfma_f32_16x16x32_bf16 spin: %.3f ms  %.1f TFLOP/s\n", ms, flops / (ms * 1e-3) / 1e12);
    }
    return 0;
}
