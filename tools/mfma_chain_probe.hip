// mfma_chain_probe.hip — MFMA issue rate of ONE 512-thread workgroup per CU (2 waves per SIMD: the residency of the kernels that
// hold a 128 KiB LDS image), as a function of the number of independent accumulator chains per wave and of how often the waves
// meet at a barrier.  v_mfma_f32_16x16x32_bf16 from registers, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_chain_probe.hip -o tools/mfma_chain_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int CH, int PER_BARRIER>
__global__ __launch_bounds__(512, 2) void spin(float* out, int iters) {
    bf16x8_t a[4], b[CH];
    for (int k = 0; k < 4; ++k) for (int i = 0; i < 8; ++i) a[k][i] = (__bf16)(0.001f * (threadIdx.x + i + k));
    for (int k = 0; k < CH; ++k) for (int i = 0; i < 8; ++i) b[k][i] = (__bf16)(0.002f * (threadIdx.x - i + k));
    f32x4_t c[CH];
    for (int k = 0; k < CH; ++k) c[k] = f32x4_t{0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int s = 0; s < PER_BARRIER / CH; ++s)
#pragma unroll
            for (int k = 0; k < CH; ++k) c[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s & 3], b[k], c[k], 0, 0, 0);
        if (PER_BARRIER < 100000) __builtin_amdgcn_s_barrier();
    }
    f32x4_t s = c[0];
    for (int k = 1; k < CH; ++k) s += c[k];
    if (s[0] == 123.456f) out[blockIdx.x * 512 + threadIdx.x] = s[1];
}

template <int CH, int PB> static void run(float* out, const char* what) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int total = 64 * 2000, iters = total / PB;
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((spin<CH, PB>), dim3(256), dim3(512), 0, 0, out, iters);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double n = (double)iters * PB;                      // MFMAs per wave
    printf("%-44s %7.1f TFLOP/s   %5.1f cycles per MFMA per SIMD at 2.4 GHz\n", what, 256.0 * 8 * n * 16384 / (best * 1e-3) / 1e12,
           best * 1e-3 * 2.4e9 / (n * 2));
}

int main() {
    float* out; if (hipMalloc(&out, 256 * 512 * 4) != hipSuccess) return 1;
    printf("one 512-thread workgroup per CU (2 waves per SIMD), v_mfma_f32_16x16x32_bf16 from registers\n");
    run<4, 64>(out, "4 chains, barrier every 64 MFMAs");
    run<4, 100032>(out, "4 chains, no barrier");
    run<8, 64>(out, "8 chains, barrier every 64 MFMAs");
    run<8, 100032>(out, "8 chains, no barrier");
    run<2, 64>(out, "2 chains, barrier every 64 MFMAs");
    run<16, 64>(out, "16 chains, barrier every 64 MFMAs");
    return 0;
}
