// Does hipGraph run independent branches concurrently?  Two chains of small kernels forked from one capture stream.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(float* p, int iters) {
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    p[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
int main() {
    float *a, *b; hipMalloc(&a, 1 << 22); hipMalloc(&b, 1 << 22);
    hipStream_t s0, s1, s2; hipStreamCreate(&s0); hipStreamCreate(&s1); hipStreamCreate(&s2);
    hipEvent_t fork, j1, j2, t0, t1; hipEventCreate(&fork); hipEventCreate(&j1); hipEventCreate(&j2); hipEventCreate(&t0); hipEventCreate(&t1);
    const int N = 40, IT = 4000;
    for (int mode = 0; mode < 3; ++mode) {   // 0: one chain only, 1: two chains serial on one stream, 2: two chains on two forked streams
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal);
        if (mode == 2) {
            hipEventRecord(fork, s0); hipStreamWaitEvent(s1, fork, 0); hipStreamWaitEvent(s2, fork, 0);
            for (int i = 0; i < N; ++i) { spin<<<80, 256, 0, s1>>>(a, IT); spin<<<80, 256, 0, s2>>>(b, IT); }
            hipEventRecord(j1, s1); hipEventRecord(j2, s2); hipStreamWaitEvent(s0, j1, 0); hipStreamWaitEvent(s0, j2, 0);
        } else {
            for (int i = 0; i < N; ++i) { spin<<<80, 256, 0, s0>>>(a, IT); if (mode == 1) spin<<<80, 256, 0, s0>>>(b, IT); }
        }
        hipStreamEndCapture(s0, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int w = 0; w < 3; ++w) hipGraphLaunch(ge, s0);
        hipStreamSynchronize(s0);
        hipEventRecord(t0, s0);
        for (int w = 0; w < 10; ++w) hipGraphLaunch(ge, s0);
        hipEventRecord(t1, s0); hipEventSynchronize(t1);
        float ms; hipEventElapsedTime(&ms, t0, t1);
        printf("mode %d: %.1f us per graph launch\n", mode, ms * 100);
    }
    // eager two streams
    hipEventRecord(t0, s0);
    hipDeviceSynchronize();
    auto run = [&]() { for (int i = 0; i < N; ++i) { spin<<<80, 256, 0, s1>>>(a, IT); spin<<<80, 256, 0, s2>>>(b, IT); } };
    run(); hipDeviceSynchronize();
    hipEventRecord(t0, s1); run(); hipEventRecord(t1, s1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, t0, t1); printf("eager 2 streams: %.1f us (stream-1 span)\n", ms * 1000);
    return 0;
}
