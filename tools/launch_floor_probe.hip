// launch_floor_probe.hip — what does ONE dependent kernel of the step's hipGraph cost before it does anything?
// A captured chain of 200 dependent launches of an (almost) empty kernel — every thread writes one flag word so that the launch
// is not elided — replayed 20 times: microseconds per launch as a function of the dynamic LDS size, the workgroup size and the
// grid (the step's kernels: 128-138 KiB of LDS, 256 or 512 threads, 80-1280 workgroups).
//   hipcc --offload-arch=gfx950 -O3 tools/launch_floor_probe.hip -o tools/launch_floor_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

extern __shared__ unsigned char smem[];
__global__ void almost_empty(int* flag, int touch_lds) {
    if (touch_lds) { smem[threadIdx.x] = (unsigned char)threadIdx.x; __syncthreads(); }
    if (threadIdx.x == 0 && blockIdx.x == 0) flag[0] += touch_lds ? smem[1] : 1;
}

static float chain_us(hipStream_t st, int* flag, int grid, int block, int lds, int touch) {
    CK(hipFuncSetAttribute((const void*)almost_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int N = 200;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(almost_empty, dim3(grid), dim3(block), lds, st, flag, touch);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < 20; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return ms * 1e3f / (20.f * N);
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    int* flag; CK(hipMalloc(&flag, 4)); CK(hipMemset(flag, 0, 4));
    printf("us per dependent launch inside a captured chain of 200 (20 replays)\n");
    printf("%-28s %8s %8s %8s %8s %8s %8s\n", "grid x block", "LDS 0", "16 KiB", "64 KiB", "96 KiB", "128 KiB", "160 KiB");
    const int ldss[6] = {0, 16 << 10, 64 << 10, 96 << 10, 128 << 10, 160 << 10};
    for (int block : {256, 512})
        for (int grid : {80, 256, 512, 1280}) {
            char name[64]; snprintf(name, sizeof name, "%d x %d", grid, block);
            printf("%-28s", name);
            for (int l = 0; l < 6; ++l) printf(" %8.2f", chain_us(st, flag, grid, block, ldss[l], 0));
            printf("\n");
        }
    printf("(same, every thread also writes one LDS byte + one barrier)\n");
    for (int block : {256, 512}) {
        char name[64]; snprintf(name, sizeof name, "%d x %d", 256, block);
        printf("%-28s", name);
        for (int l = 0; l < 6; ++l) printf(" %8.2f", l == 0 ? 0.f : chain_us(st, flag, 256, block, ldss[l], 1));
        printf("\n");
    }
    return 0;
}
