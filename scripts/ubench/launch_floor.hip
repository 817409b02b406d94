// What a chain of K dependent small kernels costs on this chip, replayed as one hipGraph on one stream: the floor under
// the one-batch path (8 launches).  Kernels: (a) empty, 1 workgroup; (b) `wg` workgroups of 256 threads that each load
// 4 KB, add, and store 4 KB (one global round trip); (c) the same with 1024-thread workgroups.
//   hipcc --offload-arch=gfx950 -O3 launch_floor.hip -o launch_floor && ./launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>

__global__ void k_empty() {}
__global__ void k_touch(const float4* __restrict__ in, float4* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float4 v = in[i];
    v.x += 1.f;
    out[i] = v;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static double run(hipStream_t s, int K, int mode, int wg, int threads, float4* a, float4* b, int reps) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    for (int k = 0; k < K; ++k) {
        if (mode == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s);
        else hipLaunchKernelGGL(k_touch, dim3(wg), dim3(threads), 0, s, (k & 1) ? b : a, (k & 1) ? a : b);
    }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return us;
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    float4 *a, *b; const size_t n = (size_t)1024 * 1024;
    CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMemset(a, 0, n * 16)); CK(hipMemset(b, 0, n * 16));
    const int Ks[] = {1, 2, 4, 8, 16};
    printf("us per graph replay (back-to-back replays on one stream), K dependent kernels per graph\n");
    for (int K : Ks) {
        printf("K=%2d  empty %.2f | touch 192x256 %.2f | touch 768x256 %.2f | touch 188x1024 %.2f | touch 736x1024 %.2f\n", K,
               run(s, K, 0, 1, 64, a, b, 2000), run(s, K, 1, 192, 256, a, b, 2000), run(s, K, 1, 768, 256, a, b, 2000),
               run(s, K, 1, 188, 1024, a, b, 2000), run(s, K, 1, 736, 1024, a, b, 2000));
    }
    return 0;
}
