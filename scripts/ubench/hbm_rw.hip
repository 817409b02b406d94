// What HBM sustains for the three traffic shapes of the STFT pair on this chip: write-only (the forward kernel writes 6x what it
// reads), read-mostly (the inverse reads 3x what it writes), copy.  16-byte accesses, 1 KB per wave instruction.
//   hipcc --offload-arch=gfx950 -O3 hbm_rw.hip -o hbm_rw && ./hbm_rw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_fill(float4* out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = float4{1.f, 2.f, 3.f, (float)i};
}
__global__ void k_copy(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = in[i];
}
// reads `r` streams, writes one (r = 3: the inverse STFT's ratio)
__global__ void k_read_r(const float4* __restrict__ in, float4* __restrict__ out, size_t n, int r) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float4 a = in[i];
        for (int j = 1; j < r; ++j) { const float4 b = in[i + (size_t)j * n]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
        out[i] = a;
    }
}
__global__ void k_sum(const float4* __restrict__ in, float* out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { const float4 a = in[i]; s += a.x + a.y + a.z + a.w; }
    if (s == 123.456f) out[0] = s;
}

int main() {
    const size_t bytes = (size_t)256 << 20, n = bytes / 16;
    float4 *a, *b; float* o;
    CK(hipMalloc(&a, bytes * 4)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 4));
    CK(hipMemset(a, 0, bytes * 4)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wg : {1024, 2048, 4096, 16384}) {
        for (int mode = 0; mode < 4; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k_fill, dim3(wg), dim3(256), 0, 0, b, n);
                if (mode == 1) hipLaunchKernelGGL(k_copy, dim3(wg), dim3(256), 0, 0, a, b, n);
                if (mode == 2) hipLaunchKernelGGL(k_read_r, dim3(wg), dim3(256), 0, 0, a, b, n, 3);
                if (mode == 3) hipLaunchKernelGGL(k_sum, dim3(wg), dim3(256), 0, 0, a, o, n * 4);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            const double moved = mode == 0 ? bytes : mode == 1 ? 2.0 * bytes : mode == 2 ? 4.0 * bytes : 4.0 * bytes;
            const char* names[] = {"write only 256 MB", "copy 256 -> 256 MB", "read 768 + write 256 MB", "read only 1024 MB"};
            printf("%5d workgroups  %-26s %.1f us  %.2f TB/s\n", wg, names[mode], best * 1e3, moved / (best * 1e-3) / 1e12);
        }
    }
    return 0;
}
