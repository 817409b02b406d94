// global_load_lds_dwordx4 on gfx950: does lane l's 16 bytes land at M0-base + 16 l, is a partial (exec-masked) issue safe, and
// is "s_waitcnt vmcnt(0)" what makes the data visible to the wave's own ds_reads?  (the staging scheme of istft_seq4_kernel)
//   hipcc --offload-arch=gfx950 -O3 lds_async.hip -o lds_async && ./lds_async
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k(const float4* __restrict__ g, float* out, int n16) {
    __shared__ float4 s[4 * 80];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4* mine = s + wave * 80;
    const float4* src = g + (size_t)(blockIdx.x * 4 + wave) * n16;
    for (int i = lane; i < 80; i += 64) mine[i] = float4{-1.f, -1.f, -1.f, -1.f};
    // n16 = 65: one full wave load + one chunk by lane 0 alone
    __builtin_amdgcn_global_load_lds(src + lane, mine, 16, 0, 0);
    if (lane < n16 - 64) __builtin_amdgcn_global_load_lds(src + 64 + lane, mine + 64, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float acc = 0.f;
    for (int i = 0; i < 80; ++i) {
        const float4 v = mine[(i + lane) % 80];     // every lane reads everything
        acc += v.x + v.y * 0.f;
        if (lane == 0) out[(size_t)(blockIdx.x * 4 + wave) * 80 + i] = mine[i].x;
    }
    if (acc == 12345.f) out[0] = acc;
}

int main() {
    const int n16 = 65, waves = 4 * 512;
    float4* g; float* out;
    CK(hipMalloc(&g, (size_t)waves * n16 * 16)); CK(hipMalloc(&out, (size_t)waves * 80 * 4));
    float4* h = (float4*)malloc((size_t)waves * n16 * 16);
    for (size_t i = 0; i < (size_t)waves * n16; ++i) h[i] = float4{(float)i, 0.f, 0.f, 0.f};
    CK(hipMemcpy(g, h, (size_t)waves * n16 * 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, g, out, n16);
    CK(hipDeviceSynchronize());
    float* o = (float*)malloc((size_t)waves * 80 * 4);
    CK(hipMemcpy(o, out, (size_t)waves * 80 * 4, hipMemcpyDeviceToHost));
    long bad = 0;
    for (int w = 0; w < waves; ++w)
        for (int i = 0; i < 80; ++i) {
            const float want = i < n16 ? (float)((size_t)w * n16 + i) : -1.f;
            if (o[(size_t)w * 80 + i] != want) { if (bad < 5) printf("wave %d slot %d: %g want %g\n", w, i, o[(size_t)w * 80 + i], want); ++bad; }
        }
    printf("global_load_lds_dwordx4: %ld mismatches of %d\n", bad, waves * 80);
    return bad != 0;
}
