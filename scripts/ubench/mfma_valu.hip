// Microbenchmark (experiment, not product): do fp32 MFMAs and VALU ops of the same wave / of other waves
// on the same SIMD overlap on gfx950?  Each wave runs ITER x [ NM x ( MFMA 16x16x4 f32 ; V x v_fma_f32 ) ].
// Build: hipcc --offload-arch=gfx950 -O3 mfma_valu.hip -o mfma_valu ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V, bool MF, bool RCP>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0;
    float x[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) x[i] = threadIdx.x * 0.001f + i;
    float av = threadIdx.x * 1e-3f, bv = 1.0001f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 13; ++q) {
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                if (MF) {
                    f32x4& acc = s == 0 ? a0 : (s == 1 ? a1 : a2);
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(av), "v"(bv));
                }
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    if (RCP && v == 0)
                        asm volatile("v_rcp_f32 %0, %0" : "+v"(x[0]));
                    else
                        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[v % 12]) : "v"(av), "v"(bv));
                }
            }
        }
    }
    float r = a0[0] + a1[1] + a2[2];
#pragma unroll
    for (int i = 0; i < 12; ++i) r += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int V, bool MF, bool RCP>
void run(const char* name, int wg_per_cu, int waves) {
    int ncu = 256, iters = 2000;
    float* out;
    hipMalloc(&out, sizeof(float) * ncu * wg_per_cu * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<V, MF, RCP><<<ncu * wg_per_cu, waves * 64>>>(out, 10);
    hipEventRecord(e0);
    k<V, MF, RCP><<<ncu * wg_per_cu, waves * 64>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per wave: iters * 39 slots
    double ns_slot = ms * 1e6 / (iters * 39.0);
    printf("%-28s wg/cu %d waves/wg %d : %.3f ms  %.1f ns per (MFMA+%dV) slot per wave\n", name, wg_per_cu, waves, ms, ns_slot, V);
    hipFree(out);
}

int main() {
    // waves/wg = 4 -> one wave per SIMD per workgroup; wg_per_cu = waves per SIMD
    for (int occ = 1; occ <= 3; ++occ) {
        run<0, true, false>("mfma only", occ, 4);
        run<4, true, false>("mfma + 4 fma", occ, 4);
        run<8, true, false>("mfma + 8 fma", occ, 4);
        run<12, true, false>("mfma + 12 fma", occ, 4);
        run<4, false, false>("4 fma only", occ, 4);
        run<8, false, false>("8 fma only", occ, 4);
        run<12, false, false>("12 fma only", occ, 4);
        run<4, true, true>("mfma + rcp + 3 fma", occ, 4);
        run<4, false, true>("rcp + 3 fma only", occ, 4);
    }
    return 0;
}
