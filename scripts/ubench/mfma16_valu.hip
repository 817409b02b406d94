// Microbenchmark (experiment, not product) for the next round's decisions on gfx950:
//   * does the 16-bit-input MFMA (v_mfma_f32_16x16x32_bf16 / _f16, 16 384 flop, nominally 16 cycles) overlap with
//     VALU work of the same / other waves on a SIMD, unlike the fp32 one (mfma_valu.hip)?
//   * what do v_permlane32_swap / v_permlane16_swap cost (a radix-4 FFT pass across the four 16-lane rows of a wave
//     instead of an LDS round trip)?
// Each wave runs ITER x [ 39 x ( MFMA ; V x v_fma_f32 ) ].   Build: hipcc --offload-arch=gfx950 -O3 mfma16_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// KIND 0: no matrix op, 1: bf16 MFMA, 2: f16 MFMA, 3: fp32 MFMA 16x16x4, 4: permlane32_swap, 5: permlane16_swap
template <int KIND, int V>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x4 acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float x[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) x[i] = threadIdx.x * 0.001f + i;
    const float av = threadIdx.x * 1e-3f, bv = 1.0001f;
    bf16x8 ab, bb;
    f16x8 ah, bh;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        ab[i] = (__bf16)(av + i);
        bb[i] = (__bf16)(bv + i);
        ah[i] = (_Float16)(av + i);
        bh[i] = (_Float16)(bv + i);
    }
    unsigned p0 = threadIdx.x, p1 = threadIdx.x * 3u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 13; ++q) {
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                // inline asm: the accumulators are only read after the loop, so no hazard nops sit between the slots (an
                // empty asm "use" of the builtin's result makes the compiler wait out every MFMA: 25 instead of 13.6 ns)
                if (KIND == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[s]) : "v"(ab), "v"(bb));
                if (KIND == 2) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[s]) : "v"(ah), "v"(bh));
                if (KIND == 3) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[s]) : "v"(av), "v"(bv));
                if (KIND == 4) {
                    const u32x2 r = __builtin_amdgcn_permlane32_swap(p0, p1, false, false);
                    p0 = r[0];
                    p1 = r[1];
                }
                if (KIND == 5) {
                    const u32x2 r = __builtin_amdgcn_permlane16_swap(p0, p1, false, false);
                    p0 = r[0];
                    p1 = r[1];
                }
                if (KIND >= 4) asm volatile("" : "+v"(p0), "+v"(p1));
#pragma unroll
                for (int v = 0; v < V; ++v) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[v % 12]) : "v"(av), "v"(bv));
            }
        }
    }
    float r = acc[0][0] + acc[1][1] + acc[2][2] + (float)(p0 ^ p1);
#pragma unroll
    for (int i = 0; i < 12; ++i) r += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND, int V>
void run(const char* name, int wg_per_cu) {
    const int ncu = 256, iters = 2000;
    float* out;
    hipMalloc(&out, sizeof(float) * ncu * wg_per_cu * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<KIND, V><<<ncu * wg_per_cu, 256>>>(out, 10);
    hipEventRecord(e0);
    k<KIND, V><<<ncu * wg_per_cu, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-22s + %2d fma, %d waves/SIMD: %8.3f ms  %6.2f ns per slot per wave  (%.2f ns per slot per SIMD)\n", name, V,
           wg_per_cu, ms, ms * 1e6 / (iters * 39.0), ms * 1e6 / (iters * 39.0) / wg_per_cu);
    hipFree(out);
}

template <int KIND>
void sweep(const char* name) {
    for (int occ = 1; occ <= 3; occ += 2) {
        run<KIND, 0>(name, occ);
        run<KIND, 2>(name, occ);
        run<KIND, 4>(name, occ);
        run<KIND, 8>(name, occ);
    }
}

int main(int argc, char**) {
    sweep<0>("no matrix op");
    sweep<3>("mfma f32 16x16x4");
    sweep<1>("mfma bf16 16x16x32");
    sweep<2>("mfma f16 16x16x32");
    if (argc > 1) return 0;
    sweep<4>("permlane32_swap");
    sweep<5>("permlane16_swap");
    return 0;
}
