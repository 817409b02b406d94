// Round-4 microbenchmark (experiment, not product): does v_mfma_f32_32x32x16_bf16 (32 768 flop, nominally 32 cycles) leave
// MORE room for VALU work per flop than v_mfma_f32_16x16x32_bf16 (16 384 flop, 16 cycles; mfma16_valu.hip: ~2 v_fma ride
// along per MFMA, every further one costs its full issue time)?  If the issue-blocking part of an MFMA is a fixed ~8 cycles,
// the 32x32 form hides ~6 VALU instructions per 32 cycles (3 per 16 384 flop); if it is proportional, nothing is gained.
// Each wave runs ITER x [ 39 x ( MFMA ; V x v_fma_f32 ) ] with 3 independent accumulators.
//   hipcc --offload-arch=gfx950 -O3 mfma32_valu.hip -o mfma32_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int V>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x4 a16[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f32x16 a32[3];
    for (int s = 0; s < 3; ++s)
        for (int i = 0; i < 16; ++i) a32[s][i] = 0.f;
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
    const float av = threadIdx.x * 1e-3f, bv = 1.0001f;
    bf16x8 ab, bb;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        ab[i] = (__bf16)(av + i);
        bb[i] = (__bf16)(bv + i);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 13; ++q) {
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                if (KIND == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(a16[s]) : "v"(ab), "v"(bb));
                if (KIND == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a32[s]) : "v"(ab), "v"(bb));
#pragma unroll
                for (int v = 0; v < V; ++v) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[v % 16]) : "v"(av), "v"(bv));
            }
        }
    }
    float r = a16[0][0] + a16[1][1] + a16[2][2] + a32[0][0] + a32[1][5] + a32[2][9];
#pragma unroll
    for (int i = 0; i < 16; ++i) r += x[i];
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
    const double ns = ms * 1e6 / (iters * 39.0) / wg_per_cu;
    printf("%-22s + %2d fma, %d waves/SIMD: %8.3f ms  %6.2f ns per slot per SIMD  = %5.2f ns per 16 384 flop, %4.1f v_fma per 16 384 flop\n",
           name, V, wg_per_cu, ms, ns, KIND == 2 ? ns / 2 : ns, KIND == 2 ? V / 2.0 : (double)V);
    hipFree(out);
}

template <int KIND>
void sweep(const char* name) {
    for (int occ = 1; occ <= 3; occ += 2) {
        run<KIND, 0>(name, occ);
        run<KIND, 2>(name, occ);
        run<KIND, 4>(name, occ);
        run<KIND, 6>(name, occ);
        run<KIND, 8>(name, occ);
        run<KIND, 12>(name, occ);
        run<KIND, 16>(name, occ);
    }
}

int main() {
    sweep<1>("mfma bf16 16x16x32");
    sweep<2>("mfma bf16 32x32x16");
    return 0;
}
