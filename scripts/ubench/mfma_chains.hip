// Microbenchmark (experiment): fp32 MFMA 16x16x4 issue rate vs number of independent accumulator chains and
// waves per SIMD, over a run long enough (tens of ms) for the clocks to settle.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CH>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x4 acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = f32x4{0, 0, 0, 0};
    float av = threadIdx.x * 1e-3f, bv = 1.0001f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 24 / CH; ++q)
#pragma unroll
            for (int c = 0; c < CH; ++c)
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(av), "v"(bv));
    }
    float r = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) r += acc[c][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int CH>
void run() {
    for (int occ = 1; occ <= 3; ++occ) {
        int ncu = 256, iters = 40000;
        float* out;
        (void)hipMalloc(&out, sizeof(float) * ncu * occ * 256);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        k<CH><<<ncu * occ, 256>>>(out, 100);
        (void)hipEventRecord(e0);
        k<CH><<<ncu * occ, 256>>>(out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double n = (double)iters * 24 * occ;  // MFMAs per SIMD
        printf("chains %d waves/SIMD %d : %.2f ms, %.2f ns per MFMA per SIMD, %.1f TFLOP/s\n", CH, occ, ms, ms * 1e6 / n,
               n * 1024 * 2048.0 / (ms * 1e-3) / 1e12);
        (void)hipFree(out);
    }
}
int main() { run<1>(); run<2>(); run<3>(); run<4>(); run<6>(); return 0; }
