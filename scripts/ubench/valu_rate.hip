// Microbenchmark (experiment): issue rate of scalar vs packed fp32 VALU ops, LDS reads, per SIMD on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float lds[4096];
    float x[8];
    f32x2 y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 0.001f + i; y[i] = f32x2{x[i], x[i] + 1.f}; }
    lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 256] = 1.f;
    __syncthreads();
    float av = threadIdx.x * 1e-3f, bv = 1.0001f;
    f32x2 a2 = {av, av}, b2 = {bv, bv};
    const float* lp = lds + (threadIdx.x & 63) * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[q]) : "v"(av), "v"(bv));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(y[q]) : "v"(a2), "v"(b2));
            if (KIND == 2) asm volatile("v_add_f32 %0, %1, %0" : "+v"(x[q]) : "v"(av));
            if (KIND == 3) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(y[q]) : "v"(a2));
            if (KIND == 4) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[q]) : "v"(bv));
            if (KIND == 5) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(y[q]) : "v"(b2));
            if (KIND == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[q]));
            if (KIND == 7) asm volatile("v_max_f32 %0, %1, %0" : "+v"(x[q]) : "v"(av));
            if (KIND == 8) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(x[q]) : "v"(av));
            if (KIND == 9) asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(x[q]) : "v"((int)(size_t)lp));
            if (KIND == 10) asm volatile("v_mov_b32 %0, %1" : "=v"(x[q]) : "v"(av));
            if (KIND == 11) asm volatile("v_add_u32 %0, %1, %0" : "+v"(x[q]) : "v"(av));
            if (KIND == 12) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "+v"(y[q]) : "v"(a2), "v"(b2));
        }
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += x[i] + y[i][0] + y[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND>
void run(const char* name) {
    for (int occ = 1; occ <= 4; occ *= 2) {
        int ncu = 256, iters = 4000;
        float* out;
        (void)hipMalloc(&out, sizeof(float) * ncu * occ * 256);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        k<KIND><<<ncu * occ, 256>>>(out, 10);
        (void)hipEventRecord(e0);
        k<KIND><<<ncu * occ, 256>>>(out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double ns = ms * 1e6 / (iters * 8.0 * occ);  // per instruction per SIMD
        printf("%-14s waves/SIMD %d : %.3f ns per wave-instruction per SIMD\n", name, occ, ns);
        (void)hipFree(out);
    }
}

int main() {
    run<0>("v_fma_f32"); run<1>("v_pk_fma_f32"); run<12>("v_pk_fma opsel"); run<2>("v_add_f32"); run<3>("v_pk_add_f32");
    run<4>("v_mul_f32"); run<5>("v_pk_mul_f32"); run<6>("v_rcp_f32"); run<7>("v_max_f32"); run<8>("v_cndmask");
    run<10>("v_mov_b32"); run<11>("v_add_u32"); run<9>("ds_read_b32+wait");
    return 0;
}
