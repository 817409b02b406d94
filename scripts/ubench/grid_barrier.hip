// What a grid-wide barrier INSIDE one persistent kernel costs on this chip, against the ~3.2 us a kernel boundary costs
// on a dependent chain (launch_floor.hip): the question behind fusing the one-batch path's nine launches.
//   (a) barrier only: one arrive (agent-scope fetch_add) per workgroup, everybody polls the counter
//   (b) + hand-over: every workgroup stores `bytes` of plain data before the barrier (release fence) and reads ANOTHER
//       workgroup's data after it (acquire fence), values checked
//   (c) as (b) with the data written / read by write-through / L2-bypassing accesses instead of fences around plain ones
//   hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier && ./grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr unsigned kSpinLimit = 1u << 22;

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, bool fences) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        if (fences) __atomic_thread_fence(__ATOMIC_RELEASE);   // agent scope by default in HIP: buffer_wbl2 + waits
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > kSpinLimit) { ok = false; break; }
        }
        if (fences) __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
    return ok;
}

// mode 0: barrier only; 1: plain stores / loads + fences; 2: agent-scope relaxed atomic stores / loads (write-through), no fences
__global__ void __launch_bounds__(1024) k_persist(unsigned* counter, unsigned base, int iters, int mode, float* data,
                                                  int floats_per_wg, unsigned* errors) {
    const int nb = gridDim.x, b = blockIdx.x;
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        float* mine = data + ((size_t)(it & 1) * nb + b) * floats_per_wg;
        const float* theirs = data + ((size_t)(it & 1) * nb + (b + 97) % nb) * floats_per_wg;
        const float want = (float)(it + 1);
        if (mode == 1) {
            for (int i = threadIdx.x; i < floats_per_wg; i += blockDim.x) mine[i] = want + (float)(b);
        } else if (mode == 2) {
            for (int i = threadIdx.x; i < floats_per_wg; i += blockDim.x)
                __hip_atomic_store(mine + i, want + (float)(b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!grid_barrier(counter, base + (unsigned)(it + 1) * nb, mode == 1)) { bad |= 0x80000000u; break; }
        if (mode == 1) {
            for (int i = threadIdx.x; i < floats_per_wg; i += blockDim.x)
                if (theirs[i] != want + (float)((b + 97) % nb)) ++bad;
        } else if (mode == 2) {
            for (int i = threadIdx.x; i < floats_per_wg; i += blockDim.x)
                if (__hip_atomic_load(theirs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want + (float)((b + 97) % nb)) ++bad;
        }
    }
    if (bad) atomicAdd(errors, bad & 0x7fffffffu ? 1u : 0x10000u);
}


// (d) the pattern a fused one-batch kernel would use instead of full barriers: stage s has n[s] producing workgroups, each
// arrives on counter[s] when its (write-through) stores are issued; the workgroups of stage s + 1 -- and only those -- wait for
// counter[s] to reach n[s] * (step + 1).  Work per stage: read 1 KB another producer wrote, write 1 KB.  Steps are chained
// (stage 0 of step i + 1 waits for the last stage of step i), so time / steps / stages = one hand-over incl. one global
// round trip, comparable with launch_floor's "touch" chains.
constexpr int kStages = 9;
struct Chain { int n[kStages]; };
__global__ void __launch_bounds__(1024) k_chain(unsigned* counters /* [kStages][32] */, unsigned base_steps, int steps, Chain c,
                                                float* data, int sleep, unsigned* errors) {
    const int b = blockIdx.x;
    unsigned bad = 0;
    __shared__ int ok_s;
    for (int it = 0; it < steps; ++it) {
        for (int s = 0; s < kStages; ++s) {
            if (b >= c.n[s]) continue;
            const int ps = (s + kStages - 1) % kStages;
            const unsigned target = (unsigned)c.n[ps] * (base_steps + it + (s == 0 ? 0 : 1));
            if (threadIdx.x == 0) {
                unsigned spins = 0; int ok = 1;
                while (__hip_atomic_load(counters + ps * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    if (++spins > kSpinLimit) { ok = 0; break; }
                    if (sleep) __builtin_amdgcn_s_sleep(1);
                }
                ok_s = ok;
            }
            __syncthreads();
            if (!ok_s) { bad = 0x80000000u; goto out; }
            {
                const float want = (float)(base_steps + it) * 16.f + (float)ps;
                const bool first = (base_steps + it == 0 && s == 0);
                const float* src = data + ((size_t)ps * 256 + (b + 5) % c.n[ps]) * 256;
                float* dst = data + ((size_t)s * 256 + b) * 256;
                if (threadIdx.x < 256) {
                    const float v = __hip_atomic_load(src + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const float expect = (s == 0) ? (float)(base_steps + it - 1) * 16.f + (float)ps : want;
                    if (!first && v != expect) ++bad;
                    __hip_atomic_store(dst + threadIdx.x, (float)(base_steps + it) * 16.f + (float)s + (v - expect) * 0.f,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            __syncthreads();    // compiler waits for the stores to be acknowledged (vmcnt) before the barrier? see below
            if (threadIdx.x == 0) {
                __builtin_amdgcn_s_waitcnt(0);
                __hip_atomic_fetch_add(counters + s * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
out:
    if (bad && threadIdx.x < 256) atomicAdd(errors, bad & 0x7fffffffu ? 1u : 0x10000u);
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    unsigned *counter, *errors; float* data;
    CK(hipMalloc(&counter, 256)); CK(hipMalloc(&errors, 4)); CK(hipMalloc(&data, (size_t)2 * 1024 * 4096 * 4));
    CK(hipMemset(counter, 0, 256)); CK(hipMemset(errors, 0, 4)); CK(hipMemset(data, 0, (size_t)2 * 1024 * 4096 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned base = 0;
    const int grids[][2] = {{256, 1024}, {256, 256}, {64, 1024}, {512, 512}};
    const char* names[] = {"barrier only", "plain stores + release / acquire fences", "agent-scope atomic stores / loads, no fences"};
    for (auto& g : grids) {
        for (int mode = 0; mode < 3; ++mode) {
            for (int fl : {256, 4096}) {
                if (mode == 0 && fl != 256) continue;
                double us[2];
                const int its[2] = {50, 1050};
                for (int r = 0; r < 2; ++r) {
                    for (int warm = 0; warm < 2; ++warm) {
                        if (warm == 1) CK(hipEventRecord(e0, s));
                        hipLaunchKernelGGL(k_persist, dim3(g[0]), dim3(g[1]), 0, s, counter, base, its[r], mode, data, fl, errors);
                        base += (unsigned)its[r] * g[0];
                        if (warm == 1) CK(hipEventRecord(e1, s));
                    }
                    CK(hipStreamSynchronize(s));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    us[r] = ms * 1e3;
                }
                unsigned err; CK(hipMemcpy(&err, errors, 4, hipMemcpyDeviceToHost));
                printf("grid %3d x %4d  %-46s %5d B/wg: %.3f us per barrier (kernel of 50: %.1f us)  errors %u\n", g[0], g[1],
                       names[mode], mode ? fl * 4 : 0, (us[1] - us[0]) / 1000.0, us[0], err);
                CK(hipMemset(errors, 0, 4));
            }
        }
    }
    {
        unsigned* counters; CK(hipMalloc(&counters, kStages * 32 * 4)); CK(hipMemset(counters, 0, kStages * 32 * 4));
        const Chain chains[] = {{{47, 48, 44, 16, 256, 256, 94, 185, 188}}, {{47, 48, 44, 16, 156, 96, 47, 185, 94}},
                                {{256, 256, 256, 256, 256, 256, 256, 256, 256}}, {{32, 32, 32, 32, 32, 32, 32, 32, 32}}};
        for (auto& c : chains) for (int sleep = 0; sleep < 2; ++sleep) {
            unsigned base_steps = 0;
            CK(hipMemset(counters, 0, kStages * 32 * 4));
            double us[2]; const int its[2] = {20, 520};
            for (int r = 0; r < 2; ++r) {
                for (int warm = 0; warm < 2; ++warm) {
                    if (warm == 1) CK(hipEventRecord(e0, s));
                    hipLaunchKernelGGL(k_chain, dim3(256), dim3(1024), 0, s, counters, base_steps, its[r], c, data, sleep, errors);
                    base_steps += its[r];
                    if (warm == 1) CK(hipEventRecord(e1, s));
                }
                CK(hipStreamSynchronize(s));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                us[r] = ms * 1e3;
            }
            unsigned err; CK(hipMemcpy(&err, errors, 4, hipMemcpyDeviceToHost)); CK(hipMemset(errors, 0, 4));
            printf("chain of 9 stages, producers {%d %d %d %d %d %d %d %d %d}, sleep %d: %.2f us per step = %.3f us per hand-over  errors %u\n",
                   c.n[0], c.n[1], c.n[2], c.n[3], c.n[4], c.n[5], c.n[6], c.n[7], c.n[8], sleep, (us[1] - us[0]) / 500.0,
                   (us[1] - us[0]) / 500.0 / kStages, err);
        }
    }
    return 0;
}
