// Micro-benchmark: cost of a software grid barrier on gfx950 (one atomic per block, agent-scope fences).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

#ifndef SLEEP
#define SLEEP 2
#endif
// arrival counter (one RMW per block) + separate generation flag that waiters poll read-only
__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        unsigned* flag = counter + 32;                       // different 128-byte line
        const unsigned gen = target / gridDim.x;             // 1, 2, ...
        const unsigned prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1 == target) {
            __hip_atomic_store(flag, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned spins = 0;
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) {
                __builtin_amdgcn_s_sleep(SLEEP);
                if (++spins > (1u << 22)) { ok = false; break; }
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
    }
    __syncthreads();
    return ok;
}

__global__ void bar_kernel(unsigned* counter, int nbar, float* buf, int* fail) {
    const unsigned nb = gridDim.x;
    float v = buf[blockIdx.x * blockDim.x + threadIdx.x];
    for (int i = 0; i < nbar; ++i) {
        v = v * 1.0001f + 1.f;
        buf[blockIdx.x * blockDim.x + threadIdx.x] = v;
        if (!grid_barrier(counter, nb * (unsigned)(i + 1))) { if (threadIdx.x == 0) atomicAdd(fail, 1); return; }
        v += buf[((blockIdx.x + 1) % nb) * blockDim.x + threadIdx.x];
    }
    buf[blockIdx.x * blockDim.x + threadIdx.x] = v;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("CUs %d\n", p.multiProcessorCount);
    unsigned* counter; float* buf; int* fail;
    CK(hipMalloc(&counter, 256)); CK(hipMalloc(&fail, 4));
    CK(hipMalloc(&buf, 4096 * 1024 * 4)); CK(hipMemset(buf, 0, 4096 * 1024 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int bpc : {1, 2, 4}) for (int threads : {256, 1024}) {
        if (bpc * threads > 2048) continue;
        const int grid = p.multiProcessorCount * bpc;
        for (int nbar : {0, 1, 2, 10, 50}) {
            float best = 1e9;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipMemsetAsync(counter, 0, 256, 0)); CK(hipMemsetAsync(fail, 0, 4, 0));
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(bar_kernel, dim3(grid), dim3(threads), 0, 0, counter, nbar, buf, fail);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            int f; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
            printf("blocks/CU %d threads %4d grid %4d barriers %2d: %.2f us  (fail %d)\n", bpc, threads, grid, nbar, best * 1e3, f);
        }
    }
    return 0;
}
