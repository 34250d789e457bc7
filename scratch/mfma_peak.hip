// Micro-benchmark: what does v_mfma_f32_16x16x4_f32 sustain on gfx950?  Every wavefront issues MFMAs on NACC independent accumulator
// tiles (NACC = 1: a dependent chain) from registers, no memory traffic; W wavefronts per SIMD.  Also: the same with VALU work
// (fused multiply-adds on other registers) interleaved, to see whether the vector pipe and the matrix pipe of one SIMD overlap.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak scratch/mfma_peak.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int NVALU>
__global__ __launch_bounds__(256) void mfma_kernel(float* out, int iters, float seed) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{seed, seed, seed, seed};
    float a = seed + threadIdx.x * 1e-6f, b = seed * 0.5f;
    float v[NVALU > 0 ? NVALU : 1];
#pragma unroll
    for (int i = 0; i < (NVALU > 0 ? NVALU : 1); ++i) v[i] = seed + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NVALU; ++i) v[i] = __builtin_fmaf(v[i], a, b);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < (NVALU > 0 ? NVALU : 1); ++i) s += v[i];
    if (s == 12345.678f) out[0] = s;
}

template <int NACC, int NVALU>
int run(int wgs_per_cu, float* d, int cus, double ghz) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((mfma_kernel<NACC, NVALU>), dim3(cus * wgs_per_cu), dim3(256), 0, 0, d, 10, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((mfma_kernel<NACC, NVALU>), dim3(cus * wgs_per_cu), dim3(256), 0, 0, d, iters, 1.0f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double mfma_per_simd = (double)wgs_per_cu * iters * 4 * NACC;      // one wavefront of each workgroup per SIMD
    const double valu_per_simd = (double)wgs_per_cu * iters * 4 * NVALU;
    const double cyc = ms * 1e-3 * ghz * 1e9;
    printf("accumulators %d  valu/mfma %4.1f  wavefronts/SIMD %d : %8.3f ms  %6.1f cycles per MFMA (at %.2f GHz)  %6.1f TFLOP/s  [valu %.1f cycles each if alone]\n",
           NACC, NACC ? (double)NVALU / NACC : 0.0, wgs_per_cu, ms, cyc / mfma_per_simd, ghz, mfma_per_simd * 1024 * 2048 / (ms * 1e-3) / 1e12,
           valu_per_simd > 0 ? cyc / valu_per_simd : 0.0);
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const double ghz = p.clockRate / 1e6;
    printf("%s: %d CUs, clockRate %.2f GHz\n", p.name, p.multiProcessorCount, ghz);
    float* d;
    CK(hipMalloc(&d, 1024));
    const int cus = p.multiProcessorCount;
    for (int w = 1; w <= 4; w *= 2) {
        run<1, 0>(w, d, cus, ghz); run<2, 0>(w, d, cus, ghz); run<4, 0>(w, d, cus, ghz); run<8, 0>(w, d, cus, ghz);
    }
    for (int w = 1; w <= 2; ++w) {
        run<8, 8>(w, d, cus, ghz); run<8, 16>(w, d, cus, ghz); run<8, 32>(w, d, cus, ghz); run<8, 64>(w, d, cus, ghz);
    }
    return 0;
}
