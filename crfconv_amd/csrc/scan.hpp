// Exclusive scan of int32 in three small launches (local scan + block totals | scan of the totals by ONE block | add),
// shared by the reverse-CSR builder (graph.hip) and the kNN grid build (knn.hip).  No scratch memory, no library call:
// the rocPRIM scans / sorts these replaced are not safe to replay from a captured hipGraph on ROCm 7.2 (graph.hip).
#pragma once
#include "common.hpp"

namespace crf {

// exclusive scan of int32, 1024 elements per 256-thread block: local scan + block totals | scan of the totals | add
constexpr int SCAN_EPB = 1024;
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void scan_local_kernel(const int32_t* __restrict__ in, int32_t* __restrict__ out, int64_t n,
                                                         int32_t* __restrict__ block_sum) {
    __shared__ int32_t s_w[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_EPB + 4 * threadIdx.x;
    int32_t v[4], t = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = base + k < n ? in[base + k] : 0; t += v[k]; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t inc = t;                                   // inclusive scan over the wavefront
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int32_t u = __shfl_up(inc, o, WAVE); if (lane >= o) inc += u; }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    int32_t wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += s_w[w];
    int32_t run = wbase + inc - t;
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (base + k < n) out[base + k] = run; run += v[k]; }
    if (threadIdx.x == 255) block_sum[blockIdx.x] = wbase + inc;
}
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void scan_sums_kernel(int32_t* __restrict__ block_sum, int64_t nb) {   // ONE block, in place
    __shared__ int32_t s_w[4];
    __shared__ int32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t base = 0; base < nb; base += 256) {
        const int64_t i = base + threadIdx.x;
        const int32_t t = i < nb ? block_sum[i] : 0;
        int32_t inc = t;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int32_t u = __shfl_up(inc, o, WAVE); if (lane >= o) inc += u; }
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        int32_t wbase = s_carry;
        for (int w = 0; w < wave; ++w) wbase += s_w[w];
        if (i < nb) block_sum[i] = wbase + inc - t;
        __syncthreads();
        if (threadIdx.x == 255) s_carry = wbase + inc;
        __syncthreads();
    }
}
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void scan_add_kernel(int32_t* __restrict__ out, int64_t n, const int32_t* __restrict__ block_sum) {
    const int64_t base = (int64_t)blockIdx.x * SCAN_EPB + 4 * threadIdx.x;
    const int32_t add = block_sum[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 4; ++k) if (base + k < n) out[base + k] += add;
}


// scan_sums + scan_add in one launch for up to SCAN_FUSED_MAX blocks: every block sums the totals of the blocks before it
// itself (integer sums: any order gives the same result) -- one launch less per scan, and the scans here are a few thousand
// to a few hundred thousand elements, i.e. pure launch latency.
constexpr int SCAN_FUSED_MAX = 8192;
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void scan_add_fused_kernel(int32_t* __restrict__ out, int64_t n, const int32_t* __restrict__ block_sum) {
    __shared__ int32_t s_w[4];
    int32_t t = 0;
    for (int i = threadIdx.x; i < (int)blockIdx.x; i += 256) t += block_sum[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, WAVE);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = t;
    __syncthreads();
    const int32_t add = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    if (blockIdx.x == 0) return;
    const int64_t base = (int64_t)blockIdx.x * SCAN_EPB + 4 * threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (base + k < n) out[base + k] += add;
}

// out[i] = sum_{k < i} in[k], i in [0, n).  block_sums: cdiv(n, SCAN_EPB) int32 of scratch.  in != out.
inline size_t scan_block_sums(int64_t n) { return (size_t)cdiv(n, SCAN_EPB); }
inline void exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, int32_t* block_sums, hipStream_t st) {
    const int64_t nb = cdiv(n, SCAN_EPB);
    hipLaunchKernelGGL(scan_local_kernel<0>, dim3((unsigned)nb), dim3(256), 0, st, in, out, n, block_sums);
    if (nb == 1) return;                               // one block: its local scan is the scan
    if (nb <= SCAN_FUSED_MAX) {
        hipLaunchKernelGGL(scan_add_fused_kernel<0>, dim3((unsigned)nb), dim3(256), 0, st, out, n, block_sums);
        return;
    }
    hipLaunchKernelGGL(scan_sums_kernel<0>, dim3(1), dim3(256), 0, st, block_sums, nb);
    hipLaunchKernelGGL(scan_add_kernel<0>, dim3((unsigned)nb), dim3(256), 0, st, out, n, block_sums);
}

}  // namespace crf
