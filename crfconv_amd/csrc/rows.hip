// Row plumbing of the training step that is neither a contraction nor a normalisation: column concatenation of two per-point
// feature tensors and its inverse (the fusion layers' torch.cat, models/continuous_crf_conv_big.py:71 and
// models/point_conv_big.py:107, on the levels where the two-pointer Linear does not apply) and the step-counter increment of
// the BatchNorm layers (the batched gradient copy into the all-reduce bucket is crfconv_copy_jobs of graph.hip).  All HBM-bound streaming passes;
// they exist so that a captured training step launches no framework kernel.
#include "common.hpp"

namespace crf {

constexpr int RW_BLOCK = 256;

// out [m, ca + cb] = [xa | xb]   (ca, cb multiples of 4: one float4 per thread)
__global__ __launch_bounds__(RW_BLOCK) void cat2_kernel(const float* __restrict__ xa, const float* __restrict__ xb, int64_t n4,
                                                        int ca4, int cb4, float* __restrict__ out) {
    const int c4 = ca4 + cb4;
    for (int64_t t = (int64_t)blockIdx.x * RW_BLOCK + threadIdx.x; t < n4; t += (int64_t)gridDim.x * RW_BLOCK) {
        const int64_t r = t / c4;
        const int q = (int)(t - r * c4);
        const float4 v = q < ca4 ? *reinterpret_cast<const float4*>(xa + 4 * (r * ca4 + q))
                                 : *reinterpret_cast<const float4*>(xb + 4 * (r * cb4 + (q - ca4)));
        *reinterpret_cast<float4*>(out + 4 * t) = v;
    }
}

// ga [m, ca] = g[:, :ca],  gb [m, cb] = g[:, ca:]
__global__ __launch_bounds__(RW_BLOCK) void split2_kernel(const float* __restrict__ g, int64_t n4, int ca4, int cb4,
                                                          float* __restrict__ ga, float* __restrict__ gb) {
    const int c4 = ca4 + cb4;
    for (int64_t t = (int64_t)blockIdx.x * RW_BLOCK + threadIdx.x; t < n4; t += (int64_t)gridDim.x * RW_BLOCK) {
        const int64_t r = t / c4;
        const int q = (int)(t - r * c4);
        const float4 v = *reinterpret_cast<const float4*>(g + 4 * t);
        if (q < ca4) *reinterpret_cast<float4*>(ga + 4 * (r * ca4 + q)) = v;
        else *reinterpret_cast<float4*>(gb + 4 * (r * cb4 + (q - ca4))) = v;
    }
}

__global__ __launch_bounds__(RW_BLOCK) void add_i64_kernel(long long* __restrict__ x, int n, long long delta) {
    const int t = blockIdx.x * RW_BLOCK + threadIdx.x;
    if (t < n) x[t] += delta;
}

// A BOUNDED device-side gate between two streams (round 5): the training graph MARKS the point from which its launches leave most of
// the chip idle (the coarse levels), the collate graph on the side stream WAITS for the mark before its first kernel -- the overlap of
// two hipGraphs on two streams is otherwise wherever the launches happen to fall.  gate [4] device words: mark | consumed | enabled | timeouts.
// The wait is one wavefront sleeping between polls, for at most max_wait_us (then it goes ahead and counts a timeout): it cannot
// hang, whatever the other stream does.
__global__ void gate_mark_kernel(unsigned long long* __restrict__ gate) {
    __hip_atomic_fetch_add(gate, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void gate_wait_kernel(unsigned long long* __restrict__ gate, unsigned long long max_ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long on = __hip_atomic_load(gate + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (on == 0ull) return;                                                     // disabled: no wait
    const unsigned long long consumed = __hip_atomic_load(gate + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = wall_clock64();
    unsigned long long mark;
    for (;;) {
        mark = __hip_atomic_load(gate, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if (mark > consumed) break;
        if (wall_clock64() - t0 > max_ticks) {                 // the other stream is not coming: go ahead
            __hip_atomic_fetch_add(gate + 3, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // the enabled word counts the timeouts IN A ROW in its upper bits; the third switches the gate off: a marking stream
            // that shares this stream's hardware queue cannot run while this wavefront waits, and every wait would cost max_wait
            const unsigned long long run = (on >> 8) + 1ull;
            __hip_atomic_store(gate + 2, run >= 3ull ? 0ull : (1ull | (run << 8)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        __builtin_amdgcn_s_sleep(32);
    }
    __hip_atomic_store(gate + 1, mark, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // one mark opens the gate once
    __hip_atomic_store(gate + 2, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static unsigned stream_grid(int64_t n) {
    int64_t nb = (n + RW_BLOCK - 1) / RW_BLOCK;
    return (unsigned)(nb < 1 ? 1 : (nb > 2048 ? 2048 : nb));
}

}  // namespace crf

extern "C" int crfconv_cat2(const float* xa, const float* xb, int64_t m, int ca, int cb, float* out, void* stream) {
    CRF_REQUIRE(m >= 0 && ca >= 4 && cb >= 4 && ca % 4 == 0 && cb % 4 == 0, CRF_ERR_UNSUPPORTED,
                "cat2: widths must be multiples of 4 (ca=%d cb=%d)", ca, cb);
    if (m == 0) return CRF_OK;
    CRF_REQUIRE(xa != nullptr && xb != nullptr && out != nullptr, CRF_ERR_ARG, "null operand");
    const int64_t n4 = m * (int64_t)((ca + cb) / 4);
    hipLaunchKernelGGL(crf::cat2_kernel, dim3(crf::stream_grid(n4)), dim3(crf::RW_BLOCK), 0, crf::as_stream(stream), xa, xb, n4,
                       ca / 4, cb / 4, out);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_split2(const float* g, int64_t m, int ca, int cb, float* ga, float* gb, void* stream) {
    CRF_REQUIRE(m >= 0 && ca >= 4 && cb >= 4 && ca % 4 == 0 && cb % 4 == 0, CRF_ERR_UNSUPPORTED,
                "split2: widths must be multiples of 4 (ca=%d cb=%d)", ca, cb);
    if (m == 0) return CRF_OK;
    CRF_REQUIRE(g != nullptr && ga != nullptr && gb != nullptr, CRF_ERR_ARG, "null operand");
    const int64_t n4 = m * (int64_t)((ca + cb) / 4);
    hipLaunchKernelGGL(crf::split2_kernel, dim3(crf::stream_grid(n4)), dim3(crf::RW_BLOCK), 0, crf::as_stream(stream), g, n4,
                       ca / 4, cb / 4, ga, gb);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_add_i64(int64_t* x, int64_t n, int64_t delta, void* stream) {
    CRF_REQUIRE(n >= 0 && n < ((int64_t)1 << 31), CRF_ERR_ARG, "bad length");
    if (n == 0) return CRF_OK;
    CRF_REQUIRE(x != nullptr, CRF_ERR_ARG, "null operand");
    hipLaunchKernelGGL(crf::add_i64_kernel, dim3((unsigned)((n + crf::RW_BLOCK - 1) / crf::RW_BLOCK)), dim3(crf::RW_BLOCK), 0,
                       crf::as_stream(stream), reinterpret_cast<long long*>(x), (int)n, (long long)delta);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

// gate: 4 uint64 device words, zero-initialised (mark | consumed | enabled | timeouts).  crfconv_gate_mark: mark += 1 (one tiny
// launch inside the marking stream's work).  crfconv_gate_wait: returns at once while gate[2] == 0; else holds the stream until a mark
// it has not consumed yet exists, for at most max_wait_us microseconds (<= 100 000).
extern "C" int crfconv_gate_mark(uint64_t* gate, void* stream) {
    CRF_REQUIRE(gate != nullptr, CRF_ERR_ARG, "null gate");
    hipLaunchKernelGGL(crf::gate_mark_kernel, dim3(1), dim3(1), 0, crf::as_stream(stream), reinterpret_cast<unsigned long long*>(gate));
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}
extern "C" int crfconv_gate_wait(uint64_t* gate, int max_wait_us, void* stream) {
    CRF_REQUIRE(gate != nullptr && max_wait_us >= 0 && max_wait_us <= 100000, CRF_ERR_ARG, "null gate or max_wait_us=%d outside [0, 100000]", max_wait_us);
    hipLaunchKernelGGL(crf::gate_wait_kernel, dim3(1), dim3(64), 0, crf::as_stream(stream), reinterpret_cast<unsigned long long*>(gate),
                       (unsigned long long)max_wait_us * 100ull);      // wall_clock64(): 100 MHz
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}
