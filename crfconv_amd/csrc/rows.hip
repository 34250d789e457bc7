// Row plumbing of the training step that is neither a contraction nor a normalisation: column concatenation of two per-point
// feature tensors and its inverse (the fusion layers' torch.cat, models/continuous_crf_conv_big.py:71 and
// models/point_conv_big.py:107, on the levels where the two-pointer Linear does not apply), the batched copy of gradients
// into the flat all-reduce bucket, and the step-counter increment of the BatchNorm layers.  All HBM-bound streaming passes;
// they exist so that a captured training step launches no framework kernel.
#include "common.hpp"

#define CRF_COPY_MAX_JOBS 96      /* 96 x (8 + 8 + 4) bytes of kernel arguments */

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

struct CopyJobs {
    const float* src[CRF_COPY_MAX_JOBS];
    float* dst[CRF_COPY_MAX_JOBS];
    int n[CRF_COPY_MAX_JOBS];
};

// blockIdx.y = job; grid-stride over its elements
__global__ __launch_bounds__(RW_BLOCK) void copy_jobs_kernel(CopyJobs jobs) {
    const int j = blockIdx.y;
    const float* __restrict__ s = jobs.src[j];
    float* __restrict__ d = jobs.dst[j];
    const int n = jobs.n[j];
    for (int t = blockIdx.x * RW_BLOCK + threadIdx.x; t < n; t += gridDim.x * RW_BLOCK) d[t] = s[t];
}

__global__ __launch_bounds__(RW_BLOCK) void add_i64_kernel(long long* __restrict__ x, int n, long long delta) {
    const int t = blockIdx.x * RW_BLOCK + threadIdx.x;
    if (t < n) x[t] += delta;
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

extern "C" int crfconv_copy_batched_max_jobs(void) { return CRF_COPY_MAX_JOBS; }

// dst[j][0 .. n[j]) = src[j][...] for njobs <= CRF_COPY_MAX_JOBS contiguous float ranges, one launch.
extern "C" int crfconv_copy_batched(const float* const* src, float* const* dst, const int64_t* n, int njobs, void* stream) {
    CRF_REQUIRE(njobs >= 0 && njobs <= CRF_COPY_MAX_JOBS, CRF_ERR_ARG, "at most %d copy jobs per launch (got %d)",
                CRF_COPY_MAX_JOBS, njobs);
    if (njobs == 0) return CRF_OK;
    crf::CopyJobs jobs;
    int64_t longest = 0;
    for (int j = 0; j < njobs; ++j) {
        CRF_REQUIRE(n[j] >= 0 && n[j] < ((int64_t)1 << 31) && (n[j] == 0 || (src[j] != nullptr && dst[j] != nullptr)), CRF_ERR_ARG,
                    "copy job %d: bad range", j);
        jobs.src[j] = src[j];
        jobs.dst[j] = dst[j];
        jobs.n[j] = (int)n[j];
        if (n[j] > longest) longest = n[j];
    }
    if (longest == 0) return CRF_OK;
    int64_t gx = (longest + crf::RW_BLOCK - 1) / crf::RW_BLOCK;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(crf::copy_jobs_kernel, dim3((unsigned)gx, (unsigned)njobs), dim3(crf::RW_BLOCK), 0, crf::as_stream(stream),
                       jobs);
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
