// Fused BatchNorm (+ LeakyReLU) over per-point rows [M, C] -- the normalisation inside every MLP of the path
// (models/common.py:31,36-37: FastBatchNorm1d, statistics over all B*N rows; followed by LeakyReLU(0.1)
// in most layers).  HBM-bound streaming kernels:
//   forward  : stats (1 read)            -> finalize (tiny) -> apply + activation (1 read, 1 write)
//   backward : reduce (2 reads)          -> finalize (tiny) -> apply (2 reads, 1 write)
// against the stock sequence of 5 forward / 6 backward passes.  Reductions: per-thread fp32 over a strided
// row slice (shifted by row 0 to avoid cancellation), fixed-order block and grid combination in float64 --
// bitwise reproducible, no atomics.
#include "common.hpp"

namespace crf {

constexpr int BN_BLOCK = 256;
constexpr int BN_MAXBLK = 512;

__device__ __forceinline__ float4 ld4g(const float* p) { return *reinterpret_cast<const float4*>(p); }

// partial[blk][0][C] = sum(x - shift), partial[blk][1][C] = sum((x - shift)^2), shift = row 0
__global__ __launch_bounds__(BN_BLOCK) void bn_stats_kernel(const float* __restrict__ x, int64_t M, int C,
                                                            float* __restrict__ partial) {
    extern __shared__ float sred[];  // [rows_in_block][2][C]
    const int C4 = C >> 2;
    const int rpi = BN_BLOCK / C4;               // rows handled per iteration by this block
    const int q = threadIdx.x % C4, rl = threadIdx.x / C4;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    if (rl < rpi) {
        const float4 sh = ld4g(x + 4 * q);
        auto add = [&](const float4 v) {
            const float4 d = make_float4(v.x - sh.x, v.y - sh.y, v.z - sh.z, v.w - sh.w);
            s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
            s2.x = fmaf(d.x, d.x, s2.x); s2.y = fmaf(d.y, d.y, s2.y); s2.z = fmaf(d.z, d.z, s2.z); s2.w = fmaf(d.w, d.w, s2.w);
        };
        const int64_t stride = (int64_t)gridDim.x * rpi;
        int64_t r = (int64_t)blockIdx.x * rpi + rl;
        for (; r + 3 * stride < M; r += 4 * stride) {            // four rows in flight per thread
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = ld4g(x + (r + u * stride) * C + 4 * q);
#pragma unroll
            for (int u = 0; u < 4; ++u) add(v[u]);
        }
        for (; r < M; r += stride) add(ld4g(x + r * C + 4 * q));
        st4(sred + (rl * 2 + 0) * C + 4 * q, s1);
        st4(sred + (rl * 2 + 1) * C + 4 * q, s2);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * C; t += BN_BLOCK) {
        float a = 0.f;
        for (int r = 0; r < rpi; ++r) a += sred[r * 2 * C + t];
        partial[(int64_t)blockIdx.x * 2 * C + t] = a;
    }
}

// coef[0] = a = gamma * rstd, coef[1] = b = beta - a * mean, coef[2] = mean, coef[3] = rstd   (each [C])
// running statistics updated in place when given (momentum; unbiased variance), as torch.nn.BatchNorm1d.
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partial, int nblk,
                                                          const float* __restrict__ x_row0, int64_t M, int C,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps,
                                                          float* __restrict__ run_mean,
                                                          float* __restrict__ run_var, float momentum,
                                                          float* __restrict__ coef) {
    // one wavefront per channel: lanes stride over the block partials, fixed-order shuffle tree
    const int c = blockIdx.x * (256 / WAVE) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int b = lane; b < nblk; b += WAVE) {
        s1 += (double)partial[(int64_t)b * 2 * C + c];
        s2 += (double)partial[(int64_t)b * 2 * C + C + c];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o, WAVE);
        s2 += __shfl_xor(s2, o, WAVE);
    }
    if (lane != 0) return;
    const double m1 = s1 / (double)M;
    const double mean = (double)x_row0[c] + m1;
    double var = s2 / (double)M - m1 * m1;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    const double a = (double)gamma[c] * rstd;
    coef[c] = (float)a;
    coef[C + c] = (float)((double)beta[c] - a * mean);
    coef[2 * C + c] = (float)mean;
    coef[3 * C + c] = (float)rstd;
    if (run_mean != nullptr) {
        const double unb = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
        run_mean[c] = (float)((1.0 - (double)momentum) * (double)run_mean[c] + (double)momentum * mean);
        run_var[c] = (float)((1.0 - (double)momentum) * (double)run_var[c] + (double)momentum * unb);
    }
}

// eval mode: coefficients from running statistics
__global__ __launch_bounds__(256) void bn_coef_eval_kernel(const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ run_mean,
                                                           const float* __restrict__ run_var, float eps, int C,
                                                           float* __restrict__ coef) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const double rstd = 1.0 / sqrt((double)run_var[c] + (double)eps);
    const double a = (double)gamma[c] * rstd;
    coef[c] = (float)a;
    coef[C + c] = (float)((double)beta[c] - a * (double)run_mean[c]);
    coef[2 * C + c] = run_mean[c];
    coef[3 * C + c] = (float)rstd;
}

// y = lrelu(a x + b, slope)   (slope == 1: no activation)
__global__ __launch_bounds__(BN_BLOCK) void bn_apply_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ coef, int64_t n4, int C4,
                                                            float slope, float* __restrict__ y) {
    for (int64_t t = (int64_t)blockIdx.x * BN_BLOCK + threadIdx.x; t < n4; t += (int64_t)gridDim.x * BN_BLOCK) {
        const int q = (int)(t % C4);
        const float4 a = ld4g(coef + 4 * q), b = ld4g(coef + 4 * C4 + 4 * q);
        const float4 v = ld4g(x + 4 * t);
        float4 o = make_float4(fmaf(a.x, v.x, b.x), fmaf(a.y, v.y, b.y), fmaf(a.z, v.z, b.z), fmaf(a.w, v.w, b.w));
        o.x = o.x > 0.f ? o.x : slope * o.x;
        o.y = o.y > 0.f ? o.y : slope * o.y;
        o.z = o.z > 0.f ? o.z : slope * o.z;
        o.w = o.w > 0.f ? o.w : slope * o.w;
        st4(y + 4 * t, o);
    }
}

// backward reductions: partial[blk][0][C] = sum g_pre, [1][C] = sum g_pre * xhat
// The ResNet join  out = lrelu(BN(y) + skip, slope)  (models/point_conv_big.py:84-88: lin_out has no activation, then
// F.leaky_relu(x + shortcut)): BatchNorm's affine, the residual add and the activation in ONE pass -- the normalised
// tensor never reaches memory.  Same arithmetic, operation for operation, as bn_apply (slope 1) followed by add_lrelu.
__global__ __launch_bounds__(BN_BLOCK) void bn_apply_add_kernel(const float* __restrict__ x, const float* __restrict__ coef,
                                                                const float* __restrict__ skip, int64_t n4, int C4,
                                                                float slope, float* __restrict__ y) {
    for (int64_t t = (int64_t)blockIdx.x * BN_BLOCK + threadIdx.x; t < n4; t += (int64_t)gridDim.x * BN_BLOCK) {
        const int q = (int)(t % C4);
        const float4 a = ld4g(coef + 4 * q), b = ld4g(coef + 4 * C4 + 4 * q);
        const float4 v = ld4g(x + 4 * t), k = ld4g(skip + 4 * t);
        float4 o = make_float4(add_rn(fmaf(a.x, v.x, b.x), k.x), add_rn(fmaf(a.y, v.y, b.y), k.y),
                               add_rn(fmaf(a.z, v.z, b.z), k.z), add_rn(fmaf(a.w, v.w, b.w), k.w));
        o.x = o.x > 0.f ? o.x : slope * o.x;
        o.y = o.y > 0.f ? o.y : slope * o.y;
        o.z = o.z > 0.f ? o.z : slope * o.z;
        o.w = o.w > 0.f ? o.w : slope * o.w;
        st4(y + 4 * t, o);
    }
}

// y = dropout(lrelu(a x + b, slope), p) in one pass (models/point_conv_big.py:131-134: MLP -> nn.Dropout(0.5)).
__global__ __launch_bounds__(BN_BLOCK) void bn_apply_dropout_kernel(const float* __restrict__ x, const float* __restrict__ coef,
                                                                    int64_t n4, int C4, float slope, unsigned long long seed,
                                                                    const long long* __restrict__ counter, unsigned threshold,
                                                                    float scale, float* __restrict__ y,
                                                                    long long* __restrict__ counter_used) {
    const unsigned long long ctr = (unsigned long long)counter[0];
    // the counter value this call masked with, for ITS backward: the live word may have advanced by then (a second training
    // forward before the first backward -- multi-view losses, forward-all-then-backward accumulation)
    if (counter_used != nullptr && blockIdx.x == 0 && threadIdx.x == 0) counter_used[0] = (long long)ctr;
    for (int64_t t = (int64_t)blockIdx.x * BN_BLOCK + threadIdx.x; t < n4; t += (int64_t)gridDim.x * BN_BLOCK) {
        const int q = (int)(t % C4);
        const float4 a = ld4g(coef + 4 * q), b = ld4g(coef + 4 * C4 + 4 * q);
        const float4 v = ld4g(x + 4 * t);
        float4 o = make_float4(fmaf(a.x, v.x, b.x), fmaf(a.y, v.y, b.y), fmaf(a.z, v.z, b.z), fmaf(a.w, v.w, b.w));
        o.x = o.x > 0.f ? o.x : slope * o.x;
        o.y = o.y > 0.f ? o.y : slope * o.y;
        o.z = o.z > 0.f ? o.z : slope * o.z;
        o.w = o.w > 0.f ? o.w : slope * o.w;
        const unsigned long long e = 4ull * (unsigned long long)t;
        o.x = dropout_keep(seed, ctr, e, threshold) ? o.x * scale : 0.f;
        o.y = dropout_keep(seed, ctr, e + 1, threshold) ? o.y * scale : 0.f;
        o.z = dropout_keep(seed, ctr, e + 2, threshold) ? o.z * scale : 0.f;
        o.w = dropout_keep(seed, ctr, e + 3, threshold) ? o.w * scale : 0.f;
        st4(y + 4 * t, o);
    }
}

// gin = g * keep * scale with the mask of the matching forward call.
__global__ __launch_bounds__(BN_BLOCK) void dropout_bwd_kernel(const float* __restrict__ g, int64_t n4, unsigned long long seed,
                                                               const long long* __restrict__ counter, unsigned threshold,
                                                               float scale, float* __restrict__ gin) {
    const unsigned long long ctr = (unsigned long long)counter[0];
    for (int64_t t = (int64_t)blockIdx.x * BN_BLOCK + threadIdx.x; t < n4; t += (int64_t)gridDim.x * BN_BLOCK) {
        float4 o = ld4g(g + 4 * t);
        const unsigned long long e = 4ull * (unsigned long long)t;
        o.x = dropout_keep(seed, ctr, e, threshold) ? o.x * scale : 0.f;
        o.y = dropout_keep(seed, ctr, e + 1, threshold) ? o.y * scale : 0.f;
        o.z = dropout_keep(seed, ctr, e + 2, threshold) ? o.z * scale : 0.f;
        o.w = dropout_keep(seed, ctr, e + 3, threshold) ? o.w * scale : 0.f;
        st4(gin + 4 * t, o);
    }
}

__global__ __launch_bounds__(BN_BLOCK) void bn_bwd_reduce_kernel(const float* __restrict__ gy,
                                                                 const float* __restrict__ x,
                                                                 const float* __restrict__ coef, int64_t M, int C,
                                                                 float slope, float* __restrict__ partial) {
    extern __shared__ float sred[];
    const int C4 = C >> 2;
    const int rpi = BN_BLOCK / C4;
    const int q = threadIdx.x % C4, rl = threadIdx.x / C4;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    if (rl < rpi) {
        const float4 a = ld4g(coef + 4 * q), b = ld4g(coef + C + 4 * q);
        const float4 mu = ld4g(coef + 2 * C + 4 * q), rs = ld4g(coef + 3 * C + 4 * q);
        auto add = [&](const float4 v, float4 g) {
            g.x *= fmaf(a.x, v.x, b.x) > 0.f ? 1.f : slope;
            g.y *= fmaf(a.y, v.y, b.y) > 0.f ? 1.f : slope;
            g.z *= fmaf(a.z, v.z, b.z) > 0.f ? 1.f : slope;
            g.w *= fmaf(a.w, v.w, b.w) > 0.f ? 1.f : slope;
            s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
            s2.x = fmaf(g.x, (v.x - mu.x) * rs.x, s2.x);
            s2.y = fmaf(g.y, (v.y - mu.y) * rs.y, s2.y);
            s2.z = fmaf(g.z, (v.z - mu.z) * rs.z, s2.z);
            s2.w = fmaf(g.w, (v.w - mu.w) * rs.w, s2.w);
        };
        const int64_t stride = (int64_t)gridDim.x * rpi;
        int64_t r = (int64_t)blockIdx.x * rpi + rl;
        for (; r + 3 * stride < M; r += 4 * stride) {            // four rows (eight 16-byte loads) in flight per thread
            float4 v[4], g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = ld4g(x + (r + u * stride) * C + 4 * q);
                g[u] = ld4g(gy + (r + u * stride) * C + 4 * q);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) add(v[u], g[u]);
        }
        for (; r < M; r += stride) add(ld4g(x + r * C + 4 * q), ld4g(gy + r * C + 4 * q));
        st4(sred + (rl * 2 + 0) * C + 4 * q, s1);
        st4(sred + (rl * 2 + 1) * C + 4 * q, s2);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * C; t += BN_BLOCK) {
        float acc = 0.f;
        for (int r = 0; r < rpi; ++r) acc += sred[r * 2 * C + t];
        partial[(int64_t)blockIdx.x * 2 * C + t] = acc;
    }
}

// dgamma, dbeta and the per-channel terms of dx = c1 * (g_pre - c2 - xhat * c3); bcoef = [c1 | c2 | c3]
// training: c1 = gamma*rstd, c2 = dbeta/M, c3 = dgamma/M;  eval: c1 = gamma*rstd, c2 = c3 = 0
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblk,
                                                              const float* __restrict__ coef, int64_t M, int C,
                                                              int training, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta,
                                                              float* __restrict__ bcoef) {
    const int c = blockIdx.x * (256 / WAVE) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int b = lane; b < nblk; b += WAVE) {
        s1 += (double)partial[(int64_t)b * 2 * C + c];
        s2 += (double)partial[(int64_t)b * 2 * C + C + c];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o, WAVE);
        s2 += __shfl_xor(s2, o, WAVE);
    }
    if (lane != 0) return;
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
    bcoef[c] = coef[c];   // a = gamma * rstd
    bcoef[C + c] = training ? (float)(s1 / (double)M) : 0.f;
    bcoef[2 * C + c] = training ? (float)(s2 / (double)M) : 0.f;
}

__global__ __launch_bounds__(BN_BLOCK) void bn_bwd_apply_kernel(const float* __restrict__ gy,
                                                                const float* __restrict__ x,
                                                                const float* __restrict__ coef,
                                                                const float* __restrict__ bcoef, int64_t n4, int C4,
                                                                float slope, float* __restrict__ gx) {
    const int C = 4 * C4;
    for (int64_t t = (int64_t)blockIdx.x * BN_BLOCK + threadIdx.x; t < n4; t += (int64_t)gridDim.x * BN_BLOCK) {
        const int q = (int)(t % C4);
        const float4 a = ld4g(coef + 4 * q), b = ld4g(coef + C + 4 * q);
        const float4 mu = ld4g(coef + 2 * C + 4 * q), rs = ld4g(coef + 3 * C + 4 * q);
        const float4 c1 = ld4g(bcoef + 4 * q), c2 = ld4g(bcoef + C + 4 * q), c3 = ld4g(bcoef + 2 * C + 4 * q);
        const float4 v = ld4g(x + 4 * t);
        float4 g = ld4g(gy + 4 * t);
        g.x *= fmaf(a.x, v.x, b.x) > 0.f ? 1.f : slope;
        g.y *= fmaf(a.y, v.y, b.y) > 0.f ? 1.f : slope;
        g.z *= fmaf(a.z, v.z, b.z) > 0.f ? 1.f : slope;
        g.w *= fmaf(a.w, v.w, b.w) > 0.f ? 1.f : slope;
        float4 o;
        o.x = c1.x * (g.x - c2.x - (v.x - mu.x) * rs.x * c3.x);
        o.y = c1.y * (g.y - c2.y - (v.y - mu.y) * rs.y * c3.y);
        o.z = c1.z * (g.z - c2.z - (v.z - mu.z) * rs.z * c3.z);
        o.w = c1.w * (g.w - c2.w - (v.w - mu.w) * rs.w * c3.w);
        st4(gx + 4 * t, o);
    }
}

// ------------------------------------------------------------------ small tensors: the whole BatchNorm in ONE launch
// The coarse levels (2560 / 640 points, 64..512 channels) are launch-bound: stats -> finalize -> apply cost three
// launches of 4.5-6 us for a few hundred KB.  Here one 1024-thread workgroup owns ONE channel quad (16 bytes of every
// row) and keeps its column in registers (NR rows per thread, every load in flight at once): statistics (fp32 per
// thread over <= NR rows, float64 across threads, fixed order) -> coefficients -> apply, one memory round trip, no
// cross-workgroup dependency.  A quad per workgroup (C/4 workgroups) rather than a wider slice: a workgroup streams at
// one CU's ~60 GB/s, so the work has to spread over many CUs (16-channel slices measured 11-22 us, this 5-6 us).
constexpr int BNS_BLOCK = 1024, BNS_CH = 4, BNS_ROWS = BNS_BLOCK, BNS_NR = 3, BNS_MAXM = BNS_ROWS * BNS_NR;

// sum over the 1024 threads of 8 per-thread floats {a.xyzw, b.xyzw}; every thread returns the totals in out[0..8).
// Fixed order: xor-shuffles inside a wave, then waves 0..15.
__device__ __forceinline__ void bns_block_sum(const float4 a, const float4 b, double (&out)[8], double (*s_red)[8]) {
    double v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int o = 1; o < WAVE; o <<= 1) v[i] += __shfl_xor(v[i], o, WAVE);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) s_red[wave][i] = v[i];
    }
    __syncthreads();
    if (threadIdx.x < 8) {                                       // thread i folds the 16 waves, fixed order
        double t = 0.0;
        for (int w = 0; w < BNS_BLOCK / WAVE; ++w) t += s_red[w][threadIdx.x];
        s_red[0][threadIdx.x] = t;                               // slot [0][i] is read by this thread only, above
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = s_red[0][i];
}

template <int NR>
__global__ __launch_bounds__(BNS_BLOCK) void bn_small_fwd_kernel(const float* __restrict__ x, int64_t M, int C,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float eps,
                                                                 float* __restrict__ run_mean,
                                                                 float* __restrict__ run_var, float momentum,
                                                                 float slope, float* __restrict__ coef,
                                                                 float* __restrict__ y) {
    __shared__ double s_red[BNS_BLOCK / WAVE][8];
    const int rl = threadIdx.x;
    const int c = blockIdx.x * BNS_CH;
    const float4 sh = ld4g(x + c);                               // shift = row 0 (a sample: no cancellation in the sums)
    float4 v[NR];
#pragma unroll
    for (int u = 0; u < NR; ++u) {
        const int64_t r = rl + (int64_t)u * BNS_ROWS;
        v[u] = r < M ? ld4g(x + r * C + c) : sh;                  // rows past the end contribute d = 0
    }
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
#pragma unroll
    for (int u = 0; u < NR; ++u) {
        const float4 d = make_float4(v[u].x - sh.x, v[u].y - sh.y, v[u].z - sh.z, v[u].w - sh.w);
        s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
        s2.x = fmaf(d.x, d.x, s2.x); s2.y = fmaf(d.y, d.y, s2.y); s2.z = fmaf(d.z, d.z, s2.z); s2.w = fmaf(d.w, d.w, s2.w);
    }
    double t[8];
    bns_block_sum(s1, s2, t, s_red);
    float av[4], bv[4];
    const float shv[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double m1 = t[i] / (double)M;
        const double mean = (double)shv[i] + m1;
        double var = (t[4 + i] - t[i] * m1) / (double)M;
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + (double)eps);
        const double a = (double)gamma[c + i] * rstd;
        av[i] = (float)a;
        bv[i] = (float)((double)beta[c + i] - a * mean);
        if (rl == 0) {                                           // one thread per channel quad publishes
            coef[c + i] = av[i];
            coef[C + c + i] = bv[i];
            coef[2 * C + c + i] = (float)mean;
            coef[3 * C + c + i] = (float)rstd;
            if (run_mean != nullptr) {
                const double unb = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
                run_mean[c + i] = (float)((1.0 - (double)momentum) * (double)run_mean[c + i] + (double)momentum * mean);
                run_var[c + i] = (float)((1.0 - (double)momentum) * (double)run_var[c + i] + (double)momentum * unb);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < NR; ++u) {
        const int64_t r = rl + (int64_t)u * BNS_ROWS;
        float4 o;
        o.x = fmaf(av[0], v[u].x, bv[0]); o.y = fmaf(av[1], v[u].y, bv[1]);
        o.z = fmaf(av[2], v[u].z, bv[2]); o.w = fmaf(av[3], v[u].w, bv[3]);
        o.x = o.x > 0.f ? o.x : o.x * slope; o.y = o.y > 0.f ? o.y : o.y * slope;
        o.z = o.z > 0.f ? o.z : o.z * slope; o.w = o.w > 0.f ? o.w : o.w * slope;
        if (r < M) st4(y + r * C + c, o);
    }
}

template <int NR>
__global__ __launch_bounds__(BNS_BLOCK) void bn_small_bwd_kernel(const float* __restrict__ gy,
                                                                 const float* __restrict__ x,
                                                                 const float* __restrict__ coef, int64_t M, int C,
                                                                 int training, float slope, float* __restrict__ gx,
                                                                 float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta) {
    __shared__ double s_red[BNS_BLOCK / WAVE][8];
    const int rl = threadIdx.x;
    const int c = blockIdx.x * BNS_CH;
    const float4 a = ld4g(coef + c), b = ld4g(coef + C + c), mu = ld4g(coef + 2 * C + c), rs = ld4g(coef + 3 * C + c);
    float4 xh[NR], g[NR];                                        // x-hat and the gradient in front of the activation
#pragma unroll
    for (int u = 0; u < NR; ++u) {
        const int64_t r = rl + (int64_t)u * BNS_ROWS;
        const bool in = r < M;
        xh[u] = in ? ld4g(x + r * C + c) : mu;
        g[u] = in ? ld4g(gy + r * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
#pragma unroll
    for (int u = 0; u < NR; ++u) {
        const float4 v = xh[u];
        g[u].x *= fmaf(a.x, v.x, b.x) > 0.f ? 1.f : slope;
        g[u].y *= fmaf(a.y, v.y, b.y) > 0.f ? 1.f : slope;
        g[u].z *= fmaf(a.z, v.z, b.z) > 0.f ? 1.f : slope;
        g[u].w *= fmaf(a.w, v.w, b.w) > 0.f ? 1.f : slope;
        xh[u] = make_float4((v.x - mu.x) * rs.x, (v.y - mu.y) * rs.y, (v.z - mu.z) * rs.z, (v.w - mu.w) * rs.w);
        s1.x += g[u].x; s1.y += g[u].y; s1.z += g[u].z; s1.w += g[u].w;
        s2.x = fmaf(g[u].x, xh[u].x, s2.x); s2.y = fmaf(g[u].y, xh[u].y, s2.y);
        s2.z = fmaf(g[u].z, xh[u].z, s2.z); s2.w = fmaf(g[u].w, xh[u].w, s2.w);
    }
    double t[8];
    bns_block_sum(s1, s2, t, s_red);
    float c2[4], c3[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        c2[i] = training ? (float)(t[i] / (double)M) : 0.f;
        c3[i] = training ? (float)(t[4 + i] / (double)M) : 0.f;
        if (rl == 0) {
            dbeta[c + i] = (float)t[i];
            dgamma[c + i] = (float)t[4 + i];
        }
    }
#pragma unroll
    for (int u = 0; u < NR; ++u) {
        const int64_t r = rl + (int64_t)u * BNS_ROWS;
        float4 o;
        o.x = a.x * (g[u].x - c2[0] - xh[u].x * c3[0]);
        o.y = a.y * (g[u].y - c2[1] - xh[u].y * c3[1]);
        o.z = a.z * (g[u].z - c2[2] - xh[u].z * c3[2]);
        o.w = a.w * (g[u].w - c2[3] - xh[u].w * c3[3]);
        if (r < M) st4(gx + r * C + c, o);
    }
}

static bool bn_small_ok(int64_t M, int C) { return M <= BNS_MAXM && C % BNS_CH == 0; }

static int bn_check(int64_t M, int C) {
    CRF_REQUIRE(M > 0 && M < ((int64_t)1 << 40), CRF_ERR_ARG, "M=%lld out of range", (long long)M);
    CRF_REQUIRE(C >= 4 && C % 4 == 0 && C <= 1024, CRF_ERR_UNSUPPORTED, "C=%d must be a multiple of 4 in [4, 1024]", C);
    return CRF_OK;
}

static int bn_nblk(int64_t M, int C) {
    const int rpi = BN_BLOCK / (C / 4);
    // ~8 rows per thread (two unrolled trips of four) before paying for another partial
    int64_t nb = (M + 8 * (int64_t)rpi - 1) / (8 * (int64_t)rpi);
    if (nb < 1) nb = 1;
    return (int)(nb > BN_MAXBLK ? BN_MAXBLK : nb);
}

static unsigned ew_grid(int64_t n4) {
    int64_t g = (n4 + BN_BLOCK - 1) / BN_BLOCK;
    return (unsigned)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace crf

using namespace crf;

extern "C" size_t crfconv_bn_workspace(int64_t M, int C) {
    if (M <= 0 || C < 4) return 0;
    return sizeof(float) * (2 * (size_t)C * BN_MAXBLK + 3 * (size_t)C) + 512;
}

// training (use_batch_stats != 0): statistics of x, coef out, running stats updated (may be NULL);
// eval: coef from running stats.  y = lrelu(BN(x), slope); slope = 1 -> plain BatchNorm.  y may alias x.
extern "C" int crfconv_bn_forward(const float* x, int64_t M, int C, const float* gamma, const float* beta,
                                  float* run_mean, float* run_var, float momentum, float eps, int use_batch_stats,
                                  float slope, float* coef, float* y, void* workspace, size_t workspace_bytes,
                                  crf_stream_t stream) {
    if (int rc = bn_check(M, C)) return rc;
    CRF_REQUIRE(x && gamma && beta && coef && y && workspace, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(use_batch_stats || (run_mean && run_var), CRF_ERR_ARG, "eval mode needs running statistics");
    CRF_REQUIRE(workspace_bytes >= crfconv_bn_workspace(M, C), CRF_ERR_WORKSPACE, "workspace too small");
    hipStream_t st = as_stream(stream);
    float* partial = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    if (use_batch_stats && bn_small_ok(M, C)) {
        if (M <= BNS_ROWS)
            hipLaunchKernelGGL(bn_small_fwd_kernel<1>, dim3(C / BNS_CH), dim3(BNS_BLOCK), 0, st, x, M, C, gamma, beta, eps,
                               run_mean, run_var, momentum, slope, coef, y);
        else
            hipLaunchKernelGGL(bn_small_fwd_kernel<BNS_NR>, dim3(C / BNS_CH), dim3(BNS_BLOCK), 0, st, x, M, C, gamma, beta, eps,
                               run_mean, run_var, momentum, slope, coef, y);
        CRF_LAUNCH_CHECK();
        return CRF_OK;
    }
    if (use_batch_stats) {
        const int nblk = bn_nblk(M, C);
        const int rpi = BN_BLOCK / (C / 4);
        hipLaunchKernelGGL(bn_stats_kernel, dim3(nblk), dim3(BN_BLOCK), sizeof(float) * 2 * C * (rpi > 0 ? rpi : 1), st, x, M,
                           C, partial);
        CRF_LAUNCH_CHECK();
        hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, st, partial, nblk, x, M, C, gamma, beta,
                           eps, run_mean, run_var, momentum, coef);
        CRF_LAUNCH_CHECK();
    } else {
        hipLaunchKernelGGL(bn_coef_eval_kernel, dim3((C + 255) / 256), dim3(256), 0, st, gamma, beta, run_mean, run_var, eps,
                           C, coef);
        CRF_LAUNCH_CHECK();
    }
    const int64_t n4 = M * (C / 4);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(n4)), dim3(BN_BLOCK), 0, st, x, coef, n4, C / 4, slope, y);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_bn_backward(const float* gy, const float* x, const float* coef, int64_t M, int C,
                                   int training, float slope, float* gx, float* dgamma, float* dbeta, void* workspace,
                                   size_t workspace_bytes, crf_stream_t stream) {
    if (int rc = bn_check(M, C)) return rc;
    CRF_REQUIRE(gy && x && coef && gx && dgamma && dbeta && workspace, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(workspace_bytes >= crfconv_bn_workspace(M, C), CRF_ERR_WORKSPACE, "workspace too small");
    hipStream_t st = as_stream(stream);
    float* partial = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    float* bcoef = partial + 2 * (size_t)C * BN_MAXBLK;
    if (bn_small_ok(M, C)) {
        if (M <= BNS_ROWS)
            hipLaunchKernelGGL(bn_small_bwd_kernel<1>, dim3(C / BNS_CH), dim3(BNS_BLOCK), 0, st, gy, x, coef, M, C, training,
                               slope, gx, dgamma, dbeta);
        else
            hipLaunchKernelGGL(bn_small_bwd_kernel<BNS_NR>, dim3(C / BNS_CH), dim3(BNS_BLOCK), 0, st, gy, x, coef, M, C,
                               training, slope, gx, dgamma, dbeta);
        CRF_LAUNCH_CHECK();
        return CRF_OK;
    }
    const int nblk = bn_nblk(M, C);
    const int rpi = BN_BLOCK / (C / 4);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nblk), dim3(BN_BLOCK), sizeof(float) * 2 * C * (rpi > 0 ? rpi : 1), st, gy,
                       x, coef, M, C, slope, partial);
    CRF_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, st, partial, nblk, coef, M, C, training,
                       dgamma, dbeta, bcoef);
    CRF_LAUNCH_CHECK();
    const int64_t n4 = M * (C / 4);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(n4)), dim3(BN_BLOCK), 0, st, gy, x, coef, bcoef, n4, C / 4, slope,
                       gx);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

// y = lrelu(a x + b, slope) with coefficients that already exist (e.g. from crfconv_bn_coef_from_records).
extern "C" int crfconv_bn_apply(const float* x, int64_t M, int C, const float* coef, float slope, float* y,
                                crf_stream_t stream) {
    if (int rc = bn_check(M, C)) return rc;
    CRF_REQUIRE(x && coef && y, CRF_ERR_ARG, "null pointer");
    const int64_t n4 = M * (C / 4);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(n4)), dim3(BN_BLOCK), 0, as_stream(stream), x, coef, n4, C / 4, slope, y);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

// out = lrelu(a x + b + skip, slope) with existing coefficients: BatchNorm (no activation of its own), the residual add and
// the join's LeakyReLU in one pass.  out may alias skip.
extern "C" int crfconv_bn_apply_add(const float* x, int64_t M, int C, const float* coef, const float* skip, float slope,
                                    float* out, crf_stream_t stream) {
    if (int rc = bn_check(M, C)) return rc;
    CRF_REQUIRE(x && coef && skip && out, CRF_ERR_ARG, "null pointer");
    const int64_t n4 = M * (C / 4);
    hipLaunchKernelGGL(bn_apply_add_kernel, dim3(ew_grid(n4)), dim3(BN_BLOCK), 0, as_stream(stream), x, coef, skip, n4, C / 4,
                       slope, out);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

// out = dropout(lrelu(a x + b, slope), p): the mask of element e is a hash of (seed, *counter, e); counter = one int64 DEVICE word
// the caller advances between steps.  crfconv_dropout_backward with the same (seed, counter value, p) applies the same mask.
extern "C" int crfconv_bn_apply_dropout(const float* x, int64_t M, int C, const float* coef, float slope, float p,
                                        uint64_t seed, const int64_t* counter, float* out, int64_t* counter_used,
                                        crf_stream_t stream) {
    if (int rc = bn_check(M, C)) return rc;
    CRF_REQUIRE(x && coef && counter && out, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(p >= 0.f && p < 1.f, CRF_ERR_ARG, "dropout probability %g outside [0, 1)", (double)p);
    const int64_t n4 = M * (C / 4);
    hipLaunchKernelGGL(bn_apply_dropout_kernel, dim3(ew_grid(n4)), dim3(BN_BLOCK), 0, as_stream(stream), x, coef, n4, C / 4, slope,
                       (unsigned long long)seed, reinterpret_cast<const long long*>(counter), crf::dropout_threshold(p), 1.f / (1.f - p), out,
                       reinterpret_cast<long long*>(counter_used));
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_dropout_backward(const float* g, int64_t n, float p, uint64_t seed, const int64_t* counter, float* gin,
                                        crf_stream_t stream) {
    CRF_REQUIRE(g && counter && gin && n > 0 && n % 4 == 0, CRF_ERR_ARG, "null pointer / n=%lld not a positive multiple of 4", (long long)n);
    CRF_REQUIRE(p >= 0.f && p < 1.f, CRF_ERR_ARG, "dropout probability %g outside [0, 1)", (double)p);
    const int64_t n4 = n / 4;
    hipLaunchKernelGGL(dropout_bwd_kernel, dim3(ew_grid(n4)), dim3(BN_BLOCK), 0, as_stream(stream), g, n4, (unsigned long long)seed,
                       reinterpret_cast<const long long*>(counter), crf::dropout_threshold(p), 1.f / (1.f - p), gin);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}
