// Discrete (label-space) CRF layer, models/discrete_crf_conv.py:40-63 -- the pieces that are not already in crf.hip:
//
//   w_e = sum_g Wg[g] * exp(-| f_g[j] - f_g[i] |^2)        edge e = (target i <- source j), f_g = f F_g  (:49-54)
//
// and its backward.  The mean-field step itself ( q <- softmax(-u - (sum_e w_e q_j) C), :57-61 ) is
// crfconv_meanfield_step (crf.hip) with z = -u, Q = I, P = -C followed by a row soft-max.
//
// Layout: fk [m, G * H] row-major (kernel g, hidden h at g * H + h); one 64-lane wavefront per target row, lane l
// owns hidden channels l, l + 64, ...; a row of G * H floats is read as contiguous 256-byte segments.  Padded table
// entries (< 0) mean "no neighbour" (weight 0).  Scatter-free: the source-side gradient walks the reverse CSR.
#include "common.hpp"

namespace crf {

constexpr int DW_BLOCK = 256, DW_ROWS = DW_BLOCK / WAVE, DW_GMAX = 8;

// squared distances of the G kernels between rows a and b of fk, every lane receives all of them
template <int HPL>
__device__ __forceinline__ void kernel_dists(const float (&fa)[DW_GMAX][HPL], const float* __restrict__ rb, int G, int H,
                                             int lane, float (&delta)[DW_GMAX][HPL], float (&d)[DW_GMAX]) {
#pragma unroll
    for (int g = 0; g < DW_GMAX; ++g) {
        float part = 0.f;
        if (g < G) {
#pragma unroll
            for (int u = 0; u < HPL; ++u) {
                const int h = lane + 64 * u;
                const float v = h < H ? rb[g * H + h] - fa[g][u] : 0.f;       // f[col] - f[row]  (:52)
                delta[g][u] = v;
                part = fmaf(v, v, part);
            }
            part = wave_sum(part);
        }
        d[g] = part;
    }
}

template <int HPL>
__device__ __forceinline__ void load_row_g(const float* __restrict__ r, int G, int H, int lane, float (&f)[DW_GMAX][HPL]) {
#pragma unroll
    for (int g = 0; g < DW_GMAX; ++g)
#pragma unroll
        for (int u = 0; u < HPL; ++u) {
            const int h = lane + 64 * u;
            f[g][u] = (g < G && h < H) ? r[g * H + h] : 0.f;
        }
}

template <int HPL>
__global__ __launch_bounds__(DW_BLOCK) void kw_forward_kernel(const float* __restrict__ fk,
                                                              const int32_t* __restrict__ idx, int K,
                                                              const float* __restrict__ Wg, int G, int H, int64_t m,
                                                              float* __restrict__ w) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * DW_ROWS + (threadIdx.x >> 6);
    if (i >= m) return;                                   // whole waves leave together
    const int GH = G * H;
    float fi[DW_GMAX][HPL], delta[DW_GMAX][HPL], d[DW_GMAX];
    load_row_g<HPL>(fk + i * GH, G, H, lane, fi);
    for (int k = 0; k < K; ++k) {
        const int j = idx[i * K + k];
        float acc = 0.f;
        if (j >= 0) {                                     // wave-uniform
            kernel_dists<HPL>(fi, fk + (int64_t)j * GH, G, H, lane, delta, d);
#pragma unroll
            for (int g = 0; g < DW_GMAX; ++g)
                if (g < G) acc = fmaf(Wg[g], expf(-d[g]), acc);
        }
        if (lane == 0) w[i * K + k] = acc;
    }
}

// target side of the backward: dfk_self[i] = sum_k gw_ik sum_g Wg[g] e_g * 2 (f_g[j] - f_g[i]) * (-1) ... see below;
// dW partial per block.   d/d f_i of exp(-|f_j - f_i|^2) = +2 e (f_j - f_i);  d/d f_j = -2 e (f_j - f_i).
template <int HPL>
__global__ __launch_bounds__(DW_BLOCK) void kw_backward_target_kernel(const float* __restrict__ gw,
                                                                      const float* __restrict__ fk,
                                                                      const int32_t* __restrict__ idx, int K,
                                                                      const float* __restrict__ Wg, int G, int H,
                                                                      int64_t m, float* __restrict__ dfk_self,
                                                                      double* __restrict__ dW_partial) {
    __shared__ double s_dw[DW_ROWS][DW_GMAX];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * DW_ROWS + wv;
    const int GH = G * H;
    float acc[DW_GMAX][HPL];
    double dw[DW_GMAX];
#pragma unroll
    for (int g = 0; g < DW_GMAX; ++g) {
        dw[g] = 0.0;
#pragma unroll
        for (int u = 0; u < HPL; ++u) acc[g][u] = 0.f;
    }
    if (i < m) {
        float fi[DW_GMAX][HPL], delta[DW_GMAX][HPL], d[DW_GMAX];
        load_row_g<HPL>(fk + i * GH, G, H, lane, fi);
        for (int k = 0; k < K; ++k) {
            const int j = idx[i * K + k];
            if (j < 0) continue;
            const float g_e = gw[i * K + k];
            kernel_dists<HPL>(fi, fk + (int64_t)j * GH, G, H, lane, delta, d);
#pragma unroll
            for (int g = 0; g < DW_GMAX; ++g)
                if (g < G) {
                    const float e = expf(-d[g]);
                    dw[g] += (double)(g_e * e);
                    const float coef = 2.0f * g_e * Wg[g] * e;
#pragma unroll
                    for (int u = 0; u < HPL; ++u) acc[g][u] = fmaf(coef, delta[g][u], acc[g][u]);
                }
        }
#pragma unroll
        for (int g = 0; g < DW_GMAX; ++g)
#pragma unroll
            for (int u = 0; u < HPL; ++u) {
                const int h = lane + 64 * u;
                if (g < G && h < H) dfk_self[i * GH + g * H + h] = acc[g][u];
            }
    }
    if (lane == 0)
#pragma unroll
        for (int g = 0; g < DW_GMAX; ++g) s_dw[wv][g] = dw[g];
    __syncthreads();
    if (threadIdx.x < DW_GMAX) {
        double a = 0.0;
        for (int r = 0; r < DW_ROWS; ++r) a += s_dw[r][threadIdx.x];
        dW_partial[(int64_t)blockIdx.x * DW_GMAX + threadIdx.x] = a;
    }
}

// source side: dfk[j] = dfk_self[j] - sum_{e in rev(j)} gw_e sum_g 2 Wg[g] e_g (f_g[j] - f_g[i(e)])
template <int HPL>
__global__ __launch_bounds__(DW_BLOCK) void kw_backward_source_kernel(const float* __restrict__ gw,
                                                                      const float* __restrict__ fk,
                                                                      const int32_t* __restrict__ rev_ptr,
                                                                      const int32_t* __restrict__ rev_eid, int K,
                                                                      const float* __restrict__ Wg, int G, int H,
                                                                      int64_t m, const float* __restrict__ dfk_self,
                                                                      float* __restrict__ dfk) {
    const int lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * DW_ROWS + (threadIdx.x >> 6);
    if (j >= m) return;
    const int GH = G * H;
    float fj[DW_GMAX][HPL], delta[DW_GMAX][HPL], d[DW_GMAX], acc[DW_GMAX][HPL];
    load_row_g<HPL>(fk + j * GH, G, H, lane, fj);
    load_row_g<HPL>(dfk_self + j * GH, G, H, lane, acc);
    const int beg = rev_ptr[j], end = rev_ptr[j + 1];
    for (int p = beg; p < end; ++p) {
        const int e = rev_eid[p];
        const int64_t i = e / K;
        const float g_e = gw[e];
        // kernel_dists gives f[b] - f[a] with a = the row held in registers: here a = j, b = i  ->  -(f_j - f_i)
        kernel_dists<HPL>(fj, fk + i * GH, G, H, lane, delta, d);
#pragma unroll
        for (int g = 0; g < DW_GMAX; ++g)
            if (g < G) {
                const float coef = 2.0f * g_e * Wg[g] * expf(-d[g]);
#pragma unroll
                for (int u = 0; u < HPL; ++u) acc[g][u] = fmaf(coef, delta[g][u], acc[g][u]);   // -2 c (f_j - f_i)
            }
    }
#pragma unroll
    for (int g = 0; g < DW_GMAX; ++g)
#pragma unroll
        for (int u = 0; u < HPL; ++u) {
            const int h = lane + 64 * u;
            if (g < G && h < H) dfk[j * GH + g * H + h] = acc[g][u];
        }
}

static int kw_check(int64_t m, int K, int G, int H) {
    CRF_REQUIRE(m > 0 && m * (int64_t)K < ((int64_t)1 << 31), CRF_ERR_ARG, "m=%lld K=%d out of range", (long long)m, K);
    CRF_REQUIRE(K >= 1 && K <= 64, CRF_ERR_ARG, "K=%d not in [1, 64]", K);
    CRF_REQUIRE(G >= 1 && G <= DW_GMAX, CRF_ERR_UNSUPPORTED, "num_kernels=%d not in [1, %d]", G, DW_GMAX);
    CRF_REQUIRE(H >= 1 && H <= 256, CRF_ERR_UNSUPPORTED, "hidden_channels=%d not in [1, 256]", H);
    return CRF_OK;
}

#define DISPATCH_HPL(H, ...)                                          \
    if ((H) <= 64) { constexpr int HPL = 1; __VA_ARGS__; }            \
    else if ((H) <= 128) { constexpr int HPL = 2; __VA_ARGS__; }      \
    else { constexpr int HPL = 4; __VA_ARGS__; }

}  // namespace crf

using namespace crf;

extern "C" int crfconv_kernel_weights_forward(const float* fk, const int32_t* idx32, int K, const float* Wg, int G,
                                              int H, int64_t m, float* w, crf_stream_t stream) {
    if (int rc = kw_check(m, K, G, H)) return rc;
    CRF_REQUIRE(fk && idx32 && Wg && w, CRF_ERR_ARG, "null pointer");
    const dim3 grid((unsigned)cdiv(m, DW_ROWS)), blk(DW_BLOCK);
    DISPATCH_HPL(H, hipLaunchKernelGGL(kw_forward_kernel<HPL>, grid, blk, 0, as_stream(stream), fk, idx32, K, Wg, G, H, m, w));
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" size_t crfconv_kernel_weights_partials(int64_t m) { return (size_t)cdiv(m, DW_ROWS) * DW_GMAX; }

extern "C" int crfconv_kernel_weights_backward(const float* gw, const float* fk, const int32_t* idx32,
                                               const int32_t* rev_ptr, const int32_t* rev_eid, int K, const float* Wg,
                                               int G, int H, int64_t m, float* dfk_self, float* dfk,
                                               double* dW_partial, crf_stream_t stream) {
    if (int rc = kw_check(m, K, G, H)) return rc;
    CRF_REQUIRE(gw && fk && idx32 && rev_ptr && rev_eid && Wg && dfk_self && dfk && dW_partial, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(dfk_self != dfk, CRF_ERR_ARG, "dfk must not alias dfk_self");
    const dim3 grid((unsigned)cdiv(m, DW_ROWS)), blk(DW_BLOCK);
    hipStream_t st = as_stream(stream);
    DISPATCH_HPL(H, hipLaunchKernelGGL(kw_backward_target_kernel<HPL>, grid, blk, 0, st, gw, fk, idx32, K, Wg, G, H, m,
                                       dfk_self, dW_partial));
    CRF_LAUNCH_CHECK();
    DISPATCH_HPL(H, hipLaunchKernelGGL(kw_backward_source_kernel<HPL>, grid, blk, 0, st, gw, fk, rev_ptr, rev_eid, K, Wg, G,
                                       H, m, dfk_self, dfk));
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}
