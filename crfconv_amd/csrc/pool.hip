// Neighbour max-pooling (strided ResNet shortcut, point_conv_big.py:74-77) and nearest
// up-sampling (row gather, :97-101 / continuous_crf_conv_big.py:60), forward and backward.
// One thread per (row, 4-channel quad): 16-byte loads, a row's C/4 quads are adjacent lanes.
#include "common.hpp"

namespace crf {

// AFFINE: the pooled quantity is a x + b per channel (coef = [a | b | ..] rows of C floats: a BatchNorm without activation
// applied while gathering -- the strided shortcut of a ResNet block, models/point_conv_big.py:74-83 -- so the normalised
// fine-level tensor never reaches memory; fmaf(a, x, b) is exactly what bn_apply would have stored).
template <bool AFFINE>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x,
                                                          const int32_t* __restrict__ idx, int K,
                                                          int64_t m_tgt, int C4,
                                                          float* __restrict__ out,
                                                          int32_t* __restrict__ arg,
                                                          const float* __restrict__ coef) {
    const int64_t t = (int64_t)xcd_block_id() * 256 + threadIdx.x;
    if (t >= m_tgt * C4) return;
    const int64_t i = t / C4;
    const int q = (int)(t - i * C4);
    const int32_t* irow = idx + i * K;
    float4 ca = make_float4(1.f, 1.f, 1.f, 1.f), cb = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (AFFINE) { ca = ld4(coef + 4 * q); cb = ld4(coef + 4 * C4 + 4 * q); }
    const float ninf = -__builtin_inff();
    float4 best = make_float4(ninf, ninf, ninf, ninf);
    int4 who = make_int4(-1, -1, -1, -1);
    for (int k0 = 0; k0 < K; k0 += 4) {                         // four neighbours per trip: index entries, then rows
        int jn[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) jn[u] = k0 + u < K ? irow[k0 + u] : -1;
        float4 vn[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) vn[u] = ld4(x + ((int64_t)(jn[u] < 0 ? 0 : jn[u]) * C4 + q) * 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (jn[u] < 0) continue;                             // "no neighbour" entry of a padded table
            float4 v = vn[u];
            if constexpr (AFFINE) v = make_float4(fmaf(ca.x, v.x, cb.x), fmaf(ca.y, v.y, cb.y), fmaf(ca.z, v.z, cb.z), fmaf(ca.w, v.w, cb.w));
            const int k = k0 + u;
            if (v.x > best.x || who.x < 0) { best.x = v.x; who.x = k; }
            if (v.y > best.y || who.y < 0) { best.y = v.y; who.y = k; }
            if (v.z > best.z || who.z < 0) { best.z = v.z; who.z = k; }
            if (v.w > best.w || who.w < 0) { best.w = v.w; who.w = k; }
        }
    }
    if (who.x < 0) best = make_float4(0.f, 0.f, 0.f, 0.f);      // empty group -> 0 (scatter_max convention)
    st4(out + t * 4, best);
    *reinterpret_cast<int4*>(arg + t * 4) = who;
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ gout,
                                                          const int32_t* __restrict__ arg,
                                                          const int32_t* __restrict__ rev_ptr,
                                                          const int32_t* __restrict__ rev_eid, int K,
                                                          int64_t m_src, int C4,
                                                          float* __restrict__ dx) {
    const int64_t t = (int64_t)xcd_block_id() * 256 + threadIdx.x;
    if (t >= m_src * C4) return;
    const int64_t j = t / C4;
    const int q = (int)(t - j * C4);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int beg = rev_ptr[j], end = rev_ptr[j + 1];
    // four reverse edges per trip: their edge ids first, then the four (arg, gradient) row pairs -- two dependent memory
    // phases per four edges instead of two per edge; summed in edge order as before
    const int kshift = (K & (K - 1)) == 0 ? __ffs(K) - 1 : -1;
    for (int p = beg; p < end; p += 4) {
        int e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) e[u] = p + u < end ? rev_eid[p + u] : -1;
        int4 who[4];
        float4 g[4];
        int kk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ee = e[u] < 0 ? 0 : e[u];
            const int i = kshift >= 0 ? (ee >> kshift) : (ee / K);
            kk[u] = e[u] < 0 ? -2 : ee - i * K;                 // -2 matches no arg entry
            const int64_t o = ((int64_t)i * C4 + q) * 4;
            who[u] = *reinterpret_cast<const int4*>(arg + o);
            g[u] = ld4(gout + o);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (who[u].x == kk[u]) acc.x += g[u].x;
            if (who[u].y == kk[u]) acc.y += g[u].y;
            if (who[u].z == kk[u]) acc.z += g[u].z;
            if (who[u].w == kk[u]) acc.w += g[u].w;
        }
    }
    st4(dx + t * 4, acc);
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ x,
                                                          const int32_t* __restrict__ idx,
                                                          int64_t m_tgt, int C4,
                                                          float* __restrict__ out) {
    const int64_t t = (int64_t)xcd_block_id() * 256 + threadIdx.x;
    if (t >= m_tgt * C4) return;
    const int64_t i = t / C4;
    const int q = (int)(t - i * C4);
    st4(out + t * 4, ld4(x + ((int64_t)idx[i] * C4 + q) * 4));
}

__global__ __launch_bounds__(256) void gather_rows_bwd_kernel(const float* __restrict__ gout,
                                                              const int32_t* __restrict__ rev_ptr,
                                                              const int32_t* __restrict__ rev_eid,
                                                              int64_t m_src, int C4,
                                                              float* __restrict__ dx) {
    const int64_t t = (int64_t)xcd_block_id() * 256 + threadIdx.x;
    if (t >= m_src * C4) return;
    const int64_t j = t / C4;
    const int q = (int)(t - j * C4);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int beg = rev_ptr[j], end = rev_ptr[j + 1];
    for (int p = beg; p < end; p += 4) {              // four reverse edges per trip (ids, then rows), summed in edge order
        int e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) e[u] = p + u < end ? rev_eid[p + u] : -1;
        float4 g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) g[u] = ld4(gout + ((int64_t)(e[u] < 0 ? 0 : e[u]) * C4 + q) * 4);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (e[u] >= 0) { acc.x += g[u].x; acc.y += g[u].y; acc.z += g[u].z; acc.w += g[u].w; }
    }
    st4(dx + t * 4, acc);
}

static int check(int64_t rows, int C) {
    CRF_REQUIRE(rows > 0 && rows < ((int64_t)1 << 31), CRF_ERR_ARG, "rows=%lld out of range", (long long)rows);
    CRF_REQUIRE(C > 0 && C % 4 == 0, CRF_ERR_UNSUPPORTED, "channels C=%d must be a positive multiple of 4", C);
    return CRF_OK;
}

// Residual join of the ResNet block (point_conv_big.py:86-88): out = lrelu(a + b, slope); its backward
// g_in = g * (out > 0 ? 1 : slope) serves both addends (slope > 0, so sign(out) = sign(a + b)).
__global__ __launch_bounds__(256) void add_lrelu_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        int64_t n4, float slope, float* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n4) return;
    const float4 u = ld4(a + 4 * t), v = ld4(b + 4 * t);
    float4 o = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
    o.x = o.x > 0.f ? o.x : slope * o.x;
    o.y = o.y > 0.f ? o.y : slope * o.y;
    o.z = o.z > 0.f ? o.z : slope * o.z;
    o.w = o.w > 0.f ? o.w : slope * o.w;
    st4(out + 4 * t, o);
}

__global__ __launch_bounds__(256) void lrelu_bwd_kernel(const float* __restrict__ g, const float* __restrict__ out,
                                                        int64_t n4, float slope, float* __restrict__ gin) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n4) return;
    const float4 u = ld4(g + 4 * t), o = ld4(out + 4 * t);
    st4(gin + 4 * t, make_float4(o.x > 0.f ? u.x : slope * u.x, o.y > 0.f ? u.y : slope * u.y,
                                 o.z > 0.f ? u.z : slope * u.z, o.w > 0.f ? u.w : slope * u.w));
}

}  // namespace crf

using namespace crf;

extern "C" int crfconv_neighbor_maxpool_forward(const float* x, const int32_t* idx32, int K, int64_t m_tgt,
                                                int C, float* out, int32_t* arg, crf_stream_t stream) {
    if (int rc = check(m_tgt, C)) return rc;
    CRF_REQUIRE(x && idx32 && out && arg && K >= 1, CRF_ERR_ARG, "null pointer / K");
    const int64_t n = m_tgt * (C / 4);
    hipLaunchKernelGGL(maxpool_fwd_kernel<false>, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, as_stream(stream), x,
                       idx32, K, m_tgt, C / 4, out, arg, (const float*)nullptr);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

// out[i,c] = max_k (a[c] x[idx32[i,k], c] + b[c]),  coef = [a | b | ...] (the [4, C] block of crfconv_bn_coef_from_records).
extern "C" int crfconv_neighbor_maxpool_affine_forward(const float* x, const float* coef, const int32_t* idx32, int K,
                                                       int64_t m_tgt, int C, float* out, int32_t* arg, crf_stream_t stream) {
    if (int rc = check(m_tgt, C)) return rc;
    CRF_REQUIRE(x && coef && idx32 && out && arg && K >= 1, CRF_ERR_ARG, "null pointer / K");
    const int64_t n = m_tgt * (C / 4);
    hipLaunchKernelGGL(maxpool_fwd_kernel<true>, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, as_stream(stream), x,
                       idx32, K, m_tgt, C / 4, out, arg, coef);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_neighbor_maxpool_backward(const float* gout, const int32_t* arg,
                                                 const int32_t* rev_ptr, const int32_t* rev_eid, int K,
                                                 int64_t m_src, int C, float* dx, crf_stream_t stream) {
    if (int rc = check(m_src, C)) return rc;
    CRF_REQUIRE(gout && arg && rev_ptr && rev_eid && dx && K >= 1, CRF_ERR_ARG, "null pointer / K");
    const int64_t n = m_src * (C / 4);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, as_stream(stream),
                       gout, arg, rev_ptr, rev_eid, K, m_src, C / 4, dx);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_gather_rows(const float* x, const int32_t* idx32, int64_t m_tgt, int C, float* out,
                                   crf_stream_t stream) {
    if (int rc = check(m_tgt, C)) return rc;
    CRF_REQUIRE(x && idx32 && out, CRF_ERR_ARG, "null pointer");
    const int64_t n = m_tgt * (C / 4);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, as_stream(stream), x,
                       idx32, m_tgt, C / 4, out);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_gather_rows_backward(const float* gout, const int32_t* rev_ptr,
                                            const int32_t* rev_eid, int64_t m_src, int C, float* dx,
                                            crf_stream_t stream) {
    if (int rc = check(m_src, C)) return rc;
    CRF_REQUIRE(gout && rev_ptr && rev_eid && dx, CRF_ERR_ARG, "null pointer");
    const int64_t n = m_src * (C / 4);
    hipLaunchKernelGGL(gather_rows_bwd_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, as_stream(stream),
                       gout, rev_ptr, rev_eid, m_src, C / 4, dx);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_add_lrelu(const float* a, const float* b, int64_t n, float slope, float* out, crf_stream_t stream) {
    CRF_REQUIRE(a && b && out, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(n > 0 && n % 4 == 0 && n < ((int64_t)1 << 40), CRF_ERR_ARG, "n=%lld must be a positive multiple of 4", (long long)n);
    CRF_REQUIRE(slope > 0.f, CRF_ERR_ARG, "slope must be positive");
    hipLaunchKernelGGL(crf::add_lrelu_kernel, dim3((unsigned)crf::cdiv(n / 4, 256)), dim3(256), 0, crf::as_stream(stream), a, b,
                       n / 4, slope, out);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_add_lrelu_backward(const float* gout, const float* out, int64_t n, float slope, float* gin,
                                          crf_stream_t stream) {
    CRF_REQUIRE(gout && out && gin, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(n > 0 && n % 4 == 0 && n < ((int64_t)1 << 40), CRF_ERR_ARG, "n=%lld must be a positive multiple of 4", (long long)n);
    hipLaunchKernelGGL(crf::lrelu_bwd_kernel, dim3((unsigned)crf::cdiv(n / 4, 256)), dim3(256), 0, crf::as_stream(stream), gout,
                       out, n / 4, slope, gin);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}
