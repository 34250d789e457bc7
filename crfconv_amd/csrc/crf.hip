// CRF mean-field kernels for gfx950 (forward + the two halves of each backward step).
//
// Thread mapping (all kernels): a point's H channels are spread over L = H/4 adjacent lanes,
// one float4 (16-byte load) per lane, so a 64-lane wavefront carries 64/L points and a
// neighbour row is fetched by L lanes as one contiguous 4H-byte segment.  Per-point reductions
// (squared distance, dot products) are xor-shuffles over the L lanes; the H x H products use
// shuffles for the vector and LDS (float4 rows) for the matrix.
//
// Reference semantics: models/continuous_crf_conv_big.py:49-54 (similarity), :63-72 (loop).
#include "common.hpp"

namespace crf {

constexpr int BLOCK = 256;

// v (float4 per lane, quad q of the point's H-vector)  ->  out += v_full * Mat, where
// sM holds Mat [H][H] row-major as float4 rows: sM[h * L + q] = Mat[h][4q .. 4q+3].
template <int H>
__device__ __forceinline__ float4 matvec_acc(float4 v, const float4* sM, int lane, int q, float4 acc) {
    constexpr int L = H / 4;
    const int base = lane - q;
#pragma unroll
    for (int hq = 0; hq < L; ++hq) {
        const float v0 = __shfl(v.x, base + hq, WAVE);
        const float v1 = __shfl(v.y, base + hq, WAVE);
        const float v2 = __shfl(v.z, base + hq, WAVE);
        const float v3 = __shfl(v.w, base + hq, WAVE);
        acc = fma4(v0, sM[(4 * hq + 0) * L + q], acc);
        acc = fma4(v1, sM[(4 * hq + 1) * L + q], acc);
        acc = fma4(v2, sM[(4 * hq + 2) * L + q], acc);
        acc = fma4(v3, sM[(4 * hq + 3) * L + q], acc);
    }
    return acc;
}

template <int H>
__device__ __forceinline__ void load_matrix(float4* sM, const float* __restrict__ Mat, bool transpose) {
    // sM[h][c] = transpose ? Mat[c][h] : Mat[h][c]
    float* s = reinterpret_cast<float*>(sM);
    for (int t = threadIdx.x; t < H * H; t += BLOCK) {
        const int h = t / H, c = t % H;
        s[t] = transpose ? Mat[c * H + h] : Mat[t];
    }
}

// ------------------------------------------------------------------ similarity + z Q
// KN > 0: neighbour count known at compile time, distances stay in registers.
// KN == 0: any count; distances are recomputed in a second sweep (rows are L1/L2 hot by then).
template <int H, int KN>
__global__ __launch_bounds__(BLOCK) void sim_kernel(const float* __restrict__ y,
                                                    const float* __restrict__ z,
                                                    const int32_t* __restrict__ idx, int K, int k0,
                                                    const float* __restrict__ Q,
                                                    float* __restrict__ s, float* __restrict__ zq,
                                                    int64_t m) {
    constexpr int L = H / 4, PPW = WAVE / L, PPB = PPW * (BLOCK / WAVE);
    __shared__ float4 sQ[H * L];
    load_matrix<H>(sQ, Q, false);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane % L;
    const int64_t row = (int64_t)blockIdx.x * PPB + wave * PPW + lane / L;
    const bool valid = row < m;
    const int64_t r = valid ? row : m - 1;
    const int Kn = KN > 0 ? KN : K - k0;

    const float4 yi = ld4(y + r * H + 4 * q);
    const int32_t* irow = idx + r * K + k0;
    float* srow = s + r * Kn;

    auto dist_to = [&](int k) {
        const int j = irow[k];
        const float4 yj = ld4(y + (int64_t)j * H + 4 * q);
        const float4 df = make_float4(yi.x - yj.x, yi.y - yj.y, yi.z - yj.z, yi.w - yj.w);
        return group_sum<L>(dot4(df, df));
    };

    if constexpr (KN > 0) {
        float d[KN];
        float dmin = 3.4e38f;
#pragma unroll
        for (int k = 0; k < KN; ++k) {
            d[k] = dist_to(k);
            dmin = fminf(dmin, d[k]);
        }
        float den = 0.f;
#pragma unroll
        for (int k = 0; k < KN; ++k) {
            d[k] = expf(dmin - d[k]);
            den += d[k];
        }
        const float inv = 1.0f / den;
#pragma unroll
        for (int k = 0; k < KN; ++k)
            if (valid && q == (k % L)) srow[k] = d[k] * inv;
    } else {
        float dmin = 3.4e38f;
        for (int k = 0; k < Kn; ++k) dmin = fminf(dmin, dist_to(k));
        float den = 0.f;
        for (int k = 0; k < Kn; ++k) {
            const float e = expf(dmin - dist_to(k));
            den += e;
            if (valid && q == 0) srow[k] = e;
        }
        const float inv = 1.0f / den;
        if (valid && q == 0)  // same lane re-reads what it wrote
            for (int k = 0; k < Kn; ++k) srow[k] *= inv;
    }

    // zq = z Q
    const float4 zi = ld4(z + r * H + 4 * q);
    const float4 o = matvec_acc<H>(zi, sQ, lane, q, make_float4(0.f, 0.f, 0.f, 0.f));
    if (valid) st4(zq + r * H + 4 * q, o);
}

// ------------------------------------------------------------------ one mean-field step
template <int H>
__global__ __launch_bounds__(BLOCK) void step_kernel(const float* __restrict__ xin,
                                                     const float* __restrict__ zq,
                                                     const float* __restrict__ s,
                                                     const int32_t* __restrict__ idx, int K, int k0,
                                                     const float* __restrict__ P,
                                                     float* __restrict__ xout, int64_t m) {
    constexpr int L = H / 4, PPW = WAVE / L, PPB = PPW * (BLOCK / WAVE);
    __shared__ float4 sP[H * L];
    load_matrix<H>(sP, P, false);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane % L;
    const int64_t row = (int64_t)blockIdx.x * PPB + wave * PPW + lane / L;
    const bool valid = row < m;
    const int64_t r = valid ? row : m - 1;
    const int Kn = K - k0;
    const int32_t* irow = idx + r * K + k0;
    const float* srow = s + r * Kn;

    float4 msg = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 5
    for (int k = 0; k < Kn; ++k) {
        const int j = irow[k];
        msg = fma4(srow[k], ld4(xin + (int64_t)j * H + 4 * q), msg);
    }
    const float4 o = matvec_acc<H>(msg, sP, lane, q, ld4(zq + r * H + 4 * q));
    if (valid) st4(xout + r * H + 4 * q, o);
}

// ------------------------------------------------------------------ backward, edge half
template <int H>
__global__ __launch_bounds__(BLOCK) void bwd_edge_kernel(const float* __restrict__ G,
                                                         const float* __restrict__ xprev,
                                                         const float* __restrict__ s,
                                                         const int32_t* __restrict__ idx, int K,
                                                         int k0, const float* __restrict__ P,
                                                         float* __restrict__ gm,
                                                         float* __restrict__ ds,
                                                         float* __restrict__ mt, int accumulate,
                                                         int64_t m) {
    constexpr int L = H / 4, PPW = WAVE / L, PPB = PPW * (BLOCK / WAVE);
    __shared__ float4 sPT[H * L];
    load_matrix<H>(sPT, P, true);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane % L;
    const int64_t row = (int64_t)blockIdx.x * PPB + wave * PPW + lane / L;
    const bool valid = row < m;
    const int64_t r = valid ? row : m - 1;
    const int Kn = K - k0;
    const int32_t* irow = idx + r * K + k0;
    const float* srow = s + r * Kn;
    float* dsrow = ds + r * Kn;

    const float4 g = ld4(G + r * H + 4 * q);
    const float4 gmi = matvec_acc<H>(g, sPT, lane, q, make_float4(0.f, 0.f, 0.f, 0.f));
    if (valid) st4(gm + r * H + 4 * q, gmi);
    float4 msg = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 5
    for (int k = 0; k < Kn; ++k) {
        const int j = irow[k];
        const float4 xj = ld4(xprev + (int64_t)j * H + 4 * q);
        msg = fma4(srow[k], xj, msg);
        const float dotv = group_sum<L>(dot4(gmi, xj));
        if (valid && q == (k % L)) dsrow[k] = accumulate ? dsrow[k] + dotv : dotv;
    }
    if (mt != nullptr && valid) st4(mt + r * H + 4 * q, msg);
}

// ------------------------------------------------------------------ backward, scatter half
// One source row per L lanes; walks the row's incoming edges (ascending edge id).
template <int H>
__global__ __launch_bounds__(BLOCK) void bwd_scatter_kernel(const float* __restrict__ gm,
                                                            const float* __restrict__ s,
                                                            const int32_t* __restrict__ rev_ptr,
                                                            const int32_t* __restrict__ rev_eid,
                                                            int K, int k0,
                                                            const float* __restrict__ add,
                                                            float* __restrict__ Gprev,
                                                            int64_t m_src) {
    constexpr int L = H / 4, PPW = WAVE / L, PPB = PPW * (BLOCK / WAVE);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane % L;
    const int64_t row = (int64_t)blockIdx.x * PPB + wave * PPW + lane / L;
    if (row >= m_src) return;  // no cross-lane traffic below
    const int Kn = K - k0;
    float4 acc = add ? ld4(add + row * H + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int beg = rev_ptr[row], end = rev_ptr[row + 1];
    for (int p = beg; p < end; ++p) {
        const int e = rev_eid[p];
        const int i = e / K, k = e - i * K;
        if (k < k0) continue;
        acc = fma4(s[(int64_t)i * Kn + (k - k0)], ld4(gm + (int64_t)i * H + 4 * q), acc);
    }
    st4(Gprev + row * H + 4 * q, acc);
}

// ------------------------------------------------------------------ softmax / distance backward
template <int H>
__global__ __launch_bounds__(BLOCK) void sim_bwd_kernel(const float* __restrict__ ds,
                                                        const float* __restrict__ s,
                                                        const float* __restrict__ y,
                                                        const int32_t* __restrict__ idx, int K,
                                                        int k0, float* __restrict__ w,
                                                        float* __restrict__ dy_self, int64_t m) {
    constexpr int L = H / 4, PPW = WAVE / L, PPB = PPW * (BLOCK / WAVE);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane % L;
    const int64_t row = (int64_t)blockIdx.x * PPB + wave * PPW + lane / L;
    const bool valid = row < m;
    const int64_t r = valid ? row : m - 1;
    const int Kn = K - k0;
    const int32_t* irow = idx + r * K + k0;
    const float* srow = s + r * Kn;
    const float* dsrow = ds + r * Kn;
    float* wrow = w + r * Kn;

    float dotv = 0.f;
    for (int k = 0; k < Kn; ++k) dotv = fmaf(srow[k], dsrow[k], dotv);
    const float4 yi = ld4(y + r * H + 4 * q);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < Kn; ++k) {
        // d loss / d logit_k = s_k (ds_k - dot);  logit = -dist  =>  d/d dist = -that; w = 2 * d/d dist
        const float wk = -2.0f * srow[k] * (dsrow[k] - dotv);
        const int j = irow[k];
        const float4 yj = ld4(y + (int64_t)j * H + 4 * q);
        acc = fma4(wk, make_float4(yi.x - yj.x, yi.y - yj.y, yi.z - yj.z, yi.w - yj.w), acc);
        if (valid && q == (k % L)) wrow[k] = wk;
    }
    if (valid) st4(dy_self + r * H + 4 * q, acc);
}

template <int H>
__global__ __launch_bounds__(BLOCK) void sim_bwd_scatter_kernel(const float* __restrict__ w,
                                                                const float* __restrict__ y,
                                                                const float* __restrict__ dy_self,
                                                                const int32_t* __restrict__ rev_ptr,
                                                                const int32_t* __restrict__ rev_eid,
                                                                int K, int k0,
                                                                float* __restrict__ dy,
                                                                int64_t m_src) {
    constexpr int L = H / 4, PPW = WAVE / L, PPB = PPW * (BLOCK / WAVE);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane % L;
    const int64_t row = (int64_t)blockIdx.x * PPB + wave * PPW + lane / L;
    if (row >= m_src) return;
    const int Kn = K - k0;
    const float4 yj = ld4(y + row * H + 4 * q);
    float4 acc = ld4(dy_self + row * H + 4 * q);
    const int beg = rev_ptr[row], end = rev_ptr[row + 1];
    for (int p = beg; p < end; ++p) {
        const int e = rev_eid[p];
        const int i = e / K, k = e - i * K;
        if (k < k0) continue;
        const float4 yi = ld4(y + (int64_t)i * H + 4 * q);
        acc = fma4(w[(int64_t)i * Kn + (k - k0)],
                   make_float4(yj.x - yi.x, yj.y - yi.y, yj.z - yi.z, yj.w - yi.w), acc);
    }
    st4(dy + row * H + 4 * q, acc);
}

template <int H>
constexpr int points_per_block() { return (WAVE / (H / 4)) * (BLOCK / WAVE); }

static int check_common(int64_t m, int H, int K, int k0) {
    CRF_REQUIRE(m > 0 && m < (int64_t)1 << 31, CRF_ERR_ARG, "rows m=%lld out of range", (long long)m);
    CRF_REQUIRE(H == 4 || H == 8 || H == 16 || H == 32 || H == 64, CRF_ERR_UNSUPPORTED,
                "hidden channels H=%d not in {4,8,16,32,64}", H);
    CRF_REQUIRE(K >= 1 && K <= 64 && k0 >= 0 && k0 < K, CRF_ERR_ARG, "K=%d k0=%d invalid", K, k0);
    return CRF_OK;
}

#define DISPATCH_H(H, ...)                         \
    switch (H) {                                   \
        case 4: { constexpr int HH = 4; __VA_ARGS__; break; }   \
        case 8: { constexpr int HH = 8; __VA_ARGS__; break; }   \
        case 16: { constexpr int HH = 16; __VA_ARGS__; break; } \
        case 32: { constexpr int HH = 32; __VA_ARGS__; break; } \
        default: { constexpr int HH = 64; __VA_ARGS__; break; } \
    }

}  // namespace crf

using namespace crf;

extern "C" int crfconv_meanfield_forward(const float* z, const float* y, const int32_t* idx32, int K,
                                         int k0, int64_t m, int H, const float* Q, const float* P,
                                         int T, float* s, float* zq, float* xs, crf_stream_t stream) {
    if (int rc = check_common(m, H, K, k0)) return rc;
    CRF_REQUIRE(z && y && idx32 && Q && P && s && zq && (xs || T == 0), CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(T >= 0, CRF_ERR_ARG, "T=%d < 0", T);
    hipStream_t st = as_stream(stream);
    DISPATCH_H(H, {
        const int ppb = points_per_block<HH>();
        const dim3 grid((unsigned)cdiv(m, ppb));
        switch (K - k0) {
            case 15: hipLaunchKernelGGL((sim_kernel<HH, 15>), grid, dim3(BLOCK), 0, st, y, z, idx32, K, k0, Q, s, zq, m); break;
            case 16: hipLaunchKernelGGL((sim_kernel<HH, 16>), grid, dim3(BLOCK), 0, st, y, z, idx32, K, k0, Q, s, zq, m); break;
            case 31: hipLaunchKernelGGL((sim_kernel<HH, 31>), grid, dim3(BLOCK), 0, st, y, z, idx32, K, k0, Q, s, zq, m); break;
            default: hipLaunchKernelGGL((sim_kernel<HH, 0>), grid, dim3(BLOCK), 0, st, y, z, idx32, K, k0, Q, s, zq, m); break;
        }
        CRF_LAUNCH_CHECK();
        const float* xin = z;
        for (int t = 0; t < T; ++t) {
            float* xout = xs + (int64_t)t * m * H;
            hipLaunchKernelGGL(step_kernel<HH>, grid, dim3(BLOCK), 0, st, xin, zq, s, idx32, K, k0, P,
                               xout, m);
            CRF_LAUNCH_CHECK();
            xin = xout;
        }
    });
    return CRF_OK;
}

extern "C" int crfconv_meanfield_bwd_edge(const float* G, const float* xprev, const float* s,
                                          const int32_t* idx32, int K, int k0, int64_t m, int H,
                                          const float* P, float* gm, float* ds, float* mt,
                                          int accumulate, crf_stream_t stream) {
    if (int rc = check_common(m, H, K, k0)) return rc;
    CRF_REQUIRE(G && xprev && s && idx32 && P && gm && ds, CRF_ERR_ARG, "null pointer");
    DISPATCH_H(H, {
        const dim3 grid((unsigned)cdiv(m, points_per_block<HH>()));
        hipLaunchKernelGGL(bwd_edge_kernel<HH>, grid, dim3(BLOCK), 0, as_stream(stream), G, xprev, s,
                           idx32, K, k0, P, gm, ds, mt, accumulate, m);
        CRF_LAUNCH_CHECK();
    });
    return CRF_OK;
}

extern "C" int crfconv_meanfield_bwd_scatter(const float* gm, const float* s, const int32_t* rev_ptr,
                                             const int32_t* rev_eid, int K, int k0, int64_t m_src,
                                             int H, const float* add, float* Gprev,
                                             crf_stream_t stream) {
    if (int rc = check_common(m_src, H, K, k0)) return rc;
    CRF_REQUIRE(gm && s && rev_ptr && rev_eid && Gprev, CRF_ERR_ARG, "null pointer");
    DISPATCH_H(H, {
        const dim3 grid((unsigned)cdiv(m_src, points_per_block<HH>()));
        hipLaunchKernelGGL(bwd_scatter_kernel<HH>, grid, dim3(BLOCK), 0, as_stream(stream), gm, s,
                           rev_ptr, rev_eid, K, k0, add, Gprev, m_src);
        CRF_LAUNCH_CHECK();
    });
    return CRF_OK;
}

extern "C" int crfconv_similarity_bwd(const float* ds, const float* s, const float* y,
                                      const int32_t* idx32, int K, int k0, int64_t m, int H, float* w,
                                      float* dy_self, crf_stream_t stream) {
    if (int rc = check_common(m, H, K, k0)) return rc;
    CRF_REQUIRE(ds && s && y && idx32 && w && dy_self, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(w != ds, CRF_ERR_ARG, "w must not alias ds");
    DISPATCH_H(H, {
        const dim3 grid((unsigned)cdiv(m, points_per_block<HH>()));
        hipLaunchKernelGGL(sim_bwd_kernel<HH>, grid, dim3(BLOCK), 0, as_stream(stream), ds, s, y, idx32,
                           K, k0, w, dy_self, m);
        CRF_LAUNCH_CHECK();
    });
    return CRF_OK;
}

extern "C" int crfconv_similarity_bwd_scatter(const float* w, const float* y, const float* dy_self,
                                              const int32_t* rev_ptr, const int32_t* rev_eid, int K,
                                              int k0, int64_t m_src, int H, float* dy,
                                              crf_stream_t stream) {
    if (int rc = check_common(m_src, H, K, k0)) return rc;
    CRF_REQUIRE(w && y && dy_self && rev_ptr && rev_eid && dy, CRF_ERR_ARG, "null pointer");
    DISPATCH_H(H, {
        const dim3 grid((unsigned)cdiv(m_src, points_per_block<HH>()));
        hipLaunchKernelGGL(sim_bwd_scatter_kernel<HH>, grid, dim3(BLOCK), 0, as_stream(stream), w, y,
                           dy_self, rev_ptr, rev_eid, K, k0, dy, m_src);
        CRF_LAUNCH_CHECK();
    });
    return CRF_OK;
}
