// CRF mean-field kernels for gfx950 (forward + the two halves of each backward step).
//
// Thread mapping (all kernels): a point's H channels are spread over L = H/4 adjacent lanes,
// one float4 (16-byte load) per lane, so a 64-lane wavefront carries 64/L points and a
// neighbour row is fetched by L lanes as one contiguous 4H-byte segment.  Per-point reductions
// (squared distance, dot products) are xor-shuffles over the L lanes; the H x H products use
// shuffles for the vector and LDS (float4 rows) for the matrix.
//
// Layout: every per-edge array (s, ds, w) is [m, K] with the SAME column index as the neighbour
// table (columns < k0 hold 0), i.e. it is addressed by the edge id e = i*K + k.  With K = 16 a
// row of indices and a row of weights are each one aligned 64-byte segment: the fast kernels
// (K = 16 / 32, k0 = 1) pull them with dwordx4 loads and keep them in registers, so all K-1
// neighbour gathers of a point are in flight together.
//
// Reference semantics: models/continuous_crf_conv_big.py:49-54 (similarity), :63-72 (loop).
#include "common.hpp"

#include <cstdlib>

namespace crf {

constexpr int BLOCK = 256;

// v (float4 per lane, quad q of the point's H-vector)  ->  acc + v_full * Mat, where
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

template <int H>
struct Geo {
    static constexpr int L = H / 4, PPW = WAVE / L, PPB = PPW * (BLOCK / WAVE);
};

template <int H>
__device__ __forceinline__ int64_t my_point(int64_t m, int& lane, int& q, bool& valid) {
    lane = threadIdx.x & 63;
    q = lane % Geo<H>::L;
    const int64_t row = (int64_t)xcd_block_id() * Geo<H>::PPB + (threadIdx.x >> 6) * Geo<H>::PPW + lane / Geo<H>::L;
    valid = row < m;
    return valid ? row : m - 1;
}

__device__ __forceinline__ float4 sub4(float4 a, float4 b) {
    return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
}

// K-wide row of 32-bit values (K % 4 == 0) as K/4 aligned dwordx4 loads.
template <int K, typename T4, typename T>
__device__ __forceinline__ void load_row(const T* __restrict__ p, T (&out)[K]) {
#pragma unroll
    for (int c = 0; c < K / 4; ++c) {
        const T4 v = reinterpret_cast<const T4*>(p)[c];
        out[4 * c + 0] = v.x; out[4 * c + 1] = v.y; out[4 * c + 2] = v.z; out[4 * c + 3] = v.w;
    }
}

// Index row in either layout: int32 global rows, or uint16 per-cloud local ids (half the bytes; valid when
// every cloud has <= 65536 source points) decoded as  cloud * n_src + id  with cloud = row / n_tgt.
template <int K>
__device__ __forceinline__ void load_index_row(const int32_t* __restrict__ idx32, const uint16_t* __restrict__ idx16,
                                               int64_t r, int n_tgt, int n_src, int (&j)[K]) {
    if (idx16 != nullptr) {
        const int base = (int)(r / n_tgt) * n_src;
        const uint4* p = reinterpret_cast<const uint4*>(idx16 + r * K);
#pragma unroll
        for (int c = 0; c < K / 8; ++c) {
            const uint4 v = p[c];
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                j[8 * c + 2 * e] = base + (int)(w[e] & 0xffffu);
                j[8 * c + 2 * e + 1] = base + (int)(w[e] >> 16);
            }
        }
    } else {
        load_row<K, int4>(idx32 + r * K, j);
    }
}

// K-wide rows of the wave's PPW points (PPW * K contiguous floats in memory) written as 1 KiB-contiguous stores.
// Stored straight from the owning lanes, each store instruction touches 64/L rows with 16 bytes each (measured on
// the level-0 first kernel: +2.1 us); routed through a per-wave LDS tile every instruction writes consecutive bytes.
// One call per kernel (the tile is a single static array per instantiation).
template <int H, int K>
__device__ __forceinline__ void store_rows_coalesced(const float (&d)[K], float* __restrict__ dst, int lane, int q,
                                                     int64_t m) {
    constexpr int L = Geo<H>::L, PPW = Geo<H>::PPW, CPR = K / 4, NCH = PPW * CPR;   // 16-byte chunks per row / wave
    __shared__ float4 tile[BLOCK / WAVE][NCH];
    float4* mine = tile[threadIdx.x >> 6];
    const int p = lane / L;
#pragma unroll
    for (int c = 0; c < CPR; ++c)
        if ((c % L) == q) mine[p * CPR + c] = make_float4(d[4 * c], d[4 * c + 1], d[4 * c + 2], d[4 * c + 3]);
    __builtin_amdgcn_wave_barrier();     // LDS operations of one wave complete in order
    const int64_t row0 = (int64_t)xcd_block_id() * Geo<H>::PPB + (threadIdx.x >> 6) * PPW;
#pragma unroll
    for (int c = lane; c < NCH; c += WAVE)
        if (row0 + c / CPR < m) st4(dst + row0 * K + 4 * c, mine[c]);
}

// ====================================================================== fast forward kernels
// (K in {16, 32}, k0 == 1).  FIRST = similarity + z Q + first step fused: the index row is read
// once, s never round-trips through memory before its first use.
template <int H, int K, bool WITH_STEP>
__global__ __launch_bounds__(BLOCK, (H == 8 && K == 16) ? 4 : 1) void sim_step_fast_kernel(const float* __restrict__ y,
                                                              const float* __restrict__ z,
                                                              const int32_t* __restrict__ idx,
                                                              const uint16_t* __restrict__ idx16, int n_tgt, int n_src,
                                                              const float* __restrict__ Q,
                                                              const float* __restrict__ P,
                                                              float* __restrict__ s,
                                                              float* __restrict__ x1, int64_t m) {
    constexpr int L = Geo<H>::L;
    __shared__ float4 sQ[WITH_STEP ? H * L : 1];
    __shared__ float4 sP[WITH_STEP ? H * L : 1];
    if constexpr (WITH_STEP) load_matrix<H>(sQ, Q, false);
    if constexpr (WITH_STEP) load_matrix<H>(sP, P, false);     // barrier deferred to the first matvec: the
    int lane, q;                                              // matrix fetch overlaps the row loads and gathers
    bool valid;
    const int64_t r = my_point<H>(m, lane, q, valid);

    int j[K];
    load_index_row<K>(idx, idx16, r, n_tgt, n_src, j);
    const float4 yi = ld4(y + r * H + 4 * q);
    float4 nb[K];
#pragma unroll
    for (int k = 1; k < K; ++k) nb[k] = ld4(y + (int64_t)j[k] * H + 4 * q);
    float d[K];
    float dmin = 3.4e38f;
#pragma unroll
    for (int k = 1; k < K; ++k) {
        const float4 df = sub4(yi, nb[k]);
        d[k] = group_sum<L>(dot4(df, df));
        dmin = fminf(dmin, d[k]);
    }
    if constexpr (WITH_STEP) {   // issue the z-row gathers before the exp chain
#pragma unroll
        for (int k = 1; k < K; ++k) nb[k] = ld4(z + (int64_t)j[k] * H + 4 * q);
    }
    float den = 0.f;
#pragma unroll
    for (int k = 1; k < K; ++k) {
        d[k] = __expf(dmin - d[k]);        // v_exp_f32 path: ~1e-7 relative, far inside the 1e-4 budget
        den += d[k];
    }
    const float inv = 1.0f / den;
    d[0] = 0.f;
#pragma unroll
    for (int k = 1; k < K; ++k) d[k] *= inv;
    if (s != nullptr) store_rows_coalesced<H, K>(d, s, lane, q, m);       // NULL: inference with T <= 1, nobody re-reads s

    if constexpr (WITH_STEP) {
        const float4 zi = ld4(z + r * H + 4 * q);
        float4 msg = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 1; k < K; ++k) msg = fma4(d[k], nb[k], msg);
        __syncthreads();
        const float4 zqi = matvec_acc<H>(zi, sQ, lane, q, make_float4(0.f, 0.f, 0.f, 0.f));
        const float4 o = matvec_acc<H>(msg, sP, lane, q, zqi);
        if (valid) st4(x1 + r * H + 4 * q, o);
    }
}

template <int H, int K>
__global__ __launch_bounds__(BLOCK) void step_fast_kernel(const float* __restrict__ xin,
                                                          const float* __restrict__ z,
                                                          const float* __restrict__ s,
                                                          const int32_t* __restrict__ idx,
                                                          const uint16_t* __restrict__ idx16, int n_tgt, int n_src,
                                                          const float* __restrict__ Q,
                                                          const float* __restrict__ P,
                                                          float* __restrict__ xout, int64_t m) {
    constexpr int L = Geo<H>::L;
    __shared__ float4 sP[H * L];
    __shared__ float4 sQ[H * L];
    load_matrix<H>(sP, P, false);
    load_matrix<H>(sQ, Q, false);                               // barrier deferred to the first matvec
    int lane, q;
    bool valid;
    const int64_t r = my_point<H>(m, lane, q, valid);
    int j[K];
    float w[K];
    load_index_row<K>(idx, idx16, r, n_tgt, n_src, j);
    load_row<K, float4>(s + r * K, w);
    const float4 zi = ld4(z + r * H + 4 * q);
    // gathers in two batches of K/2 (fewer live registers: 8 waves/SIMD at H = 8, K = 16; measured 0.2-0.3 us faster)
    float4 msg = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        float4 nb[K / 2];
#pragma unroll
        for (int k = 1; k < K / 2; ++k) nb[k] = ld4(xin + (int64_t)j[k] * H + 4 * q);
#pragma unroll
        for (int k = 1; k < K / 2; ++k) msg = fma4(w[k], nb[k], msg);
#pragma unroll
        for (int k = 0; k < K / 2; ++k) nb[k] = ld4(xin + (int64_t)j[K / 2 + k] * H + 4 * q);
#pragma unroll
        for (int k = 0; k < K / 2; ++k) msg = fma4(w[K / 2 + k], nb[k], msg);
    }
    __syncthreads();
    // z Q recomputed from z (same bytes as reading a stored z Q, and nothing extra to write)
    const float4 zqi = matvec_acc<H>(zi, sQ, lane, q, make_float4(0.f, 0.f, 0.f, 0.f));
    const float4 o = matvec_acc<H>(msg, sP, lane, q, zqi);
    if (valid) st4(xout + r * H + 4 * q, o);
}

// ====================================================================== LDS-window forward kernels
// With spatially sorted clouds (the device collate emits Morton order) ~80 % of a block's
// neighbour rows lie within +-HALO rows of the block itself.  The block stages that contiguous
// window with coalesced 16-byte loads and serves in-window neighbours from LDS (ds_read_b128);
// only the rest goes through the vector-memory gather path.  Correct for any point order -- an
// unsorted cloud just finds fewer neighbours in its window.
template <int H, int NT, int HALO>
struct Win {
    static constexpr int L = H / 4, PPW = WAVE / L, PPB = NT / L, ROWS = PPB + 2 * HALO;
    __device__ static __forceinline__ int64_t base(int64_t m) {
        int64_t first = (int64_t)xcd_block_id() * PPB - HALO;
        const int64_t hi = m - ROWS;
        if (first > hi) first = hi;
        if (first < 0) first = 0;
        return first;
    }
    __device__ static __forceinline__ void stage(float4* dst, const float* __restrict__ src, int64_t w0, int64_t m) {
        const float4* s4 = reinterpret_cast<const float4*>(src) + w0 * L;
        const int64_t lim = (m - w0) * L;
        for (int t = threadIdx.x; t < ROWS * L; t += NT)
            dst[t] = t < lim ? s4[t] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __device__ static __forceinline__ float4 fetch(const float4* lds, const float* __restrict__ src, int w0, int j, int q) {
        const int jl = j - w0;
        return (unsigned)jl < (unsigned)ROWS ? lds[jl * L + q] : ld4(src + (int64_t)j * H + 4 * q);
    }
    __device__ static __forceinline__ int64_t point(int64_t m, int& lane, int& q, bool& valid) {
        lane = threadIdx.x & 63;
        q = lane % L;
        const int64_t row = (int64_t)xcd_block_id() * PPB + (threadIdx.x >> 6) * PPW + lane / L;
        valid = row < m;
        return valid ? row : m - 1;
    }
};

template <int H, int NT>
__device__ __forceinline__ void load_matrix_nt(float4* sM, const float* __restrict__ Mat, bool transpose) {
    float* s = reinterpret_cast<float*>(sM);
    for (int t = threadIdx.x; t < H * H; t += NT) {
        const int h = t / H, c = t % H;
        s[t] = transpose ? Mat[c * H + h] : Mat[t];
    }
}

template <int H, int K, int NT, int HALO, bool WITH_STEP>
__global__ __launch_bounds__(NT) void sim_step_win_kernel(const float* __restrict__ y,
                                                          const float* __restrict__ z,
                                                          const int32_t* __restrict__ idx,
                                                          const float* __restrict__ Q,
                                                          const float* __restrict__ P,
                                                          float* __restrict__ s,
                                                          float* __restrict__ x1, int64_t m) {
    using W = Win<H, NT, HALO>;
    constexpr int L = W::L;
    __shared__ float4 sQ[H * L];
    __shared__ float4 sP[WITH_STEP ? H * L : 1];
    __shared__ float4 sY[W::ROWS * L];
    __shared__ float4 sZ[WITH_STEP ? W::ROWS * L : 1];
    const int64_t w0 = W::base(m);
    load_matrix_nt<H, NT>(sQ, Q, false);
    if constexpr (WITH_STEP) load_matrix_nt<H, NT>(sP, P, false);
    W::stage(sY, y, w0, m);
    if constexpr (WITH_STEP) W::stage(sZ, z, w0, m);
    int lane, q;
    bool valid;
    const int64_t r = W::point(m, lane, q, valid);
    int j[K];
    load_row<K, int4>(idx + r * K, j);
    const float4 yi = ld4(y + r * H + 4 * q);
    const float4 zi = ld4(z + r * H + 4 * q);
    __syncthreads();

    float4 nb[K];
#pragma unroll
    for (int k = 1; k < K; ++k) nb[k] = W::fetch(sY, y, (int)w0, j[k], q);
    float d[K];
    float dmin = 3.4e38f;
#pragma unroll
    for (int k = 1; k < K; ++k) {
        const float4 df = sub4(yi, nb[k]);
        d[k] = group_sum<L>(dot4(df, df));
        dmin = fminf(dmin, d[k]);
    }
    if constexpr (WITH_STEP) {
#pragma unroll
        for (int k = 1; k < K; ++k) nb[k] = W::fetch(sZ, z, (int)w0, j[k], q);
    }
    float den = 0.f;
#pragma unroll
    for (int k = 1; k < K; ++k) {
        d[k] = __expf(dmin - d[k]);        // v_exp_f32 path: ~1e-7 relative, far inside the 1e-4 budget
        den += d[k];
    }
    const float inv = 1.0f / den;
    d[0] = 0.f;
#pragma unroll
    for (int k = 1; k < K; ++k) d[k] *= inv;
#pragma unroll
    for (int c = 0; c < K / 4; ++c)
        if (valid && (c % L) == q) st4(s + r * K + 4 * c, make_float4(d[4 * c], d[4 * c + 1], d[4 * c + 2], d[4 * c + 3]));
    if constexpr (WITH_STEP) {
        const float4 zqi = matvec_acc<H>(zi, sQ, lane, q, make_float4(0.f, 0.f, 0.f, 0.f));
        float4 msg = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 1; k < K; ++k) msg = fma4(d[k], nb[k], msg);
        const float4 o = matvec_acc<H>(msg, sP, lane, q, zqi);
        if (valid) st4(x1 + r * H + 4 * q, o);
    }
}

template <int H, int K, int NT, int HALO>
__global__ __launch_bounds__(NT) void step_win_kernel(const float* __restrict__ xin,
                                                      const float* __restrict__ z,
                                                      const float* __restrict__ s,
                                                      const int32_t* __restrict__ idx,
                                                      const float* __restrict__ Q,
                                                      const float* __restrict__ P,
                                                      float* __restrict__ xout, int64_t m) {
    using W = Win<H, NT, HALO>;
    constexpr int L = W::L;
    __shared__ float4 sP[H * L];
    __shared__ float4 sQ[H * L];
    __shared__ float4 sX[W::ROWS * L];
    const int64_t w0 = W::base(m);
    load_matrix_nt<H, NT>(sP, P, false);
    load_matrix_nt<H, NT>(sQ, Q, false);
    W::stage(sX, xin, w0, m);
    int lane, q;
    bool valid;
    const int64_t r = W::point(m, lane, q, valid);
    int j[K];
    float w[K];
    load_row<K, int4>(idx + r * K, j);
    load_row<K, float4>(s + r * K, w);
    const float4 zi = ld4(z + r * H + 4 * q);
    __syncthreads();
    const float4 zqi = matvec_acc<H>(zi, sQ, lane, q, make_float4(0.f, 0.f, 0.f, 0.f));
    float4 msg = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 1; k < K; ++k) msg = fma4(w[k], W::fetch(sX, xin, (int)w0, j[k], q), msg);
    const float4 o = matvec_acc<H>(msg, sP, lane, q, zqi);
    if (valid) st4(xout + r * H + 4 * q, o);
}

// ====================================================================== generic forward kernels
// any K <= 64, any k0: distances recomputed in a second sweep (rows are L1/L2 hot by then).
template <int H>
__global__ __launch_bounds__(BLOCK) void sim_kernel(const float* __restrict__ y,
                                                    const int32_t* __restrict__ idx, int K, int k0,
                                                    float* __restrict__ s, int64_t m) {
    constexpr int L = Geo<H>::L;
    int lane, q;
    bool valid;
    const int64_t r = my_point<H>(m, lane, q, valid);
    const float4 yi = ld4(y + r * H + 4 * q);
    const int32_t* irow = idx + r * K;
    float* srow = s + r * K;

    // entries < 0 mark "no neighbour" (padded variable-degree tables of the sparse operators)
    auto dist_to = [&](int k, bool& have) {
        const int j = irow[k];
        have = j >= 0;
        const float4 df = sub4(yi, ld4(y + (int64_t)(have ? j : 0) * H + 4 * q));
        return group_sum<L>(dot4(df, df));
    };
    float dmin = 3.4e38f;
    for (int k = k0; k < K; ++k) {
        bool have;
        const float dk = dist_to(k, have);
        if (have) dmin = fminf(dmin, dk);
    }
    float den = 0.f;
    for (int k = k0; k < K; ++k) {
        bool have;
        const float dk = dist_to(k, have);
        const float e = have ? expf(dmin - dk) : 0.f;
        den += e;
        if (valid && q == 0) srow[k] = e;
    }
    const float inv = den > 0.f ? 1.0f / den : 0.f;      // isolated point: no message
    if (valid && q == 0) {  // same lane re-reads what it wrote
        for (int k = 0; k < k0; ++k) srow[k] = 0.f;
        for (int k = k0; k < K; ++k) srow[k] *= inv;
    }
}

template <int H>
__global__ __launch_bounds__(BLOCK) void step_kernel(const float* __restrict__ xin,
                                                     const float* __restrict__ z,
                                                     const float* __restrict__ s,
                                                     const int32_t* __restrict__ idx, int K, int k0,
                                                     const float* __restrict__ Q,
                                                     const float* __restrict__ P,
                                                     float* __restrict__ xout, int64_t m) {
    constexpr int L = Geo<H>::L;
    __shared__ float4 sP[H * L];
    __shared__ float4 sQ[H * L];
    load_matrix<H>(sP, P, false);
    load_matrix<H>(sQ, Q, false);
    __syncthreads();
    int lane, q;
    bool valid;
    const int64_t r = my_point<H>(m, lane, q, valid);
    const int32_t* irow = idx + r * K;
    const float* srow = s + r * K;
    float4 msg = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int k = k0; k < K; ++k) {
        const int j = irow[k];
        if (j >= 0) msg = fma4(srow[k], ld4(xin + (int64_t)j * H + 4 * q), msg);
    }
    const float4 zqi = matvec_acc<H>(ld4(z + r * H + 4 * q), sQ, lane, q, make_float4(0.f, 0.f, 0.f, 0.f));
    const float4 o = matvec_acc<H>(msg, sP, lane, q, zqi);
    if (valid) st4(xout + r * H + 4 * q, o);
}

// ====================================================================== backward, edge half
// gm = G P^T ; ds[i,k] (+)= <gm_i, xprev_j> ; mt_i = sum_k s_ik xprev_j.   KT > 0: fast row form.
template <int H, int KT>
__global__ __launch_bounds__(BLOCK) void bwd_edge_kernel(const float* __restrict__ G,
                                                         const float* __restrict__ xprev,
                                                         const float* __restrict__ s,
                                                         const int32_t* __restrict__ idx, int K,
                                                         int k0, const float* __restrict__ P,
                                                         float* __restrict__ gm,
                                                         float* __restrict__ ds,
                                                         float* __restrict__ mt, int accumulate,
                                                         int64_t m) {
    constexpr int L = Geo<H>::L;
    __shared__ float4 sPT[H * L];
    load_matrix<H>(sPT, P, true);
    __syncthreads();
    int lane, q;
    bool valid;
    const int64_t r = my_point<H>(m, lane, q, valid);
    const float4 g = ld4(G + r * H + 4 * q);
    const float4 gmi = matvec_acc<H>(g, sPT, lane, q, make_float4(0.f, 0.f, 0.f, 0.f));
    if (valid) st4(gm + r * H + 4 * q, gmi);
    float4 msg = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (KT > 0) {
        int j[KT];
        float w[KT], dd[KT];
        load_row<KT, int4>(idx + r * KT, j);
        load_row<KT, float4>(s + r * KT, w);
        float4 nb[KT];
#pragma unroll
        for (int k = 1; k < KT; ++k) nb[k] = ld4(xprev + (int64_t)j[k] * H + 4 * q);
        if (accumulate) load_row<KT, float4>(ds + r * KT, dd);
        else {
#pragma unroll
            for (int k = 0; k < KT; ++k) dd[k] = 0.f;
        }
#pragma unroll
        for (int k = 1; k < KT; ++k) {
            msg = fma4(w[k], nb[k], msg);
            dd[k] += group_sum<L>(dot4(gmi, nb[k]));
        }
        dd[0] = 0.f;
        store_rows_coalesced<H, KT>(dd, ds, lane, q, m);
    } else {
        const int32_t* irow = idx + r * K;
        const float* srow = s + r * K;
        float* dsrow = ds + r * K;
#pragma unroll 4
        for (int k = k0; k < K; ++k) {
            const int j = irow[k];
            const bool have = j >= 0;
            float4 xj = ld4(xprev + (int64_t)(have ? j : 0) * H + 4 * q);
            if (!have) xj = make_float4(0.f, 0.f, 0.f, 0.f);
            msg = fma4(srow[k], xj, msg);
            const float dotv = group_sum<L>(dot4(gmi, xj));
            if (valid && q == (k % L)) dsrow[k] = accumulate ? dsrow[k] + dotv : dotv;
        }
        if (valid && q == 0 && !accumulate)
            for (int k = 0; k < k0; ++k) dsrow[k] = 0.f;
    }
    if (mt != nullptr && valid) st4(mt + r * H + 4 * q, msg);
}

// ====================================================================== backward, scatter half
// One source row per L lanes; walks the row's incoming edges (ascending edge id, fixed order).
// Gprev[j] = add[j] + sum_{e in rev(j)} s[e] gm[e / K]     (s is 0 on columns < k0)
// EP edge-lanes per row walk the row's incoming edges EP at a time (edge p = beg + lane-group, += EP) and fold their
// partial sums by xor-shuffles (fixed tree): in-degrees of a kNN graph spread from 0 to ~40, and with one lane
// group per row a wavefront iterates to the LARGEST in-degree of its 64/L rows.
template <int H>
struct Scat {
    static constexpr int L = H / 4, EP = (L <= 4) ? 4 : 1, RPW = WAVE / (L * EP), RPB = RPW * (BLOCK / WAVE);
};

template <int H>
__device__ __forceinline__ float4 fold_edge_lanes(float4 a) {
    constexpr int L = Scat<H>::L;
#pragma unroll
    for (int o = L; o < L * Scat<H>::EP; o <<= 1) {
        a.x += __shfl_xor(a.x, o, WAVE); a.y += __shfl_xor(a.y, o, WAVE);
        a.z += __shfl_xor(a.z, o, WAVE); a.w += __shfl_xor(a.w, o, WAVE);
    }
    return a;
}

template <int H>
__global__ __launch_bounds__(BLOCK) void bwd_scatter_kernel(const float* __restrict__ gm,
                                                            const float* __restrict__ s,
                                                            const int32_t* __restrict__ rev_ptr,
                                                            const int32_t* __restrict__ rev_eid,
                                                            int K, int kshift,
                                                            const float* __restrict__ add,
                                                            float* __restrict__ Gprev,
                                                            int64_t m_src) {
    constexpr int L = Scat<H>::L, EP = Scat<H>::EP;
    const int lane = threadIdx.x & 63;
    const int q = lane % L, el = (lane / L) % EP;
    int64_t row = (int64_t)xcd_block_id() * Scat<H>::RPB + (threadIdx.x >> 6) * Scat<H>::RPW + lane / (L * EP);
    const bool valid = row < m_src;
    if (!valid) row = m_src - 1;                    // keep every lane in the shuffles below
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int beg = rev_ptr[row], end = valid ? rev_ptr[row + 1] : beg;
#pragma unroll 2
    for (int p = beg + el; p < end; p += EP) {
        const int e = rev_eid[p];
        const int i = kshift >= 0 ? (e >> kshift) : (e / K);
        acc = fma4(s[e], ld4(gm + (int64_t)i * H + 4 * q), acc);
    }
    acc = fold_edge_lanes<H>(acc);
    if (valid && el == 0) {
        if (add) {
            const float4 a0 = ld4(add + row * H + 4 * q);
            acc = make_float4(acc.x + a0.x, acc.y + a0.y, acc.z + a0.z, acc.w + a0.w);
        }
        st4(Gprev + row * H + 4 * q, acc);
    }
}

// ====================================================================== softmax / distance backward
template <int H>
__global__ __launch_bounds__(BLOCK) void sim_bwd_kernel(const float* __restrict__ ds,
                                                        const float* __restrict__ s,
                                                        const float* __restrict__ y,
                                                        const int32_t* __restrict__ idx, int K,
                                                        int k0, float* __restrict__ w,
                                                        float* __restrict__ dy_self, int64_t m) {
    constexpr int L = Geo<H>::L;
    int lane, q;
    bool valid;
    const int64_t r = my_point<H>(m, lane, q, valid);
    const int32_t* irow = idx + r * K;
    const float* srow = s + r * K;
    const float* dsrow = ds + r * K;
    float* wrow = w + r * K;
    float dotv = 0.f;
    for (int k = k0; k < K; ++k) dotv = fmaf(srow[k], dsrow[k], dotv);
    const float4 yi = ld4(y + r * H + 4 * q);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int k = k0; k < K; ++k) {
        // d loss / d logit_k = s_k (ds_k - dot);  logit = -dist  =>  d/d dist = -that;  w = 2 * d/d dist
        const float wk = -2.0f * srow[k] * (dsrow[k] - dotv);      // 0 on missing entries (s == 0)
        const int j = irow[k];
        acc = fma4(wk, sub4(yi, ld4(y + (int64_t)(j >= 0 ? j : 0) * H + 4 * q)), acc);
        if (valid && q == (k % L)) wrow[k] = wk;
    }
    if (valid && q == 0)
        for (int k = 0; k < k0; ++k) wrow[k] = 0.f;
    if (valid) st4(dy_self + r * H + 4 * q, acc);
}

// fast form (K = 16, k0 = 1): index / s / ds rows as aligned dwordx4 loads, all K-1 gathers in flight, w rows
// written through the per-wave LDS tile
template <int H, int K>
__global__ __launch_bounds__(BLOCK) void sim_bwd_fast_kernel(const float* __restrict__ ds, const float* __restrict__ s,
                                                             const float* __restrict__ y,
                                                             const int32_t* __restrict__ idx, float* __restrict__ w,
                                                             float* __restrict__ dy_self, int64_t m) {
    int lane, q;
    bool valid;
    const int64_t r = my_point<H>(m, lane, q, valid);
    int j[K];
    float sw[K], dd[K];
    load_row<K, int4>(idx + r * K, j);
    const float4 yi = ld4(y + r * H + 4 * q);
    float4 nb[K];
#pragma unroll
    for (int k = 1; k < K; ++k) nb[k] = ld4(y + (int64_t)j[k] * H + 4 * q);
    load_row<K, float4>(s + r * K, sw);
    load_row<K, float4>(ds + r * K, dd);
    float dotv = 0.f;
#pragma unroll
    for (int k = 1; k < K; ++k) dotv = fmaf(sw[k], dd[k], dotv);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    dd[0] = 0.f;
#pragma unroll
    for (int k = 1; k < K; ++k) {
        const float wk = -2.0f * sw[k] * (dd[k] - dotv);
        acc = fma4(wk, sub4(yi, nb[k]), acc);
        dd[k] = wk;
    }
    store_rows_coalesced<H, K>(dd, w, lane, q, m);
    if (valid) st4(dy_self + r * H + 4 * q, acc);
}

// dy[j] = dy_self[j] + sum_{e in rev(j)} w[e] (y_j - y_{e/K})       (w is 0 on columns < k0)
template <int H>
__global__ __launch_bounds__(BLOCK) void sim_bwd_scatter_kernel(const float* __restrict__ w,
                                                                const float* __restrict__ y,
                                                                const float* __restrict__ dy_self,
                                                                const int32_t* __restrict__ rev_ptr,
                                                                const int32_t* __restrict__ rev_eid,
                                                                int K, int kshift,
                                                                float* __restrict__ dy,
                                                                int64_t m_src) {
    constexpr int L = Scat<H>::L, EP = Scat<H>::EP;
    const int lane = threadIdx.x & 63;
    const int q = lane % L, el = (lane / L) % EP;
    int64_t row = (int64_t)xcd_block_id() * Scat<H>::RPB + (threadIdx.x >> 6) * Scat<H>::RPW + lane / (L * EP);
    const bool valid = row < m_src;
    if (!valid) row = m_src - 1;
    const float4 yj = ld4(y + row * H + 4 * q);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int beg = rev_ptr[row], end = valid ? rev_ptr[row + 1] : beg;
#pragma unroll 2
    for (int p = beg + el; p < end; p += EP) {
        const int e = rev_eid[p];
        const int i = kshift >= 0 ? (e >> kshift) : (e / K);
        acc = fma4(w[e], sub4(yj, ld4(y + (int64_t)i * H + 4 * q)), acc);
    }
    acc = fold_edge_lanes<H>(acc);
    if (valid && el == 0) {
        const float4 a0 = ld4(dy_self + row * H + 4 * q);
        st4(dy + row * H + 4 * q, make_float4(acc.x + a0.x, acc.y + a0.y, acc.z + a0.z, acc.w + a0.w));
    }
}

static int check_common(int64_t m, int H, int K, int k0) {
    CRF_REQUIRE(m > 0 && m < (int64_t)1 << 31, CRF_ERR_ARG, "rows m=%lld out of range", (long long)m);
    CRF_REQUIRE(H == 4 || H == 8 || H == 16 || H == 32 || H == 64, CRF_ERR_UNSUPPORTED,
                "hidden channels H=%d not in {4,8,16,32,64}", H);
    CRF_REQUIRE(K >= 1 && K <= 64 && k0 >= 0 && k0 < K, CRF_ERR_ARG, "K=%d k0=%d invalid", K, k0);
    CRF_REQUIRE(m * K < (int64_t)1 << 31, CRF_ERR_ARG, "edge ids exceed int32 (m=%lld K=%d)", (long long)m, K);
    return CRF_OK;
}

// The LDS-window kernels measured SLOWER than the plain gather kernels on MI355X (43.0 vs 39.5 us for the
// level-0 forward, profiles/r1b): after Morton ordering + XCD-contiguous blocks the gathers are L1/L2 hits
// and the step is bound by the idx / weight / state streams, not by gather issue.  Kept for A/B runs:
// CRFCONV_WINDOW=1 enables them.
static const bool g_use_window = (getenv("CRFCONV_WINDOW") != nullptr);

static int kshift_of(int K) {
    for (int sft = 0; sft < 7; ++sft)
        if ((1 << sft) == K) return sft;
    return -1;
}

#define DISPATCH_H(H, ...)                                      \
    switch (H) {                                                \
        case 4: { constexpr int HH = 4; __VA_ARGS__; break; }   \
        case 8: { constexpr int HH = 8; __VA_ARGS__; break; }   \
        case 16: { constexpr int HH = 16; __VA_ARGS__; break; } \
        case 32: { constexpr int HH = 32; __VA_ARGS__; break; } \
        default: { constexpr int HH = 64; __VA_ARGS__; break; } \
    }

}  // namespace crf

using namespace crf;

static int meanfield_forward_impl(const float* z, const float* y, const int32_t* idx32, const uint16_t* idx16,
                                  int n_tgt, int n_src, int K, int k0, int64_t m, int H, const float* Q,
                                  const float* P, int T, float* s, float* xs, crf_stream_t stream);

extern "C" int crfconv_meanfield_forward(const float* z, const float* y, const int32_t* idx32, int K,
                                         int k0, int64_t m, int H, const float* Q, const float* P,
                                         int T, float* s, float* xs, crf_stream_t stream) {
    return meanfield_forward_impl(z, y, idx32, nullptr, 1, 1, K, k0, m, H, Q, P, T, s, xs, stream);
}

extern "C" int crfconv_meanfield_forward_u16(const float* z, const float* y, const int32_t* idx32,
                                             const uint16_t* idx16, int n_tgt, int n_src, int K, int k0,
                                             int64_t m, int H, const float* Q, const float* P, int T, float* s,
                                             float* xs, crf_stream_t stream) {
    CRF_REQUIRE(idx16 == nullptr || (n_tgt > 0 && n_src > 0 && n_src <= 65536 && m % n_tgt == 0), CRF_ERR_ARG,
                "u16 table needs n_src <= 65536 and m a multiple of n_tgt (n_tgt=%d n_src=%d)", n_tgt, n_src);
    return meanfield_forward_impl(z, y, idx32, idx16, n_tgt, n_src, K, k0, m, H, Q, P, T, s, xs, stream);
}

static int meanfield_forward_impl(const float* z, const float* y, const int32_t* idx32, const uint16_t* idx16,
                                  int n_tgt, int n_src, int K, int k0, int64_t m, int H, const float* Q,
                                  const float* P, int T, float* s, float* xs, crf_stream_t stream) {
    if (int rc = check_common(m, H, K, k0)) return rc;
    CRF_REQUIRE(z && y && idx32 && Q && P && (xs || T == 0), CRF_ERR_ARG, "null pointer");
    // s may be NULL when nothing reads it back: a single fused step (T == 1) on the fast path, no backward pass
    CRF_REQUIRE(s || (T == 1 && k0 == 1 && (K == 16 || K == 32) && !g_use_window), CRF_ERR_ARG,
                "s == NULL needs T == 1 on the fused first-step kernel (K in {16, 32}, k0 == 1)");
    CRF_REQUIRE(T >= 0, CRF_ERR_ARG, "T=%d < 0", T);
    hipStream_t st = as_stream(stream);
    const bool fast = (k0 == 1) && (K == 16 || K == 32);
    DISPATCH_H(H, {
        const dim3 grid((unsigned)cdiv(m, Geo<HH>::PPB)), blk(BLOCK);
        int t0 = 0;
        constexpr bool HAS_WIN = (HH == 8 || HH == 16);
        constexpr int NT = HH == 8 ? 512 : 256, HALO = HH == 8 ? 128 : 96;
        const bool win = fast && K == 16 && HAS_WIN && g_use_window;
        if (win) {
            if constexpr (HAS_WIN) {
                using W = Win<HH, NT, HALO>;
                const dim3 wgrid((unsigned)cdiv(m, W::PPB)), wblk(NT);
                if (T > 0) hipLaunchKernelGGL((sim_step_win_kernel<HH, 16, NT, HALO, true>), wgrid, wblk, 0, st, y, z, idx32, Q, P, s, xs, m);
                else hipLaunchKernelGGL((sim_step_win_kernel<HH, 16, NT, HALO, false>), wgrid, wblk, 0, st, y, z, idx32, Q, P, s, xs, m);
                CRF_LAUNCH_CHECK();
                for (int t = 1; t < T; ++t) {
                    hipLaunchKernelGGL((step_win_kernel<HH, 16, NT, HALO>), wgrid, wblk, 0, st, xs + (int64_t)(t - 1) * m * HH,
                                       z, s, idx32, Q, P, xs + (int64_t)t * m * HH, m);
                    CRF_LAUNCH_CHECK();
                }
            }
            return CRF_OK;
        }
        if (fast) {
            float* x1 = T > 0 ? xs : nullptr;
            if (K == 16) {
                if (T > 0) hipLaunchKernelGGL((sim_step_fast_kernel<HH, 16, true>), grid, blk, 0, st, y, z, idx32, idx16, n_tgt, n_src, Q, P, s, x1, m);
                else hipLaunchKernelGGL((sim_step_fast_kernel<HH, 16, false>), grid, blk, 0, st, y, z, idx32, idx16, n_tgt, n_src, Q, P, s, x1, m);
            } else {
                if (T > 0) hipLaunchKernelGGL((sim_step_fast_kernel<HH, 32, true>), grid, blk, 0, st, y, z, idx32, idx16, n_tgt, n_src, Q, P, s, x1, m);
                else hipLaunchKernelGGL((sim_step_fast_kernel<HH, 32, false>), grid, blk, 0, st, y, z, idx32, idx16, n_tgt, n_src, Q, P, s, x1, m);
            }
            t0 = 1;
        } else {
            hipLaunchKernelGGL(sim_kernel<HH>, grid, blk, 0, st, y, idx32, K, k0, s, m);
        }
        CRF_LAUNCH_CHECK();
        for (int t = t0; t < T; ++t) {
            const float* xin = t == 0 ? z : xs + (int64_t)(t - 1) * m * H;
            float* xout = xs + (int64_t)t * m * H;
            if (fast && K == 16) hipLaunchKernelGGL((step_fast_kernel<HH, 16>), grid, blk, 0, st, xin, z, s, idx32, idx16, n_tgt, n_src, Q, P, xout, m);
            else if (fast) hipLaunchKernelGGL((step_fast_kernel<HH, 32>), grid, blk, 0, st, xin, z, s, idx32, idx16, n_tgt, n_src, Q, P, xout, m);
            else hipLaunchKernelGGL(step_kernel<HH>, grid, blk, 0, st, xin, z, s, idx32, K, k0, Q, P, xout, m);
            CRF_LAUNCH_CHECK();
        }
    });
    return CRF_OK;
}

// One step with GIVEN edge weights: xout = z Q + (sum_k s_ik xin_{j(i,k)}) P  (generic kernel: any K <= 64, k0,
// entries < 0 = no neighbour).  The discrete CRF layer (models/discrete_crf_conv.py:57-61) runs on this.
extern "C" int crfconv_meanfield_step(const float* xin, const float* z, const float* s, const int32_t* idx32, int K,
                                      int k0, int64_t m, int H, const float* Q, const float* P, float* xout,
                                      crf_stream_t stream) {
    if (int rc = check_common(m, H, K, k0)) return rc;
    CRF_REQUIRE(xin && z && s && idx32 && Q && P && xout, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(xin != xout, CRF_ERR_ARG, "xout must not alias xin");
    DISPATCH_H(H, {
        const dim3 grid((unsigned)cdiv(m, Geo<HH>::PPB)), blk(BLOCK);
        hipLaunchKernelGGL(step_kernel<HH>, grid, blk, 0, as_stream(stream), xin, z, s, idx32, K, k0, Q, P, xout, m);
        CRF_LAUNCH_CHECK();
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
        const dim3 grid((unsigned)cdiv(m, Geo<HH>::PPB)), blk(BLOCK);
        if (k0 == 1 && K == 16)
            hipLaunchKernelGGL((bwd_edge_kernel<HH, 16>), grid, blk, 0, as_stream(stream), G, xprev, s, idx32, K, k0, P, gm, ds, mt, accumulate, m);
        else
            hipLaunchKernelGGL((bwd_edge_kernel<HH, 0>), grid, blk, 0, as_stream(stream), G, xprev, s, idx32, K, k0, P, gm, ds, mt, accumulate, m);
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
        const dim3 grid((unsigned)cdiv(m_src, Scat<HH>::RPB));
        hipLaunchKernelGGL(bwd_scatter_kernel<HH>, grid, dim3(BLOCK), 0, as_stream(stream), gm, s,
                           rev_ptr, rev_eid, K, kshift_of(K), add, Gprev, m_src);
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
        const dim3 grid((unsigned)cdiv(m, Geo<HH>::PPB));
        if constexpr (HH <= 16) {
            if (k0 == 1 && K == 16) {
                hipLaunchKernelGGL((sim_bwd_fast_kernel<HH, 16>), grid, dim3(BLOCK), 0, as_stream(stream), ds, s, y, idx32,
                                   w, dy_self, m);
                CRF_LAUNCH_CHECK();
                return CRF_OK;
            }
        }
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
        const dim3 grid((unsigned)cdiv(m_src, Scat<HH>::RPB));
        hipLaunchKernelGGL(sim_bwd_scatter_kernel<HH>, grid, dim3(BLOCK), 0, as_stream(stream), w, y,
                           dy_self, rev_ptr, rev_eid, K, kshift_of(K), dy, m_src);
        CRF_LAUNCH_CHECK();
    });
    return CRF_OK;
}
