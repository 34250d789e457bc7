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
#include "crf_common.hpp"

#ifndef CRF_GATHER_SRD
#define CRF_GATHER_SRD 1      // neighbour rows through a buffer resource with 32-bit byte offsets (0: generic 64-bit pointers, the A/B baseline)
#endif

#ifndef SIM_WAVES_
#define SIM_WAVES_ 4          // wavefronts per SIMD the level-0 first kernel is compiled for (A/B: 5 = the whole 4 x 40960 grid co-resident)
#endif

namespace crf {


// ====================================================================== fast forward kernels
// (K in {16, 32}, k0 == 1).  FIRST = similarity + z Q + first step fused: the index row is read
// once, s never round-trips through memory before its first use.
template <int H, int K, bool WITH_STEP, bool U16>
__global__ __launch_bounds__(BLOCK, (H == 8 && K == 16) ? SIM_WAVES_ : 1) void sim_step_fast_kernel(const float* __restrict__ y,
                                                              const float* __restrict__ z,
                                                              const int32_t* __restrict__ idx,
                                                              const uint16_t* __restrict__ idx16, int n_tgt, int n_src,
                                                              const float* __restrict__ Q,
                                                              const float* __restrict__ P,
                                                              float* __restrict__ s,
                                                              float* __restrict__ x1, int64_t m) {
    constexpr int L = Geo<H>::L;
    __shared__ float4 sQ[WITH_STEP ? MatStage<H>::F4 : 1];
    __shared__ float4 sP[WITH_STEP ? MatStage<H>::F4 : 1];
    int lane, q;
    bool valid;
    const int64_t r = my_point<H>(m, lane, q, valid);

    // issue order = arrival order: the index row first (the gathers wait for nothing else), then the own rows and the
    // matrices, which are only needed behind the gathers
#if CRF_GATHER_SRD
    int off[K];                                   // byte offsets of the neighbour rows: buffer loads, 32-bit address arithmetic
    load_index_offsets_t<K, U16, H>(idx, idx16, r, n_tgt, n_src, q, off);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(y, (int)(m * H * 4));
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rz = make_rsrc(z, (int)(m * H * 4));
#else
    int j[K];
    load_index_row_t<K, U16>(idx, idx16, r, n_tgt, n_src, j);
#endif
    const float4 yi = ld4(y + r * H + 4 * q);
    [[maybe_unused]] float4 zi = make_float4(0.f, 0.f, 0.f, 0.f);
    [[maybe_unused]] MatStage<H> mq, mp;
    if constexpr (WITH_STEP) {
        zi = ld4(z + r * H + 4 * q);
        mq.fetch(Q, false);
        mp.fetch(P, false);
    }
    float4 nb[K];
#pragma unroll
#if CRF_GATHER_SRD
    for (int k = 1; k < K; ++k) nb[k] = ld4_buf(ry, off[k]);
#else
    for (int k = 1; k < K; ++k) nb[k] = ld4(y + (int64_t)j[k] * H + 4 * q);
#endif
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
#if CRF_GATHER_SRD
        for (int k = 1; k < K; ++k) nb[k] = ld4_buf(rz, off[k]);
#else
        for (int k = 1; k < K; ++k) nb[k] = ld4(z + (int64_t)j[k] * H + 4 * q);
#endif
        mq.park(sQ);
        mp.park(sP);
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
        float4 msg = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 1; k < K; ++k) msg = fma4(d[k], nb[k], msg);
        __syncthreads();
        const float4 zqi = matvec_acc<H>(zi, sQ, lane, q, make_float4(0.f, 0.f, 0.f, 0.f));
        const float4 o = matvec_acc<H>(msg, sP, lane, q, zqi);
        if (valid) st4(x1 + r * H + 4 * q, o);
    }
}

template <int H, int K, bool U16>
__global__ __launch_bounds__(BLOCK) void step_fast_kernel(const float* __restrict__ xin,
                                                          const float* __restrict__ z,
                                                          const float* __restrict__ s,
                                                          const int32_t* __restrict__ idx,
                                                          const uint16_t* __restrict__ idx16, int n_tgt, int n_src,
                                                          const float* __restrict__ Q,
                                                          const float* __restrict__ P,
                                                          float* __restrict__ xout, int64_t m) {
    __shared__ float4 sP[MatStage<H>::F4];
    __shared__ float4 sQ[MatStage<H>::F4];
    int lane, q;
    bool valid;
    const int64_t r = my_point<H>(m, lane, q, valid);
    float w[K];
#if CRF_GATHER_SRD
    int off[K];
    load_index_offsets_t<K, U16, H>(idx, idx16, r, n_tgt, n_src, q, off);  // first: the gathers wait for this row only
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(xin, (int)(m * H * 4));
#define CRF_GATHER_X(k) ld4_buf(rx, off[k])
#else
    int j[K];
    load_index_row_t<K, U16>(idx, idx16, r, n_tgt, n_src, j);              // first: the gathers wait for this row only
#define CRF_GATHER_X(k) ld4(xin + (int64_t)j[k] * H + 4 * q)
#endif
    load_row<K, float4>(s + r * K, w);
    const float4 zi = ld4(z + r * H + 4 * q);
    MatStage<H> mp, mq;
    mp.fetch(P, false);
    mq.fetch(Q, false);
    // gathers in two batches of K/2 (fewer live registers: 8 waves/SIMD at H = 8, K = 16; measured 0.2-0.3 us faster)
    float4 msg = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        float4 nb[K / 2];
#pragma unroll
        for (int k = 1; k < K / 2; ++k) nb[k] = CRF_GATHER_X(k);
        mp.park(sP);
        mq.park(sQ);
#pragma unroll
        for (int k = 1; k < K / 2; ++k) msg = fma4(w[k], nb[k], msg);
#pragma unroll
        for (int k = 0; k < K / 2; ++k) nb[k] = CRF_GATHER_X(K / 2 + k);
#pragma unroll
        for (int k = 0; k < K / 2; ++k) msg = fma4(w[K / 2 + k], nb[k], msg);
    }
    __syncthreads();
    // z Q recomputed from z (same bytes as reading a stored z Q, and nothing extra to write)
    const float4 zqi = matvec_acc<H>(zi, sQ, lane, q, make_float4(0.f, 0.f, 0.f, 0.f));
    const float4 o = matvec_acc<H>(msg, sP, lane, q, zqi);
    if (valid) st4(xout + r * H + 4 * q, o);
#undef CRF_GATHER_X
}

// (The one-launch forward -- all T steps behind grid barriers, mf_fused_kernel -- and the LDS-window kernels of rounds 1-3 were
// measured slower than the per-step launches (39-40 us against 25 us; 43.0 against 39.5 us: DESIGN.md 5c, profiles/r2a_*) and
// were removed in round 4; `git log -- crfconv_amd/csrc/crf.hip` has them.)

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
    constexpr int UB = SCAT_UB;                     // edge ids, then weights + rows, in batches (see bwd_chain_kernel)
    for (int p0 = beg + el; p0 < end; p0 += EP * UB) {
        int e[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int p = p0 + u * EP;
            e[u] = p < end ? rev_eid[p] : -1;
        }
        float we[UB];
        float4 g[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int ee = e[u] >= 0 ? e[u] : 0;
            const int i = kshift >= 0 ? (ee >> kshift) : (ee / K);
            we[u] = e[u] >= 0 ? w[ee] : 0.f;
            g[u] = ld4(y + (int64_t)i * H + 4 * q);
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) acc = fma4(we[u], sub4(yj, g[u]), acc);
    }
    acc = fold_edge_lanes<H>(acc);
    if (valid && el == 0) {
        const float4 a0 = ld4(dy_self + row * H + 4 * q);
        st4(dy + row * H + 4 * q, make_float4(acc.x + a0.x, acc.y + a0.y, acc.z + a0.z, acc.w + a0.w));
    }
}

// ====================================================================== wide rows (H = 128, 256)
// The 128- and 256-channel decoder stages of the sparse networks (models/point_conv.py:318-339: GCRFConv(512, 256),
// GCRFConv(256, 128)) sit on the coarsest point sets (a few thousand points at most).  Their H x H matrices no longer
// fit the LDS tiles of the kernels above, and the H x H products are genuinely dense [m, H] x [H, H] contractions: they
// run as plain library GEMMs (ops/crf.py), while everything that touches the graph stays here -- ONE point per wavefront,
// VW = H / 64 channels per lane, any K <= 64 / k0, entries < 0 = no neighbour:
//   wide_sim      s = softmax_k(-|y_i - y_j|^2)            wide_agg          m_i = sum_k s_ik x_j
//   wide_bwd_edge ds (+)= <gm_i, x_j>                      wide_bwd_scatter  G'[j] = sum_{e in rev(j)} s[e] gm[e / K]
//   wide_sim_bwd  w = -2 s (ds - <s, ds>), dy_self         wide_sim_bwd_scatter  dy[j] = dy_self[j] + sum w[e] (y_j - y_i)
template <int VW>
struct WideVec {
    float v[VW];
    __device__ __forceinline__ void load(const float* __restrict__ p, int lane) {
        if constexpr (VW == 4) { const float4 t = ld4(p + 4 * lane); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
        else if constexpr (VW == 2) { const float2 t = *reinterpret_cast<const float2*>(p + 2 * lane); v[0] = t.x; v[1] = t.y; }
        else v[0] = p[lane];
    }
    __device__ __forceinline__ void store(float* __restrict__ p, int lane) const {
        if constexpr (VW == 4) st4(p + 4 * lane, make_float4(v[0], v[1], v[2], v[3]));
        else if constexpr (VW == 2) *reinterpret_cast<float2*>(p + 2 * lane) = make_float2(v[0], v[1]);
        else p[lane] = v[0];
    }
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int c = 0; c < VW; ++c) v[c] = 0.f;
    }
};

template <int VW>
__global__ __launch_bounds__(BLOCK) void wide_sim_kernel(const float* __restrict__ y, const int32_t* __restrict__ idx,
                                                         int K, int k0, float* __restrict__ s, int64_t m) {
    constexpr int H = 64 * VW;
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6);
    if (i >= m) return;                                    // wave-uniform
    WideVec<VW> yi;
    yi.load(y + i * H, lane);
    float mine = 3.4e38f;                                  // lane k keeps the distance of column k
    bool have_mine = false;
    for (int k = k0; k < K; ++k) {
        const int j = idx[i * K + k];
        if (j < 0) continue;                               // wave-uniform
        WideVec<VW> yj;
        yj.load(y + (int64_t)j * H, lane);
        float part = 0.f;
#pragma unroll
        for (int c = 0; c < VW; ++c) { const float d = yi.v[c] - yj.v[c]; part = fmaf(d, d, part); }
        const float d2 = wave_sum(part);
        if (lane == k) { mine = d2; have_mine = true; }
    }
    float dmin = mine;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dmin = fminf(dmin, __shfl_xor(dmin, o, WAVE));
    const float e = have_mine ? expf(dmin - mine) : 0.f;
    const float den = wave_sum(e);
    if (lane < K) s[i * K + lane] = den > 0.f ? e / den : 0.f;     // isolated point: no message
}

template <int VW>
__global__ __launch_bounds__(BLOCK) void wide_agg_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                         const int32_t* __restrict__ idx, int K, int k0,
                                                         float* __restrict__ out, int64_t m) {
    constexpr int H = 64 * VW;
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6);
    if (i >= m) return;
    WideVec<VW> acc;
    acc.zero();
    for (int k = k0; k < K; ++k) {
        const int j = idx[i * K + k];
        if (j < 0) continue;
        const float w = s[i * K + k];
        WideVec<VW> xj;
        xj.load(x + (int64_t)j * H, lane);
#pragma unroll
        for (int c = 0; c < VW; ++c) acc.v[c] = fmaf(w, xj.v[c], acc.v[c]);
    }
    acc.store(out + i * H, lane);
}

template <int VW>
__global__ __launch_bounds__(BLOCK) void wide_bwd_edge_kernel(const float* __restrict__ gm, const float* __restrict__ xprev,
                                                              const int32_t* __restrict__ idx, int K, int k0,
                                                              float* __restrict__ ds, int accumulate, int64_t m) {
    constexpr int H = 64 * VW;
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6);
    if (i >= m) return;
    WideVec<VW> g;
    g.load(gm + i * H, lane);
    float mine = (accumulate && lane < K) ? ds[i * K + lane] : 0.f;
    for (int k = k0; k < K; ++k) {
        const int j = idx[i * K + k];
        if (j < 0) continue;
        WideVec<VW> xj;
        xj.load(xprev + (int64_t)j * H, lane);
        float part = 0.f;
#pragma unroll
        for (int c = 0; c < VW; ++c) part = fmaf(g.v[c], xj.v[c], part);
        const float dotv = wave_sum(part);
        if (lane == k) mine += dotv;
    }
    if (lane < K) ds[i * K + lane] = mine;
}

// rows = source rows; coef[e] weights gathered rows of `src` addressed by the edge's TARGET row e / K:
//   out[j] = (add ? add[j] : 0) + sum_{p in rev(j)} coef[e_p] * (SIM ? (y_j - src[e_p / K]) : src[e_p / K])
template <int VW, bool SIM>
__global__ __launch_bounds__(BLOCK) void wide_scatter_kernel(const float* __restrict__ src, const float* __restrict__ coef,
                                                             const int32_t* __restrict__ rev_ptr,
                                                             const int32_t* __restrict__ rev_eid, int K,
                                                             const float* __restrict__ add, float* __restrict__ out,
                                                             int64_t m_src) {
    constexpr int H = 64 * VW;
    const int lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6);
    if (j >= m_src) return;
    WideVec<VW> acc, self;
    acc.zero();
    if constexpr (SIM) self.load(src + j * H, lane);
    const int beg = rev_ptr[j], end = rev_ptr[j + 1];
    for (int p = beg; p < end; ++p) {
        const int e = rev_eid[p];
        const float w = coef[e];
        WideVec<VW> r;
        r.load(src + (int64_t)(e / K) * H, lane);
#pragma unroll
        for (int c = 0; c < VW; ++c) acc.v[c] = fmaf(w, SIM ? (self.v[c] - r.v[c]) : r.v[c], acc.v[c]);
    }
    if (add != nullptr) {
        WideVec<VW> a;
        a.load(add + j * H, lane);
#pragma unroll
        for (int c = 0; c < VW; ++c) acc.v[c] += a.v[c];
    }
    acc.store(out + j * H, lane);
}

template <int VW>
__global__ __launch_bounds__(BLOCK) void wide_sim_bwd_kernel(const float* __restrict__ ds, const float* __restrict__ s,
                                                             const float* __restrict__ y, const int32_t* __restrict__ idx,
                                                             int K, int k0, float* __restrict__ w,
                                                             float* __restrict__ dy_self, int64_t m) {
    constexpr int H = 64 * VW;
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6);
    if (i >= m) return;
    const bool col = lane >= k0 && lane < K;
    const float sk = col ? s[i * K + lane] : 0.f, dk = col ? ds[i * K + lane] : 0.f;
    const float dotv = wave_sum(sk * dk);
    const float wk = -2.0f * sk * (dk - dotv);             // 0 on missing entries (s == 0) and columns < k0
    if (lane < K) w[i * K + lane] = wk;
    WideVec<VW> yi, acc;
    yi.load(y + i * H, lane);
    acc.zero();
    for (int k = k0; k < K; ++k) {
        const int j = idx[i * K + k];
        if (j < 0) continue;
        const float wv = __shfl(wk, k, WAVE);
        WideVec<VW> yj;
        yj.load(y + (int64_t)j * H, lane);
#pragma unroll
        for (int c = 0; c < VW; ++c) acc.v[c] = fmaf(wv, yi.v[c] - yj.v[c], acc.v[c]);
    }
    acc.store(dy_self + i * H, lane);
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

// the index layout (uint16 local ids / int32 global rows) is a template argument of the fast kernels
template <int HH, int KK, bool WS>
static void launch_sim_step(dim3 grid, hipStream_t st, const float* y, const float* z, const int32_t* idx32,
                            const uint16_t* idx16, int n_tgt, int n_src, const float* Q, const float* P, float* s,
                            float* x1, int64_t m) {
    if (idx16) hipLaunchKernelGGL((sim_step_fast_kernel<HH, KK, WS, true>), grid, dim3(BLOCK), 0, st, y, z, idx32, idx16, n_tgt, n_src, Q, P, s, x1, m);
    else hipLaunchKernelGGL((sim_step_fast_kernel<HH, KK, WS, false>), grid, dim3(BLOCK), 0, st, y, z, idx32, idx16, n_tgt, n_src, Q, P, s, x1, m);
}
template <int HH, int KK>
static void launch_step(dim3 grid, hipStream_t st, const float* xin, const float* z, const float* s, const int32_t* idx32,
                        const uint16_t* idx16, int n_tgt, int n_src, const float* Q, const float* P, float* xout,
                        int64_t m) {
    if (idx16) hipLaunchKernelGGL((step_fast_kernel<HH, KK, true>), grid, dim3(BLOCK), 0, st, xin, z, s, idx32, idx16, n_tgt, n_src, Q, P, xout, m);
    else hipLaunchKernelGGL((step_fast_kernel<HH, KK, false>), grid, dim3(BLOCK), 0, st, xin, z, s, idx32, idx16, n_tgt, n_src, Q, P, xout, m);
}

static int meanfield_forward_impl(const float* z, const float* y, const int32_t* idx32, const uint16_t* idx16,
                                  int n_tgt, int n_src, int K, int k0, int64_t m, int H, const float* Q,
                                  const float* P, int T, float* s, float* xs, crf_stream_t stream) {
    if (int rc = check_common(m, H, K, k0)) return rc;
    CRF_REQUIRE(z && y && idx32 && Q && P && (xs || T == 0), CRF_ERR_ARG, "null pointer");
    // s may be NULL when nothing reads it back: a single fused step (T == 1) on the fast path, no backward pass
    CRF_REQUIRE(s || (T == 1 && k0 == 1 && (K == 16 || K == 32) && m * H * 4 < ((int64_t)1 << 31)), CRF_ERR_ARG,
                "s == NULL needs T == 1 on the fused first-step kernel (K in {16, 32}, k0 == 1)");
    CRF_REQUIRE(T >= 0, CRF_ERR_ARG, "T=%d < 0", T);
    hipStream_t st = as_stream(stream);
    // (the fast kernels address neighbour rows by 32-bit byte offsets: tables of 2 GiB and more take the generic kernels)
    const bool fast = (k0 == 1) && (K == 16 || K == 32) && m * H * 4 < ((int64_t)1 << 31);
    DISPATCH_H(H, {
        const dim3 grid((unsigned)cdiv(m, Geo<HH>::PPB)), blk(BLOCK);
        int t0 = 0;
        if (fast) {
            float* x1 = T > 0 ? xs : nullptr;
            if (K == 16) {
                if (T > 0) launch_sim_step<HH, 16, true>(grid, st, y, z, idx32, idx16, n_tgt, n_src, Q, P, s, x1, m);
                else launch_sim_step<HH, 16, false>(grid, st, y, z, idx32, idx16, n_tgt, n_src, Q, P, s, x1, m);
            } else {
                if (T > 0) launch_sim_step<HH, 32, true>(grid, st, y, z, idx32, idx16, n_tgt, n_src, Q, P, s, x1, m);
                else launch_sim_step<HH, 32, false>(grid, st, y, z, idx32, idx16, n_tgt, n_src, Q, P, s, x1, m);
            }
            t0 = 1;
        } else {
            hipLaunchKernelGGL(sim_kernel<HH>, grid, blk, 0, st, y, idx32, K, k0, s, m);
        }
        CRF_LAUNCH_CHECK();
        for (int t = t0; t < T; ++t) {
            const float* xin = t == 0 ? z : xs + (int64_t)(t - 1) * m * H;
            float* xout = xs + (int64_t)t * m * H;
            if (fast && K == 16) launch_step<HH, 16>(grid, st, xin, z, s, idx32, idx16, n_tgt, n_src, Q, P, xout, m);
            else if (fast) launch_step<HH, 32>(grid, st, xin, z, s, idx32, idx16, n_tgt, n_src, Q, P, xout, m);
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

// ---------------------------------------------------------------------- wide rows: host side
#define DISPATCH_VW(H, ...)                                     \
    switch (H) {                                                \
        case 128: { constexpr int VV = 2; __VA_ARGS__; break; } \
        default: { constexpr int VV = 4; __VA_ARGS__; break; }  \
    }

static int check_wide(int64_t m, int H, int K, int k0) {
    CRF_REQUIRE(m > 0 && m < (int64_t)1 << 31, CRF_ERR_ARG, "rows m=%lld out of range", (long long)m);
    CRF_REQUIRE(H == 128 || H == 256, CRF_ERR_UNSUPPORTED, "wide mean-field kernels take H in {128, 256}, got %d", H);
    CRF_REQUIRE(K >= 1 && K <= 64 && k0 >= 0 && k0 < K, CRF_ERR_ARG, "K=%d k0=%d invalid", K, k0);
    CRF_REQUIRE(m * K < (int64_t)1 << 31, CRF_ERR_ARG, "edge ids exceed int32 (m=%lld K=%d)", (long long)m, K);
    return CRF_OK;
}

extern "C" int crfconv_wide_similarity(const float* y, const int32_t* idx32, int K, int k0, int64_t m, int H, float* s,
                                       crf_stream_t stream) {
    if (int rc = check_wide(m, H, K, k0)) return rc;
    CRF_REQUIRE(y && idx32 && s, CRF_ERR_ARG, "null pointer");
    const dim3 grid((unsigned)cdiv(m, BLOCK / WAVE));
    DISPATCH_VW(H, hipLaunchKernelGGL(wide_sim_kernel<VV>, grid, dim3(BLOCK), 0, as_stream(stream), y, idx32, K, k0, s, m));
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_wide_aggregate(const float* x, const float* s, const int32_t* idx32, int K, int k0, int64_t m, int H,
                                      float* out, crf_stream_t stream) {
    if (int rc = check_wide(m, H, K, k0)) return rc;
    CRF_REQUIRE(x && s && idx32 && out && out != x, CRF_ERR_ARG, "null pointer or aliasing");
    const dim3 grid((unsigned)cdiv(m, BLOCK / WAVE));
    DISPATCH_VW(H, hipLaunchKernelGGL(wide_agg_kernel<VV>, grid, dim3(BLOCK), 0, as_stream(stream), x, s, idx32, K, k0, out, m));
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_wide_bwd_edge(const float* gm, const float* xprev, const int32_t* idx32, int K, int k0, int64_t m,
                                     int H, float* ds, int accumulate, crf_stream_t stream) {
    if (int rc = check_wide(m, H, K, k0)) return rc;
    CRF_REQUIRE(gm && xprev && idx32 && ds, CRF_ERR_ARG, "null pointer");
    const dim3 grid((unsigned)cdiv(m, BLOCK / WAVE));
    DISPATCH_VW(H, hipLaunchKernelGGL(wide_bwd_edge_kernel<VV>, grid, dim3(BLOCK), 0, as_stream(stream), gm, xprev, idx32, K,
                                      k0, ds, accumulate, m));
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

/* out[j] = (add ? add[j] : 0) + sum_{e in rev(j)} coef[e] * src[e / K]          (similarity == 0: A^T gm)
 * out[j] = add[j] + sum_{e in rev(j)} coef[e] * (src[j] - src[e / K])           (similarity != 0: dy, src = y, add = dy_self) */
extern "C" int crfconv_wide_scatter(const float* src, const float* coef, const int32_t* rev_ptr, const int32_t* rev_eid,
                                    int K, int64_t m_src, int H, const float* add, int similarity, float* out,
                                    crf_stream_t stream) {
    if (int rc = check_wide(m_src, H, K, 0)) return rc;
    CRF_REQUIRE(src && coef && rev_ptr && rev_eid && out && out != src, CRF_ERR_ARG, "null pointer or aliasing");
    const dim3 grid((unsigned)cdiv(m_src, BLOCK / WAVE));
    DISPATCH_VW(H, {
        if (similarity)
            hipLaunchKernelGGL((wide_scatter_kernel<VV, true>), grid, dim3(BLOCK), 0, as_stream(stream), src, coef, rev_ptr, rev_eid, K, add, out, m_src);
        else
            hipLaunchKernelGGL((wide_scatter_kernel<VV, false>), grid, dim3(BLOCK), 0, as_stream(stream), src, coef, rev_ptr, rev_eid, K, add, out, m_src);
    });
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_wide_similarity_bwd(const float* ds, const float* s, const float* y, const int32_t* idx32, int K,
                                           int k0, int64_t m, int H, float* w, float* dy_self, crf_stream_t stream) {
    if (int rc = check_wide(m, H, K, k0)) return rc;
    CRF_REQUIRE(ds && s && y && idx32 && w && dy_self && w != ds, CRF_ERR_ARG, "null pointer or aliasing");
    const dim3 grid((unsigned)cdiv(m, BLOCK / WAVE));
    DISPATCH_VW(H, hipLaunchKernelGGL(wide_sim_bwd_kernel<VV>, grid, dim3(BLOCK), 0, as_stream(stream), ds, s, y, idx32, K, k0,
                                      w, dy_self, m));
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}
