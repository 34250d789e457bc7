// The per-point classifier of the network, models/point_conv_big.py:131-134:
//     MLP(C -> 4C: Linear, BatchNorm, LeakyReLU 0.1) -> nn.Dropout(0.5) -> nn.Linear(4C -> classes)
// as kernels that never store a [M, 4C] tensor.  At the finest level (M = 163840 rows, C = 32) the 128-wide activation is
// 84 MB; the step-by-step form writes or reads such a tensor eleven times per training step (y, h = dropout(lrelu(bn(y))),
// gA, the passes of the fused MLP backward, the weight gradient of the last Linear).  Here every pass RECOMPUTES
// y = x W1^T from the 21 MB input on the matrix pipe (v_mfma_f32_16x16x4_f32: exact fp32 products, the same k order as
// linear_fwd_kernel, so y is bit-identical in every pass) and keeps everything 128-wide in registers:
//
//   forward   head_stats_kernel                           BatchNorm statistic records of y (the tuples of linear_fwd_kernel), nothing stored
//             crfconv_bn_coef_from_nrecords (linear.hip)  coefficients + running statistics
//             head_fwd_kernel                             y -> lrelu(a y + b) -> mask -> logits = h W2^T + b2; one mask WORD per
//                                                         (row, lane group) = the 32 channels a lane holds (2.6 MB instead of h)
//   backward  head_bwd_p1_kernel    y, gh = g W2 recomputed in the TRANSPOSED accumulator layout (rows in registers, channels on
//                                   lanes), so that h and g1 = lrelu'() mask gh are directly the MFMA operands of the three
//                                   row contractions dW2 += g^T h, PA += g1^T x, x^T x -- partial rows per workgroup
//             head_bwd_sum_kernel   partial rows -> float64 totals (fixed order)
//             head_bwd_dx_kernel    coefficients of gY from the totals, y and gh recomputed, dX = gY W1 from registers
//             head_bwd_params_kernel  dgamma, dbeta, dW1 = diag(a)[PA - c2 1^T x - c3 rs (W1 x^T x - mu 1^T x)], dW2, db2
//
// Layouts of v_mfma_f32_16x16x4_f32 (D[i][j] += sum_k A[i][k] B[k][j]): lane l supplies A[i = l & 15][k = l >> 4] and
// B[k = l >> 4][j = l & 15]; register e of lane l holds D[i = 4 (l >> 4) + e][j = l & 15].  With rr = l & 15, g = l >> 4:
//   "channel-major" (forward, dX):  A = weights, B = rows   ->  lane (rr, g) holds row rr, channels 16 t + 4 g + e: four
//       consecutive channels of one row = the float4 operand fragment of the NEXT product over channels;
//   "row-major" (backward pass 1):  A = rows, B = weights   ->  lane (rr, g) holds channel 16 t + rr, rows 4 g + e = the operand
//       of a product over ROWS whose k-step e takes row 4 g + e from every lane (any fixed bijection rows <-> (step, lane
//       group) serves a sum over all sixteen rows, as long as both operands use the same one).
// Every row-streaming kernel requests a wavefront's operand rows two groups ahead.  The vector ALU and the matrix pipe of a SIMD do
// not overlap (scratch/mfma_peak.hip: 32.8 + 4 V cycles per MFMA with V vector instructions beside it), so the kernels' time is
// 32 x MFMAs + 4 x vector instructions per wavefront: matrix-pipe utilisation 0.39 (forward: the mask hash) .. 0.57 (profiles/r4j_head_pmc.md).
#include "common.hpp"

#include <cstdlib>

namespace crf {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int HD_BLOCK = 256, HD_WAVES = HD_BLOCK / WAVE;
constexpr int HD_CO = 128, HD_T = HD_CO / 16;       // hidden width (4 C) and its 16-channel tiles
constexpr int HD_C2P = 16;                         // classes, padded to one tile
constexpr int HD_LD = HD_CO + 4;                   // padded row of the [., 128] LDS arrays (conflict-free float4 fragment reads)
constexpr int HD_SLAB_T = 4, HD_SLABS = HD_T / HD_SLAB_T;      // pass 1: 64 channels per workgroup

#define HD_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// W [rows][Ci] (row-major, Ci % 4 == 0) -> LDS [rows][Ci + 4]
__device__ __forceinline__ void hd_stage_rows(float* __restrict__ dst, const float* __restrict__ W, int rows, int Ci) {
    const int c4 = Ci / 4, ld = Ci + 4;
    for (int t = threadIdx.x; t < rows * c4; t += HD_BLOCK) {
        const int r = t / c4, k4 = t - r * c4;
        st4(dst + r * ld + 4 * k4, ld4(W + (int64_t)r * Ci + 4 * k4));
    }
}

// ------------------------------------------------------------------------------------------------------------ statistics
// BatchNorm statistic records of y = x W1^T, nothing stored: record [workgroup][channel] = {shift, n, sum (y - shift),
// sum (y - shift)^2} exactly as linear_fwd_kernel's epilogue writes them (crfconv_bn_coef_from_nrecords reads them).  Two channel
// slabs (blockIdx.y) of four tiles; the operand rows of the NEXT group are requested before the current group's products.
template <int NCH>
__global__ __launch_bounds__(HD_BLOCK) void head_stats_kernel(const float* __restrict__ X, const float* __restrict__ W1, int64_t M,
                                                              float* __restrict__ stat_partial /*[gridDim.x][128][4]*/) {
    constexpr int Ci = 16 * NCH, Cip = Ci + 4, ST = HD_SLAB_T;
    __shared__ float sW1[16 * ST * Cip];
    __shared__ float sSt[HD_WAVES * 4 * 16 * ST];
    const int slab = blockIdx.y, co_base = 16 * ST * slab;
    hd_stage_rows(sW1, W1 + (int64_t)co_base * Ci, 16 * ST, Ci);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, rr = lane & 15, g = lane >> 4;
    float s1[ST][4], s2[ST][4], sh[ST][4];
#pragma unroll
    for (int t = 0; t < ST; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) { s1[t][e] = 0.f; s2[t][e] = 0.f; sh[t][e] = 0.f; }
    bool have_shift = false;
    int64_t nrows = 0;
    const int64_t ngroups = (M + 15) / 16, stride = (int64_t)gridDim.x * HD_WAVES;
    int64_t grp = (int64_t)blockIdx.x * HD_WAVES + wave;
    float4 cur[NCH], nxt[NCH], nx2[NCH];          // operand rows requested TWO groups ahead (a group's products take ~0.5 us, a miss ~1-2)
    auto load = [&](int64_t gq, float4 (&xv)[NCH]) {
        const int64_t r = gq * 16 + rr;
        const bool ok = gq < ngroups && r < M;
#pragma unroll
        for (int c = 0; c < NCH; ++c) xv[c] = ok ? ld4(X + r * Ci + 16 * c + 4 * g) : f4zero();
    };
    load(grp, cur);
    load(grp + stride, nxt);
    for (; grp < ngroups; grp += stride) {
        load(grp + 2 * stride, nx2);
        const int64_t row0 = grp * 16;
        const bool rv = row0 + rr < M;
        f32x4 acc[ST];
#pragma unroll
        for (int t = 0; t < ST; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int t = 0; t < ST; ++t) {
                const float4 wv = ld4(sW1 + (16 * t + rr) * Cip + 16 * c + 4 * g);
                acc[t] = HD_MFMA(wv.x, cur[c].x, acc[t]);
                acc[t] = HD_MFMA(wv.y, cur[c].y, acc[t]);
                acc[t] = HD_MFMA(wv.z, cur[c].z, acc[t]);
                acc[t] = HD_MFMA(wv.w, cur[c].w, acc[t]);
            }
        }
        if (!have_shift) {       // shift = this wave's first row (the lane with rr == 0 of each channel group)
#pragma unroll
            for (int t = 0; t < ST; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) sh[t][e] = __shfl(acc[t][e], 16 * g, WAVE);
            have_shift = true;
        }
        if (rv) {
#pragma unroll
            for (int t = 0; t < ST; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = acc[t][e] - sh[t][e];
                    s1[t][e] += d;
                    s2[t][e] = fmaf(d, d, s2[t][e]);
                }
        }
        nrows += (M - row0) < 16 ? (M - row0) : 16;
#pragma unroll
        for (int c = 0; c < NCH; ++c) { cur[c] = nxt[c]; nxt[c] = nx2[c]; }
    }
    // one record per workgroup and channel: the sixteen row lanes fold by shuffles, the four waves through LDS, re-based on wave 0's shift
    float* sw = sSt + wave * 4 * 16 * ST;
#pragma unroll
    for (int t = 0; t < ST; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = s1[t][e], b = s2[t][e];
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) {
                a += __shfl_xor(a, o, WAVE);
                b += __shfl_xor(b, o, WAVE);
            }
            if (rr == 0) {
                const int cl = 16 * t + 4 * g + e;
                sw[cl] = sh[t][e];
                sw[16 * ST + cl] = (float)nrows;
                sw[2 * 16 * ST + cl] = a;
                sw[3 * 16 * ST + cl] = b;
            }
        }
    __syncthreads();
    for (int cl = threadIdx.x; cl < 16 * ST; cl += HD_BLOCK) {
        const float s0 = sSt[cl];
        double n = 0.0, S1 = 0.0, S2 = 0.0;
        for (int w = 0; w < HD_WAVES; ++w) {
            const float* q = sSt + w * 4 * 16 * ST;
            const double nb = q[16 * ST + cl];
            if (nb <= 0.0) continue;
            const double d = (double)q[cl] - (double)s0, a = q[2 * 16 * ST + cl], b = q[3 * 16 * ST + cl];
            n += nb;
            S1 += a + nb * d;
            S2 += b + 2.0 * d * a + nb * d * d;
        }
        st4(stat_partial + ((int64_t)blockIdx.x * HD_CO + co_base + cl) * 4, make_float4(s0, (float)n, (float)S1, (float)S2));
    }
}

// ------------------------------------------------------------------------------------------------------------ forward
// logits [M, C2] = dropout(lrelu(a (x W1^T) + b, slope)) W2^T + b2;  mask_bits [ceil(M / 16)][64]: bit 4 t + e of word
// (group, lane (rr, g)) = element (row 16 group + rr, channel 16 t + 4 g + e) was kept.
template <int NCH>
__global__ __launch_bounds__(HD_BLOCK, 2) void head_fwd_kernel(const float* __restrict__ X, const float* __restrict__ W1,
                                                               const float* __restrict__ coef, float slope, unsigned long long seed,
                                                               const long long* __restrict__ counter, unsigned threshold, float scale,
                                                               const float* __restrict__ W2, const float* __restrict__ b2, int64_t M,
                                                               int C2, float* __restrict__ logits,
                                                               unsigned* __restrict__ mask_bits, long long* __restrict__ counter_used) {
    constexpr int Ci = 16 * NCH, Cip = Ci + 4;
    __shared__ float sW1[HD_CO * Cip];              // [128][Ci + 4]
    __shared__ float sW2[HD_C2P * HD_LD];           // [16][132], rows >= C2 zero
    __shared__ float sAB[2 * HD_CO];                // a | b
    __shared__ float sOut[HD_WAVES * 256];          // [waves][16 rows][16]
    hd_stage_rows(sW1, W1, HD_CO, Ci);
    for (int t = threadIdx.x; t < HD_C2P * (HD_CO / 4); t += HD_BLOCK) {
        const int r = t / (HD_CO / 4), k4 = t - r * (HD_CO / 4);
        st4(sW2 + r * HD_LD + 4 * k4, r < C2 ? ld4(W2 + (int64_t)r * HD_CO + 4 * k4) : f4zero());
    }
    for (int t = threadIdx.x; t < 2 * HD_CO; t += HD_BLOCK) sAB[t] = coef[t];
    const unsigned long long ctr = (unsigned long long)counter[0];
    if (counter_used != nullptr && blockIdx.x == 0 && threadIdx.x == 0) counter_used[0] = (long long)ctr;
    const unsigned long long keys = dropout_keys(seed, ctr);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, rr = lane & 15, g = lane >> 4;
    float bias2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bias2[e] = (b2 != nullptr && 4 * g + e < C2) ? b2[4 * g + e] : 0.f;
    float* tile = sOut + wave * 256;
    const int64_t ngroups = (M + 15) / 16, stride = (int64_t)gridDim.x * HD_WAVES;
    int64_t grp = (int64_t)blockIdx.x * HD_WAVES + wave;
    float4 cur[NCH], nxt[NCH], nx2[NCH];
    auto load = [&](int64_t gq, float4 (&xv)[NCH]) {
        const int64_t r = gq * 16 + rr;
        const bool ok = gq < ngroups && r < M;
#pragma unroll
        for (int c = 0; c < NCH; ++c) xv[c] = ok ? ld4(X + r * Ci + 16 * c + 4 * g) : f4zero();
    };
    load(grp, cur);
    load(grp + stride, nxt);
    for (; grp < ngroups; grp += stride) {
        load(grp + 2 * stride, nx2);
        const int64_t row0 = grp * 16, r = row0 + rr;
        f32x4 acc[HD_T];
#pragma unroll
        for (int t = 0; t < HD_T; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NCH; ++c) {          // the eight tiles' products of a k-step side by side: no MFMA waits for its predecessor
            float4 wv[HD_T];
#pragma unroll
            for (int t = 0; t < HD_T; ++t) wv[t] = ld4(sW1 + (16 * t + rr) * Cip + 16 * c + 4 * g);
#pragma unroll
            for (int t = 0; t < HD_T; ++t) acc[t] = HD_MFMA(wv[t].x, cur[c].x, acc[t]);
#pragma unroll
            for (int t = 0; t < HD_T; ++t) acc[t] = HD_MFMA(wv[t].y, cur[c].y, acc[t]);
#pragma unroll
            for (int t = 0; t < HD_T; ++t) acc[t] = HD_MFMA(wv[t].z, cur[c].z, acc[t]);
#pragma unroll
            for (int t = 0; t < HD_T; ++t) acc[t] = HD_MFMA(wv[t].w, cur[c].w, acc[t]);
        }
        unsigned bits = 0u;
        f32x4 lg = f32x4{bias2[0], bias2[1], bias2[2], bias2[3]};
        const unsigned long long ebase = (unsigned long long)r * HD_CO + (unsigned long long)(4 * g);
#pragma unroll
        for (int t = 0; t < HD_T; ++t) {
            const float4 a4 = ld4(sAB + 16 * t + 4 * g), b4 = ld4(sAB + HD_CO + 16 * t + 4 * g);
            const float av[4] = {a4.x, a4.y, a4.z, a4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
            float h[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float u = fmaf(av[e], acc[t][e], bv[e]);
                u = u > 0.f ? u : slope * u;
                const bool keep = dropout_keep_keyed(keys, ebase + (unsigned long long)(16 * t + e), threshold);
                bits |= (keep ? 1u : 0u) << (4 * t + e);
                h[e] = keep ? u * scale : 0.f;
            }
            const float4 w2 = ld4(sW2 + rr * HD_LD + 16 * t + 4 * g);
            lg = HD_MFMA(w2.x, h[0], lg);
            lg = HD_MFMA(w2.y, h[1], lg);
            lg = HD_MFMA(w2.z, h[2], lg);
            lg = HD_MFMA(w2.w, h[3], lg);
        }
        mask_bits[grp * 64 + lane] = bits;
        // lane holds logits[row rr][class 4 g + e]: through a per-wave tile, so that the sixteen rows leave as one contiguous run
        st4(tile + rr * 16 + 4 * g, make_float4(lg[0], lg[1], lg[2], lg[3]));
        __builtin_amdgcn_wave_barrier();
        const int nrows = (M - row0) < 16 ? (int)(M - row0) : 16;
        for (int idx = lane; idx < nrows * C2; idx += WAVE) {
            const int row = idx / C2, c = idx - row * C2;
            logits[row0 * C2 + idx] = tile[row * 16 + c];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < NCH; ++c) { cur[c] = nxt[c]; nxt[c] = nx2[c]; }
    }
}

// ------------------------------------------------------------------------------------------------------------ backward, pass 1
// Partial rows per workgroup column bx (both channel slabs write disjoint parts of row bx):
//   PS  [nblk][2][128]     s1 = sum g1, s2 = sum g1 yh           (g1 = lrelu'(a y + b) mask scale gh,  yh = (y - mean) rstd)
//   PA  [nblk][128][Ci]    g1^T x
//   PW2 [nblk][16][128]    g^T h                                 (rows >= C2 zero)
//   PB2 [nblk][16]         1^T g
//   PXX [nblk][Ci][Ci]     x^T x          (column tile d by slab d % 2; one input tile: slab 0)
//   PX  [nblk][Ci]         1^T x
struct HeadPartials {
    float *PS, *PA, *PW2, *PB2, *PXX, *PX;
};
__host__ __device__ inline size_t head_record_floats(int Ci) {
    return 2 * HD_CO + (size_t)HD_CO * Ci + HD_C2P * HD_CO + HD_C2P + (size_t)Ci * Ci + Ci;
}

// sum of `v` over the four wavefronts: written through LDS [wave][N][64]; afterwards thread (w, lane) returns in out[0 .. N/4)
// the totals of registers i = w N / 4 + {0 .. N/4) of its lane position (N % 4 == 0).  Two barriers.
template <int N>
__device__ __forceinline__ void hd_wg_sum(const float (&v)[N], float* __restrict__ sRed, int wave, int lane, float (&out)[N / 4]) {
#pragma unroll
    for (int i = 0; i < N; ++i) sRed[(wave * N + i) * WAVE + lane] = v[i];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < N / 4; ++q) {
        const int i = wave * (N / 4) + q;
        out[q] = (sRed[(0 * N + i) * WAVE + lane] + sRed[(1 * N + i) * WAVE + lane]) +
                 (sRed[(2 * N + i) * WAVE + lane] + sRed[(3 * N + i) * WAVE + lane]);
    }
    __syncthreads();
}

template <int NCH>      // Ci / 16
__global__ __launch_bounds__(HD_BLOCK, 2) void head_bwd_p1_kernel(const float* __restrict__ G, const float* __restrict__ X,
                                                                  const float* __restrict__ W1, const float* __restrict__ W2,
                                                                  const float* __restrict__ coef, float slope, float scale,
                                                                  const unsigned* __restrict__ mask_bits, int64_t M, int C2,
                                                                  HeadPartials P) {
    constexpr int Ci = 16 * NCH, Cip = Ci + 4, ST = HD_SLAB_T;
    constexpr int ND = NCH >= 2 ? NCH / 2 : 1;      // x^T x column tiles of this slab: d = 2 j + slab (one input tile: d = 0, slab 0)
    __shared__ float sW1[16 * ST * Cip];             // this slab's 64 rows of W1
    __shared__ float sRed[HD_WAVES * 16 * WAVE];     // 16 KB
    const int slab = blockIdx.y, co_base = 16 * ST * slab;
    hd_stage_rows(sW1, W1 + (int64_t)co_base * Ci, 16 * ST, Ci);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, rr = lane & 15, g = lane >> 4;
    // B fragments of gh = g W2 (k = class 4 g + s, j = channel 16 t + rr) and the channel coefficients of this lane's channels
    float w2f[ST][4], ca[ST], cb[ST], cmu[ST], crs[ST];
#pragma unroll
    for (int t = 0; t < ST; ++t) {
        const int co = co_base + 16 * t + rr;
#pragma unroll
        for (int s = 0; s < 4; ++s) w2f[t][s] = 4 * g + s < C2 ? W2[(int64_t)(4 * g + s) * HD_CO + co] : 0.f;
        ca[t] = coef[co];
        cb[t] = coef[HD_CO + co];
        cmu[t] = coef[2 * HD_CO + co];
        crs[t] = coef[3 * HD_CO + co];
    }
    __syncthreads();
    const bool do_xx = NCH >= 2 || slab == 0;
    f32x4 dw2[ST], pa[ST][NCH], xx[NCH][ND];
    float s1[ST], s2[ST], sb2 = 0.f, sx[NCH];
#pragma unroll
    for (int t = 0; t < ST; ++t) {
        dw2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        s1[t] = 0.f;
        s2[t] = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) pa[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        sx[c] = 0.f;
#pragma unroll
        for (int j = 0; j < ND; ++j) xx[c][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int mshift = 4 * ST * slab + (rr & 3);          // bit of (tile t, this lane's channel) = mshift + 4 t
    const int64_t ngroups = (M + 15) / 16, stride = (int64_t)gridDim.x * HD_WAVES;
    int64_t grp = (int64_t)blockIdx.x * HD_WAVES + wave;
    // operands with the ROW on the lane (A of the recomputed products) + the mask words: requested one group ahead
    struct Ops { float4 xv[NCH]; float g4[4]; uint4 mw; };
    Ops cur, nxt;
    auto load = [&](int64_t gq, Ops& o) {
        const int64_t r = gq * 16 + rr;
        const bool gv = gq < ngroups, ok = gv && r < M;
#pragma unroll
        for (int c = 0; c < NCH; ++c) o.xv[c] = ok ? ld4(X + r * Ci + 16 * c + 4 * g) : f4zero();
#pragma unroll
        for (int s = 0; s < 4; ++s) o.g4[s] = (ok && 4 * g + s < C2) ? G[r * C2 + 4 * g + s] : 0.f;
        o.mw = gv ? *reinterpret_cast<const uint4*>(mask_bits + gq * 64 + 16 * (rr >> 2) + 4 * g) : make_uint4(0u, 0u, 0u, 0u);
    };
    load(grp, cur);
    for (; grp < ngroups; grp += stride) {
        const int64_t row0 = grp * 16;
        // operands with the row as k (step e takes row 4 g + e): the lines were fetched with this group's row operands
        float gk[4], xk[NCH][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t re = row0 + 4 * g + e;
            const bool rve = re < M;
            gk[e] = (rve && rr < C2) ? G[re * C2 + rr] : 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) xk[c][e] = rve ? X[re * Ci + 16 * c + rr] : 0.f;
        }
        load(grp + stride, nxt);
        const unsigned mwv[4] = {cur.mw.x, cur.mw.y, cur.mw.z, cur.mw.w};
        f32x4 y[ST], gh[ST];
#pragma unroll
        for (int t = 0; t < ST; ++t) {
            y[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            gh[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const float xs[4] = {cur.xv[c].x, cur.xv[c].y, cur.xv[c].z, cur.xv[c].w};
#pragma unroll
            for (int t = 0; t < ST; ++t) {
                const float4 wv = ld4(sW1 + (16 * t + rr) * Cip + 16 * c + 4 * g);
                y[t] = HD_MFMA(xs[0], wv.x, y[t]);
                y[t] = HD_MFMA(xs[1], wv.y, y[t]);
                y[t] = HD_MFMA(xs[2], wv.z, y[t]);
                y[t] = HD_MFMA(xs[3], wv.w, y[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < ST; ++t)
#pragma unroll
            for (int s = 0; s < 4; ++s) gh[t] = HD_MFMA(cur.g4[s], w2f[t][s], gh[t]);
        // lane: channel co_base + 16 t + rr, rows row0 + 4 g + e
#pragma unroll
        for (int t = 0; t < ST; ++t) {
            float g1v[4], hv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float yv = y[t][e];
                const float u = fmaf(ca[t], yv, cb[t]);
                const bool pos = u > 0.f;
                const bool keep = ((mwv[e] >> (mshift + 4 * t)) & 1u) != 0u;
                const float gA = keep ? gh[t][e] * scale : 0.f;
                const float g1 = pos ? gA : slope * gA;
                hv[e] = keep ? (pos ? u : slope * u) * scale : 0.f;
                g1v[e] = g1;
                s1[t] += g1;
                s2[t] = fmaf(g1, (yv - cmu[t]) * crs[t], s2[t]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) dw2[t] = HD_MFMA(gk[e], hv[e], dw2[t]);               // [class i][channel j]
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) pa[t][c] = HD_MFMA(g1v[e], xk[c][e], pa[t][c]);   // [channel i][input j]
        }
        if (do_xx) {
#pragma unroll
            for (int j = 0; j < ND; ++j) {
                float xd[4];                             // column tile d = 2 j + slab
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if constexpr (NCH >= 2) xd[e] = slab ? xk[2 * j + 1][e] : xk[2 * j][e];
                    else xd[e] = xk[0][e];
                }
#pragma unroll
                for (int c = 0; c < NCH; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) xx[c][j] = HD_MFMA(xk[c][e], xd[e], xx[c][j]);
            }
        }
        if (slab == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sb2 += gk[e];
#pragma unroll
                for (int c = 0; c < NCH; ++c) sx[c] += xk[c][e];
            }
        }
        cur = nxt;
    }
    // ---- workgroup totals -> partial row blockIdx.x
    const int64_t bx = blockIdx.x;
    {   // dW2 tile t, register e': class 4 g + e', channel co_base + 16 t + rr
        float v[4 * ST], o[ST];
#pragma unroll
        for (int t = 0; t < ST; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * t + e] = dw2[t][e];
        hd_wg_sum<4 * ST>(v, sRed, wave, lane, o);
#pragma unroll
        for (int q = 0; q < ST; ++q) {
            const int i = wave * ST + q, t = i >> 2, e = i & 3;
            P.PW2[(bx * HD_C2P + 4 * g + e) * HD_CO + co_base + 16 * t + rr] = o[q];
        }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {   // PA tile (t, c), register e': channel co_base + 16 t + 4 g + e', input 16 c + rr
        float v[4 * ST], o[ST];
#pragma unroll
        for (int t = 0; t < ST; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * t + e] = pa[t][c][e];
        hd_wg_sum<4 * ST>(v, sRed, wave, lane, o);
#pragma unroll
        for (int q = 0; q < ST; ++q) {
            const int i = wave * ST + q, t = i >> 2, e = i & 3;
            P.PA[(bx * HD_CO + co_base + 16 * t + 4 * g + e) * Ci + 16 * c + rr] = o[q];
        }
    }
    if (do_xx) {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int j = 0; j < ND; ++j) {   // x^T x tile (c, d), register e': input 16 c + 4 g + e', input 16 d + rr
                const int d = NCH >= 2 ? 2 * j + slab : 0;
                float v[4], o[1];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = xx[c][j][e];
                hd_wg_sum<4>(v, sRed, wave, lane, o);
                P.PXX[(bx * Ci + 16 * c + 4 * g + wave) * Ci + 16 * d + rr] = o[0];
            }
    }
    {   // channel sums: rows sit on (register e, lane group g) -> fold the lane groups, then the wavefronts
        float v[2 * ST];
#pragma unroll
        for (int t = 0; t < ST; ++t) {
            float a = s1[t], b = s2[t];
            a += __shfl_xor(a, 16, WAVE); a += __shfl_xor(a, 32, WAVE);
            b += __shfl_xor(b, 16, WAVE); b += __shfl_xor(b, 32, WAVE);
            v[t] = a;
            v[ST + t] = b;
        }
        float o[2 * ST / 4];
        hd_wg_sum<2 * ST>(v, sRed, wave, lane, o);
        if (g == 0) {
#pragma unroll
            for (int q = 0; q < 2 * ST / 4; ++q) {
                const int i = wave * (2 * ST / 4) + q, which = i / ST, t = i - which * ST;
                P.PS[(bx * 2 + which) * HD_CO + co_base + 16 * t + rr] = o[q];
            }
        }
    }
    if (slab == 0) {   // 1^T g (class rr) and 1^T x (input 16 c + rr): register q of the family = {1^T g, 1^T x tile 0 .. }
        float v[8], o[2];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float a = q == 0 ? sb2 : (q <= NCH ? sx[q <= NCH && q > 0 ? q - 1 : 0] : 0.f);
            a += __shfl_xor(a, 16, WAVE); a += __shfl_xor(a, 32, WAVE);
            v[q] = a;
        }
        hd_wg_sum<8>(v, sRed, wave, lane, o);
        if (g == 0) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int i = wave * 2 + q;
                if (i == 0) P.PB2[bx * HD_C2P + rr] = o[q];
                else if (i <= NCH) P.PX[bx * Ci + 16 * (i - 1) + rr] = o[q];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------ totals
// tot[toff[f] + slot] = sum_b base[f][b][slot] in float64 for the six families in ONE launch (64 slots per workgroup, the
// workgroups of the families laid end to end).  Sixteen parts of a workgroup take rows b = part, part + 16, ... (all of a part's
// loads in flight at once: one round trip at <= 256 rows); the parts are then added in part order.
constexpr int HS_PARTS = 16, HS_BLOCK = HS_PARTS * WAVE;
struct HeadSumJobs {
    const float* base[6];
    int nslots[6], toff[6], blk0[7];
};
__global__ __launch_bounds__(HS_BLOCK) void head_bwd_sum_kernel(const HeadSumJobs j, int nblk, double* __restrict__ tot) {
    __shared__ double s_part[HS_PARTS][WAVE];
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    int f = 0;
#pragma unroll
    for (int q = 1; q < 6; ++q)
        if ((int)blockIdx.x >= j.blk0[q]) f = q;
    const float* __restrict__ rec = j.base[f];
    const int nslots = j.nslots[f];
    const int slot = ((int)blockIdx.x - j.blk0[f]) * WAVE + lane;
    const bool ok = slot < nslots;
    double acc = 0.0;
    if (ok) {
        for (int b0 = part; b0 < nblk; b0 += 16 * HS_PARTS) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int b = b0 + u * HS_PARTS;
                v[u] = b < nblk ? rec[(int64_t)b * nslots + slot] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += (double)v[u];
        }
    }
    s_part[part][lane] = acc;
    __syncthreads();
    if (part == 0 && ok) {
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < HS_PARTS; ++q) v += s_part[q][lane];
        tot[j.toff[f] + slot] = v;
    }
}

// offsets (in slots) of the families inside the totals
struct HeadOffsets {
    int ps, pa, pw2, pb2, pxx, px, total;
};
__host__ __device__ inline HeadOffsets head_offsets(int Ci) {
    HeadOffsets o;
    o.ps = 0;
    o.pa = o.ps + 2 * HD_CO;
    o.pw2 = o.pa + HD_CO * Ci;
    o.pb2 = o.pw2 + HD_C2P * HD_CO;
    o.pxx = o.pb2 + HD_C2P;
    o.px = o.pxx + Ci * Ci;
    o.total = o.px + Ci;
    return o;
}

// ------------------------------------------------------------------------------------------------------------ dX
// dX [M, Ci] = gY W1,  gY = alpha lrelu'(a y + b) mask scale gh + bet y + del with the coefficients of the fused MLP backward
// (linear.hip, mlp_channel_part) derived here from the totals s1, s2.
template <int NCH>
__global__ __launch_bounds__(HD_BLOCK, 2) void head_bwd_dx_kernel(const float* __restrict__ G, const float* __restrict__ X,
                                                                  const float* __restrict__ W1, const float* __restrict__ W2,
                                                                  const float* __restrict__ coef, const double* __restrict__ tot,
                                                                  float slope, float scale, const unsigned* __restrict__ mask_bits,
                                                                  int64_t M, int C2, float* __restrict__ dX) {
    constexpr int Ci = 16 * NCH, Cip = Ci + 4, W2LD = HD_C2P + 4;
    __shared__ float sW1[HD_CO * Cip];              // [128][Ci + 4]
    __shared__ float sW1T[Ci * HD_LD];              // [Ci][132]
    __shared__ float sW2T[HD_CO * W2LD];            // [128][20]: channel-major, classes >= C2 zero
    __shared__ float sPro[5 * HD_CO];               // a | b | alpha | bet | del
    hd_stage_rows(sW1, W1, HD_CO, Ci);
    for (int t = threadIdx.x; t < HD_CO * Ci; t += HD_BLOCK) {
        const int co = t / Ci, ci = t - co * Ci;
        sW1T[ci * HD_LD + co] = W1[t];
    }
    for (int t = threadIdx.x; t < HD_CO * HD_C2P; t += HD_BLOCK) {
        const int cls = t / HD_CO, co = t - cls * HD_CO;
        sW2T[co * W2LD + cls] = cls < C2 ? W2[t] : 0.f;
    }
    for (int c = threadIdx.x; c < HD_CO; c += HD_BLOCK) {
        const double a = coef[c], mu = coef[2 * HD_CO + c], rs = coef[3 * HD_CO + c];
        const double c2 = tot[c] / (double)M, c3 = tot[HD_CO + c] / (double)M;
        sPro[c] = coef[c];
        sPro[HD_CO + c] = coef[HD_CO + c];
        sPro[2 * HD_CO + c] = (float)a;
        sPro[3 * HD_CO + c] = (float)(-a * c3 * rs);
        sPro[4 * HD_CO + c] = (float)(-a * c2 + a * c3 * rs * mu);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, rr = lane & 15, g = lane >> 4;
    __syncthreads();
    const int64_t ngroups = (M + 15) / 16, stride = (int64_t)gridDim.x * HD_WAVES;
    int64_t grp = (int64_t)blockIdx.x * HD_WAVES + wave;
    struct Ops { float4 xv[NCH]; float g4[4]; unsigned bits; };
    Ops cur, nxt, nx2;
    auto load = [&](int64_t gq, Ops& o) {
        const int64_t r = gq * 16 + rr;
        const bool gv = gq < ngroups, ok = gv && r < M;
#pragma unroll
        for (int c = 0; c < NCH; ++c) o.xv[c] = ok ? ld4(X + r * Ci + 16 * c + 4 * g) : f4zero();
#pragma unroll
        for (int s = 0; s < 4; ++s) o.g4[s] = (ok && 4 * g + s < C2) ? G[r * C2 + 4 * g + s] : 0.f;
        o.bits = gv ? mask_bits[gq * 64 + lane] : 0u;
    };
    load(grp, cur);
    load(grp + stride, nxt);
    for (; grp < ngroups; grp += stride) {
        load(grp + 2 * stride, nx2);
        const int64_t r = grp * 16 + rr;
        const bool rv = r < M;
        f32x4 acc[HD_T], gh[HD_T];
#pragma unroll
        for (int t = 0; t < HD_T; ++t) {
            acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            gh[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {          // the eight tiles' products of a k-step side by side: no MFMA waits for its predecessor
            float4 wv[HD_T];
#pragma unroll
            for (int t = 0; t < HD_T; ++t) wv[t] = ld4(sW1 + (16 * t + rr) * Cip + 16 * c + 4 * g);
#pragma unroll
            for (int t = 0; t < HD_T; ++t) acc[t] = HD_MFMA(wv[t].x, cur.xv[c].x, acc[t]);
#pragma unroll
            for (int t = 0; t < HD_T; ++t) acc[t] = HD_MFMA(wv[t].y, cur.xv[c].y, acc[t]);
#pragma unroll
            for (int t = 0; t < HD_T; ++t) acc[t] = HD_MFMA(wv[t].z, cur.xv[c].z, acc[t]);
#pragma unroll
            for (int t = 0; t < HD_T; ++t) acc[t] = HD_MFMA(wv[t].w, cur.xv[c].w, acc[t]);
        }
        {
            float4 w2[HD_T];                     // channel 16 t + rr, classes 4 g + s
#pragma unroll
            for (int t = 0; t < HD_T; ++t) w2[t] = ld4(sW2T + (16 * t + rr) * W2LD + 4 * g);
#pragma unroll
            for (int t = 0; t < HD_T; ++t) gh[t] = HD_MFMA(w2[t].x, cur.g4[0], gh[t]);
#pragma unroll
            for (int t = 0; t < HD_T; ++t) gh[t] = HD_MFMA(w2[t].y, cur.g4[1], gh[t]);
#pragma unroll
            for (int t = 0; t < HD_T; ++t) gh[t] = HD_MFMA(w2[t].z, cur.g4[2], gh[t]);
#pragma unroll
            for (int t = 0; t < HD_T; ++t) gh[t] = HD_MFMA(w2[t].w, cur.g4[3], gh[t]);
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 dx[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) dx[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < HD_T; ++t) {
            const int k0 = 16 * t + 4 * g;
            const float4 pa4 = ld4(sPro + k0), pb4 = ld4(sPro + HD_CO + k0), al4 = ld4(sPro + 2 * HD_CO + k0);
            const float4 be4 = ld4(sPro + 3 * HD_CO + k0), de4 = ld4(sPro + 4 * HD_CO + k0);
            const float pav[4] = {pa4.x, pa4.y, pa4.z, pa4.w}, pbv[4] = {pb4.x, pb4.y, pb4.z, pb4.w};
            const float alv[4] = {al4.x, al4.y, al4.z, al4.w}, bev[4] = {be4.x, be4.y, be4.z, be4.w};
            const float dev[4] = {de4.x, de4.y, de4.z, de4.w};
            float gy[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float yv = acc[t][e];
                const bool keep = ((cur.bits >> (4 * t + e)) & 1u) != 0u;
                const float gA = keep ? gh[t][e] * scale : 0.f;
                gy[e] = fmaf(alv[e] * (fmaf(pav[e], yv, pbv[e]) > 0.f ? 1.f : slope), gA, fmaf(bev[e], yv, dev[e]));
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const float4 wt = ld4(sW1T + (16 * c + rr) * HD_LD + k0);
                dx[c] = HD_MFMA(wt.x, gy[0], dx[c]);
                dx[c] = HD_MFMA(wt.y, gy[1], dx[c]);
                dx[c] = HD_MFMA(wt.z, gy[2], dx[c]);
                dx[c] = HD_MFMA(wt.w, gy[3], dx[c]);
            }
            __builtin_amdgcn_sched_barrier(0);       // keeps the 20 coefficient registers of a tile from being requested eight tiles ahead (spills)
        }
        if (rv) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) st4(dX + r * Ci + 16 * c + 4 * g, make_float4(dx[c][0], dx[c][1], dx[c][2], dx[c][3]));
        }
        cur = nxt;
        nxt = nx2;
    }
}

// ------------------------------------------------------------------------------------------------------------ parameters
// dW1[co][ci] = a [ PA - c2 SX - c3 rs ( sum_k W1[co][k] XX[k][ci] - mu SX ) ],  c2 = s1 / M, c3 = s2 / M: one slot per thread, x^T x
// staged in LDS (float64); workgroup 0 also writes dgamma = s2, dbeta = s1, dW2, db2 straight from the totals.
__global__ __launch_bounds__(HD_BLOCK) void head_bwd_params_kernel(const double* __restrict__ tot, const float* __restrict__ W1,
                                                                   const float* __restrict__ coef, int64_t M, int Ci, int C2,
                                                                   float* __restrict__ dW1, float* __restrict__ dgamma,
                                                                   float* __restrict__ dbeta, float* __restrict__ dW2,
                                                                   float* __restrict__ db2) {
    extern __shared__ double sXX[];                 // [Ci][Ci] | [Ci]
    const HeadOffsets o = head_offsets(Ci);
    for (int t = threadIdx.x; t < Ci * Ci + Ci; t += HD_BLOCK) sXX[t] = tot[o.pxx + t];       // PXX and PX are adjacent
    __syncthreads();
    const int t = blockIdx.x * HD_BLOCK + threadIdx.x;
    if (t < HD_CO * Ci) {
        const int co = t / Ci, ci = t - co * Ci;
        const double a = coef[co], mu = coef[2 * HD_CO + co], rs = coef[3 * HD_CO + co];
        const double c2 = tot[o.ps + co] / (double)M, c3 = tot[o.ps + HD_CO + co] / (double)M;
        double yx = 0.0;
        for (int k = 0; k < Ci; k += 4) {
            const float4 w = ld4(W1 + co * Ci + k);
            yx += (double)w.x * sXX[k * Ci + ci] + (double)w.y * sXX[(k + 1) * Ci + ci] + (double)w.z * sXX[(k + 2) * Ci + ci] +
                  (double)w.w * sXX[(k + 3) * Ci + ci];
        }
        const double sxv = sXX[Ci * Ci + ci];
        dW1[t] = (float)(a * (tot[o.pa + t] - c2 * sxv - c3 * rs * (yx - mu * sxv)));
    }
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c < HD_CO; c += HD_BLOCK) {
            dbeta[c] = (float)tot[o.ps + c];
            dgamma[c] = (float)tot[o.ps + HD_CO + c];
        }
        for (int q = threadIdx.x; q < C2 * HD_CO; q += HD_BLOCK) dW2[q] = (float)tot[o.pw2 + q];
        if (db2 != nullptr)
            for (int c = threadIdx.x; c < C2; c += HD_BLOCK) db2[c] = (float)tot[o.pb2 + c];
    }
}

// workgroups of the row-streaming kernels: `cap` resident workgroups (a multiple of the 256 CUs), each wavefront walking its groups
static int hd_grid_cap(int64_t M, int cap) {
    const int64_t ngroups = (M + 15) / 16;
    int64_t nb = (ngroups + HD_WAVES - 1) / HD_WAVES;
    if (nb > cap) nb = cap;
    return (int)(nb < 1 ? 1 : nb);
}
// workgroups per launch (swept 256 .. 1024 on the training step, DESIGN 9 HD3: all within 2 us)
static int hd_grid(int64_t M) { return hd_grid_cap(M, 512); }          // forward
static int hd_grid_dx(int64_t M) { return hd_grid_cap(M, 512); }
static int hd_grid_p1(int64_t M) { return hd_grid_cap(M, 256); }    // x 2 slabs
static int hd_grid_st(int64_t M) { return hd_grid_cap(M, 256); }    // x 2 slabs

}  // namespace crf

using namespace crf;

extern "C" int crfconv_head_supported(int64_t M, int Ci, int Co, int C2) {
    return (M > 0 && Co == HD_CO && (Ci == 16 || Ci == 32) && C2 >= 1 && C2 <= HD_C2P) ? 1 : 0;
}

extern "C" size_t crfconv_head_mask_words(int64_t M) { return (size_t)((M + 15) / 16) * 64; }

extern "C" size_t crfconv_head_stat_records(int64_t M) { return (size_t)hd_grid_st(M); }

extern "C" size_t crfconv_head_backward_workspace(int64_t M, int Ci, int Co, int C2) {
    if (!crfconv_head_supported(M, Ci, Co, C2)) return 0;
    const size_t rec = head_record_floats(Ci);
    return sizeof(float) * rec * (size_t)hd_grid_p1(M) + sizeof(double) * rec + 256;
}

// stat_rec [crfconv_head_stat_records(M)][Co][4]: the records of y = X W1^T for crfconv_bn_coef_from_nrecords; y is not stored.
extern "C" int crfconv_head_stats(const float* X, const float* W1, int64_t M, int Ci, int Co, float* stat_rec, crf_stream_t stream) {
    CRF_REQUIRE(X && W1 && stat_rec, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(crfconv_head_supported(M, Ci, Co, 1), CRF_ERR_UNSUPPORTED, "classifier head %d -> %d not supported", Ci, Co);
    const dim3 grid((unsigned)hd_grid_st(M), HD_SLABS), blk(HD_BLOCK);
    if (Ci == 16) hipLaunchKernelGGL(head_stats_kernel<1>, grid, blk, 0, as_stream(stream), X, W1, M, stat_rec);
    else hipLaunchKernelGGL(head_stats_kernel<2>, grid, blk, 0, as_stream(stream), X, W1, M, stat_rec);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_head_forward(const float* X, const float* W1, const float* coef, float slope, float p, uint64_t seed,
                                    const int64_t* counter, const float* W2, const float* b2, int64_t M, int Ci, int Co, int C2,
                                    float* logits, uint32_t* mask_bits, int64_t* counter_used, crf_stream_t stream) {
    CRF_REQUIRE(X && W1 && coef && counter && W2 && logits && mask_bits, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(crfconv_head_supported(M, Ci, Co, C2), CRF_ERR_UNSUPPORTED, "classifier head %d -> %d -> %d not supported", Ci, Co, C2);
    CRF_REQUIRE(p >= 0.f && p < 1.f, CRF_ERR_ARG, "dropout probability %g outside [0, 1)", (double)p);
    const dim3 grid((unsigned)hd_grid(M)), blk(HD_BLOCK);
#define HD_FWD(N) hipLaunchKernelGGL(head_fwd_kernel<N>, grid, blk, 0, as_stream(stream), X, W1, coef, slope, (unsigned long long)seed, reinterpret_cast<const long long*>(counter), dropout_threshold(p), 1.f / (1.f - p), W2, b2, M, C2, logits, reinterpret_cast<unsigned*>(mask_bits), reinterpret_cast<long long*>(counter_used))
    if (Ci == 16) HD_FWD(1); else HD_FWD(2);
#undef HD_FWD
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_head_backward(const float* g, const float* X, const float* W1, const float* coef, float slope, float p,
                                     const float* W2, const uint32_t* mask_bits, int64_t M, int Ci, int Co, int C2, float* dX,
                                     float* dW1, float* dgamma, float* dbeta, float* dW2, float* db2, void* workspace,
                                     size_t workspace_bytes, crf_stream_t stream) {
    CRF_REQUIRE(g && X && W1 && coef && W2 && mask_bits && dW1 && dgamma && dbeta && dW2 && workspace, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(crfconv_head_supported(M, Ci, Co, C2), CRF_ERR_UNSUPPORTED, "classifier head %d -> %d -> %d not supported", Ci, Co, C2);
    CRF_REQUIRE(p >= 0.f && p < 1.f, CRF_ERR_ARG, "dropout probability %g outside [0, 1)", (double)p);
    CRF_REQUIRE(workspace_bytes >= crfconv_head_backward_workspace(M, Ci, Co, C2), CRF_ERR_ARG, "workspace too small");
    hipStream_t st = as_stream(stream);
    const float scale = 1.f / (1.f - p);
    const unsigned* mb = reinterpret_cast<const unsigned*>(mask_bits);
    const int nblk = hd_grid_p1(M);
    const HeadOffsets o = head_offsets(Ci);
    const size_t rec = head_record_floats(Ci);
    float* recs = reinterpret_cast<float*>(workspace);
    double* tot = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + ((sizeof(float) * rec * (size_t)nblk + 255) / 256) * 256);
    HeadPartials P;          // the six families as separate [nblk][...] arrays; ONE launch sums them all
    size_t off = 0;
    P.PS = recs + off;  off += (size_t)nblk * 2 * HD_CO;
    P.PA = recs + off;  off += (size_t)nblk * HD_CO * Ci;
    P.PW2 = recs + off; off += (size_t)nblk * HD_C2P * HD_CO;
    P.PB2 = recs + off; off += (size_t)nblk * HD_C2P;
    P.PXX = recs + off; off += (size_t)nblk * Ci * Ci;
    P.PX = recs + off;  off += (size_t)nblk * Ci;
    const dim3 grid1((unsigned)nblk, HD_SLABS), blk(HD_BLOCK);
    if (Ci == 16) hipLaunchKernelGGL(head_bwd_p1_kernel<1>, grid1, blk, 0, st, g, X, W1, W2, coef, slope, scale, mb, M, C2, P);
    else hipLaunchKernelGGL(head_bwd_p1_kernel<2>, grid1, blk, 0, st, g, X, W1, W2, coef, slope, scale, mb, M, C2, P);
    CRF_LAUNCH_CHECK();
    HeadSumJobs sj;
    const float* fam[6] = {P.PS, P.PA, P.PW2, P.PB2, P.PXX, P.PX};
    const int fslots[6] = {2 * HD_CO, HD_CO * Ci, HD_C2P * HD_CO, HD_C2P, Ci * Ci, Ci};
    const int foff[6] = {o.ps, o.pa, o.pw2, o.pb2, o.pxx, o.px};
    sj.blk0[0] = 0;
    for (int f = 0; f < 6; ++f) {
        sj.base[f] = fam[f];
        sj.nslots[f] = fslots[f];
        sj.toff[f] = foff[f];
        sj.blk0[f + 1] = sj.blk0[f] + (fslots[f] + WAVE - 1) / WAVE;
    }
    hipLaunchKernelGGL(head_bwd_sum_kernel, dim3(sj.blk0[6]), dim3(HS_BLOCK), 0, st, sj, nblk, tot);
    CRF_LAUNCH_CHECK();
    if (dX != nullptr) {
        const dim3 grid((unsigned)hd_grid_dx(M));
        if (Ci == 16) hipLaunchKernelGGL(head_bwd_dx_kernel<1>, grid, blk, 0, st, g, X, W1, W2, coef, tot, slope, scale, mb, M, C2, dX);
        else hipLaunchKernelGGL(head_bwd_dx_kernel<2>, grid, blk, 0, st, g, X, W1, W2, coef, tot, slope, scale, mb, M, C2, dX);
        CRF_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(head_bwd_params_kernel, dim3((HD_CO * Ci + HD_BLOCK - 1) / HD_BLOCK), blk, sizeof(double) * (size_t)(Ci * Ci + Ci), st,
                       tot, W1, coef, M, Ci, C2, dW1, dgamma, dbeta, dW2, db2);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}
