// Parameter gradients of the WIDE PointConv layers (d = 32, 64: the 10 240- and 2 560-point levels of PointConvBig) on the
// matrix pipe, in one pass over the edges and without any per-edge tensor in memory.
//
// What is computed (models/point_conv_big.py:20-23, 37-58, backward of the weight MLP's second layer and of its first layer's
// folded coefficients), per edge e = (target i, neighbour k) with rel = p_i - p_j:
//     h1  = lrelu(A1 rel + b1)                 [d]       layer 1 (BatchNorm-1 folded into A1, b1)
//     h2  = W2 h1                              [d]       layer 2
//     gh2 = ca (g_i x_j) + cb h2 + cc          [d]       gradient wrt h2 with BatchNorm-2's batch-statistic terms (ca, cb, cc per channel)
//     dW2 += gh2 (x) h1                        [d, d]
//     gp  = (W2^T gh2) lrelu'(h1)              [d]       gradient wrt the layer-1 pre-activation
//     dA1 | db1 += gp (x) [rel, 1]             [d, 4]
// Until round 3 the wide layers wrote h1, gh2 and rel per edge (bwd_dump_kernel: 88 MB at d = 32), a GEMM launch formed gh2 W2,
// a third pass (a1_reduce) the [d, 4] sums and the weight-gradient kernel dW2 -- 136 us of the training step for d = 32 and 64.
//
// Here ONE wavefront takes ONE target point = 16 edges at a time (K = 16: the edges of a point are exactly the 16 rows of an
// MFMA tile) and the three d x d products per edge are v_mfma_f32_16x16x4_f32:
//   (1) H2 = H1 W2^T      A = H1 computed by the lanes DIRECTLY in fragment layout (lane (e, k) holds H1[e][4 s + k]), B = W2 from LDS
//   (3) GW = GH2 W2       A = GH2 tile (LDS), B = W2 from LDS
//   (5) dW2 += GH2^T H1   A, B = the GH2 / H1 tiles read row-wise from LDS; (d/16)^2 accumulators live across the wave's points
// Everything between them (gh2, the lrelu' mask, the [d, 4] sums) happens in the MFMA result layout (lane (g, j): rows 4 g .. 4 g + 3,
// column j).  LDS row strides: d + 4 floats for W2 and the GH2 tile (conflict-free for the A-fragment reads [row = lane % 16][4 s + lane / 16]
// and for the transposed B-fragment reads of (1)), d + 16 for the H1 tile (conflict-free for the row-wise reads of (5)); the
// remaining row-wise reads of a d + 4 tile are two-way conflicted, far below the matrix pipe's pace (2 LDS reads per 8 CU cycles).
// Per workgroup: one float slab [d, d] (dW2) and one float64 slab [d, 4] (dA1 | db1); the sums over workgroups join the batched
// reductions at the end of the backward pass (crfconv_reduce_jobs, crfconv_reduce_jobs_f64).
#include "common.hpp"
#include "gridsync.hpp"

namespace crf {

using f32x4w = __attribute__((ext_vector_type(4))) float;
#ifndef WP_BLOCK_
#define WP_BLOCK_ 512         // 8 wavefronts share one W2 staging (swept on the step, one box: 128 threads 4.337 ms, 256 4.297, 512 4.284)
#endif
constexpr int WP_BLOCK = WP_BLOCK_, WP_WAVES = WP_BLOCK / WAVE, WP_MAX = 8;

struct WideJobs {
    const float* x[WP_MAX]; const float* gout[WP_MAX]; const float* pos_src[WP_MAX]; const float* pos_tgt[WP_MAX];
    const int32_t* idx[WP_MAX];
    const float* A1[WP_MAX]; const float* b1[WP_MAX]; const float* W2[WP_MAX]; const float* ca[WP_MAX]; const float* cb[WP_MAX];
    const float* cc[WP_MAX];
    float* dw2_partial[WP_MAX]; double* a1_partial[WP_MAX];
    int m_tgt[WP_MAX], nblk[WP_MAX], d[WP_MAX];
    float slope[WP_MAX];
    int blk_base[WP_MAX + 1];
    int njobs;
};

// LDS of one workgroup of the parameter pass at width D (floats): the kernel that serves both widths owns ONE buffer of the larger
template <int D>
constexpr int wide_params_lds_floats() {
    return D * (D + 4) + 4 * D + 3 * D + WP_WAVES * 16 * (D + 4) + WP_WAVES * 16 * (D + 16) + 4 * WP_WAVES * 16;
}
template <int D>
__device__ __forceinline__ void wide_params_body(const WideJobs& t, const int job, const int blk, float* __restrict__ lds) {
    constexpr int T16 = D / 16, S4 = D / 4, LDW = D + 4, LDG = D + 4, LDH = D + 16, K = 16;
    constexpr int UNR = S4 > 16 ? 4 : S4;           // k-steps unrolled at a time (fully unrolled, the d = 128 form runs out of registers)
    float* const s_w2 = lds;                                                            // W2[c][c'], row c padded to LDW
    float4* const s_a1 = reinterpret_cast<float4*>(s_w2 + D * LDW);                     // {A1[c'][0..2], b1[c']}
    float* const s_coef = reinterpret_cast<float*>(s_a1 + D);                           // ca | cb | cc
    float (*s_g)[K * LDG] = reinterpret_cast<float (*)[K * LDG]>(s_coef + 3 * D);       // GH2 tile of the wave's point  [edge][channel]
    float (*s_h)[K * LDH] = reinterpret_cast<float (*)[K * LDH]>(s_coef + 3 * D + WP_WAVES * K * LDG);      // H1 tile [edge][channel]
    float4 (*s_rel)[K] = reinterpret_cast<float4 (*)[K]>(s_coef + 3 * D + WP_WAVES * K * (LDG + LDH));      // {rel x, y, z, 1} per edge (zero row for a missing edge)
    float* s_red = s_w2;                                  // block sums (dW2 [d, d], then the [d, 4] sums) reuse the W2 tile once the points are done
    const int nblk = t.nblk[job], m = t.m_tgt[job];
    const float* __restrict__ x = t.x[job];
    const float* __restrict__ gout = t.gout[job];
    const float* __restrict__ pos_src = t.pos_src[job];
    const float* __restrict__ pos_tgt = t.pos_tgt[job];
    const int32_t* __restrict__ idx = t.idx[job];
    const float slope = t.slope[job];
    {
        const float* __restrict__ W2 = t.W2[job];
        for (int i = threadIdx.x; i < D * D; i += WP_BLOCK) s_w2[(i / D) * LDW + (i % D)] = W2[i];
        const float* __restrict__ A1 = t.A1[job];
        const float* __restrict__ b1 = t.b1[job];
        for (int c = threadIdx.x; c < D; c += WP_BLOCK) {
            s_a1[c] = make_float4(A1[3 * c], A1[3 * c + 1], A1[3 * c + 2], b1[c]);
            s_coef[c] = t.ca[job][c];
            s_coef[D + c] = t.cb[job][c];
            s_coef[2 * D + c] = t.cc[job][c];
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = lane & 15, kq = lane >> 4;              // fragment coordinates: row / column index, k index
    float* tg = s_g[wave];
    float* th = s_h[wave];
    float4* trel = s_rel[wave];
    f32x4w accW[T16][T16];
#pragma unroll
    for (int a = 0; a < T16; ++a)
#pragma unroll
        for (int b = 0; b < T16; ++b) accW[a][b] = f32x4w{0.f, 0.f, 0.f, 0.f};
    float a1acc[T16][4];
#pragma unroll
    for (int a = 0; a < T16; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) a1acc[a][b] = 0.f;

    // the workgroup's contiguous range of target points, dealt over its wavefronts
    const int per = (m + nblk - 1) / nblk;
    const int p0 = blk * per, p1 = p0 + per < m ? p0 + per : m;
    for (int p = p0 + wave; p < p1; p += WP_WAVES) {
        // ---- the lane's edge: lane (e, kq) evaluates layer 1 for H1[e][4 s + kq], s = 0 .. D/4 - 1 (the A-fragment layout)
        const int jraw = idx[(int64_t)p * K + e];
        const bool live = jraw >= 0;
        const int64_t j = live ? jraw : 0;
        const float rx = pos_tgt[3 * (int64_t)p] - pos_src[3 * j], ry = pos_tgt[3 * (int64_t)p + 1] - pos_src[3 * j + 1],
                    rz = pos_tgt[3 * (int64_t)p + 2] - pos_src[3 * j + 2];
        // the feature rows and the gradient row are needed behind product (1) only, but depend on nothing but the index row:
        // issued here, they travel while layer 1 and product (1) run.  Result layout of the MFMA: lane (kq, e) owns rows (edges)
        // 4 kq .. 4 kq + 3 and column (channel) 16 tj + e
        int jr[4];
        bool lr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            jr[r] = __shfl((int)j, 4 * kq + r, WAVE);            // lanes 0..15 hold the neighbour of edge `lane`
            lr[r] = __shfl(live ? 1 : 0, 4 * kq + r, WAVE) != 0;
        }
        constexpr bool PRELOAD = D <= 64;
        [[maybe_unused]] float xv[PRELOAD ? T16 : 1][4], gi[PRELOAD ? T16 : 1];
        if constexpr (PRELOAD) {
#pragma unroll
            for (int tj = 0; tj < T16; ++tj) {
                gi[tj] = gout[(int64_t)p * D + 16 * tj + e];
#pragma unroll
                for (int r = 0; r < 4; ++r) xv[tj][r] = x[(int64_t)jr[r] * D + 16 * tj + e];
            }
        }
        if (kq == 0) trel[e] = live ? make_float4(rx, ry, rz, 1.f) : make_float4(0.f, 0.f, 0.f, 0.f);
        // ---- (1) H2 = H1 W2^T: B[k = c'][j = c] = W2[c][c'].  The A fragment of step s IS layer 1 evaluated for this lane's
        // (edge, channel 4 s + kq): computed on the fly (five vector instructions against T16 MFMAs), stored to the H1 tile for (4), (5)
        f32x4w acc[T16];
#pragma unroll
        for (int tj = 0; tj < T16; ++tj) acc[tj] = f32x4w{0.f, 0.f, 0.f, 0.f};
#pragma unroll UNR
        for (int s = 0; s < S4; ++s) {
            const float4 a = s_a1[4 * s + kq];
            const float pre = fmaf(a.x, rx, fmaf(a.y, ry, fmaf(a.z, rz, a.w)));
            const float h = live ? (pre > 0.f ? pre : slope * pre) : 0.f;
            th[e * LDH + 4 * s + kq] = h;
#pragma unroll
            for (int tj = 0; tj < T16; ++tj)
                acc[tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(h, s_w2[(16 * tj + e) * LDW + 4 * s + kq], acc[tj], 0, 0, 0);
        }
        // ---- (2) gh2 in result layout: acc[tj][r] = H2[edge 4 kq + r][channel 16 tj + e]
#pragma unroll
        for (int tj = 0; tj < T16; ++tj) {
            const int c = 16 * tj + e;
            const float va = s_coef[c], vb = s_coef[D + c], vc = s_coef[2 * D + c];
            float gic, xc[4];
            if constexpr (PRELOAD) {
                gic = gi[tj];
#pragma unroll
                for (int r = 0; r < 4; ++r) xc[r] = xv[tj][r];
            } else {
                gic = gout[(int64_t)p * D + c];
#pragma unroll
                for (int r = 0; r < 4; ++r) xc[r] = x[(int64_t)jr[r] * D + c];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float gh = fmaf(va, gic * xc[r], fmaf(vb, acc[tj][r], vc));
                tg[(4 * kq + r) * LDG + c] = lr[r] ? gh : 0.f;
            }
        }
        __builtin_amdgcn_wave_barrier();                         // LDS operations of one wavefront complete in order: tiles are written
        // ---- (3) GW = GH2 W2: A[i = edge][k = c] from the GH2 tile, B[k = c][j = c'] = W2[c][c']
#pragma unroll
        for (int tj = 0; tj < T16; ++tj) acc[tj] = f32x4w{0.f, 0.f, 0.f, 0.f};
#pragma unroll UNR
        for (int s = 0; s < S4; ++s) {
            const float a = tg[e * LDG + 4 * s + kq];
#pragma unroll
            for (int tj = 0; tj < T16; ++tj)
                acc[tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, s_w2[(4 * s + kq) * LDW + 16 * tj + e], acc[tj], 0, 0, 0);
        }
        // ---- (4) gp = GW lrelu'(H1), [d, 4] sums: acc[tj][r] = GW[edge 4 kq + r][channel c' = 16 tj + e]
        {
            float4 rl[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) rl[r] = trel[4 * kq + r];
#pragma unroll
            for (int tj = 0; tj < T16; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float h = th[(4 * kq + r) * LDH + 16 * tj + e];
                    const float gp = acc[tj][r] * (h > 0.f ? 1.f : slope);
                    a1acc[tj][0] = fmaf(gp, rl[r].x, a1acc[tj][0]);
                    a1acc[tj][1] = fmaf(gp, rl[r].y, a1acc[tj][1]);
                    a1acc[tj][2] = fmaf(gp, rl[r].z, a1acc[tj][2]);
                    a1acc[tj][3] = fmaf(gp, rl[r].w, a1acc[tj][3]);
                }
        }
        // ---- (5) dW2 += GH2^T H1: A[i = c][k = edge] = GH2[edge][c], B[k = edge][j = c'] = H1[edge][c']
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float af[T16], bf[T16];
#pragma unroll
            for (int ti = 0; ti < T16; ++ti) {
                af[ti] = tg[(4 * s + kq) * LDG + 16 * ti + e];
                bf[ti] = th[(4 * s + kq) * LDH + 16 * ti + e];
            }
#pragma unroll
            for (int ti = 0; ti < T16; ++ti)
#pragma unroll
                for (int tj = 0; tj < T16; ++tj)
                    accW[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ti], bf[tj], accW[ti][tj], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();                         // the next point overwrites the tiles
    }
    // ---- block sums.  dW2: accW[ti][tj][r] = dW2[c = 16 ti + 4 kq + r][c' = 16 tj + e]; the wavefronts add in turn (fixed order)
    __syncthreads();                                             // every wavefront is done with W2
    for (int w = 0; w < WP_WAVES; ++w) {
        if (wave == w) {
#pragma unroll
            for (int ti = 0; ti < T16; ++ti)
#pragma unroll
                for (int tj = 0; tj < T16; ++tj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int o = (16 * ti + 4 * kq + r) * D + 16 * tj + e;
                        s_red[o] = (w == 0 ? 0.f : s_red[o]) + accW[ti][tj][r];
                    }
        }
        __syncthreads();
    }
    {
        float* __restrict__ out = t.dw2_partial[job] + (int64_t)blk * D * D;
        for (int i = threadIdx.x; i < D * D; i += WP_BLOCK) out[i] = s_red[i];
    }
    __syncthreads();
    // [d, 4] sums: the four k-groups of a wavefront hold different edges of the same channels -> fold over kq by shuffles, then
    // over the wavefronts through LDS in float64
#pragma unroll
    for (int tj = 0; tj < T16; ++tj)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v = a1acc[tj][q];
            v += __shfl_xor(v, 16, WAVE);
            v += __shfl_xor(v, 32, WAVE);
            if (kq == 0) s_red[(wave * D + 16 * tj + e) * 4 + q] = v;
        }
    __syncthreads();
    {
        double* __restrict__ outd = t.a1_partial[job] + (int64_t)blk * 4 * D;
        for (int i = threadIdx.x; i < 4 * D; i += WP_BLOCK) {
            double a = 0.0;
#pragma unroll
            for (int w = 0; w < WP_WAVES; ++w) a += (double)s_red[w * 4 * D + i];
            outd[i] = a;                                         // [channel][x, y, z, bias]
        }
    }
}

// Both widths in ONE launch (round 5): the d = 32 layers (854 workgroups at config 2) and the d = 64 ones (214) used to be two launches
// of ~41 us each at the end of the backward pass, neither of which fills the chip; a workgroup looks its job up and dispatches on the
// job's width.
__global__ __launch_bounds__(WP_BLOCK) void wide_params_any_kernel(const WideJobs t) {
    __shared__ __attribute__((aligned(16))) float lds[wide_params_lds_floats<64>()];
    int job = 0;
    while (job + 1 < t.njobs && t.blk_base[job + 1] <= (int)blockIdx.x) ++job;
    const int blk = (int)blockIdx.x - t.blk_base[job];
    if (t.d[job] == 32) wide_params_body<32>(t, job, blk, lds);
    else wide_params_body<64>(t, job, blk, lds);
}

// ------------------------------------------------------------------ forward statistics pass of the wide layers on the matrix pipe
// uvstats_kernel's work (pointconv.hip: U = sum_k (h2 - shift) x_j, V = sum_k x_j and the BatchNorm-2 statistic partials of h2 - shift
// from ONE pass over the edges) for K = 16, d in {32, 64, 128} with the d x d layer-2 product of a point's sixteen edges as
// v_mfma_f32_16x16x4_f32 (product (1) above; layer 1 evaluated by the lanes in fragment layout).  These launches sit ON the chain
// of the coarse levels (2 560 ... 10 240 points): on the vector ALU a wavefront walked a point's edges four at a time through an LDS
// exchange, 17-21 us per launch whatever the level; here a point is d^2 / 64 MFMAs.
template <int D>
__global__ __launch_bounds__(WP_BLOCK) void uvstats_mfma_kernel(const float* __restrict__ x, const float* __restrict__ pos_src,
                                                                const float* __restrict__ pos_tgt, const int32_t* __restrict__ idx,
                                                                int m, int nblk, const float* __restrict__ A1,
                                                                const float* __restrict__ b1, const float* __restrict__ W2, float slope,
                                                                const float* __restrict__ mean_rel, float* __restrict__ shift_out,
                                                                float* __restrict__ U, float* __restrict__ V, float* __restrict__ partial,
                                                                unsigned* __restrict__ ticket, double* __restrict__ stats) {
    constexpr int T16 = D / 16, S4 = D / 4, LDW = D + 4, K = 16;
    constexpr int UNR = S4 > 16 ? 4 : S4;
    __shared__ __attribute__((aligned(16))) float s_w2[D * LDW];
    __shared__ float4 s_a1[D];
    __shared__ float s_shift[D];
    __shared__ float s_h0[D];
    __shared__ float s_red[WP_WAVES][2][D];
    for (int i = threadIdx.x; i < D * D; i += WP_BLOCK) s_w2[(i / D) * LDW + (i % D)] = W2[i];
    for (int c = threadIdx.x; c < D; c += WP_BLOCK) {
        const float4 a = make_float4(A1[3 * c], A1[3 * c + 1], A1[3 * c + 2], b1[c]);
        s_a1[c] = a;
        const float pre = fmaf(a.x, mean_rel[0], fmaf(a.y, mean_rel[1], fmaf(a.z, mean_rel[2], a.w)));
        s_h0[c] = pre > 0.f ? pre : slope * pre;
    }
    __syncthreads();
    // shift = layer 2 at the mean relative position (the statistics are taken of h2 - shift: no cancellation in the variance);
    // ascending channel order, the summation order of the vector kernels' h2_of
    for (int c = threadIdx.x; c < D; c += WP_BLOCK) {
        float a = 0.f;
        for (int cp = 0; cp < D; ++cp) a = fmaf(s_h0[cp], s_w2[c * LDW + cp], a);
        s_shift[c] = a;
        if (blockIdx.x == 0) shift_out[c] = a;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = lane & 15, kq = lane >> 4;
    float st1[T16], st2[T16];
#pragma unroll
    for (int tj = 0; tj < T16; ++tj) st1[tj] = st2[tj] = 0.f;
    const int per = (m + nblk - 1) / nblk;
    const int p0 = (int)blockIdx.x * per, p1 = p0 + per < m ? p0 + per : m;
    for (int p = p0 + wave; p < p1; p += WP_WAVES) {
        const int jraw = idx[(int64_t)p * K + e];
        const bool live = jraw >= 0;
        const int64_t j = live ? jraw : 0;
        const float rx = pos_tgt[3 * (int64_t)p] - pos_src[3 * j], ry = pos_tgt[3 * (int64_t)p + 1] - pos_src[3 * j + 1],
                    rz = pos_tgt[3 * (int64_t)p + 2] - pos_src[3 * j + 2];
        int jr[4];
        float lv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            jr[r] = __shfl((int)j, 4 * kq + r, WAVE);
            lv[r] = __shfl(live ? 1.f : 0.f, 4 * kq + r, WAVE);
        }
        float xv[T16][4];
#pragma unroll
        for (int tj = 0; tj < T16; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) xv[tj][r] = x[(int64_t)jr[r] * D + 16 * tj + e] * lv[r];
        f32x4w acc[T16];
#pragma unroll
        for (int tj = 0; tj < T16; ++tj) acc[tj] = f32x4w{0.f, 0.f, 0.f, 0.f};
#pragma unroll UNR
        for (int s = 0; s < S4; ++s) {
            const float4 a = s_a1[4 * s + kq];
            const float pre = fmaf(a.x, rx, fmaf(a.y, ry, fmaf(a.z, rz, a.w)));
            const float h = pre > 0.f ? pre : slope * pre;
#pragma unroll
            for (int tj = 0; tj < T16; ++tj)
                acc[tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(h, s_w2[(16 * tj + e) * LDW + 4 * s + kq], acc[tj], 0, 0, 0);
        }
        // acc[tj][r] = h2[edge 4 kq + r][channel 16 tj + e]
#pragma unroll
        for (int tj = 0; tj < T16; ++tj) {
            const float sh = s_shift[16 * tj + e];
            float u = 0.f, v = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dlt = (acc[tj][r] - sh) * lv[r];
                st1[tj] += dlt;
                st2[tj] = fmaf(dlt, dlt, st2[tj]);
                u = fmaf(dlt, xv[tj][r], u);
                v += xv[tj][r];
            }
            u += __shfl_xor(u, 16, WAVE); u += __shfl_xor(u, 32, WAVE);
            v += __shfl_xor(v, 16, WAVE); v += __shfl_xor(v, 32, WAVE);
            if (kq == 0) {
                U[(int64_t)p * D + 16 * tj + e] = u;
                V[(int64_t)p * D + 16 * tj + e] = v;
            }
        }
    }
#pragma unroll
    for (int tj = 0; tj < T16; ++tj) {
        float a = st1[tj], b = st2[tj];
        a += __shfl_xor(a, 16, WAVE); a += __shfl_xor(a, 32, WAVE);
        b += __shfl_xor(b, 16, WAVE); b += __shfl_xor(b, 32, WAVE);
        if (kq == 0) { s_red[wave][0][16 * tj + e] = a; s_red[wave][1][16 * tj + e] = b; }
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t pr = make_rsrc(partial, (int)gridDim.x * 2 * D * 4);
    for (int i = threadIdx.x; i < 2 * D; i += WP_BLOCK) {
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < WP_WAVES; ++w) a += s_red[w][i / D][i % D];
        st1_sc1(pr, ((int)blockIdx.x * 2 * D + i) * 4, a);       // [block][sum | sum of squares][channel]: block_reduce_store<D, 2>'s layout
    }
    // with a ticket word the last workgroup to finish sums the rows into stats (gridsync.hpp; reduce_partials_kernel's launch otherwise)
    // (its 16 KB of LDS: at d = 128 inside the W2 tile, which every wavefront has left by then -- two workgroups per CU as before)
    constexpr bool ALIAS = sizeof(float) * D * LDW >= sizeof(double) * 4 * WP_BLOCK;
    __shared__ double s_own[ALIAS ? 1 : 4 * WP_BLOCK], s_tot[2 * D];
    __shared__ int s_flag;
    if (ticket == nullptr || !last_workgroup(ticket, gridDim.x, &s_flag)) return;
    double* s_buf = ALIAS ? reinterpret_cast<double*>(s_w2) : s_own;
    sum_partial_rows_f64<WP_BLOCK>(pr, (int)gridDim.x, 2 * D, s_buf, s_tot);
    if (threadIdx.x < 2 * D) stats[threadIdx.x] = s_tot[threadIdx.x];
}

// Points per wavefront.  Every workgroup pays a prologue (W2 into LDS, the shift vector) before its first point, and a wavefront walks
// its points one after the other (each a chain index row -> rows -> MFMAs -> stores that only OTHER wavefronts overlap): few points per
// wavefront pay the prologue too often, many leave the chip without wavefronts.  By level size, swept on the training step on one box
// (big, mid, small): (3, 3, 1) 4.292 ms, (4, 4, 1) 4.300, (4, 2, 1) 4.300, (3, 2, 1) 4.300, (3, 3, 2) 4.307, (4, 4, 4) 4.341; without the
// matrix-pipe kernels 4.388:
#ifndef WP_PPW_BIG_
#define WP_PPW_BIG_ 3         // >= 8192 target points (the 10 240-point level)
#endif
#ifndef WP_PPW_MID_
#define WP_PPW_MID_ 3         // >= 2048 (2 560 points)
#endif
#ifndef WP_PPW_SMALL_
#define WP_PPW_SMALL_ 1       // below (640 points: 160 workgroups)
#endif
static int64_t wide_points_per_wave(int64_t m_tgt) {
    return m_tgt >= 8192 ? WP_PPW_BIG_ : (m_tgt >= 2048 ? WP_PPW_MID_ : WP_PPW_SMALL_);
}

// 1 when the matrix-pipe statistics pass takes (K, d); *nblk: its workgroup count (<= max_blocks partial slabs of 2 d floats)
bool uvstats_mfma_ok(int K, int d) { return K == 16 && (d == 32 || d == 64 || d == 128); }
int uvstats_mfma_launch(const float* x, const float* pos_src, const float* pos_tgt, const int32_t* idx32, int64_t m_tgt, int d,
                        const float* A1, const float* b1, const float* W2, float slope, const float* mean_rel3, float* shift, float* U,
                        float* V, float* partial, int64_t max_blocks, int64_t* nblk_out, unsigned* ticket, double* stats, hipStream_t st) {
    int64_t nb = cdiv(m_tgt, wide_points_per_wave(m_tgt) * WP_WAVES);
    if (nb > max_blocks) nb = max_blocks;
    if (nb < 1) nb = 1;
    *nblk_out = nb;
    if (d == 32) hipLaunchKernelGGL(uvstats_mfma_kernel<32>, dim3((unsigned)nb), dim3(WP_BLOCK), 0, st, x, pos_src, pos_tgt, idx32, (int)m_tgt, (int)nb, A1, b1, W2, slope, mean_rel3, shift, U, V, partial, ticket, stats);
    else if (d == 64) hipLaunchKernelGGL(uvstats_mfma_kernel<64>, dim3((unsigned)nb), dim3(WP_BLOCK), 0, st, x, pos_src, pos_tgt, idx32, (int)m_tgt, (int)nb, A1, b1, W2, slope, mean_rel3, shift, U, V, partial, ticket, stats);
    else hipLaunchKernelGGL(uvstats_mfma_kernel<128>, dim3((unsigned)nb), dim3(WP_BLOCK), 0, st, x, pos_src, pos_tgt, idx32, (int)m_tgt, (int)nb, A1, b1, W2, slope, mean_rel3, shift, U, V, partial, ticket, stats);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

static int64_t wide_nblk(int64_t m_tgt, int d) {
    (void)d;
    int64_t nb = cdiv(m_tgt, wide_points_per_wave(m_tgt) * WP_WAVES);
    if (nb < 1) nb = 1;
    if (nb > 2048) nb = 2048;
    return nb;
}

}  // namespace crf

using namespace crf;

extern "C" int crfconv_pointconv_wide_params_supported(int64_t m_tgt, int K, int d) {
    // (d = 128 compiles to 256 + 256 registers with a few spilled to scratch -- and scratch kernels cannot sit in a replayed hipGraph
    // on ROCm 7.2: the 640-point level keeps the dump + GEMM passes)
    return (K == 16 && (d == 32 || d == 64) && m_tgt > 0 && m_tgt < ((int64_t)1 << 27)) ? 1 : 0;
}

extern "C" int64_t crfconv_pointconv_wide_params_nblk(int64_t m_tgt, int d) {
    return m_tgt > 0 ? wide_nblk(m_tgt, d) : 0;
}

extern "C" int crfconv_pointconv_wide_params_jobs(const crf_pc_wide_job* jobs, int njobs, crf_stream_t stream) {
    CRF_REQUIRE(jobs || njobs == 0, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(njobs >= 0, CRF_ERR_ARG, "njobs=%d < 0", njobs);
    hipStream_t st = as_stream(stream);
    for (int j = 0; j < njobs; ++j) {
        const crf_pc_wide_job& jb = jobs[j];
        CRF_REQUIRE(crfconv_pointconv_wide_params_supported(jb.m_tgt, jb.K, jb.d) == 1, CRF_ERR_UNSUPPORTED,
                    "job %d: K=%d d=%d m_tgt=%lld outside the matrix-pipe parameter pass (K = 16, d in {32, 64})", j, jb.K, jb.d,
                    (long long)jb.m_tgt);
        CRF_REQUIRE(jb.x && jb.gout && jb.pos_src && jb.pos_tgt && jb.idx32 && jb.A1 && jb.b1 && jb.W2 && jb.ca && jb.cb && jb.cc &&
                        jb.dw2_partial && jb.a1_partial, CRF_ERR_ARG, "job %d: null pointer", j);
    }
    // the jobs in the caller's order, WP_MAX per launch, both widths together (the longest-running width first: its workgroups start first)
    int order[2] = {64, 32};
    int idx[64];
    int nidx = 0;
    for (int w = 0; w < 2; ++w)
        for (int j = 0; j < njobs && nidx < 64; ++j)
            if (jobs[j].d == order[w]) idx[nidx++] = j;
    CRF_REQUIRE(nidx == njobs, CRF_ERR_UNSUPPORTED, "at most 64 jobs per call");
    for (int j0 = 0; j0 < nidx; j0 += WP_MAX) {
        WideJobs t;
        const int n = nidx - j0 < WP_MAX ? nidx - j0 : WP_MAX;
        int64_t blocks = 0;
        for (int k = 0; k < n; ++k) {
            const crf_pc_wide_job& jb = jobs[idx[j0 + k]];
            t.x[k] = jb.x; t.gout[k] = jb.gout; t.pos_src[k] = jb.pos_src; t.pos_tgt[k] = jb.pos_tgt; t.idx[k] = jb.idx32;
            t.A1[k] = jb.A1; t.b1[k] = jb.b1; t.W2[k] = jb.W2; t.ca[k] = jb.ca; t.cb[k] = jb.cb; t.cc[k] = jb.cc;
            t.dw2_partial[k] = jb.dw2_partial; t.a1_partial[k] = jb.a1_partial;
            t.m_tgt[k] = (int)jb.m_tgt; t.nblk[k] = (int)wide_nblk(jb.m_tgt, jb.d); t.slope[k] = jb.slope; t.d[k] = jb.d;
            t.blk_base[k] = (int)blocks;
            blocks += t.nblk[k];
        }
        for (int k = n; k <= WP_MAX; ++k) t.blk_base[k] = (int)blocks;
        for (int k = n; k < WP_MAX; ++k) {
            t.x[k] = t.gout[k] = t.pos_src[k] = t.pos_tgt[k] = t.A1[k] = t.b1[k] = t.W2[k] = t.ca[k] = t.cb[k] = t.cc[k] = nullptr;
            t.idx[k] = nullptr; t.dw2_partial[k] = nullptr; t.a1_partial[k] = nullptr; t.m_tgt[k] = 0; t.nblk[k] = 1; t.slope[k] = 1.f; t.d[k] = 32;
        }
        t.njobs = n;
        hipLaunchKernelGGL(wide_params_any_kernel, dim3((unsigned)blocks), dim3(WP_BLOCK), 0, st, t);
        CRF_LAUNCH_CHECK();
    }
    return CRF_OK;
}
