// Weight gradient of the per-point Linear layers:  dW[co, ci] = sum_m G[m, co] * X[m, ci]
// (+ optional bias gradient db[co] = sum_m G[m, co]).
//
// These are the path's genuinely dense contractions (models/common.py:30,35 -- every MLP.lin), but
// with a reduction dimension of m = 10^4..10^5 rows and a tiny [Co, Ci] output, which is exactly
// the shape vendor GEMMs handle worst (rocBLAS: 140-420 us at m = 163840; streaming the operands once
// takes 5-20 us).  Here each wavefront streams its slice of rows straight from HBM into fp32 MFMA
// (v_mfma_f32_16x16x4_f32: exact f32, fmaf-chain numerics): lane l of a k-step holds
//   A[i = l & 15][k = l >> 4] = G[row0 + (l >> 4)][co0 + (l & 15)]
//   B[k = l >> 4][j = l & 15] = X[row0 + (l >> 4)][ci0 + (l & 15)]
// i.e. the row-major operands ARE the fragment layout -- no transpose, no LDS staging.  Accumulator
// tiles stay in registers for the whole slice; waves of a block combine through LDS; block partials
// are summed in a fixed order by a second kernel (bitwise reproducible, no float atomics).
#include "common.hpp"
#include "gridsync.hpp"
#include "wgrad_body.hpp"
#include "crf_matrices_body.hpp"
#include "reduce64_body.hpp"

#include <cstdlib>

namespace crf {

template <int TCO, int TCI>
__global__ __launch_bounds__(WG_BLOCK) void wgrad_kernel(const float* __restrict__ G,
                                                         const float* __restrict__ X, int64_t M, int Co,
                                                         int Ci, int rows_per_block,
                                                         float* __restrict__ partial /*[nblk][Co][Ci]*/,
                                                         float* __restrict__ partial_b /*[nblk][Co] or null*/) {
    __shared__ float s_red[WG_RED_BUFS * TCO * TCI * 256];
    __shared__ float s_b[WG_WAVES * TCO * 16];
    wgrad_body<TCO, TCI>(G, X, M, Co, Ci, rows_per_block, partial, partial_b, blockIdx.x, blockIdx.y, blockIdx.z, s_red, s_b);
}

// Every tile class in ONE launch (round 4): the jobs of the small classes -- five launches of 40-240 workgroups, 7-12 us each, behind
// the <4, 4> launch of the step -- run beside the large ones.  The workgroup looks its job up as above and dispatches on the job's
// class; one LDS buffer of the largest class (32 KB: wgrad_body's two-round sum), the register budget of the largest (the small classes' jobs are few).
__global__ __launch_bounds__(WG_BLOCK) void wgrad_jobs_any_kernel(const WgJobTable t) {
    __shared__ float s_red[WG_RED_BUFS * 4 * 4 * 256];
    __shared__ float s_b[WG_WAVES * 4 * 16];
    wgrad_any_run(t, (int)blockIdx.x, s_red, s_b);
}

// ====================================================================== backward of Linear -> BatchNorm -> LeakyReLU
// (models/common.py:34-40, training mode) in TWO passes over the activations instead of four.  With
//   g1 = gA * lrelu'(a y + b),   yh = (y - mean) rstd,   dbeta = sum g1,   dgamma = sum g1 yh,
//   gY = a (g1 - dbeta / M - yh dgamma / M),   dX = gY W,   dW = gY^T X
// the weight gradient expands to   dW = diag(a) [ G1^T X - (dbeta / M) (1^T X) - diag(dgamma / M) Yh^T X ]:
// G1^T X, Yh^T X, 1^T X, sum g1 and sum g1 yh are all plain row reductions, so ONE streaming pass over (gA, Y, X)
// produces every partial (mlp_bwd_p1_kernel: two MFMA accumulator sets sharing the X fragment); a small finalize turns
// them into dgamma, dbeta, dW and the per-channel coefficients of gY; and dX = gY W is one more pass in which gY is
// formed in registers from (gA, Y) while loading the operand (linear_fwd_kernel<.., true>).  The step-by-step form
// (bn_bwd_reduce -> finalize -> bn_bwd_apply -> dX -> wgrad) reads or writes nine [M, C] arrays; this one six, with
// three launches instead of five, and gY never reaches memory.  (Two launches where the last workgroup of the first pass does the
// finalize's channel part: crfconv_mlp_backward's ticket.)

// dgamma, dbeta of channel c and the coefficients of  gY = alpha * lrelu'(a y + b) * gA + bet * y + del  (bcoef [5][Co] =
// a | b | alpha | bet | del) from the two channel sums s1 = sum g1, s2 = sum g1 yh
__device__ __forceinline__ void mlp_channel_part(int c, double s1, double s2, const float* __restrict__ coef, int64_t M, int Co,
                                                 float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ bcoef) {
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
    const double a = coef[c], mu = coef[2 * Co + c], rs = coef[3 * Co + c];
    const double c2 = s1 / (double)M, c3 = s2 / (double)M;
    bcoef[c] = coef[c];
    bcoef[Co + c] = coef[Co + c];
    bcoef[2 * Co + c] = (float)a;
    bcoef[3 * Co + c] = (float)(-a * c3 * rs);
    bcoef[4 * Co + c] = (float)(-a * c2 + a * c3 * rs * mu);
}

template <int TCO, int TCI>
__global__ __launch_bounds__(WG_BLOCK) void mlp_bwd_p1_kernel(const float* __restrict__ GA, const float* __restrict__ Y,
                                                              const float* __restrict__ X,
                                                              const float* __restrict__ X2, int split /*X = [X | X2] at column split (X2 may be null)*/,
                                                              const float* __restrict__ coef /*[4][Co]: a, b, mean, rstd*/,
                                                              float slope, int64_t M, int Co, int Ci, int rows_per_block,
                                                              float* __restrict__ PA /*[nblk][Co][Ci]*/,
                                                              float* __restrict__ PB /*[nblk][Co][Ci]*/,
                                                              float* __restrict__ PG /*[nblk][2][Co]*/,
                                                              float* __restrict__ PX /*[nblk][Ci]*/,
                                                              unsigned* __restrict__ ticket /*null: mlp_bwd_finalize_kernel does the channel part*/,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ bcoef) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int co_base = blockIdx.y * 16 * TCO, ci_base = blockIdx.z * 16 * TCI;
    const int kk = lane >> 4, cc = lane & 15;
    f32x4 accA[TCO][TCI], accB[TCO][TCI];
#pragma unroll
    for (int a = 0; a < TCO; ++a)
#pragma unroll
        for (int b = 0; b < TCI; ++b) { accA[a][b] = f32x4{0.f, 0.f, 0.f, 0.f}; accB[a][b] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float ca[TCO], cb[TCO], cm[TCO], cr[TCO], sg[TCO], sgy[TCO], sx[TCI];
#pragma unroll
    for (int a = 0; a < TCO; ++a) {
        const int co = co_base + 16 * a + cc;
        const bool ok = co < Co;
        ca[a] = ok ? coef[co] : 0.f;
        cb[a] = ok ? coef[Co + co] : 0.f;
        cm[a] = ok ? coef[2 * Co + co] : 0.f;
        cr[a] = ok ? coef[3 * Co + co] : 0.f;
        sg[a] = 0.f;
        sgy[a] = 0.f;
    }
#pragma unroll
    for (int b = 0; b < TCI; ++b) sx[b] = 0.f;

    const int64_t row_begin = (int64_t)blockIdx.x * rows_per_block;
    const int64_t row_end = row_begin + rows_per_block < M ? row_begin + rows_per_block : M;
#ifndef P1_PREFETCH_
#define P1_PREFETCH_ 0
#endif
    struct Frag { float gv[4][TCO], yv[4][TCO], bv[4][TCI]; };
    auto load = [&](int64_t r0, Frag& f) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t r = r0 + 4 * u + kk;
            const bool rv = r < row_end;
#pragma unroll
            for (int a = 0; a < TCO; ++a) {
                const int co = co_base + 16 * a + cc;
                const bool ok = rv && co < Co;
                f.gv[u][a] = ok ? GA[r * Co + co] : 0.f;
                f.yv[u][a] = ok ? Y[r * Co + co] : cm[a];          // yh = 0 on padding
            }
#pragma unroll
            for (int b = 0; b < TCI; ++b) {
                const int ci = ci_base + 16 * b + cc;
                float xv = 0.f;
                if (rv && ci < Ci) xv = (X2 == nullptr || ci < split) ? X[r * (X2 ? split : Ci) + ci] : X2[r * (Ci - split) + (ci - split)];
                f.bv[u][b] = xv;
            }
        }
    };
    auto compute = [&](const Frag& f) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int b = 0; b < TCI; ++b) sx[b] += f.bv[u][b];
#pragma unroll
            for (int a = 0; a < TCO; ++a) {
                const float g1 = f.gv[u][a] * (fmaf(ca[a], f.yv[u][a], cb[a]) > 0.f ? 1.f : slope);
                const float yh = (f.yv[u][a] - cm[a]) * cr[a];
                sg[a] += g1;
                sgy[a] = fmaf(g1, yh, sgy[a]);
#pragma unroll
                for (int b = 0; b < TCI; ++b) {
                    accA[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(g1, f.bv[u][b], accA[a][b], 0, 0, 0);
                    accB[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(yh, f.bv[u][b], accB[a][b], 0, 0, 0);
                }
            }
        }
    };
    if constexpr (P1_PREFETCH_ != 0) {
        // the next 16-row group's fragments are requested before this group's products (rows past the slice load nothing)
        Frag cur, nxt;
        load(row_begin + 16 * wave, cur);
        for (int64_t r0 = row_begin + 16 * wave; r0 < row_end; r0 += 16 * WG_WAVES) {
            load(r0 + 16 * WG_WAVES, nxt);
            compute(cur);
            cur = nxt;
        }
    } else {
        for (int64_t r0 = row_begin + 16 * wave; r0 < row_end; r0 += 16 * WG_WAVES) {
            Frag f;
            load(r0, f);
            compute(f);
        }
    }
    // C/D layout of 16x16x4: col = lane & 15 (j = ci), row = 4 * (lane >> 4) + reg (i = co)
    __shared__ __attribute__((aligned(16))) float s_red[WG_WAVES][TCO * TCI * 256];
    __shared__ float s_v[WG_WAVES][(2 * TCO + TCI) * 16];
    const int64_t pb = (int64_t)blockIdx.x;
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) __syncthreads();
#pragma unroll
        for (int a = 0; a < TCO; ++a)
#pragma unroll
            for (int b = 0; b < TCI; ++b)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    s_red[wave][(a * TCI + b) * 256 + (4 * kk + g) * 16 + cc] = pass ? accB[a][b][g] : accA[a][b][g];
        if (pass == 0) {
            // per-channel sums: lanes with the same cc over the 4 k-groups
#pragma unroll
            for (int a = 0; a < TCO; ++a) {
                float t1 = sg[a], t2 = sgy[a];
                t1 += __shfl_xor(t1, 16, WAVE); t1 += __shfl_xor(t1, 32, WAVE);
                t2 += __shfl_xor(t2, 16, WAVE); t2 += __shfl_xor(t2, 32, WAVE);
                if (kk == 0) { s_v[wave][a * 16 + cc] = t1; s_v[wave][(TCO + a) * 16 + cc] = t2; }
            }
#pragma unroll
            for (int b = 0; b < TCI; ++b) {
                float t1 = sx[b];
                t1 += __shfl_xor(t1, 16, WAVE); t1 += __shfl_xor(t1, 32, WAVE);
                if (kk == 0) s_v[wave][(2 * TCO + b) * 16 + cc] = t1;
            }
        }
        __syncthreads();
        float* dst = pass ? PB : PA;
        for (int t = threadIdx.x; t < TCO * TCI * 256; t += WG_BLOCK) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < WG_WAVES; ++w) v += s_red[w][t];
            const int tile = t >> 8, a = tile / TCI, b = tile % TCI, i = (t >> 4) & 15, j = t & 15;
            const int co = co_base + 16 * a + i, ci = ci_base + 16 * b + j;
            if (co < Co && ci < Ci) dst[(pb * Co + co) * Ci + ci] = v;
        }
    }
    const __amdgpu_buffer_rsrc_t pgr = make_rsrc(PG, (int)gridDim.x * 2 * Co * 4);
    for (int t = threadIdx.x; t < (2 * TCO + TCI) * 16; t += WG_BLOCK) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < WG_WAVES; ++w) v += s_v[w][t];
        const int grp = t >> 4, c16 = t & 15;
        if (grp < 2 * TCO) {                                   // sum g1 | sum g1 yh: slabs of ci-slab 0 only
            const int which = grp / TCO, co = co_base + 16 * (grp % TCO) + c16;
            if (blockIdx.z == 0 && co < Co) st1_sc1(pgr, (((int)pb * 2 + which) * Co + co) * 4, v);   // write-through: summed in this launch
        } else {                                               // column sums of X: slabs of co-slab 0 only
            const int ci = ci_base + 16 * (grp - 2 * TCO) + c16;
            if (blockIdx.y == 0 && ci < Ci) PX[pb * Ci + ci] = v;
        }
    }
    // The channel part of the finalize (dgamma, dbeta, the coefficients of gY for the dX pass) by the LAST workgroup of this launch
    // to finish (gridsync.hpp): the launch between the two passes of the block's backward disappears.  2 Co <= WG_BLOCK slots.
    constexpr bool ALIAS = TCO * TCI >= 3;                     // the 10 KB of sums inside s_red (everybody has left it by then)
    __shared__ double s_own[ALIAS ? 1 : 5 * WG_BLOCK];
    __shared__ int s_flag;
    if (ticket == nullptr || !last_workgroup(ticket, gridDim.x * gridDim.y * gridDim.z, &s_flag)) return;
    double* s_buf = ALIAS ? reinterpret_cast<double*>(&s_red[0][0]) : s_own;
    double* s_tot = s_buf + 4 * WG_BLOCK;
    sum_partial_rows_f64<WG_BLOCK>(pgr, (int)gridDim.x, 2 * Co, s_buf, s_tot);
    if ((int)threadIdx.x < Co) mlp_channel_part(threadIdx.x, s_tot[threadIdx.x], s_tot[Co + threadIdx.x], coef, M, Co, dgamma, dbeta, bcoef);
}

// dW slots [64 block, 64 block + 64) of one MLP block's backward from its partial slabs (see mlp_bwd_finalize_kernel): shared by
// the per-layer finalize launch and the batched launch over all layers of a backward pass (mlp_dw_jobs_kernel).
constexpr int MF_BLOCK = 1024, MF_WAVES = MF_BLOCK / WAVE;
__device__ __forceinline__ void mlp_dw_slots(const float* __restrict__ PA, const float* __restrict__ PB,
                                             const float* __restrict__ PG, const float* __restrict__ PX, int nblk,
                                             const float* __restrict__ coef, int64_t M, int Co, int Ci, int block,
                                             float* __restrict__ dW) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // 64 slots per workgroup; the sixteen wavefronts split the slabs (b = w, w + 16, ..), four slabs of each of the five
    // streams in flight per lane: a 512-slab reduction is eight dependent round trips (the 256-thread form
    // measured 16.5 us per layer, more than the pass that produced the slabs)
    __shared__ float s_part[MF_WAVES][5][64];
    const int nslots = Co * Ci;
    const int slot = block * 64 + lane;
    const bool ok = slot < nslots;
    const int sl = ok ? slot : nslots - 1;
    const int co = sl / Ci, ci = sl - co * Ci;
    float a0 = 0.f, b0 = 0.f, x0 = 0.f, g1 = 0.f, g2 = 0.f;
    // four slabs of each stream in flight (20 loads per lane; eight spill at the 128-register budget of a 1024-thread block)
    const int64_t sa = (int64_t)MF_WAVES * Co * Ci, sx = (int64_t)MF_WAVES * Ci, sg = (int64_t)MF_WAVES * 2 * Co;
    const float* pa = PA + ((int64_t)w * Co + co) * Ci + ci;
    const float* pb = PB + ((int64_t)w * Co + co) * Ci + ci;
    const float* px = PX + (int64_t)w * Ci + ci;
    const float* pg = PG + (int64_t)w * 2 * Co + co;
#pragma unroll 1
    for (int b = w; b < nblk; b += 4 * MF_WAVES) {
        float va[4], vb[4], vx[4], vg[4], vh[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool in = b + MF_WAVES * u < nblk;
            va[u] = in ? pa[u * sa] : 0.f;
            vb[u] = in ? pb[u * sa] : 0.f;
            vx[u] = in ? px[u * sx] : 0.f;
            vg[u] = in ? pg[u * sg] : 0.f;
            vh[u] = in ? pg[u * sg + Co] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { a0 += va[u]; b0 += vb[u]; x0 += vx[u]; g1 += vg[u]; g2 += vh[u]; }
        pa += 4 * sa; pb += 4 * sa; px += 4 * sx; pg += 4 * sg;
    }
    s_part[w][0][lane] = a0; s_part[w][1][lane] = b0; s_part[w][2][lane] = x0;
    s_part[w][3][lane] = g1; s_part[w][4][lane] = g2;
    __syncthreads();
    if (w == 0 && ok) {
        double A = 0.0, B = 0.0, Xs = 0.0, G1 = 0.0, G2 = 0.0;
        for (int k = 0; k < MF_WAVES; ++k) {
            A += s_part[k][0][lane]; B += s_part[k][1][lane]; Xs += s_part[k][2][lane];
            G1 += s_part[k][3][lane]; G2 += s_part[k][4][lane];
        }
        const double a = coef[co], c2 = G1 / (double)M, c3 = G2 / (double)M;
        dW[slot] = (float)(a * (A - c2 * Xs - c3 * B));
    }
}

// Finalize of the pass above, ONE launch of two kinds of workgroups:
//   blockIdx.x <  nw : 64 slots (co, ci) of dW = a [ sum A - c2 sum sx - c3 sum B ],  c2 = dbeta / M, c3 = dgamma / M
//                      (each workgroup re-derives c2 / c3 of the one or two co rows it touches: no ordering between the
//                      two kinds of workgroups is needed)
//   blockIdx.x >= nw : four channels each: dgamma, dbeta and the coefficients of
//                      gY = alpha * lrelu'(a y + b) * gA + bet * y + del   (bcoef [5][Co] = a | b | alpha | bet | del)
__global__ __launch_bounds__(MF_BLOCK) void mlp_bwd_finalize_kernel(const float* __restrict__ PA, const float* __restrict__ PB,
                                                                    const float* __restrict__ PG, const float* __restrict__ PX,
                                                                    int nblk, const float* __restrict__ coef, int64_t M, int Co,
                                                                    int Ci, int nw, float* __restrict__ dW,
                                                                    float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                    float* __restrict__ bcoef) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if ((int)blockIdx.x >= nw) {                               // one wavefront per channel
        const int c = ((int)blockIdx.x - nw) * MF_WAVES + w;
        if (c >= Co) return;
        double s1 = 0.0, s2 = 0.0;
        // eight slabs of both sums in flight per lane (nblk <= 512: ONE round trip; the rolled loop was eight dependent ones --
        // this launch sits between the two passes of every MLP block's backward)
        for (int b0 = lane; b0 < nblk; b0 += 8 * WAVE) {
            float v1[8], v2[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = b0 + u * WAVE;
                const bool in = b < nblk;
                v1[u] = in ? PG[((int64_t)b * 2 + 0) * Co + c] : 0.f;
                v2[u] = in ? PG[((int64_t)b * 2 + 1) * Co + c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { s1 += (double)v1[u]; s2 += (double)v2[u]; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s1 += __shfl_xor(s1, o, WAVE);
            s2 += __shfl_xor(s2, o, WAVE);
        }
        if (lane == 0) mlp_channel_part(c, s1, s2, coef, M, Co, dgamma, dbeta, bcoef);
        return;
    }
    mlp_dw_slots(PA, PB, PG, PX, nblk, coef, M, Co, Ci, (int)blockIdx.x, dW);
}

// The dW parts of ALL MLP blocks of a backward pass in ONE launch: nothing inside the pass reads a weight gradient, so the
// per-layer finalize launch only does its channel part (dgamma, dbeta, the dX coefficients) and the slab reductions --
// 32 launches of ~8 dependent round trips each -- run side by side at the end.  group_begin = prefix sum of ceil(Co Ci / 64).
constexpr int MDW_MAX = 40;
struct MlpDwTable {
    const float* PA[MDW_MAX];
    const float* PB[MDW_MAX];
    const float* PG[MDW_MAX];
    const float* PX[MDW_MAX];
    const float* coef[MDW_MAX];
    float* dW[MDW_MAX];
    long long M[MDW_MAX];
    int nblk[MDW_MAX], Co[MDW_MAX], Ci[MDW_MAX];
    int group_begin[MDW_MAX + 1];
    int njobs;
};
__global__ __launch_bounds__(MF_BLOCK) void mlp_dw_jobs_kernel(const MlpDwTable t) {
    const int g = blockIdx.x;
    int lo = 0, hi = t.njobs;                          // largest j with group_begin[j] <= g
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (t.group_begin[mid] <= g) lo = mid; else hi = mid;
    }
    mlp_dw_slots(t.PA[lo], t.PB[lo], t.PG[lo], t.PX[lo], t.nblk[lo], t.coef[lo], (int64_t)t.M[lo], t.Co[lo], t.Ci[lo],
                 g - t.group_begin[lo], t.dW[lo]);
}

// out[slot] = sum_b partial[b][slot] for 64 consecutive slots per workgroup: lanes run along the slots (256-byte
// coalesced rows of the partial slabs), the 4 wavefronts take b = w, w + 4, ... with four loads in flight each, and
// combine through LDS in the fixed order w = 0..3 -- bitwise reproducible, and identical between the single and the
// batched entry point.
__device__ __forceinline__ void reduce_slab64(const float* __restrict__ partial, int nblk, int nslots, int slot0,
                                              float* __restrict__ out, float (*s_part)[64]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int slot = slot0 + lane;
    const bool ok = slot < nslots;
    const float* p = partial + (ok ? slot : 0);
    // eight slabs in flight per wavefront: the reduction is a chain of dependent L2 / HBM round trips, not bandwidth
    // (512 slabs of an 8 x 8 gradient took 10.5 us with four in flight)
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int b = w;
    for (; b + 28 < nblk; b += 32) {
        const float v0 = p[(int64_t)b * nslots], v1 = p[(int64_t)(b + 4) * nslots], v2 = p[(int64_t)(b + 8) * nslots];
        const float v3 = p[(int64_t)(b + 12) * nslots], v4 = p[(int64_t)(b + 16) * nslots], v5 = p[(int64_t)(b + 20) * nslots];
        const float v6 = p[(int64_t)(b + 24) * nslots], v7 = p[(int64_t)(b + 28) * nslots];
        a0 += v0; a1 += v1; a2 += v2; a3 += v3;
        a0 += v4; a1 += v5; a2 += v6; a3 += v7;
    }
    for (; b + 12 < nblk; b += 16) {
        a0 += p[(int64_t)b * nslots];
        a1 += p[(int64_t)(b + 4) * nslots];
        a2 += p[(int64_t)(b + 8) * nslots];
        a3 += p[(int64_t)(b + 12) * nslots];
    }
    for (; b < nblk; b += 4) a0 += p[(int64_t)b * nslots];
    s_part[w][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (w == 0 && ok) out[slot] = ((s_part[0][lane] + s_part[1][lane]) + s_part[2][lane]) + s_part[3][lane];
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int nblk,
                                                           int nslots, float* __restrict__ out) {
    __shared__ float s_part[4][64];
    reduce_slab64(partial, nblk, nslots, blockIdx.x * 64, out, s_part);
}

// Many reductions in one launch: job j sums nblk[j] partial slabs of nslots[j] floats into out[j].  The table travels
// in the kernel arguments (no device copy, capturable into a hipGraph); workgroup g serves one 64-slot group of one
// job (group_begin = prefix sum of ceil(nslots / 64)).
constexpr int RJ_MAX = 96;
struct ReduceJobTable {
    const float* partial[RJ_MAX];
    float* out[RJ_MAX];
    int nblk[RJ_MAX];
    int nslots[RJ_MAX];
    int group_begin[RJ_MAX + 1];
    int njobs;
};

__global__ __launch_bounds__(256) void reduce_jobs_kernel(const ReduceJobTable tbl) {
    __shared__ float s_part[4][64];
    const int g = blockIdx.x;
    int lo = 0, hi = tbl.njobs;                       // largest j with group_begin[j] <= g
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (tbl.group_begin[mid] <= g) lo = mid; else hi = mid;
    }
    reduce_slab64(tbl.partial[lo], tbl.nblk[lo], tbl.nslots[lo], (g - tbl.group_begin[lo]) * 64, tbl.out[lo], s_part);
}


// crfconv_reduce_jobs AND crfconv_reduce_jobs_f64 in one launch (the end of a backward pass runs both, on independent inputs: the
// float64 sums are ~650 wavefront-per-slot workgroups of latency, the float ones ~12 000 workgroups of bandwidth): the first n64
// workgroups take the float64 table -- they start first --, the others the float one.
__global__ __launch_bounds__(256) void reduce_both_kernel(const ReduceJobTable tbl, const Reduce64Table t64, const int n64) {
    __shared__ float s_part[4][64];
    if ((int)blockIdx.x < n64) {
        reduce_jobs_f64_body(t64, blockIdx.x);
        return;
    }
    const int g = (int)blockIdx.x - n64;
    int lo = 0, hi = tbl.njobs;                       // largest j with group_begin[j] <= g
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (tbl.group_begin[mid] <= g) lo = mid; else hi = mid;
    }
    reduce_slab64(tbl.partial[lo], tbl.nblk[lo], tbl.nslots[lo], (g - tbl.group_begin[lo]) * 64, tbl.out[lo], s_part);
}

}  // namespace crf

using namespace crf;

extern "C" size_t crfconv_linear_wgrad_workspace(int64_t M, int Co, int Ci) {
    if (M <= 0 || Co <= 0 || Ci <= 0) return 0;
    const WgPlan p = wg_plan(M, Co, Ci);
    return sizeof(float) * (size_t)p.nblk * ((size_t)Co * Ci + (size_t)Co) + 256;
}

static int wgrad_launch(const float* G, const float* X, int64_t M, int Co, int Ci, const WgPlan& p, float* partial,
                        float* partial_b, hipStream_t st) {
    const dim3 grid((unsigned)p.nblk, (unsigned)p.gy, (unsigned)p.gz), blk(WG_BLOCK);
#define WG(TA, TB) hipLaunchKernelGGL((wgrad_kernel<TA, TB>), grid, blk, 0, st, G, X, M, Co, Ci, p.rows_per_block, partial, partial_b)
    switch (p.tco * 10 + p.tci) {
        case 11: WG(1, 1); break;
        case 12: WG(1, 2); break;
        case 14: WG(1, 4); break;
        case 21: WG(2, 1); break;
        case 22: WG(2, 2); break;
        case 24: WG(2, 4); break;
        case 41: WG(4, 1); break;
        case 42: WG(4, 2); break;
        default: WG(4, 4); break;
    }
#undef WG
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_linear_wgrad_partial(const float* G, const float* X, int64_t M, int Co, int Ci, int want_bias,
                                            void* workspace, size_t workspace_bytes, int* nblk_out, crf_stream_t stream) {
    CRF_REQUIRE(G && X && workspace && nblk_out, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(M > 0 && Co > 0 && Ci > 0 && Co <= 4096 && Ci <= 4096, CRF_ERR_ARG, "bad shape M=%lld Co=%d Ci=%d",
                (long long)M, Co, Ci);
    CRF_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, CRF_ERR_ARG, "workspace must be 256-byte aligned");
    CRF_REQUIRE(workspace_bytes >= crfconv_linear_wgrad_workspace(M, Co, Ci), CRF_ERR_WORKSPACE, "workspace too small");
    const WgPlan p = wg_plan(M, Co, Ci);
    float* partial = reinterpret_cast<float*>(workspace);
    float* partial_b = want_bias ? partial + (size_t)p.nblk * Co * Ci : nullptr;
    *nblk_out = p.nblk;
    return wgrad_launch(G, X, M, Co, Ci, p, partial, partial_b, as_stream(stream));
}

extern "C" int crfconv_linear_wgrad_nblk(int64_t M, int Co, int Ci) {
    if (M <= 0 || Co <= 0 || Ci <= 0) return 0;
    return wg_plan(M, Co, Ci).nblk;
}

// crfconv_linear_wgrad_partial for several layers at once (jobs: host array): one launch per tile class present among the jobs
// (at most nine, typically one or two) instead of one per layer; identical partial slabs.
extern "C" int crfconv_linear_wgrad_partial_jobs(const crf_wgrad_job* jobs, int njobs, crf_stream_t stream) {
    CRF_REQUIRE(jobs || njobs == 0, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(njobs >= 0, CRF_ERR_ARG, "njobs=%d < 0", njobs);
    hipStream_t st = as_stream(stream);
    for (int j = 0; j < njobs; ++j) {
        const crf_wgrad_job& jb = jobs[j];
        CRF_REQUIRE(jb.G && jb.X && jb.workspace, CRF_ERR_ARG, "job %d: null pointer", j);
        CRF_REQUIRE(jb.M > 0 && jb.M < ((int64_t)1 << 31) && jb.Co > 0 && jb.Ci > 0 && jb.Co <= 4096 && jb.Ci <= 4096, CRF_ERR_ARG,
                    "job %d: bad shape M=%lld Co=%d Ci=%d", j, (long long)jb.M, jb.Co, jb.Ci);
        CRF_REQUIRE((reinterpret_cast<uintptr_t>(jb.workspace) & 255) == 0, CRF_ERR_ARG, "job %d: workspace must be 256-byte aligned", j);
        CRF_REQUIRE(jb.workspace_bytes >= crfconv_linear_wgrad_workspace(jb.M, jb.Co, jb.Ci), CRF_ERR_WORKSPACE, "job %d: workspace too small", j);
    }
    // jobs in the caller's order (longest first), WJ_MAX per launch
    for (int j0 = 0; j0 < njobs; j0 += WJ_MAX) {
        WgJobTable t;
        int64_t blocks = 0;
        const int n = njobs - j0 < WJ_MAX ? njobs - j0 : WJ_MAX;
        wg_fill_table(jobs + j0, n, t, blocks);
        CRF_REQUIRE(blocks < ((int64_t)1 << 31), CRF_ERR_UNSUPPORTED, "too many workgroups in one batch");
        hipLaunchKernelGGL(wgrad_jobs_any_kernel, dim3((unsigned)blocks), dim3(WG_BLOCK), 0, st, t);
        CRF_LAUNCH_CHECK();
    }
    return CRF_OK;
}

extern "C" int crfconv_reduce_jobs(const crf_reduce_job* jobs, int njobs, crf_stream_t stream) {
    CRF_REQUIRE(jobs || njobs == 0, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(njobs >= 0, CRF_ERR_ARG, "njobs=%d < 0", njobs);
    hipStream_t st = as_stream(stream);
    for (int j0 = 0; j0 < njobs; j0 += RJ_MAX) {
        ReduceJobTable tbl;
        const int n = njobs - j0 < RJ_MAX ? njobs - j0 : RJ_MAX;
        int64_t total = 0;
        for (int j = 0; j < n; ++j) {
            const crf_reduce_job& jb = jobs[j0 + j];
            CRF_REQUIRE(jb.partial && jb.out && jb.nblk > 0 && jb.nslots > 0, CRF_ERR_ARG, "job %d is malformed", j0 + j);
            tbl.partial[j] = jb.partial;
            tbl.out[j] = jb.out;
            tbl.nblk[j] = jb.nblk;
            tbl.nslots[j] = jb.nslots;
            tbl.group_begin[j] = (int)total;
            total += (jb.nslots + 63) / 64;
            CRF_REQUIRE(total < ((int64_t)1 << 30), CRF_ERR_ARG, "too many slots in one batch");
        }
        for (int j = n; j <= RJ_MAX; ++j) tbl.group_begin[j] = (int)total;
        for (int j = n; j < RJ_MAX; ++j) { tbl.partial[j] = nullptr; tbl.out[j] = nullptr; tbl.nblk[j] = 0; tbl.nslots[j] = 0; }
        tbl.njobs = n;
        hipLaunchKernelGGL(reduce_jobs_kernel, dim3((unsigned)total), dim3(256), 0, st, tbl);
        CRF_LAUNCH_CHECK();
    }
    return CRF_OK;
}

extern "C" int crfconv_reduce_jobs_f64(const crf_reduce64_job* jobs, int njobs, crf_stream_t stream);      // pointconv.hip
// Both kinds of sums of a backward pass in ONE launch when each fits one table (96 float jobs, 32 float64 jobs): same results as the
// two calls.  Larger batches: the two calls.
extern "C" int crfconv_reduce_jobs_both(const crf_reduce_job* jobs, int njobs, const crf_reduce64_job* jobs64, int njobs64,
                                        crf_stream_t stream) {
    CRF_REQUIRE((jobs || njobs == 0) && (jobs64 || njobs64 == 0) && njobs >= 0 && njobs64 >= 0, CRF_ERR_ARG, "null pointer or negative count");
    if (njobs == 0 || njobs64 == 0 || njobs > RJ_MAX || njobs64 > R64_MAX) {
        if (njobs64 > 0)
            if (int rc = crfconv_reduce_jobs_f64(jobs64, njobs64, stream)) return rc;
        return njobs > 0 ? crfconv_reduce_jobs(jobs, njobs, stream) : CRF_OK;
    }
    ReduceJobTable tbl;
    int64_t total = 0;
    for (int j = 0; j < njobs; ++j) {
        const crf_reduce_job& jb = jobs[j];
        CRF_REQUIRE(jb.partial && jb.out && jb.nblk > 0 && jb.nslots > 0, CRF_ERR_ARG, "job %d is malformed", j);
        tbl.partial[j] = jb.partial; tbl.out[j] = jb.out; tbl.nblk[j] = jb.nblk; tbl.nslots[j] = jb.nslots;
        tbl.group_begin[j] = (int)total;
        total += (jb.nslots + 63) / 64;
    }
    for (int j = njobs; j <= RJ_MAX; ++j) tbl.group_begin[j] = (int)total;
    for (int j = njobs; j < RJ_MAX; ++j) { tbl.partial[j] = nullptr; tbl.out[j] = nullptr; tbl.nblk[j] = 0; tbl.nslots[j] = 0; }
    tbl.njobs = njobs;
    Reduce64Table t;
    int64_t waves = 0;
    for (int j = 0; j <= R64_MAX; ++j) {
        t.wave_base[j] = (int)waves;
        if (j < njobs64) {
            const crf_reduce64_job& jb = jobs64[j];
            CRF_REQUIRE(jb.partial && jb.out && jb.nblk > 0 && jb.nblk < ((int64_t)1 << 31) && jb.nslots > 0, CRF_ERR_ARG, "float64 job %d is malformed", j);
            t.partial[j] = jb.partial; t.out[j] = jb.out; t.is_float[j] = jb.is_float; t.nblk[j] = (int)jb.nblk; t.nslots[j] = jb.nslots;
            waves += jb.nslots;
        } else if (j < R64_MAX) {
            t.partial[j] = nullptr; t.out[j] = nullptr; t.is_float[j] = 0; t.nblk[j] = 0; t.nslots[j] = 0;
        }
    }
    t.njobs = njobs64;
    const int64_t n64 = cdiv(waves, 256 / WAVE);
    CRF_REQUIRE(total + n64 < ((int64_t)1 << 30), CRF_ERR_UNSUPPORTED, "too many slots in one batch");
    hipLaunchKernelGGL(reduce_both_kernel, dim3((unsigned)(total + n64)), dim3(256), 0, as_stream(stream), tbl, t, (int)n64);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_linear_wgrad(const float* G, const float* X, int64_t M, int Co, int Ci, float* dW,
                                    float* db, void* workspace, size_t workspace_bytes, crf_stream_t stream) {
    CRF_REQUIRE(G && X && dW && workspace, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(M > 0 && Co > 0 && Ci > 0 && Co <= 4096 && Ci <= 4096, CRF_ERR_ARG, "bad shape M=%lld Co=%d Ci=%d",
                (long long)M, Co, Ci);
    CRF_REQUIRE(workspace_bytes >= crfconv_linear_wgrad_workspace(M, Co, Ci), CRF_ERR_WORKSPACE, "workspace too small");
    const WgPlan p = wg_plan(M, Co, Ci);
    hipStream_t st = as_stream(stream);
    float* partial = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    float* partial_b = db ? partial + (size_t)p.nblk * Co * Ci : nullptr;
    if (int rc = wgrad_launch(G, X, M, Co, Ci, p, partial, partial_b, st)) return rc;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv((int64_t)Co * Ci, 64)), dim3(256), 0, st, partial, p.nblk,
                       Co * Ci, dW);
    CRF_LAUNCH_CHECK();
    if (db) {
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv((int64_t)Co, 64)), dim3(256), 0, st, partial_b, p.nblk,
                           Co, db);
        CRF_LAUNCH_CHECK();
    }
    return CRF_OK;
}

// ------------------------------------------------------------------ (I + C)^-1 for the CRF layers
// In-place Gauss-Jordan on the H x H (H <= 64) symmetric positive definite matrix M = I + c^T c
// (models/continuous_crf_conv_big.py:72 calls .inverse() inside the loop; it is loop invariant).  No pivoting
// needed (eigenvalues >= 1), float64 throughout.  One workgroup of 16 x 16 threads; thread (tr, tc) keeps the 4 x 4
// cyclic sub-tile rows tr + 16 i, columns tc + 16 j in REGISTERS for the whole elimination, and only the old pivot
// row / column travel through (double-buffered) LDS: one barrier and 16 fused multiply-adds per thread per pivot,
// ~0.1 us a pivot instead of the ~3 us of an all-in-LDS sweep.  Replaces torch.linalg.inv, whose rocSOLVER path
// synchronises and therefore cannot be captured into a hipGraph.
namespace crf {
// In-place inverse of the 64 x 64 register-tiled matrix (rows / columns >= H must be identity).
__device__ __forceinline__ void gauss_jordan_tiles(double (&t)[4][4], int H, double (*s_row)[64], double (*s_col)[64]) {
    const int tr = threadIdx.x >> 4, tc = threadIdx.x & 15;
#pragma unroll
    for (int ip = 0; ip < 4; ++ip) {                     // pivot p = 16 ip + pp lives in local row / column ip
        for (int pp = 0; pp < 16; ++pp) {
            const int p = 16 * ip + pp;
            if (p >= H) break;                           // uniform: rows beyond H are identity already
            const int b = p & 1;
            if (tr == pp) {
#pragma unroll
                for (int j = 0; j < 4; ++j) s_row[b][tc + 16 * j] = t[ip][j];
            }
            if (tc == pp) {
#pragma unroll
                for (int i = 0; i < 4; ++i) s_col[b][tr + 16 * i] = t[i][ip];
            }
            __syncthreads();
            const double piv = 1.0 / s_row[b][p];
            double rowv[4], colv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) rowv[j] = s_row[b][tc + 16 * j] * piv;
#pragma unroll
            for (int i = 0; i < 4; ++i) colv[i] = s_col[b][tr + 16 * i];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool rp = (i == ip) && (tr == pp), cp = (j == ip) && (tc == pp);
                    const double upd = t[i][j] - colv[i] * rowv[j];
                    t[i][j] = rp ? (cp ? piv : rowv[j]) : (cp ? -colv[i] * piv : upd);
                }
        }
    }
}

__global__ __launch_bounds__(256) void spd_inverse_kernel(const float* __restrict__ Min, int H,
                                                          float* __restrict__ Qout) {
    __shared__ double s_row[2][64], s_col[2][64];
    const int tr = threadIdx.x >> 4, tc = threadIdx.x & 15;
    double t[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = tr + 16 * i, c = tc + 16 * j;
            t[i][j] = (r < H && c < H) ? (double)Min[r * H + c] : (r == c ? 1.0 : 0.0);
        }
    gauss_jordan_tiles(t, H, s_row, s_col);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = tr + 16 * i, c = tc + 16 * j;
            if (r < H && c < H) Qout[r * H + c] = (float)t[i][j];
        }
}

__global__ __launch_bounds__(CMF_BLOCK) void crf_matrices_kernel(const float* __restrict__ cmat, int H,
                                                           float* __restrict__ Qout, float* __restrict__ Pout) {
    __shared__ __attribute__((aligned(16))) char lds[CMF_LDS_BYTES];
    crf_matrices_body(cmat, H, Qout, Pout, lds);
}

// All CRF layers of a network in ONE launch (one workgroup each; CrfMatJobs: crf_matrices_body.hpp)
__global__ __launch_bounds__(CMF_BLOCK) void crf_matrices_batched_kernel(const CrfMatJobs j) {
    const int b = blockIdx.x;
    __shared__ __attribute__((aligned(16))) char lds[CMF_LDS_BYTES];
    crf_matrices_body(j.c[b], j.H[b], j.Q[b], j.P[b], lds);
}

// dc from dQ and dP (either may be NULL = zero).  With D = dQ - dP (P = I - Q), M = I + c^T c:
//   dM = -Q^T D Q^T,   dc = c (dM + dM^T) = -c (S + S^T),   S = Q^T D Q^T.
// Q is symmetric (the inverse of a symmetric matrix; its float rounding included, to ~1e-8), so S + S^T = Q (D + D^T) Q and
//   dc = -((c Q) (D + D^T)) Q
// is a chain of three products in which every ROW of the result depends on the same row of c only: a workgroup takes a slab
// of CMB_ROWS rows through all three products without ever meeting another workgroup (H / CMB_ROWS workgroups per layer,
// float64 accumulation, operands as float in LDS).  The one-workgroup form (T = Q^T D, S = T Q^T, c (S + S^T): three full
// H^3 float64 products on 1024 threads of ONE CU, 35 us for the four layers of PointConvBig) was instruction-bound.
constexpr int CMB_ROWS = 8, CMB_BLOCK = CMB_ROWS * 64;
__device__ __forceinline__ void crf_matrices_bwd_slab(const float* __restrict__ cmat, const float* __restrict__ Q,
                                                      const float* __restrict__ dQ, const float* __restrict__ dP,
                                                      int H, int row0, float* __restrict__ dc) {
    __shared__ float s_q[64 * 65], s_d[64 * 65];
    __shared__ double s_t[2][CMB_ROWS][65];
    for (int e = threadIdx.x; e < H * H; e += (int)blockDim.x) {              // (a rider in a launch of larger workgroups: mlp_dw_jobs_hosting_kernel)
        const int r = e / H, c = e % H;
        s_q[r * 65 + c] = Q[e];
        s_d[r * 65 + c] = (dQ ? dQ[e] : 0.f) - (dP ? dP[e] : 0.f);            // D
    }
    const int r = (threadIdx.x >> 6) & (CMB_ROWS - 1), j = threadIdx.x & 63, row = row0 + r;
    const bool live = (int)threadIdx.x < CMB_BLOCK && row < H && j < H;
    if (live) s_t[0][r][j] = (double)cmat[row * H + j];
    __syncthreads();
    double acc = 0.0;
    if (live)
        for (int k = 0; k < H; ++k) acc += s_t[0][r][k] * (double)s_q[k * 65 + j];                 // (c Q)[row][j]
    if (live) s_t[1][r][j] = acc;
    __syncthreads();
    acc = 0.0;
    if (live)
        for (int k = 0; k < H; ++k) acc += s_t[1][r][k] * ((double)s_d[k * 65 + j] + (double)s_d[j * 65 + k]);   // . (D + D^T)
    if (live) s_t[0][r][j] = acc;
    __syncthreads();
    acc = 0.0;
    if (live) {
        for (int k = 0; k < H; ++k) acc += s_t[0][r][k] * (double)s_q[k * 65 + j];                 // . Q
        dc[row * H + j] = (float)(-acc);
    }
}
__global__ __launch_bounds__(CMB_BLOCK) void crf_matrices_bwd_kernel(const float* __restrict__ cmat, const float* __restrict__ Q,
                                                               const float* __restrict__ dQ, const float* __restrict__ dP,
                                                               int H, float* __restrict__ dc) {
    crf_matrices_bwd_slab(cmat, Q, dQ, dP, H, (int)blockIdx.x * CMB_ROWS, dc);
}
__global__ __launch_bounds__(CMB_BLOCK) void crf_matrices_bwd_batched_kernel(const CrfMatJobs j) {
    int b = 0;                                                  // layer of this workgroup: slab_base is a prefix over the layers
    while (b + 1 < CM_MAX && (int)blockIdx.x >= j.slab_base[b + 1]) ++b;
    crf_matrices_bwd_slab(j.c[b], j.Q_in[b], j.gQ[b], j.gP[b], j.H[b], ((int)blockIdx.x - j.slab_base[b]) * CMB_ROWS, j.dc[b]);
}
// mlp_dw_jobs_kernel CARRYING the slabs of crf_matrices_bwd_batched_kernel as its first n_side workgroups (round 6): both are
// end-of-pass parameter work that nothing waits for; on its own the matrices' backward is a 13 us chain of three dependent float64
// products on 15 workgroups.  A rider uses the first CMB_BLOCK threads of its (MF_BLOCK-thread) workgroup.
static_assert(MF_BLOCK >= CMB_BLOCK, "a rider fits the host's workgroup");
__global__ __launch_bounds__(MF_BLOCK) void mlp_dw_jobs_hosting_kernel(const MlpDwTable t, const CrfMatJobs j, const int n_side) {
    if ((int)blockIdx.x < n_side) {
        int b = 0;
        while (b + 1 < CM_MAX && (int)blockIdx.x >= j.slab_base[b + 1]) ++b;
        crf_matrices_bwd_slab(j.c[b], j.Q_in[b], j.gQ[b], j.gP[b], j.H[b], ((int)blockIdx.x - j.slab_base[b]) * CMB_ROWS, j.dc[b]);
        return;
    }
    const int g = (int)blockIdx.x - n_side;
    int lo = 0, hi = t.njobs;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (t.group_begin[mid] <= g) lo = mid; else hi = mid;
    }
    mlp_dw_slots(t.PA[lo], t.PB[lo], t.PG[lo], t.PX[lo], t.nblk[lo], t.coef[lo], (int64_t)t.M[lo], t.Co[lo], t.Ci[lo],
                 g - t.group_begin[lo], t.dW[lo]);
}
}  // namespace crf

extern "C" int crfconv_crf_matrices_batched(const float* const* c, const int* H, int n, float* const* Q, float* const* P,
                                            crf_stream_t stream) {
    CRF_REQUIRE(c && H && Q && P && n >= 1 && n <= crf::CM_MAX, CRF_ERR_ARG, "null pointer or n=%d outside [1, %d]", n, crf::CM_MAX);
    crf::CrfMatJobs j = {};
    for (int i = 0; i < n; ++i) {
        CRF_REQUIRE(c[i] && Q[i] && P[i] && H[i] >= 1 && H[i] <= 64, CRF_ERR_ARG, "job %d: null pointer or H=%d outside [1, 64]", i, H[i]);
        j.c[i] = c[i]; j.Q[i] = Q[i]; j.P[i] = P[i]; j.H[i] = H[i];
    }
    hipLaunchKernelGGL(crf::crf_matrices_batched_kernel, dim3((unsigned)n), dim3(crf::CMF_BLOCK), 0, crf::as_stream(stream), j);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

static int crf_matrices_bwd_jobs(const float* const* c, const float* const* Q, const float* const* gQ, const float* const* gP, const int* H, int n,
                                 float* const* dc, crf::CrfMatJobs& j, int& nslab) {
    CRF_REQUIRE(c && Q && gQ && gP && H && dc && n >= 1 && n <= crf::CM_MAX, CRF_ERR_ARG, "null pointer or n=%d outside [1, %d]", n, crf::CM_MAX);
    j = crf::CrfMatJobs();
    for (int i = 0; i < n; ++i) {
        CRF_REQUIRE(c[i] && Q[i] && dc[i] && H[i] >= 1 && H[i] <= 64, CRF_ERR_ARG, "job %d: null pointer or H=%d outside [1, 64]", i, H[i]);
        j.c[i] = c[i]; j.Q_in[i] = Q[i]; j.gQ[i] = gQ[i]; j.gP[i] = gP[i]; j.dc[i] = dc[i]; j.H[i] = H[i];
        j.slab_base[i + 1] = j.slab_base[i] + (H[i] + crf::CMB_ROWS - 1) / crf::CMB_ROWS;
    }
    nslab = j.slab_base[n];
    for (int i = n; i < crf::CM_MAX; ++i) j.slab_base[i + 1] = 0x7fffffff;      // (never reached by a block index)
    return CRF_OK;
}
extern "C" int crfconv_crf_matrices_backward_batched(const float* const* c, const float* const* Q, const float* const* gQ,
                                                     const float* const* gP, const int* H, int n, float* const* dc,
                                                     crf_stream_t stream) {
    crf::CrfMatJobs j;
    int nslab = 0;
    if (int rc = crf_matrices_bwd_jobs(c, Q, gQ, gP, H, n, dc, j, nslab)) return rc;
    hipLaunchKernelGGL(crf::crf_matrices_bwd_batched_kernel, dim3((unsigned)nslab), dim3(crf::CMB_BLOCK), 0, crf::as_stream(stream), j);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

namespace crf {
// Q = M^-1 for a symmetric positive definite M [H, H] with 64 < H <= 512 (the 128- / 256-channel CRF stages of the sparse networks,
// models/point_conv.py:318-339: M = I + c^T c, eigenvalues >= 1): in-place Gauss-Jordan WITHOUT pivoting on the copy in Q, one
// workgroup, the matrix in global memory behind this CU's caches (256 KB at H = 256 -- it does not fit LDS), pivot row and column
// staged in LDS per step.  H steps of one read-modify-write pass each: ~0.1 ms at H = 128, ~1 ms at H = 256 -- once per forward of
// a layer whose reference recomputes torch.inverse in every mean-field step (continuous_crf_conv.py:66).
constexpr int SW_BLOCK = 1024, SW_MAXH = 512;
__global__ __launch_bounds__(SW_BLOCK) void spd_inverse_wide_kernel(const float* __restrict__ M, int H, float* __restrict__ Q) {
    __shared__ float s_row[SW_MAXH], s_col[SW_MAXH];
    const int n = H * H;
    for (int e = threadIdx.x; e < n; e += SW_BLOCK) Q[e] = M[e];
    __syncthreads();
    for (int k = 0; k < H; ++k) {
        for (int t = threadIdx.x; t < H; t += SW_BLOCK) {
            s_row[t] = Q[k * H + t];
            s_col[t] = Q[t * H + k];
        }
        __syncthreads();
        const float inv = 1.0f / s_row[k];
        for (int e = threadIdx.x; e < n; e += SW_BLOCK) {
            const int i = e / H, j = e - i * H;
            float v;
            if (i == k) v = j == k ? inv : s_row[j] * inv;
            else if (j == k) v = -s_col[i] * inv;
            else v = fmaf(-s_col[i] * inv, s_row[j], Q[e]);
            Q[e] = v;
        }
        __syncthreads();
    }
}
}  // namespace crf

extern "C" int crfconv_spd_inverse_wide(const float* M, int H, float* Q, crf_stream_t stream) {
    CRF_REQUIRE(M && Q && M != Q, CRF_ERR_ARG, "null pointer / aliased operands");
    CRF_REQUIRE(H >= 1 && H <= crf::SW_MAXH, CRF_ERR_UNSUPPORTED, "H=%d outside [1, %d]", H, crf::SW_MAXH);
    hipLaunchKernelGGL(crf::spd_inverse_wide_kernel, dim3(1), dim3(crf::SW_BLOCK), 0, crf::as_stream(stream), M, H, Q);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_spd_inverse(const float* M, int H, float* Q, crf_stream_t stream) {
    CRF_REQUIRE(M && Q, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(H >= 1 && H <= 64, CRF_ERR_UNSUPPORTED, "H=%d outside [1, 64]", H);
    hipLaunchKernelGGL(crf::spd_inverse_kernel, dim3(1), dim3(256), 0, crf::as_stream(stream), M, H, Q);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_crf_matrices(const float* c, int H, float* Q, float* P, crf_stream_t stream) {
    CRF_REQUIRE(c && Q && P, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(H >= 1 && H <= 64, CRF_ERR_UNSUPPORTED, "H=%d outside [1, 64]", H);
    hipLaunchKernelGGL(crf::crf_matrices_kernel, dim3(1), dim3(crf::CMF_BLOCK), 0, crf::as_stream(stream), c, H, Q, P);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_crf_matrices_backward(const float* c, const float* Q, const float* dQ, const float* dP, int H,
                                             float* dc, crf_stream_t stream) {
    CRF_REQUIRE(c && Q && dc, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(H >= 1 && H <= 64, CRF_ERR_UNSUPPORTED, "H=%d outside [1, 64]", H);
    hipLaunchKernelGGL(crf::crf_matrices_bwd_kernel, dim3((unsigned)((H + crf::CMB_ROWS - 1) / crf::CMB_ROWS)), dim3(crf::CMB_BLOCK), 0,
                       crf::as_stream(stream), c, Q, dQ, dP, H, dc);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

// ====================================================================== Y = X W^T (+ b) with BN statistics
// The per-point Linear layers at the fine levels: m = 10^4..10^5 rows, Ci, Co <= 128.  One wavefront owns 16 rows
// and all Co outputs; X rows are read ONCE as float4 (lane l: row l & 15, k = 16 c + 4 (l >> 4) + {0..3}), W sits in
// LDS (rows padded by 4 floats: conflict-free ds_read_b128), and each float4 pair feeds four
// v_mfma_f32_16x16x4_f32 steps (step s takes component s of every lane's float4: the k order inside a 16-chunk is
// permuted identically for both operands, which a sum does not notice).  Output tile D[co][row]: lane holds 4
// consecutive co of one row -> one float4 store.  Optional epilogue: per-block shifted sums / sums of squares of
// every output channel, so BatchNorm needs no separate statistics pass over Y.
namespace crf {

#ifndef LF_BLOCK_
#define LF_BLOCK_ 256
#endif
constexpr int LF_BLOCK = LF_BLOCK_, LF_WAVES = LF_BLOCK / WAVE;

// PRO: the operand is not read but formed while loading (dX of the fused MLP backward): row r, channel k of
//   gY = alpha[k] * lrelu'(a[k] y + b[k]) * X[r][k] + bet[k] * Y2[r][k] + del[k]     (X = gA, Y2 = the Linear's output y)
// pro = [5][Ci] floats a | b | alpha | bet | del, staged in LDS behind the weight slab (Ci % 4 == 0 required).
// VEC4: Ci % 4 == 0 and Co % 4 == 0 -- every access is a 16-byte one and the element-wise tail code does not exist.  (With
// both forms in one kernel the compiler merges the float4 store into the four predicated dword stores of the tail path:
// 4x the store instructions and 3x the write requests, 983 k instead of 328 k per 21 MB -- TCP_TCC_WRITE_REQ.)
// EPI: 0 plain store; 1 (PRO) Y += addend; 2 (not PRO) dropout mask on Y.  Template forms, so that the common kernels keep their
// register budget (as run-time branches the two epilogues cost <2, false> and <4, true> one wavefront per SIMD each).
constexpr int EPI_NONE = 0, EPI_ADD = 1, EPI_DROPOUT = 2;
// NCH > 0 (round 4; Ci <= 16 NCH, VEC4): the operand fragments of ALL k chunks of a row group are requested at once and those of
// the wavefront's NEXT row group before the current group's products (NCH <= LF_PF_MAX) -- the rolled loop (NCH = 0) pays one
// dependent memory round trip per chunk, eight per group at 128 inputs, with two to four wavefronts per SIMD to hide them.
#ifndef LF_PF_MAX_
#define LF_PF_MAX_ 4
#endif
template <int TCO, bool PRO = false, bool VEC4 = true, int EPI = EPI_NONE, int NCH = 0>  // 16 * TCO output channels per block slab (blockIdx.y picks the slab)
__global__ __launch_bounds__(LF_BLOCK) void linear_fwd_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                              const float* __restrict__ bias, int64_t M, int Ci, int Co,
                                                              int transpose_w, float* __restrict__ Y,
                                                              float* __restrict__ stat_partial /*[nblk][Co][4] or null*/,
                                                              const float* __restrict__ Y2 = nullptr,
                                                              const float* __restrict__ pro = nullptr, float slope = 1.f,
                                                              const float* __restrict__ Xb = nullptr, int xsplit = 0,
                                                              float* __restrict__ Yb = nullptr, int ysplit = 0,
                                                              const float* __restrict__ addend = nullptr,
                                                              const long long* __restrict__ drop_counter = nullptr,
                                                              unsigned long long drop_seed = 0ull, unsigned drop_threshold = 0u,
                                                              float drop_scale = 1.f) {
    // drop_counter (not PRO, one-pointer output): Y = dropout_mask .* (X W^T) * drop_scale with the counter-based mask of
    // common.hpp (element e = row * Co + column) -- the backward of nn.Dropout applied while the gradient of the Linear
    // BEHIND the dropout is written, instead of in a pass of its own over [M, Co].
    // addend [M, Co] (PRO only, one-pointer output): Y = X W^T + addend -- the gradient the other consumer of the block's input
    // sent back, so that autograd's accumulation pass over three [M, Co] tensors never runs.
    // Xb / xsplit: the operand is the column concatenation [X | Xb] split at column xsplit (the fusion layers' torch.cat,
    // never materialised); Yb / ysplit: the output columns >= ysplit go to Yb [M, Co - ysplit] (dX of such a layer).
    // Both splits are multiples of 4.
    extern __shared__ float sW[];                 // [16*TCO][Cip] (+ [5][Cik] prologue coefficients)
    const int Cip = ((Ci + 15) / 16) * 16 + 4;
    const int Cik = ((Ci + 15) / 16) * 16;
    float* sPro = sW + 16 * TCO * Cip;
    [[maybe_unused]] float* sTile = sPro + (PRO ? 5 * Cik : 0);          // [4 waves][16][16 TCO + 4] output staging (TCO >= 2)
    if constexpr (PRO) {
        for (int t = threadIdx.x; t < 5 * Cik; t += LF_BLOCK) {
            const int which = t / Cik, k = t - which * Cik;
            sPro[t] = k < Ci ? pro[which * Ci + k] : 0.f;
        }
    }
    const int co_base = blockIdx.y * 16 * TCO;
    if (!transpose_w && (Ci % 4) == 0) {          // rows of W are contiguous: 16-byte loads
        const int Cip4 = Cip / 4, Ci4 = Ci / 4;
        for (int t = threadIdx.x; t < 16 * TCO * Cip4; t += LF_BLOCK) {
            const int r = t / Cip4, k4 = t - r * Cip4;
            const int co = co_base + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (co < Co && k4 < Ci4) v = *reinterpret_cast<const float4*>(W + (int64_t)co * Ci + 4 * k4);
            *reinterpret_cast<float4*>(sW + r * Cip + 4 * k4) = v;
        }
    } else {
        for (int t = threadIdx.x; t < 16 * TCO * Cip; t += LF_BLOCK) {
            const int r = t / Cip, k = t - r * Cip;
            const int co = co_base + r;
            float v = 0.f;
            if (co < Co && k < Ci) v = transpose_w ? W[(int64_t)k * Co + co] : W[(int64_t)co * Ci + k];
            sW[t] = v;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rr = lane & 15, g = lane >> 4;
    const bool vec = (Ci % 4) == 0;
    float bsel[TCO][4];
#pragma unroll
    for (int t = 0; t < TCO; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int co = co_base + 16 * t + 4 * g + e;
            bsel[t][e] = (bias != nullptr && co < Co) ? bias[co] : 0.f;
        }
    float s1[TCO][4], s2[TCO][4], sh[TCO][4];
#pragma unroll
    for (int t = 0; t < TCO; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) { s1[t][e] = 0.f; s2[t][e] = 0.f; sh[t][e] = 0.f; }
    bool have_shift = false;
    const int nchunk = (Ci + 15) / 16;

    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // raw operand fragment(s) of chunk c for row r: X (either layout / the two-pointer form), or (gA, y) with PRO
    auto load_raw = [&](int64_t r, bool rv, int c, float4& xa, float4& xb2) {
        const int k0 = 16 * c + 4 * g;
        xa = zero4;
        xb2 = zero4;
        if (!rv || k0 >= Ci) return;
        if constexpr (PRO) {
            xa = *reinterpret_cast<const float4*>(X + r * Ci + k0);
            xb2 = *reinterpret_cast<const float4*>(Y2 + r * Ci + k0);
        } else if (Xb != nullptr) {
            if (k0 < xsplit) xa = *reinterpret_cast<const float4*>(X + r * xsplit + k0);
            else xa = *reinterpret_cast<const float4*>(Xb + r * (Ci - xsplit) + (k0 - xsplit));
        } else if (VEC4 || vec) {
            xa = *reinterpret_cast<const float4*>(X + r * Ci + k0);
        } else if constexpr (!VEC4) {
            const float* xp = X + r * Ci;
            xa.x = xp[k0];
            xa.y = k0 + 1 < Ci ? xp[k0 + 1] : 0.f;
            xa.z = k0 + 2 < Ci ? xp[k0 + 2] : 0.f;
            xa.w = k0 + 3 < Ci ? xp[k0 + 3] : 0.f;
        }
    };
    // the MFMA operand of chunk c: the raw fragment, or gY formed from (gA, y) and the staged coefficients
    auto operand = [&](bool rv, int c, float4 gv, float4 yv) -> float4 {
        if constexpr (PRO) {
            const int k0 = 16 * c + 4 * g;
            float4 xv = zero4;
            if (rv && k0 < Ci) {
                const float4 pa = *reinterpret_cast<const float4*>(sPro + k0), pb = *reinterpret_cast<const float4*>(sPro + Cik + k0);
                const float4 al = *reinterpret_cast<const float4*>(sPro + 2 * Cik + k0), be = *reinterpret_cast<const float4*>(sPro + 3 * Cik + k0);
                const float4 de = *reinterpret_cast<const float4*>(sPro + 4 * Cik + k0);
                xv.x = fmaf(al.x * (fmaf(pa.x, yv.x, pb.x) > 0.f ? 1.f : slope), gv.x, fmaf(be.x, yv.x, de.x));
                xv.y = fmaf(al.y * (fmaf(pa.y, yv.y, pb.y) > 0.f ? 1.f : slope), gv.y, fmaf(be.y, yv.y, de.y));
                xv.z = fmaf(al.z * (fmaf(pa.z, yv.z, pb.z) > 0.f ? 1.f : slope), gv.z, fmaf(be.z, yv.z, de.z));
                xv.w = fmaf(al.w * (fmaf(pa.w, yv.w, pb.w) > 0.f ? 1.f : slope), gv.w, fmaf(be.w, yv.w, de.w));
            }
            return xv;
        } else {
            return gv;
        }
    };
    // (Issuing the operand loads of four row groups in one burst, or prefetching the next group, measured no faster: the
    // write-heavy shapes run at the ~2.7 TB/s HBM WRITE rate -- 163840 x 8 -> 32 moves 21 MB out in 13 us -- not at a
    // per-wavefront latency limit.)
    constexpr int NCA = NCH > 0 ? NCH : 1;
    constexpr bool PF = NCH > 0 && NCH <= LF_PF_MAX_;                 // next group's fragments in flight too
    const int64_t row_stride = (int64_t)gridDim.x * (LF_BLOCK / WAVE) * 16;
    [[maybe_unused]] float4 fa[NCA], fb[PRO ? NCA : 1], na[PF ? NCA : 1], nb[(PF && PRO) ? NCA : 1];
    [[maybe_unused]] auto load_group = [&](int64_t rw0, float4 (&xa)[NCA], float4 (&xb)[PRO ? NCA : 1]) {
        const int64_t rq = rw0 + rr;
        const bool ok = rw0 < M && rq < M;
#pragma unroll
        for (int c = 0; c < NCA; ++c) {
            float4 t0, t1;
            load_raw(rq, ok, c, t0, t1);
            xa[c] = t0;
            if constexpr (PRO) xb[c] = t1;
        }
    };
    if constexpr (PF) load_group(((int64_t)blockIdx.x * (LF_BLOCK / WAVE) + wave) * 16, fa, fb);
    for (int64_t row0 = ((int64_t)blockIdx.x * (LF_BLOCK / WAVE) + wave) * 16; row0 < M; row0 += row_stride) {
        const int64_t r = row0 + rr;
        const bool rv = r < M;
        f32x4 acc[TCO];
#pragma unroll
        for (int t = 0; t < TCO; ++t) acc[t] = f32x4{bsel[t][0], bsel[t][1], bsel[t][2], bsel[t][3]};
        if constexpr (NCH > 0) {
            if constexpr (PF) {
                if constexpr (PRO) load_group(row0 + row_stride, na, nb);
                else load_group(row0 + row_stride, na, fb);
            } else {
                load_group(row0, fa, fb);
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int k0 = 16 * c + 4 * g;
                const float4 xv = operand(rv, c, fa[c], fb[PRO ? c : 0]);
#pragma unroll
                for (int t = 0; t < TCO; ++t) {
                    const float4 wv = *reinterpret_cast<const float4*>(sW + (16 * t + rr) * Cip + k0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.x, xv.x, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.y, xv.y, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.z, xv.z, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.w, xv.w, acc[t], 0, 0, 0);
                }
            }
            if constexpr (PF) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    fa[c] = na[c];
                    if constexpr (PRO) fb[c] = nb[c];
                }
            }
        } else {
        for (int c = 0; c < nchunk; ++c) {
            const int k0 = 16 * c + 4 * g;
            float4 ra, rb;
            load_raw(r, rv, c, ra, rb);
            const float4 xv = operand(rv, c, ra, rb);
#pragma unroll
            for (int t = 0; t < TCO; ++t) {
                const float4 wv = *reinterpret_cast<const float4*>(sW + (16 * t + rr) * Cip + k0);
                // D[i = co][j = row]: A = W fragment (i = lane & 15), B = X fragment (j = lane & 15)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.x, xv.x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.y, xv.y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.z, xv.z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.w, xv.w, acc[t], 0, 0, 0);
            }
        }
        }
        // lane holds Y[row = row0 + rr][co = co_base + 16 t + 4 g + e], e = 0..3
        if constexpr (TCO >= 2) {
            // Stored straight from the accumulators every store instruction writes 64 bytes into each of 16 rows (measured
            // 1.5-1.7 TB/s on write-heavy shapes); through a per-wave LDS tile [16 rows][16 TCO] every instruction writes
            // whole 128 / 256-byte row segments, consecutive lanes consecutive addresses.
            constexpr int TW = 16 * TCO, TLD = TW + 4, F4R = TW / 4;       // tile width, padded row, float4 per row
            float* tile = sTile + wave * 16 * TLD;
#pragma unroll
            for (int t = 0; t < TCO; ++t)
                *reinterpret_cast<float4*>(tile + rr * TLD + 16 * t + 4 * g) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
            __builtin_amdgcn_wave_barrier();               // LDS operations of one wave complete in order
#pragma unroll
            for (int i = 0; i < (16 * F4R) / WAVE; ++i) {
                const int qd = lane + WAVE * i, trow = qd / F4R, tc4 = qd - trow * F4R;
                const int64_t orow = row0 + trow;
                const int co = co_base + 4 * tc4;
                if (orow < M && co < Co) {
                    float4 o4 = *reinterpret_cast<const float4*>(tile + trow * TLD + 4 * tc4);
                    if constexpr (EPI == EPI_ADD) {
                        {
                            if (VEC4 || (Co % 4) == 0) {
                                const float4 a4 = *reinterpret_cast<const float4*>(addend + orow * Co + co);
                                o4.x += a4.x; o4.y += a4.y; o4.z += a4.z; o4.w += a4.w;
                            } else {
                                o4.x += addend[orow * Co + co];
                                if (co + 1 < Co) o4.y += addend[orow * Co + co + 1];
                                if (co + 2 < Co) o4.z += addend[orow * Co + co + 2];
                                if (co + 3 < Co) o4.w += addend[orow * Co + co + 3];
                            }
                        }
                    }
                    if constexpr (EPI == EPI_DROPOUT) {
                        {
                            const unsigned long long ctr = (unsigned long long)drop_counter[0];
                            const unsigned long long e = (unsigned long long)(orow * Co + co);
                            o4.x = dropout_keep(drop_seed, ctr, e, drop_threshold) ? o4.x * drop_scale : 0.f;
                            o4.y = dropout_keep(drop_seed, ctr, e + 1, drop_threshold) ? o4.y * drop_scale : 0.f;
                            o4.z = dropout_keep(drop_seed, ctr, e + 2, drop_threshold) ? o4.z * drop_scale : 0.f;
                            o4.w = dropout_keep(drop_seed, ctr, e + 3, drop_threshold) ? o4.w * drop_scale : 0.f;
                        }
                    }
                    if (Yb != nullptr) {
                        if (co < ysplit) *reinterpret_cast<float4*>(Y + orow * ysplit + co) = o4;
                        else *reinterpret_cast<float4*>(Yb + orow * (Co - ysplit) + (co - ysplit)) = o4;
                    } else if (VEC4 || (Co % 4) == 0) {
                        *reinterpret_cast<float4*>(Y + orow * Co + co) = o4;
                    } else if constexpr (!VEC4) {
                        const float ov[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (co + e < Co) Y[orow * Co + co + e] = ov[e];
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();               // the tile is rewritten by the next row group
        } else {
#pragma unroll
        for (int t = 0; t < TCO; ++t) {
            const int co = co_base + 16 * t + 4 * g;
            if constexpr (EPI == EPI_ADD) {
                if (rv) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (co + e < Co) acc[t][e] += addend[r * Co + co + e];
                }
            }
            if constexpr (EPI == EPI_DROPOUT) {
                if (rv) {
                    const unsigned long long ctr = (unsigned long long)drop_counter[0];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (co + e < Co)
                            acc[t][e] = dropout_keep(drop_seed, ctr, (unsigned long long)(r * Co + co + e), drop_threshold)
                                            ? acc[t][e] * drop_scale : 0.f;
                }
            }
            if (rv) {
                if (Yb != nullptr) {
                    const float4 o4 = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
                    if (co < ysplit) *reinterpret_cast<float4*>(Y + r * ysplit + co) = o4;
                    else if (co < Co) *reinterpret_cast<float4*>(Yb + r * (Co - ysplit) + (co - ysplit)) = o4;
                } else if (VEC4) {
                    if (co < Co) *reinterpret_cast<float4*>(Y + r * Co + co) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
                } else if (co + 3 < Co && (Co % 4) == 0) {
                    *reinterpret_cast<float4*>(Y + r * Co + co) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
                } else if constexpr (!VEC4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (co + e < Co) Y[r * Co + co + e] = acc[t][e];
                }
            }
        }
        }
        if (stat_partial != nullptr) {
            if (!have_shift) {   // shift = this wave's first row (lane with rr == 0 of each co group)
#pragma unroll
                for (int t = 0; t < TCO; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) sh[t][e] = __shfl(acc[t][e], 16 * g, WAVE);
                have_shift = true;
            }
            if (rv) {
#pragma unroll
                for (int t = 0; t < TCO; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float dlt = acc[t][e] - sh[t][e];
                        s1[t][e] += dlt;
                        s2[t][e] = fmaf(dlt, dlt, s2[t][e]);
                    }
            }
        }
    }
    if (stat_partial != nullptr) {
        // one record {shift, n, sum, sumsq} per BLOCK and channel: the 16 row-lanes fold by shuffles, the 4 waves
        // through LDS, re-based on wave 0's shift (sum (v - s0) = a + n d, sum (v - s0)^2 = b + 2 d a + n d^2)
        __syncthreads();                                 // sW is dead: reuse it as [4 waves][4][16*TCO]
        float* sw = sW + wave * 4 * 16 * TCO;
        // rows this wave actually accumulated
        int64_t nrows = 0;
        for (int64_t row0 = ((int64_t)blockIdx.x * (LF_BLOCK / WAVE) + wave) * 16; row0 < M;
             row0 += (int64_t)gridDim.x * (LF_BLOCK / WAVE) * 16)
            nrows += (M - row0) < 16 ? (M - row0) : 16;
#pragma unroll
        for (int t = 0; t < TCO; ++t)
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
                    sw[16 * TCO + cl] = (float)nrows;
                    sw[2 * 16 * TCO + cl] = a;
                    sw[3 * 16 * TCO + cl] = b;
                }
            }
        __syncthreads();
        for (int cl = threadIdx.x; cl < 16 * TCO; cl += LF_BLOCK) {
            const int co = co_base + cl;
            if (co >= Co) continue;
            const float s0 = sW[cl];
            double n = 0.0, S1 = 0.0, S2 = 0.0;
            for (int w = 0; w < LF_BLOCK / WAVE; ++w) {
                const float* q = sW + w * 4 * 16 * TCO;
                const double nb = q[16 * TCO + cl];
                if (nb <= 0.0) continue;
                const double d = (double)q[cl] - (double)s0, a = q[2 * 16 * TCO + cl], b = q[3 * 16 * TCO + cl];
                n += nb;
                S1 += a + nb * d;
                S2 += b + 2.0 * d * a + nb * d * d;
            }
            // record layout [block][channel][4]: one aligned 16-byte tuple per (block, channel)
            *reinterpret_cast<float4*>(stat_partial + ((int64_t)blockIdx.x * Co + co) * 4) =
                make_float4(s0, (float)n, (float)S1, (float)S2);
        }
    }
}

// Combine the per-block {shift, n, sum, sumsq} records into BatchNorm coefficients (Chan's parallel variance in
// float64), same outputs as bn_finalize_kernel.  rec [nrec][C][4]: a thread reads whole 16-byte tuples, 16 adjacent
// channels per record-lane (256 contiguous bytes), 64 record-lanes per workgroup; record-lanes fold by shuffles
// inside a wavefront and through LDS across the 16 wavefronts, always in the same order.
constexpr int FR_BLOCK = 1024, FR_CH = 16, FR_RL = FR_BLOCK / FR_CH;
// (device body: `slab` = which 16 channels; write != 0: this caller publishes coef / running statistics.  Threads < FR_CH of a valid
// channel return with ab = {a, b}; every other thread returns false.)
template <int FR_UN = 8>      // tuples of a thread in flight at once (nrec <= 1024: two round trips instead of four dependent ones; the
                               // order a thread visits its tuples in -- hence every sum -- does not depend on it)
__device__ __forceinline__ bool bn_finalize_records_body(const float* __restrict__ rec, int nrec, int64_t M, int C,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                         float* __restrict__ run_mean, float* __restrict__ run_var, float momentum,
                                                         float* __restrict__ coef, int slab, bool write, float (&ab)[2]) {
    __shared__ double s_red[FR_BLOCK / WAVE][2][FR_CH];
    const int cl = threadIdx.x & (FR_CH - 1), rl = threadIdx.x >> 4;
    const int c = slab * FR_CH + cl;
    const bool cv = c < C;
    const int cc = cv ? c : C - 1;
    // every record is re-based on the shift of record 0 (a sample value, so |shift - mean| ~ sigma: no cancellation
    // problem in float64)
    const double s0 = rec[(int64_t)cc * 4];
    double S1 = 0.0, S2 = 0.0;
    for (int r0 = rl; r0 < nrec; r0 += FR_UN * FR_RL) {
        float4 v[FR_UN];
#pragma unroll
        for (int u = 0; u < FR_UN; ++u) {
            const int r = r0 + u * FR_RL;
            v[u] = r < nrec ? *reinterpret_cast<const float4*>(rec + ((int64_t)r * C + cc) * 4)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < FR_UN; ++u) {
            const double nb = v[u].y;
            if (nb > 0.0) {
                const double d = (double)v[u].x - s0, a = v[u].z, b = v[u].w;
                S1 += a + nb * d;
                S2 += b + 2.0 * d * a + nb * d * d;
            }
        }
    }
    // lanes l, l ^ 16, l ^ 32, l ^ 48 of a wavefront hold the same channel
    S1 += __shfl_xor(S1, 16, WAVE); S2 += __shfl_xor(S2, 16, WAVE);
    S1 += __shfl_xor(S1, 32, WAVE); S2 += __shfl_xor(S2, 32, WAVE);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) < FR_CH) { s_red[wave][0][cl] = S1; s_red[wave][1][cl] = S2; }
    __syncthreads();
    if (threadIdx.x >= FR_CH || !cv) return false;
    S1 = 0.0; S2 = 0.0;
    for (int w = 0; w < FR_BLOCK / WAVE; ++w) { S1 += s_red[w][0][cl]; S2 += s_red[w][1][cl]; }
    const double m1 = S1 / (double)M;
    const double mean = s0 + m1;
    const double m2 = S2 - S1 * m1;
    double var = m2 / (double)M;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    const double a = (double)gamma[c] * rstd;
    ab[0] = (float)a;
    ab[1] = (float)((double)beta[c] - a * mean);
    if (!write) return true;
    coef[c] = ab[0];
    coef[C + c] = ab[1];
    coef[2 * C + c] = (float)mean;
    coef[3 * C + c] = (float)rstd;
    if (run_mean != nullptr) {
        const double unb = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
        run_mean[c] = (float)((1.0 - (double)momentum) * (double)run_mean[c] + (double)momentum * mean);
        run_var[c] = (float)((1.0 - (double)momentum) * (double)run_var[c] + (double)momentum * unb);
    }
    return true;
}
__global__ __launch_bounds__(FR_BLOCK) void bn_finalize_records_kernel(const float* __restrict__ rec, int nrec, int64_t M,
                                                                       int C, const float* __restrict__ gamma,
                                                                       const float* __restrict__ beta, float eps,
                                                                       float* __restrict__ run_mean, float* __restrict__ run_var,
                                                                       float momentum, float* __restrict__ coef) {
    float ab[2];
    bn_finalize_records_body(rec, nrec, M, C, gamma, beta, eps, run_mean, run_var, momentum, coef, blockIdx.x, true, ab);
}

// Coefficients AND apply in one launch (crfconv_bn_apply_from_records): a workgroup = one 16-channel slab x one row tile; it combines
// the records of ITS slab exactly as bn_finalize_records_kernel does (same threads, same order: identical coefficients; the
// workgroups of row tile 0 publish them and update the running statistics), then streams y = lrelu(a x + b) over its rows -- 64-byte
// row pieces, four lanes per row.  The 512-record combine is redundant per row tile (131 KB of L2 reads per workgroup) and buys the
// ~6 us coefficient launch that used to sit between every Linear and its BatchNorm apply pass.
template <bool ADD, int UN = 8>      // ADD: y = lrelu(a x + b + skip, slope) -- the ResNet join (crfconv_bn_apply_add's arithmetic)
__device__ __forceinline__ void bn_apply_records_body(const float* __restrict__ rec, int nrec, int64_t M, int C,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float eps, float* __restrict__ run_mean, float* __restrict__ run_var,
                                                      float momentum, float* __restrict__ coef, const float* __restrict__ x,
                                                      const float* __restrict__ skip, float slope, int rows_per_tile,
                                                      float* __restrict__ y, const int slab, const int tile) {
    __shared__ float s_ab[2][FR_CH];
    float ab[2];
    // grid = (row tiles, slabs): the workgroups that read the two / four 64-byte pieces of the same 128-byte lines are `tiles` apart in
    // dispatch order, and tiles is a multiple of 8 -- they land on the same XCD, whose L2 then fetches each line from HBM once
    const int q = threadIdx.x & 3, rl = threadIdx.x >> 2;            // 4 channel quads x 256 rows per pass
    const int c = slab * FR_CH + 4 * q;
    const bool cok = c < C;
    const int64_t r0 = (int64_t)tile * rows_per_tile;
    const int64_t r1 = r0 + rows_per_tile < M ? r0 + rows_per_tile : M;
    // the rows of the FIRST pass (the only one at <= 1024 rows per tile) are requested before the records are combined: their round
    // trip runs beside the combine's two instead of behind them
    float4 v[4];
    [[maybe_unused]] float4 k[ADD ? 4 : 1];
    auto request = [&](int64_t rb) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t r = rb + (int64_t)u * (FR_BLOCK / 4);
            const bool in = cok && r < r1;
            v[u] = in ? *reinterpret_cast<const float4*>(x + r * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (ADD) k[u] = in ? *reinterpret_cast<const float4*>(skip + r * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
#ifndef BN_APPLY_PREFETCH_
#define BN_APPLY_PREFETCH_ 1
#endif
    if (BN_APPLY_PREFETCH_) request(r0 + rl);
    if (bn_finalize_records_body<UN>(rec, nrec, M, C, gamma, beta, eps, run_mean, run_var, momentum, coef, slab, tile == 0, ab)) {
        s_ab[0][threadIdx.x] = ab[0];
        s_ab[1][threadIdx.x] = ab[1];
    }
    __syncthreads();
    if (!cok) return;
    const float4 a = *reinterpret_cast<const float4*>(&s_ab[0][4 * q]), b = *reinterpret_cast<const float4*>(&s_ab[1][4 * q]);
    for (int64_t rb = r0 + rl; rb < r1; rb += 4 * (FR_BLOCK / 4)) {
        if (!BN_APPLY_PREFETCH_ || rb != r0 + rl) request(rb);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t r = rb + (int64_t)u * (FR_BLOCK / 4);
            if (r >= r1) continue;
            float4 o = make_float4(fmaf(a.x, v[u].x, b.x), fmaf(a.y, v[u].y, b.y), fmaf(a.z, v[u].z, b.z), fmaf(a.w, v[u].w, b.w));
            if constexpr (ADD) {
                o.x = add_rn(o.x, k[u].x); o.y = add_rn(o.y, k[u].y); o.z = add_rn(o.z, k[u].z); o.w = add_rn(o.w, k[u].w);
            }
            o.x = o.x > 0.f ? o.x : slope * o.x;
            o.y = o.y > 0.f ? o.y : slope * o.y;
            o.z = o.z > 0.f ? o.z : slope * o.z;
            o.w = o.w > 0.f ? o.w : slope * o.w;
            *reinterpret_cast<float4*>(y + r * C + c) = o;
        }
    }
}

template <bool ADD>
__global__ __launch_bounds__(FR_BLOCK) void bn_apply_records_kernel(const float* __restrict__ rec, int nrec, int64_t M, int C,
                                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                    float eps, float* __restrict__ run_mean, float* __restrict__ run_var,
                                                                    float momentum, float* __restrict__ coef, const float* __restrict__ x,
                                                                    const float* __restrict__ skip, float slope, int rows_per_tile,
                                                                    float* __restrict__ y) {
    bn_apply_records_body<ADD>(rec, nrec, M, C, gamma, beta, eps, run_mean, run_var, momentum, coef, x, skip, slope, rows_per_tile, y,
                               blockIdx.y, blockIdx.x);
}
// The same for up to 4 independent layers in one launch (crfconv_bn_apply_from_records_jobs): the workgroups of the jobs laid end to end.
constexpr int BA_MAX = 4;
struct BnApplyJobs {
    const float* rec[BA_MAX]; const float* gamma[BA_MAX]; const float* beta[BA_MAX]; float* run_mean[BA_MAX]; float* run_var[BA_MAX];
    float* coef[BA_MAX]; const float* x[BA_MAX]; const float* skip[BA_MAX]; float* y[BA_MAX];
    long long M[BA_MAX];
    int nrec[BA_MAX], C[BA_MAX], rows_per_tile[BA_MAX], tiles[BA_MAX];
    float eps[BA_MAX], momentum[BA_MAX], slope[BA_MAX];
    int blk_base[BA_MAX + 1];
    int njobs;
};
__global__ __launch_bounds__(FR_BLOCK) void bn_apply_records_jobs_kernel(const BnApplyJobs t) {
    int j = 0;
    while (j + 1 < t.njobs && t.blk_base[j + 1] <= (int)blockIdx.x) ++j;
    const int local = (int)blockIdx.x - t.blk_base[j];
    const int tiles = uni(t.tiles[j]);
    const int tile = uni(local % tiles), slab = uni(local / tiles);
    const float* rec = uni(t.rec[j]); const float* gamma = uni(t.gamma[j]); const float* beta = uni(t.beta[j]);
    float* run_mean = uni(t.run_mean[j]); float* run_var = uni(t.run_var[j]); float* coef = uni(t.coef[j]);
    const float* x = uni(t.x[j]); const float* skip = uni(t.skip[j]); float* y = uni(t.y[j]);
    const long long M = uni(t.M[j]);
    const int nrec = uni(t.nrec[j]), C = uni(t.C[j]), rows = uni(t.rows_per_tile[j]);
    const float eps = uni(t.eps[j]), momentum = uni(t.momentum[j]), slope = uni(t.slope[j]);
    if (skip != nullptr) bn_apply_records_body<true, 4>(rec, nrec, M, C, gamma, beta, eps, run_mean, run_var, momentum, coef, x, skip, slope, rows, y, slab, tile);
    else bn_apply_records_body<false, 4>(rec, nrec, M, C, gamma, beta, eps, run_mean, run_var, momentum, coef, x, nullptr, slope, rows, y, slab, tile);
}

static int lf_blocks(int64_t M) {
    constexpr int cap = 512;   // swept 128..2048 on the training step: 512 (two blocks per CU, half the statistic records of 1024) is the optimum
    int64_t nb = (M + 16 * LF_WAVES - 1) / (16 * LF_WAVES);           // 16 rows per wave and iteration; two blocks per CU keep
    if (nb > cap) nb = cap;               // enough 16-byte loads in flight; one statistics record per block
    return (int)(nb < 1 ? 1 : nb);
}

}  // namespace crf

// Supported when a 16-channel weight slab fits LDS: 16 x (Ci rounded to 16 + 4) floats (+ prologue rows) <= 64 KB, i.e. Ci <= ~1000.
// Output tiles per workgroup and the dynamic LDS of linear_fwd_kernel for k = Ci inputs, Co outputs: weight slab
// [16 tco][Ci rounded to 16, + 4], the five prologue coefficient rows (dX form), the four output staging tiles (tco >= 2).
// 64 output channels per workgroup at most: the 128-channel form (tco = 8) needs 167 + 98 registers with the statistic
// accumulators, i.e. ONE wavefront per SIMD (measured 68 -> 49 us for 163840 x 32 -> 128; 5.85 -> 5.80 ms per step).
static size_t lf_lds_bytes_at(int Ci, int tco, bool pro) {
    const size_t cip = (size_t)((Ci + 15) / 16) * 16 + 4, cik = cip - 4;
    size_t floats = 16 * (size_t)tco * cip;
    const size_t stats = (size_t)crf::LF_WAVES * 4 * 16 * (size_t)tco;   // the statistics epilogue reuses the slab as [waves][4][16 tco]
    if (floats < stats) floats = stats;
    if (pro) floats += 5 * cik;
    if (tco >= 2) floats += (size_t)crf::LF_WAVES * 16 * (16 * (size_t)tco + 4);
    return sizeof(float) * floats;
}
// Output tiles per workgroup: by Co, then halved until the slab of k = Ci inputs fits 64 KB (256 inputs: 32 channels per
// workgroup, 512: 16 -- the operand rows are then read once per column slab, from L2).
static int lf_tco(int Ci, int Co, bool pro) {
    constexpr int max_tco = 4;
    const int tiles = (Co + 15) / 16;
    int tco = tiles >= 8 ? 8 : (tiles >= 4 ? 4 : (tiles >= 2 ? 2 : 1));
    if (tco > max_tco) tco = max_tco;
    while (tco > 1 && lf_lds_bytes_at(Ci, tco, pro) > 64 * 1024) tco >>= 1;
    return tco;
}
static size_t lf_lds_bytes(int Ci, int Co, bool pro) { return lf_lds_bytes_at(Ci, lf_tco(Ci, Co, pro), pro); }
// k chunks of the hoisted operand loop (linear_fwd_kernel<.., NCH>): Ci <= 128 in 16-byte pieces; else 0 = the rolled loop
static int lf_hoist_chunks(int Ci, bool vec4) {
    if (!vec4 || Ci > 128) return 0;
    const int n = (Ci + 15) / 16;
    return n <= 1 ? 1 : (n <= 2 ? 2 : (n <= 4 ? 4 : 8));
}

extern "C" int crfconv_linear_forward_supported(int Ci, int Co) {
    if (Ci < 1 || Co < 1) return 0;
    return lf_lds_bytes(Ci, Co, false) <= 64 * 1024 ? 1 : 0;
}

extern "C" size_t crfconv_linear_forward_stat_records(int64_t M) { return (size_t)crf::lf_blocks(M); }

// Y [M, Co] = X [M, Ci] W^T (+ bias);  W is [Co, Ci] row-major, or [Ci, Co] when transpose_w != 0 (the dX product).
// stat_rec (may be NULL): float [records][Co][4] receives per-workgroup {shift, n, sum(y - shift), sum (y - shift)^2}.
struct LinearDropout {
    const long long* counter = nullptr;
    unsigned long long seed = 0ull;
    unsigned threshold = 0u;
    float scale = 1.f;
};

static int linear_forward_impl(const float* X, const float* Xb, int xsplit, const float* W, const float* bias, int64_t M,
                               int Ci, int Co, int transpose_w, float* Y, float* stat_rec, crf_stream_t stream,
                               LinearDropout drop = LinearDropout()) {
    CRF_REQUIRE(X && W && Y, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(M > 0, CRF_ERR_ARG, "M must be positive");
    CRF_REQUIRE(crfconv_linear_forward_supported(Ci, Co), CRF_ERR_UNSUPPORTED, "weight slab %dx%d does not fit LDS", Co, Ci);
    CRF_REQUIRE(Xb == nullptr || (xsplit > 0 && xsplit < Ci && xsplit % 4 == 0 && Ci % 4 == 0), CRF_ERR_ARG,
                "two-operand form needs 0 < split < Ci, both multiples of 4 (split=%d Ci=%d)", xsplit, Ci);
    const int tiles = (Co + 15) / 16;
    const int tco = lf_tco(Ci, Co, false);
    const int gy = (tiles + tco - 1) / tco;
    const dim3 grid((unsigned)crf::lf_blocks(M), (unsigned)gy), blk(crf::LF_BLOCK);
    const size_t lds = lf_lds_bytes(Ci, Co, false);
    hipStream_t st = crf::as_stream(stream);
    const bool vec4 = (Ci % 4) == 0 && (Co % 4) == 0;
    const int nch = lf_hoist_chunks(Ci, vec4);       // 1 / 2 / 4 / 8 chunks: the hoisted loop; 0: the rolled one
#define LF4(T, V, E, N) hipLaunchKernelGGL((crf::linear_fwd_kernel<T, false, V, E, N>), grid, blk, lds, st, X, W, bias, M, Ci, Co, transpose_w, Y, stat_rec, (const float*)nullptr, (const float*)nullptr, 1.f, Xb, xsplit, (float*)nullptr, 0, (const float*)nullptr, drop.counter, drop.seed, drop.threshold, drop.scale)
#define LF3(T, V, E) do { if (V && nch == 1) LF4(T, V, E, 1); else if (V && nch == 2) LF4(T, V, E, 2); else if (V && nch == 4) LF4(T, V, E, 4); else if (V && nch == 8) LF4(T, V, E, 8); else LF4(T, V, E, 0); } while (0)
#define LF2(T, V) do { if (drop.counter != nullptr) LF4(T, V, crf::EPI_DROPOUT, 0); else LF3(T, V, crf::EPI_NONE); } while (0)
#define LF(T) do { if (vec4) LF2(T, true); else if (drop.counter != nullptr) LF4(T, false, crf::EPI_DROPOUT, 0); else LF4(T, false, crf::EPI_NONE, 0); } while (0)
    switch (tco) {
        case 1: LF(1); break;
        case 2: LF(2); break;
        case 4: LF(4); break;
        default: LF(8); break;
    }
#undef LF
#undef LF2
#undef LF3
#undef LF4
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

// Y [M, Co] = X [M, Ci] W^T (+ bias);  W is [Co, Ci] row-major, or [Ci, Co] when transpose_w != 0 (the dX product).
// stat_rec (may be NULL): float [records][Co][4] receives per-workgroup {shift, n, sum(y - shift), sum (y - shift)^2}.
extern "C" int crfconv_linear_forward(const float* X, const float* W, const float* bias, int64_t M, int Ci, int Co,
                                      int transpose_w, float* Y, float* stat_rec, crf_stream_t stream) {
    return linear_forward_impl(X, nullptr, 0, W, bias, M, Ci, Co, transpose_w, Y, stat_rec, stream);
}

// Y = dropout_mask .* (X W^T or X W) / (1 - p): the input gradient of a Linear that sits BEHIND an nn.Dropout, masked while it
// is written (the mask of crfconv_bn_apply_dropout with the same p, seed and *counter).
extern "C" int crfconv_linear_forward_dropout(const float* X, const float* W, int64_t M, int Ci, int Co, int transpose_w,
                                              float p, uint64_t seed, const int64_t* counter, float* Y, crf_stream_t stream) {
    CRF_REQUIRE(counter, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(p >= 0.f && p < 1.f, CRF_ERR_ARG, "dropout probability %g outside [0, 1)", (double)p);
    LinearDropout d;
    d.counter = reinterpret_cast<const long long*>(counter);
    d.seed = (unsigned long long)seed;
    d.threshold = crf::dropout_threshold(p);
    d.scale = 1.f / (1.f - p);
    return linear_forward_impl(X, nullptr, 0, W, nullptr, M, Ci, Co, transpose_w, Y, nullptr, stream, d);
}

// The same on the column concatenation [Xa | Xb] (Xa [M, split], Xb [M, Ci - split]) without materialising it.
extern "C" int crfconv_linear_forward_cat(const float* Xa, const float* Xb, int split, const float* W, const float* bias,
                                          int64_t M, int Ci, int Co, float* Y, float* stat_rec, crf_stream_t stream) {
    CRF_REQUIRE(Xb, CRF_ERR_ARG, "null pointer");
    return linear_forward_impl(Xa, Xb, split, W, bias, M, Ci, Co, 0, Y, stat_rec, stream);
}

// BatchNorm coefficients from the records written by crfconv_linear_forward (instead of a statistics pass).
extern "C" int crfconv_bn_coef_from_records(const float* stat_rec, int64_t M, int C, const float* gamma,
                                            const float* beta, float* run_mean, float* run_var, float momentum,
                                            float eps, float* coef, crf_stream_t stream) {
    CRF_REQUIRE(stat_rec && gamma && beta && coef, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(M > 0 && C > 0, CRF_ERR_ARG, "bad shape");
    return crfconv_bn_coef_from_nrecords(stat_rec, (int64_t)crfconv_linear_forward_stat_records(M), M, C, gamma, beta, run_mean, run_var,
                                         momentum, eps, coef, stream);
}

// The same with an explicit record count (records of crfconv_gemm_stats: one per 16-row group).
extern "C" int crfconv_bn_coef_from_nrecords(const float* stat_rec, int64_t nrec, int64_t M, int C, const float* gamma,
                                             const float* beta, float* run_mean, float* run_var, float momentum,
                                             float eps, float* coef, crf_stream_t stream) {
    CRF_REQUIRE(stat_rec && gamma && beta && coef, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(M > 0 && C > 0 && nrec > 0 && nrec < ((int64_t)1 << 31), CRF_ERR_ARG, "bad shape");
    hipLaunchKernelGGL(crf::bn_finalize_records_kernel, dim3((C + crf::FR_CH - 1) / crf::FR_CH), dim3(crf::FR_BLOCK), 0, crf::as_stream(stream), stat_rec,
                       (int)nrec, M, C, gamma, beta, eps, run_mean, run_var, momentum, coef);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

// crfconv_bn_coef_from_records followed by crfconv_bn_apply (skip == NULL) or crfconv_bn_apply_add (the ResNet join) in ONE launch:
// identical coef / running statistics / y.  C % 4 == 0.
// row tiles of one apply launch: ~`wgs` workgroups in all -- one per CU: 116-128 registers x 1024 threads is one workgroup per CU, and
// every workgroup pays the records' combine (step, one box: 128 workgroups 4.163 ms, 192 4.136, 256 4.136, 320 4.159, 512 4.153); a
// row tile is a multiple of the 1024 rows one pass covers
#ifndef BN_APPLY_WGS_
#define BN_APPLY_WGS_ 256
#endif
static void bn_apply_plan(int64_t M, int C, int wgs, int64_t& tiles, int64_t& rows) {
    const int slabs = (C + crf::FR_CH - 1) / crf::FR_CH;
    tiles = wgs / slabs;
    if (tiles < 1) tiles = 1;
    rows = (M + tiles - 1) / tiles;
    rows = (rows + 1023) / 1024 * 1024;
    tiles = (M + rows - 1) / rows;
    if (tiles > 8) tiles = (tiles + 7) / 8 * 8;          // (tiles past the end of the rows have nothing to apply; see the kernel for the 8)
}

extern "C" int crfconv_bn_apply_from_records(const float* stat_rec, int64_t nrec, const float* x, int64_t M, int C, const float* gamma,
                                             const float* beta, float* run_mean, float* run_var, float momentum, float eps,
                                             const float* skip, float slope, float* coef, float* y, crf_stream_t stream) {
    CRF_REQUIRE(stat_rec && x && gamma && beta && coef && y, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(M > 0 && C >= 4 && C % 4 == 0 && nrec > 0 && nrec < ((int64_t)1 << 31), CRF_ERR_ARG, "bad shape");
    const int slabs = (C + crf::FR_CH - 1) / crf::FR_CH;
    int64_t tiles, rows;
    bn_apply_plan(M, C, BN_APPLY_WGS_, tiles, rows);
    const dim3 grid((unsigned)tiles, (unsigned)slabs), blk(crf::FR_BLOCK);
    if (skip != nullptr)
        hipLaunchKernelGGL(crf::bn_apply_records_kernel<true>, grid, blk, 0, crf::as_stream(stream), stat_rec, (int)nrec, M, C, gamma, beta, eps,
                           run_mean, run_var, momentum, coef, x, skip, slope, (int)rows, y);
    else
        hipLaunchKernelGGL(crf::bn_apply_records_kernel<false>, grid, blk, 0, crf::as_stream(stream), stat_rec, (int)nrec, M, C, gamma, beta, eps,
                           run_mean, run_var, momentum, coef, x, skip, slope, (int)rows, y);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_bn_apply_from_records_jobs(const crf_bn_apply_job* jobs, int njobs, crf_stream_t stream) {
    CRF_REQUIRE(jobs && njobs >= 1 && njobs <= crf::BA_MAX, CRF_ERR_ARG, "1 .. %d jobs (got %d)", crf::BA_MAX, njobs);
    crf::BnApplyJobs t;
    int64_t blocks = 0;
    for (int j = 0; j <= crf::BA_MAX; ++j) {
        t.blk_base[j] = (int)blocks;
        if (j >= crf::BA_MAX) break;
        if (j >= njobs) {
            t.rec[j] = nullptr; t.gamma[j] = nullptr; t.beta[j] = nullptr; t.run_mean[j] = nullptr; t.run_var[j] = nullptr; t.coef[j] = nullptr;
            t.x[j] = nullptr; t.skip[j] = nullptr; t.y[j] = nullptr; t.M[j] = 0; t.nrec[j] = 0; t.C[j] = 4; t.rows_per_tile[j] = 1024; t.tiles[j] = 1;
            t.eps[j] = 0.f; t.momentum[j] = 0.f; t.slope[j] = 1.f;
            continue;
        }
        const crf_bn_apply_job& b = jobs[j];
        CRF_REQUIRE(b.stat_rec && b.x && b.gamma && b.beta && b.coef && b.y, CRF_ERR_ARG, "job %d: null pointer", j);
        CRF_REQUIRE(b.M > 0 && b.C >= 4 && b.C % 4 == 0 && b.nrec > 0 && b.nrec < ((int64_t)1 << 31), CRF_ERR_ARG, "job %d: bad shape", j);
        int64_t tiles, rows;
        bn_apply_plan(b.M, b.C, BN_APPLY_WGS_ / njobs, tiles, rows);
        t.rec[j] = b.stat_rec; t.gamma[j] = b.gamma; t.beta[j] = b.beta; t.run_mean[j] = b.run_mean; t.run_var[j] = b.run_var; t.coef[j] = b.coef;
        t.x[j] = b.x; t.skip[j] = b.skip; t.y[j] = b.y; t.M[j] = (long long)b.M; t.nrec[j] = (int)b.nrec; t.C[j] = b.C; t.rows_per_tile[j] = (int)rows;
        t.tiles[j] = (int)tiles; t.eps[j] = b.eps; t.momentum[j] = b.momentum; t.slope[j] = b.slope;
        blocks += tiles * ((b.C + crf::FR_CH - 1) / crf::FR_CH);
        CRF_REQUIRE(blocks < ((int64_t)1 << 31), CRF_ERR_UNSUPPORTED, "too many workgroups in one batch");
    }
    t.njobs = njobs;
    hipLaunchKernelGGL(crf::bn_apply_records_jobs_kernel, dim3((unsigned)blocks), dim3(crf::FR_BLOCK), 0, crf::as_stream(stream), t);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

// ---------------------------------------------------------------------- fused backward of Linear -> BatchNorm -> LeakyReLU
namespace crf {
struct MlpPlan {
    int tco, tci, gy, gz, nblk, rows_per_block;
};
static MlpPlan mlp_plan(int64_t M, int Co, int Ci) {
    MlpPlan p;
    const int t_co = (Co + 15) / 16, t_ci = (Ci + 15) / 16;
    p.tco = t_co >= 4 ? 4 : (t_co >= 2 ? 2 : 1);
    p.tci = t_ci >= 4 ? 4 : (t_ci >= 2 ? 2 : 1);
    if (p.tco * p.tci == 16) p.tci = 2;              // two accumulator sets: at most 8 tiles (64 registers) each
    p.gy = (t_co + p.tco - 1) / p.tco;
    p.gz = (t_ci + p.tci - 1) / p.tci;
#ifndef MLP_P1_TARGET_
#define MLP_P1_TARGET_ 512
#endif
    constexpr int target = MLP_P1_TARGET_;           // slices x column slabs per launch (256 / 1024 measured slower: DESIGN 9 C4)
    int64_t slices = target / ((int64_t)p.gy * p.gz);
    if (slices < 32) slices = 32;
    int64_t rows = (M + slices - 1) / slices;
    if (rows < 64) rows = 64;
    rows = (rows + 63) / 64 * 64;
    p.rows_per_block = (int)rows;
    p.nblk = (int)((M + rows - 1) / rows);
    return p;
}
static size_t mlp_ws_layout(int64_t M, int Co, int Ci, size_t off[5]) {
    const MlpPlan p = mlp_plan(M, Co, Ci);
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t o = 0;
    off[0] = o; o = up(o + sizeof(float) * (size_t)p.nblk * Co * Ci);      // PA
    off[1] = o; o = up(o + sizeof(float) * (size_t)p.nblk * Co * Ci);      // PB
    off[2] = o; o = up(o + sizeof(float) * (size_t)p.nblk * 2 * Co);       // PG
    off[3] = o; o = up(o + sizeof(float) * (size_t)p.nblk * Ci);           // PX
    off[4] = o; o = up(o + sizeof(float) * 5 * (size_t)Co);                // prologue coefficients
    return o;
}
}  // namespace crf

extern "C" size_t crfconv_ticket_bytes(void) { return sizeof(unsigned) * crf::LW_TICKET_WORDS; }

extern "C" int crfconv_mlp_backward_supported(int64_t M, int Ci, int Co) {
    if (!(M > 0 && Co % 4 == 0 && Ci >= 1 && Co >= 4 && Co <= 1024 && Ci <= 1024)) return 0;
    // dX runs on linear_fwd_kernel<., true> with k = Co inputs and Ci outputs: its LDS must fit 64 KB
    return lf_lds_bytes(Co, Ci, true) <= 64 * 1024 ? 1 : 0;
}

extern "C" size_t crfconv_mlp_backward_workspace(int64_t M, int Ci, int Co) {
    if (M <= 0 || Co <= 0 || Ci <= 0) return 0;
    size_t off[5];
    return crf::mlp_ws_layout(M, Co, Ci, off) + 256;
}

static int mlp_backward_impl(const float* gA, const float* Y, const float* X, const float* Xb, int xsplit, const float* W,
                             const float* coef, float slope, int64_t M, int Ci, int Co, float* dX, float* dXb, float* dW,
                             float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, unsigned* ticket,
                             crf_stream_t stream, const float* dX_add = nullptr);

extern "C" int crfconv_mlp_backward(const float* gA, const float* Y, const float* X, const float* W, const float* coef,
                                    float slope, int64_t M, int Ci, int Co, float* dX, float* dW, float* dgamma,
                                    float* dbeta, void* workspace, size_t workspace_bytes, unsigned* ticket,
                                    crf_stream_t stream) {
    return mlp_backward_impl(gA, Y, X, nullptr, 0, W, coef, slope, M, Ci, Co, dX, nullptr, dW, dgamma, dbeta, workspace,
                             workspace_bytes, ticket, stream);
}

// dX = (the block's input gradient) + dX_add [M, Ci]: the block's input has a second consumer (the shortcut of a ResNet block)
// whose gradient is already known -- the sum autograd would form in a pass of its own is made while dX is written.
extern "C" int crfconv_mlp_backward_add(const float* gA, const float* Y, const float* X, const float* W, const float* coef,
                                        float slope, int64_t M, int Ci, int Co, const float* dX_add, float* dX, float* dW,
                                        float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                                        unsigned* ticket, crf_stream_t stream) {
    CRF_REQUIRE(dX != nullptr || dX_add == nullptr, CRF_ERR_ARG, "dX_add without dX");
    return mlp_backward_impl(gA, Y, X, nullptr, 0, W, coef, slope, M, Ci, Co, dX, nullptr, dW, dgamma, dbeta, workspace,
                             workspace_bytes, ticket, stream, dX_add);
}

// The block's input was the column concatenation [Xa | Xb]: dXa [M, split], dXb [M, Ci - split] (both or neither NULL).
extern "C" int crfconv_mlp_backward_cat(const float* gA, const float* Y, const float* Xa, const float* Xb, int split,
                                        const float* W, const float* coef, float slope, int64_t M, int Ci, int Co,
                                        float* dXa, float* dXb, float* dW, float* dgamma, float* dbeta, void* workspace,
                                        size_t workspace_bytes, unsigned* ticket, crf_stream_t stream) {
    CRF_REQUIRE(Xb && split > 0 && split < Ci && split % 4 == 0 && Ci % 4 == 0 && ((dXa == nullptr) == (dXb == nullptr)),
                CRF_ERR_ARG, "two-operand form: split=%d Ci=%d must be multiples of 4, dXa / dXb both or neither", split, Ci);
    return mlp_backward_impl(gA, Y, Xa, Xb, split, W, coef, slope, M, Ci, Co, dXa, dXb, dW, dgamma, dbeta, workspace,
                             workspace_bytes, ticket, stream);
}

// dW of any number of MLP blocks whose crfconv_mlp_backward(_add / _cat) call was given dW = NULL, from the workspaces those
// calls left behind (untouched since), in ONE launch.
static int mlp_dw_jobs_impl(const crf_mlp_dw_job* jobs, int njobs, const crf::CrfMatJobs* side, int nside, crf_stream_t stream) {
    CRF_REQUIRE(jobs || njobs == 0, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(njobs >= 0, CRF_ERR_ARG, "njobs=%d < 0", njobs);
    hipStream_t st = crf::as_stream(stream);
    for (int j0 = 0; j0 < njobs; j0 += crf::MDW_MAX) {
        crf::MlpDwTable t;
        const int n = njobs - j0 < crf::MDW_MAX ? njobs - j0 : crf::MDW_MAX;
        int64_t total = 0;
        for (int j = 0; j < crf::MDW_MAX; ++j) {
            const crf_mlp_dw_job& jb = jobs[j0 + (j < n ? j : 0)];
            if (j < n) {
                CRF_REQUIRE(jb.workspace && jb.coef && jb.dW, CRF_ERR_ARG, "job %d: null pointer", j0 + j);
                CRF_REQUIRE(crfconv_mlp_backward_supported(jb.M, jb.Ci, jb.Co) == 1, CRF_ERR_UNSUPPORTED,
                            "job %d: shape M=%lld Ci=%d Co=%d not supported", j0 + j, (long long)jb.M, jb.Ci, jb.Co);
            }
            const char* base = reinterpret_cast<const char*>((reinterpret_cast<uintptr_t>(jb.workspace) + 255) & ~(uintptr_t)255);
            size_t off[5];
            crf::mlp_ws_layout(jb.M, jb.Co, jb.Ci, off);
            t.PA[j] = reinterpret_cast<const float*>(base + off[0]);
            t.PB[j] = reinterpret_cast<const float*>(base + off[1]);
            t.PG[j] = reinterpret_cast<const float*>(base + off[2]);
            t.PX[j] = reinterpret_cast<const float*>(base + off[3]);
            t.coef[j] = jb.coef;
            t.dW[j] = jb.dW;
            t.M[j] = (long long)jb.M;
            t.nblk[j] = crf::mlp_plan(jb.M, jb.Co, jb.Ci).nblk;
            t.Co[j] = jb.Co;
            t.Ci[j] = jb.Ci;
            t.group_begin[j] = (int)total;
            if (j < n) total += crf::cdiv((int64_t)jb.Co * jb.Ci, 64);
            CRF_REQUIRE(total < ((int64_t)1 << 30), CRF_ERR_ARG, "too many slots in one batch");
        }
        t.group_begin[crf::MDW_MAX] = (int)total;
        t.njobs = n;
        if (side != nullptr && j0 == 0)
            hipLaunchKernelGGL(crf::mlp_dw_jobs_hosting_kernel, dim3((unsigned)(total + nside)), dim3(crf::MF_BLOCK), 0, st, t, *side, nside);
        else
            hipLaunchKernelGGL(crf::mlp_dw_jobs_kernel, dim3((unsigned)total), dim3(crf::MF_BLOCK), 0, st, t);
        CRF_LAUNCH_CHECK();
    }
    return CRF_OK;
}
extern "C" int crfconv_mlp_dw_jobs(const crf_mlp_dw_job* jobs, int njobs, crf_stream_t stream) {
    return mlp_dw_jobs_impl(jobs, njobs, nullptr, 0, stream);
}
// crfconv_mlp_dw_jobs whose (first) launch also CARRIES crfconv_crf_matrices_backward_batched(c, Q, gQ, gP, H, n, dc) as its first
// workgroups: results of both are those of the two separate calls.
extern "C" int crfconv_mlp_dw_jobs_hosting(const crf_mlp_dw_job* jobs, int njobs, const float* const* c, const float* const* Q,
                                           const float* const* gQ, const float* const* gP, const int* H, int n, float* const* dc,
                                           crf_stream_t stream) {
    CRF_REQUIRE(njobs >= 1, CRF_ERR_ARG, "a hosting launch needs at least one job of its own (got %d)", njobs);
    crf::CrfMatJobs j;
    int nslab = 0;
    if (int rc = crf_matrices_bwd_jobs(c, Q, gQ, gP, H, n, dc, j, nslab)) return rc;
    return mlp_dw_jobs_impl(jobs, njobs, &j, nslab, stream);
}

static int mlp_backward_impl(const float* gA, const float* Y, const float* X, const float* Xb, int xsplit, const float* W,
                             const float* coef, float slope, int64_t M, int Ci, int Co, float* dX, float* dXb, float* dW,
                             float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, unsigned* ticket,
                             crf_stream_t stream, const float* dX_add) {
    CRF_REQUIRE(gA && Y && X && W && coef && dgamma && dbeta && workspace, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(dX_add == nullptr || dXb == nullptr, CRF_ERR_ARG, "dX_add is for the one-operand form");
    CRF_REQUIRE(crfconv_mlp_backward_supported(M, Ci, Co) == 1, CRF_ERR_UNSUPPORTED, "shape M=%lld Ci=%d Co=%d not supported",
                (long long)M, Ci, Co);
    CRF_REQUIRE(workspace_bytes >= crfconv_mlp_backward_workspace(M, Ci, Co), CRF_ERR_WORKSPACE, "workspace too small");
    hipStream_t st = crf::as_stream(stream);
    char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    size_t off[5];
    crf::mlp_ws_layout(M, Co, Ci, off);
    float* PA = reinterpret_cast<float*>(base + off[0]);
    float* PB = reinterpret_cast<float*>(base + off[1]);
    float* PG = reinterpret_cast<float*>(base + off[2]);
    float* PX = reinterpret_cast<float*>(base + off[3]);
    float* pro = reinterpret_cast<float*>(base + off[4]);
    const crf::MlpPlan p = crf::mlp_plan(M, Co, Ci);
    // the last workgroup of pass 1 does the channel part when its 2 Co sums fit one thread each in groups of whole quads
    if (!(2 * Co <= crf::WG_BLOCK && crf::WG_BLOCK % (Co / 2) == 0 && (int64_t)p.nblk * 2 * Co * 4 < ((int64_t)1 << 31))) ticket = nullptr;
    {
        const dim3 grid((unsigned)p.nblk, (unsigned)p.gy, (unsigned)p.gz), blk(crf::WG_BLOCK);
#define P1(TA, TB) hipLaunchKernelGGL((crf::mlp_bwd_p1_kernel<TA, TB>), grid, blk, 0, st, gA, Y, X, Xb, xsplit, coef, slope, M, Co, Ci, p.rows_per_block, PA, PB, PG, PX, ticket, dgamma, dbeta, pro)
        switch (p.tco * 10 + p.tci) {
            case 11: P1(1, 1); break;
            case 12: P1(1, 2); break;
            case 14: P1(1, 4); break;
            case 21: P1(2, 1); break;
            case 22: P1(2, 2); break;
            case 24: P1(2, 4); break;
            case 41: P1(4, 1); break;
            default: P1(4, 2); break;
        }
#undef P1
        CRF_LAUNCH_CHECK();
    }
    // dW == NULL: the channel part only; the caller finishes dW later from the workspace (crfconv_mlp_dw_jobs)
    const int nw = dW != nullptr ? (int)crf::cdiv((int64_t)Co * Ci, 64) : 0;
    const int nc = ticket != nullptr ? 0 : (Co + crf::MF_WAVES - 1) / crf::MF_WAVES;      // channel workgroups (none: done inside pass 1)
    if (nw + nc > 0) {
        hipLaunchKernelGGL(crf::mlp_bwd_finalize_kernel, dim3((unsigned)(nw + nc)), dim3(crf::MF_BLOCK), 0, st, PA, PB, PG, PX, p.nblk, coef, M, Co,
                           Ci, nw, dW, dgamma, dbeta, pro);
        CRF_LAUNCH_CHECK();
    }
    if (dX != nullptr) {
        // dX [M, Ci] = gY [M, Co] W [Co, Ci]: the forward kernel with k = Co, outputs = Ci, W read transposed
        const int gCi = Co, gCo = Ci;
        const int tiles = (gCo + 15) / 16;
        const int tco = lf_tco(gCi, gCo, true);
        const int gy = (tiles + tco - 1) / tco;
        const dim3 grid((unsigned)crf::lf_blocks(M), (unsigned)gy), blk(crf::LF_BLOCK);
        const size_t lds = lf_lds_bytes(gCi, gCo, true);
        const bool vec4 = (gCi % 4) == 0 && (gCo % 4) == 0;
        const int nch = lf_hoist_chunks(gCi, vec4);
#define DX4(T, V, E, N) hipLaunchKernelGGL((crf::linear_fwd_kernel<T, true, V, E, N>), grid, blk, lds, st, gA, W, (const float*)nullptr, M, gCi, gCo, 1, dX, (float*)nullptr, Y, pro, slope, (const float*)nullptr, 0, dXb, xsplit, dX_add)
#define DX3(T, V, E) do { if (V && nch == 1) DX4(T, V, E, 1); else if (V && nch == 2) DX4(T, V, E, 2); else if (V && nch == 4) DX4(T, V, E, 4); else if (V && nch == 8) DX4(T, V, E, 8); else DX4(T, V, E, 0); } while (0)
#define DX2(T, V) do { if (dX_add != nullptr) DX3(T, V, crf::EPI_ADD); else DX3(T, V, crf::EPI_NONE); } while (0)
#define DX(T) do { if (vec4) DX2(T, true); else if (dX_add != nullptr) DX4(T, false, crf::EPI_ADD, 0); else DX4(T, false, crf::EPI_NONE, 0); } while (0)
        switch (tco) {
            case 1: DX(1); break;
            case 2: DX(2); break;
            case 4: DX(4); break;
            default: DX(8); break;
        }
#undef DX
#undef DX2
#undef DX3
#undef DX4
        CRF_LAUNCH_CHECK();
    }
    return CRF_OK;
}
