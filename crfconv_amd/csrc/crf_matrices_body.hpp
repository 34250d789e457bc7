// The forward of the CRF layers' two loop-invariant matrices as a device function over caller-provided LDS, so that its workgroups can
// run as a launch of their own (linear.hip: crf_matrices_kernel / crf_matrices_batched_kernel) or ride along in a long launch whose
// inputs do not depend on them (pointconv.hip: the first PointConv's statistics pass, crfconv_pointconv_uvstats_hosting).
#pragma once
#include "common.hpp"

namespace crf {

// The two loop-invariant matrices of a CRF layer from its compatibility factor c [H, H] in one launch:
//   Q = (I + c^T c)^-1,   P = c^T c Q = I - Q            (continuous_crf_conv_big.py:67-72)
// The 64 pivots of the Gauss-Jordan sweep are sequential, and what a pivot costs is the INSTRUCTIONS every wavefront issues
// around its handful of float64 multiply-adds (barrier, LDS traffic, the reciprocal, the selects that treat pivot row /
// column / element apart) -- round 3's 1024-thread form (2 x 2 tiles, sixteen wavefronts on one CU) took 35 us for the four
// layers of PointConvBig.  This form: 256 threads, a 4 x 4 cyclic tile each (one wavefront per SIMD: the per-pivot overhead is
// paid once per SIMD, not four times), the pivot row and column in LDS in TILE-MAJOR order (a thread's four values are one
// 32-byte segment), 1 / pivot by the hardware estimate + two Newton steps, and NO special cases in the update: with
//   col'[p] = pivot - 1  (threads that own the pivot row)      row'[p] = 1 + 1 / pivot  (threads that own the pivot column)
// the one fused multiply-add  t -= col' * row'  leaves row * (1 / pivot) in the pivot row, -col / pivot in the pivot column and
// 1 / pivot in the pivot itself (pivot * (1 / pivot) = 1 up to float64 rounding).
constexpr int CMF_BLOCK = 256;
constexpr int CMF_LDS_BYTES = 4 * 64 * (int)sizeof(double) + 64 * 65 * (int)sizeof(float);      // pivot row / column (two parities each) | c, stride 65
__device__ __forceinline__ void gauss_jordan_fma(double (&t)[4][4], int H, double (*s_row)[64], double (*s_col)[64]) {
    const int tr = threadIdx.x >> 4, tc = threadIdx.x & 15;
#pragma unroll
    for (int ip = 0; ip < 4; ++ip) {                     // pivot p = 16 ip + pp lives in local row / column ip
        for (int pp = 0; pp < 16; ++pp) {
            const int p = 16 * ip + pp;
            if (p >= H) break;                           // uniform: rows beyond H are identity already
            const int b = p & 1;
            if (tr == pp) {
#pragma unroll
                for (int j = 0; j < 4; ++j) s_row[b][4 * tc + j] = t[ip][j];       // column tc + 16 j of row p
            }
            if (tc == pp) {
#pragma unroll
                for (int i = 0; i < 4; ++i) s_col[b][4 * tr + i] = t[i][ip];       // row tr + 16 i of column p
            }
            __syncthreads();
            const double pv = s_row[b][4 * pp + ip];                              // element (p, p)
            double piv = __builtin_amdgcn_rcp(pv);
            piv = fma(piv, fma(-pv, piv, 1.0), piv);
            piv = fma(piv, fma(-pv, piv, 1.0), piv);
            double rowv[4], colv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) rowv[j] = s_row[b][4 * tc + j] * piv;
#pragma unroll
            for (int i = 0; i < 4; ++i) colv[i] = s_col[b][4 * tr + i];
            if (tc == pp) rowv[ip] = 1.0 + piv;
            if (tr == pp) colv[ip] = pv - 1.0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) t[i][j] = fma(-colv[i], rowv[j], t[i][j]);
        }
    }
}
__device__ __forceinline__ void crf_matrices_body(const float* __restrict__ cmat, int H,
                                                  float* __restrict__ Qout, float* __restrict__ Pout, char* __restrict__ lds /*CMF_LDS_BYTES, 16-byte aligned*/) {
    double (*s_row)[64] = reinterpret_cast<double (*)[64]>(lds);
    double (*s_col)[64] = reinterpret_cast<double (*)[64]>(lds + 2 * 64 * sizeof(double));
    float* s_c = reinterpret_cast<float*>(lds + 4 * 64 * sizeof(double));
    for (int e = threadIdx.x; e < 64 * 65; e += CMF_BLOCK) s_c[e] = 0.f;           // columns >= H of c: zero (M stays identity there)
    __syncthreads();
    for (int e = threadIdx.x; e < H * H; e += CMF_BLOCK) s_c[(e / H) * 65 + (e % H)] = cmat[e];
    __syncthreads();
    const int tr = threadIdx.x >> 4, tc = threadIdx.x & 15;
    double t[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) t[i][j] = (tr + 16 * i == tc + 16 * j) ? 1.0 : 0.0;
    for (int k = 0; k < H; ++k) {                        // M = I + c^T c: eight LDS reads feed sixteen multiply-adds
        double a[4], bq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = (double)s_c[k * 65 + tr + 16 * i];
#pragma unroll
        for (int j = 0; j < 4; ++j) bq[j] = (double)s_c[k * 65 + tc + 16 * j];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) t[i][j] = fma(a[i], bq[j], t[i][j]);
    }
    gauss_jordan_fma(t, H, s_row, s_col);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = tr + 16 * i, c = tc + 16 * j;
            if (r < H && c < H) {
                Qout[r * H + c] = (float)t[i][j];
                Pout[r * H + c] = (float)((r == c ? 1.0 : 0.0) - t[i][j]);
            }
        }
}


// All CRF layers of a network in ONE launch (one workgroup each): the Gauss-Jordan sweep is a latency chain of ~20 us
// whatever H is, so four layers cost what one does.
constexpr int CM_MAX = 8;
struct CrfMatJobs {
    const float* c[CM_MAX];
    const float* Q_in[CM_MAX];
    const float* gQ[CM_MAX];
    const float* gP[CM_MAX];
    float* Q[CM_MAX];
    float* P[CM_MAX];
    float* dc[CM_MAX];
    int H[CM_MAX];
    int slab_base[CM_MAX + 1];                     // backward: prefix of ceil(H / CMB_ROWS) -- first workgroup of each layer
};

}  // namespace crf
