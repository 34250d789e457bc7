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

namespace crf {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int WG_BLOCK = 256;
constexpr int WG_WAVES = WG_BLOCK / WAVE;

// TCO x TCI tiles of 16x16 per block (output slab 16*TCO x 16*TCI at (co0, ci0) = blockIdx.y / z).
template <int TCO, int TCI>
__global__ __launch_bounds__(WG_BLOCK) void wgrad_kernel(const float* __restrict__ G,
                                                         const float* __restrict__ X, int64_t M, int Co,
                                                         int Ci, int rows_per_block,
                                                         float* __restrict__ partial /*[nblk][Co][Ci]*/,
                                                         float* __restrict__ partial_b /*[nblk][Co] or null*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int co_base = blockIdx.y * 16 * TCO, ci_base = blockIdx.z * 16 * TCI;
    const int kk = lane >> 4, cc = lane & 15;
    f32x4 acc[TCO][TCI];
#pragma unroll
    for (int a = 0; a < TCO; ++a)
#pragma unroll
        for (int b = 0; b < TCI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[TCO];
#pragma unroll
    for (int a = 0; a < TCO; ++a) bsum[a] = 0.f;

    const int64_t row_begin = (int64_t)blockIdx.x * rows_per_block;
    const int64_t row_end = row_begin + rows_per_block < M ? row_begin + rows_per_block : M;
    // waves interleave 4-row k-steps inside the block's slice
    for (int64_t r0 = row_begin + 4 * wave; r0 < row_end; r0 += 4 * WG_WAVES) {
        const int64_t r = r0 + kk;
        const bool rv = r < row_end;
        float av[TCO], bv[TCI];
#pragma unroll
        for (int a = 0; a < TCO; ++a) {
            const int co = co_base + 16 * a + cc;
            av[a] = (rv && co < Co) ? G[r * Co + co] : 0.f;
        }
#pragma unroll
        for (int b = 0; b < TCI; ++b) {
            const int ci = ci_base + 16 * b + cc;
            bv[b] = (rv && ci < Ci) ? X[r * Ci + ci] : 0.f;
        }
#pragma unroll
        for (int a = 0; a < TCO; ++a) {
            bsum[a] += av[a];
#pragma unroll
            for (int b = 0; b < TCI; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
    }
    // C/D layout of 16x16x4: col = lane & 15 (j = ci), row = 4 * (lane >> 4) + reg (i = co)
    __shared__ float s_red[WG_WAVES][TCO * TCI * 256];
    __shared__ float s_b[WG_WAVES][TCO * 16];
#pragma unroll
    for (int a = 0; a < TCO; ++a) {
#pragma unroll
        for (int b = 0; b < TCI; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) s_red[wave][(a * TCI + b) * 256 + (4 * kk + g) * 16 + cc] = acc[a][b][g];
        // bias: lanes with the same cc over the 4 k-groups
        float t = bsum[a];
        t += __shfl_xor(t, 16, WAVE);
        t += __shfl_xor(t, 32, WAVE);
        if (kk == 0) s_b[wave][a * 16 + cc] = t;
    }
    __syncthreads();
    const int64_t pb = (int64_t)blockIdx.x;
    for (int t = threadIdx.x; t < TCO * TCI * 256; t += WG_BLOCK) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < WG_WAVES; ++w) v += s_red[w][t];
        const int tile = t >> 8, a = tile / TCI, b = tile % TCI, i = (t >> 4) & 15, j = t & 15;
        const int co = co_base + 16 * a + i, ci = ci_base + 16 * b + j;
        if (co < Co && ci < Ci) partial[(pb * Co + co) * Ci + ci] = v;
    }
    if (partial_b != nullptr && blockIdx.z == 0) {
        for (int t = threadIdx.x; t < TCO * 16; t += WG_BLOCK) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < WG_WAVES; ++w) v += s_b[w][t];
            const int co = co_base + t;
            if (co < Co) partial_b[pb * Co + co] = v;
        }
    }
}

// out[slot] = sum_b partial[b][slot], fixed order, one wavefront per slot.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int nblk,
                                                           int nslots, float* __restrict__ out) {
    const int slot = blockIdx.x * (256 / WAVE) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (slot >= nslots) return;
    float a = 0.f;
    for (int b = lane; b < nblk; b += WAVE) a += partial[(int64_t)b * nslots + slot];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, WAVE);
    if (lane == 0) out[slot] = a;
}

struct WgPlan {
    int tco, tci, gy, gz, nblk, rows_per_block;
};

static WgPlan wg_plan(int64_t M, int Co, int Ci) {
    WgPlan p;
    const int t_co = (Co + 15) / 16, t_ci = (Ci + 15) / 16;
    p.tco = t_co >= 4 ? 4 : (t_co >= 2 ? 2 : 1);
    p.tci = t_ci >= 4 ? 4 : (t_ci >= 2 ? 2 : 1);
    p.gy = (t_co + p.tco - 1) / p.tco;
    p.gz = (t_ci + p.tci - 1) / p.tci;
    // ~256 row-slices over the chip (x gy x gz output slabs), at least 64 rows (4 k-steps per wave) each:
    // the partial slabs the second kernel sums stay a small fraction of the operand bytes
    int64_t rows = (M + 255) / 256;
    if (rows < 64) rows = 64;
    rows = (rows + 15) / 16 * 16;
    p.rows_per_block = (int)rows;
    p.nblk = (int)((M + rows - 1) / rows);
    return p;
}

}  // namespace crf

using namespace crf;

extern "C" size_t crfconv_linear_wgrad_workspace(int64_t M, int Co, int Ci) {
    if (M <= 0 || Co <= 0 || Ci <= 0) return 0;
    const WgPlan p = wg_plan(M, Co, Ci);
    return sizeof(float) * (size_t)p.nblk * ((size_t)Co * Ci + (size_t)Co) + 256;
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
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv((int64_t)Co * Ci, 4)), dim3(256), 0, st, partial, p.nblk,
                       Co * Ci, dW);
    CRF_LAUNCH_CHECK();
    if (db) {
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv((int64_t)Co, 4)), dim3(256), 0, st, partial_b, p.nblk,
                           Co, db);
        CRF_LAUNCH_CHECK();
    }
    return CRF_OK;
}

// ------------------------------------------------------------------ (I + C)^-1 for the CRF layers
// Gauss-Jordan on the H x H (H <= 64) symmetric positive definite matrix M = I + c^T c
// (models/continuous_crf_conv_big.py:72 calls .inverse() inside the loop; it is loop invariant).
// One workgroup, float64 in LDS, no pivoting needed (eigenvalues >= 1).  Replaces torch.linalg.inv,
// whose rocSOLVER path synchronises and therefore cannot be captured into a hipGraph.
namespace crf {
__global__ __launch_bounds__(256) void spd_inverse_kernel(const float* __restrict__ Min, int H,
                                                          float* __restrict__ Qout) {
    extern __shared__ double aug[];  // [H][2H]
    const int W = 2 * H;
    const int n = H * W;
    for (int t = threadIdx.x; t < n; t += 256) {
        const int r = t / W, c = t % W;
        aug[t] = c < H ? (double)Min[r * H + c] : (c - H == r ? 1.0 : 0.0);
    }
    __syncthreads();
    // each thread owns elements t = tid, tid + 256, ... (<= 32 of them at H = 64); per pivot: read the
    // old pivot row / column entries, barrier, write the updated element, barrier.
    for (int p = 0; p < H; ++p) {
        const double piv = 1.0 / aug[p * W + p];
        double nv[32];
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            const int t = threadIdx.x + 256 * e;
            if (t < n) {
                const int r = t / W, c = t - r * W;
                const double prc = aug[p * W + c] * piv;                  // normalised pivot-row entry
                nv[e] = r == p ? prc : aug[t] - aug[r * W + p] * prc;
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            const int t = threadIdx.x + 256 * e;
            if (t < n) aug[t] = nv[e];
        }
        __syncthreads();
    }
    for (int t = threadIdx.x; t < H * H; t += 256) Qout[t] = (float)aug[(t / H) * W + H + (t % H)];
}
}  // namespace crf

extern "C" int crfconv_spd_inverse(const float* M, int H, float* Q, crf_stream_t stream) {
    CRF_REQUIRE(M && Q, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(H >= 1 && H <= 64, CRF_ERR_UNSUPPORTED, "H=%d outside [1, 64]", H);
    hipLaunchKernelGGL(crf::spd_inverse_kernel, dim3(1), dim3(256), sizeof(double) * 2 * H * H, crf::as_stream(stream), M,
                       H, Q);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}
