// Linear -> BatchNorm(train) -> LeakyReLU of the COARSE levels (m <= 4096 rows, 64..512 channels) as ONE launch.
//
// Reference op chain: models/common.py:34-40 (MLP = Linear(bias=False) + FastBatchNorm1d + activation) at the encoder /
// decoder levels 3-5 of models/point_conv_big.py:113-131.  There a layer is a few hundred kFLOP..1 GFLOP and three
// launches (GEMM, statistics, apply) of 10-13 us each; here one kernel computes a 64-row x 64-channel tile of
// Y = X W^T per workgroup on the fp32 matrix cores, publishes the tile's per-channel {shift, n, sum, sum of squares}
// records, meets every other workgroup at a grid barrier (gridsync.hpp: all workgroups resident, checked by the host),
// folds the records of its 64 channels in float64 (Chan's parallel variance, fixed order: every row block derives
// bit-identical coefficients), and applies BatchNorm + LeakyReLU to the tile it still holds in registers.  Y (needed by
// the backward) and the activation are each written once; nothing is re-read.
#include "common.hpp"
#include "gridsync.hpp"

namespace crf {

using f32x4s = __attribute__((ext_vector_type(4))) float;

constexpr int SM_BLOCK = 256, SM_ROWS = 64, SM_COLS = 64, SM_KC = 64, SM_LD = SM_KC + 4;   // SM_COLS: the widest tile
#ifndef SM_MAX_ROWS_
#define SM_MAX_ROWS_ 4096
#endif
constexpr int64_t SM_MAX_ROWS = SM_MAX_ROWS_;      // (A/B: 10 240 = the third level joins the one-launch form, profiles/r4_ab_runs.md)
#ifndef SM_MIN_BLOCKS
#define SM_MIN_BLOCKS 256
#endif

// TCO: 16-channel output tiles per workgroup (tile = 64 rows x 16 TCO channels).  A wavefront issues Ci / 4 x TCO MFMAs, so
// at Ci = 512 the 64-wide tile is 7 us of matrix pipe per wavefront while most CUs idle (40 workgroups for 1280 x 128
// outputs): the host picks the narrowest tile that still gives the device >= SM_MIN_BLOCKS workgroups.
template <int TCO>
__global__ __launch_bounds__(SM_BLOCK) void mlp_small_fwd_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                                 int M, int Ci, int Co, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float* __restrict__ run_mean,
                                                                 float* __restrict__ run_var, float momentum, float eps,
                                                                 float slope, float* __restrict__ Y, float* __restrict__ A,
                                                                 float* __restrict__ coef, float* __restrict__ rec,
                                                                 unsigned* __restrict__ sync_ws,
                                                                 const float* __restrict__ skip, float join_slope) {
    __shared__ float sWbuf[2][SM_COLS * SM_LD];                // W chunk [64 co][64 k] (+4 pad: conflict-free b128 reads), double-buffered
    __shared__ double s_comb[2][SM_BLOCK];
    __shared__ float s_ab[2][SM_COLS];
    __shared__ int s_ok;
    float* const sW = sWbuf[0];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rr = lane & 15, g = lane >> 4;
    constexpr int COLS = 16 * TCO;
    const int rb = blockIdx.x, co_base = blockIdx.y * COLS;
    const int r = rb * SM_ROWS + wave * 16 + rr;
    const bool rv = r < M;
    const float* xrow = X + (int64_t)(rv ? r : 0) * Ci;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    f32x4s acc[TCO];
#pragma unroll
    for (int t = 0; t < TCO; ++t) acc[t] = f32x4s{0.f, 0.f, 0.f, 0.f};

    // software pipeline over 64-wide K chunks: the next chunk's weight slab (four float4 per thread) and X fragments are
    // in flight while the matrix cores work on the current one; one __syncthreads per chunk
    const int wr = threadIdx.x >> 4, k4 = threadIdx.x & 15;      // this thread stages W rows wr, wr + 16, wr + 32, wr + 48
    const float* wsrc = W + (int64_t)(co_base + wr) * Ci + 4 * k4;
    float4 wreg[TCO], xv[4], xn[4];
#pragma unroll
    for (int i = 0; i < TCO; ++i) wreg[i] = 4 * k4 < Ci ? ld4(wsrc + (int64_t)16 * i * Ci) : zero4;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int k0 = 16 * c + 4 * g;
        xv[c] = (rv && k0 < Ci) ? ld4(xrow + k0) : zero4;
    }
#pragma unroll
    for (int i = 0; i < TCO; ++i) st4(sWbuf[0] + (wr + 16 * i) * SM_LD + 4 * k4, wreg[i]);
    __syncthreads();
    int cur = 0;
    for (int kc = 0; kc < Ci; kc += SM_KC) {
        const int kn = kc + SM_KC;
        const bool more = kn < Ci;
        if (more) {
#pragma unroll
            for (int i = 0; i < TCO; ++i) wreg[i] = kn + 4 * k4 < Ci ? ld4(wsrc + (int64_t)16 * i * Ci + kn) : zero4;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int k0 = kn + 16 * c + 4 * g;
                xn[c] = (rv && k0 < Ci) ? ld4(xrow + k0) : zero4;
            }
        }
        const float* sWc = sWbuf[cur];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int t = 0; t < TCO; ++t) {
                const float4 wv = ld4(sWc + (16 * t + rr) * SM_LD + 16 * c + 4 * g);
                // D[i = co][j = row]: A = W fragment (i = lane & 15), B = X fragment (j = lane & 15)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.x, xv[c].x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.y, xv[c].y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.z, xv[c].z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.w, xv[c].w, acc[t], 0, 0, 0);
            }
        }
        if (more) {
#pragma unroll
            for (int i = 0; i < TCO; ++i) st4(sWbuf[cur ^ 1] + (wr + 16 * i) * SM_LD + 4 * k4, wreg[i]);
#pragma unroll
            for (int c = 0; c < 4; ++c) xv[c] = xn[c];
        }
        __syncthreads();                                        // chunk kn staged; every reader of buffer `cur` is done
        cur ^= 1;
    }
    // lane holds Y[row r][co_base + 16 t + 4 g + e], e = 0..3
    if (rv) {
#pragma unroll
        for (int t = 0; t < TCO; ++t)
            st4(Y + (int64_t)r * Co + co_base + 16 * t + 4 * g, make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]));
    }

    // ---- the tile's statistic records: shifted sums per wave (shift = the wave's first row), folded over the four waves
    // in float64 re-based on wave 0's shift, one {shift, n, sum, sumsq} tuple per (row block, channel)
    __syncthreads();                                            // sW is dead: reuse it as [4 waves][4][64]
    {
        float* sw = sW + wave * 4 * COLS;
        const int nrows = M - (rb * SM_ROWS + wave * 16) < 16 ? (M - (rb * SM_ROWS + wave * 16) < 0 ? 0 : M - (rb * SM_ROWS + wave * 16)) : 16;
#pragma unroll
        for (int t = 0; t < TCO; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float sh = __shfl(acc[t][e], 16 * g, WAVE);
                const float d = rv ? acc[t][e] - sh : 0.f;
                float a = d, b = d * d;
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) {
                    a += __shfl_xor(a, o, WAVE);
                    b += __shfl_xor(b, o, WAVE);
                }
                if (rr == 0) {
                    const int cl = 16 * t + 4 * g + e;
                    sw[cl] = sh;
                    sw[COLS + cl] = (float)nrows;
                    sw[2 * COLS + cl] = a;
                    sw[3 * COLS + cl] = b;
                }
            }
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rrec = make_rsrc(rec, (int)((size_t)gridDim.x * Co * 4 * sizeof(float)));
    if (threadIdx.x < COLS) {
        const int cl = threadIdx.x;
        const float s0 = sW[cl];
        double n = 0.0, S1 = 0.0, S2 = 0.0;
        for (int w = 0; w < 4; ++w) {
            const float* q = sW + w * 4 * COLS;
            const double nb = q[COLS + cl];
            if (nb <= 0.0) continue;
            const double d = (double)q[cl] - (double)s0, a = q[2 * COLS + cl], b = q[3 * COLS + cl];
            n += nb;
            S1 += a + nb * d;
            S2 += b + 2.0 * d * a + nb * d * d;
        }
        st4_sc1(rrec, (int)(((size_t)rb * Co + co_base + cl) * 16), make_float4(s0, (float)n, (float)S1, (float)S2));
    }

    // ---- every tile's records are in memory once every workgroup has arrived
    const unsigned nblk = gridDim.x * gridDim.y, bid = blockIdx.y * gridDim.x + blockIdx.x;
    unsigned n_in_group, n_groups;
    grid_sync_groups(nblk, bid, n_in_group, n_groups);
    if (!fused_grid_sync<false>(sync_ws, 1u, n_in_group, n_groups, &s_ok, nullptr, bid)) {
        // The barrier gave up (a workgroup of the launch was never resident: CU mask, reserved CUs, a wrong capacity).  The
        // FW_FAIL word is set (sticky: ops.check_gridsync / FlatSGD.step raise on it); poison this tile so that the loss
        // turns NaN instead of training on garbage, and still count out -- the last workgroup out zeroes the barrier
        // words, so the next launch does not inherit a broken count.
        if (rv) {
            const float qnan = __int_as_float(0x7fc00000);
#pragma unroll
            for (int t = 0; t < TCO; ++t) st4(A + (int64_t)r * Co + co_base + 16 * t + 4 * g, make_float4(qnan, qnan, qnan, qnan));
        }
        fused_exit_reset(sync_ws, nblk, 2, bid);
        return;
    }

    // ---- coefficients of this workgroup's COLS channels: 256 / COLS groups of COLS threads, group p folds row blocks p,
    // p + parts, ... (fixed order, every load in flight at once), re-based on row block 0's shift
    {
        constexpr int PARTS = SM_BLOCK / COLS, NB = (int)(SM_MAX_ROWS / SM_ROWS) / PARTS;
        const int cl = threadIdx.x % COLS, part = threadIdx.x / COLS, co = co_base + cl;
        const float4 r0 = ld4_sc1(rrec, (int)(((size_t)co) * 16), 0);
        float4 v[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int b = part + PARTS * i;
            v[i] = b < (int)gridDim.x ? ld4_sc1(rrec, (int)(((size_t)b * Co + co) * 16), 0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const double s0 = r0.x;
        double S1 = 0.0, S2 = 0.0;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const double nb = v[i].y;
            if (nb > 0.0) {
                const double d = (double)v[i].x - s0, a = v[i].z, bb = v[i].w;
                S1 += a + nb * d;
                S2 += bb + 2.0 * d * a + nb * d * d;
            }
        }
        s_comb[0][threadIdx.x] = S1;                              // slot part * COLS + cl
        s_comb[1][threadIdx.x] = S2;
        __syncthreads();
        if (threadIdx.x < COLS) {
            S1 = 0.0; S2 = 0.0;
#pragma unroll
            for (int p2 = 0; p2 < PARTS; ++p2) { S1 += s_comb[0][p2 * COLS + cl]; S2 += s_comb[1][p2 * COLS + cl]; }
            const double m1 = S1 / (double)M;
            const double mean = s0 + m1;
            double var = (S2 - S1 * m1) / (double)M;
            if (var < 0.0) var = 0.0;
            const double rstd = 1.0 / sqrt(var + (double)eps);
            const double a = (double)gamma[co] * rstd;
            const float af = (float)a, bf = (float)((double)beta[co] - a * mean);
            s_ab[0][cl] = af;
            s_ab[1][cl] = bf;
            if (rb == 0) {                                      // one row block publishes the layer's coefficients
                coef[co] = af;
                coef[Co + co] = bf;
                coef[2 * Co + co] = (float)mean;
                coef[3 * Co + co] = (float)rstd;
                if (run_mean != nullptr) {
                    const double unb = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
                    run_mean[co] = (float)((1.0 - (double)momentum) * (double)run_mean[co] + (double)momentum * mean);
                    run_var[co] = (float)((1.0 - (double)momentum) * (double)run_var[co] + (double)momentum * unb);
                }
            }
        }
        __syncthreads();
    }
    if (rv) {
#pragma unroll
        for (int t = 0; t < TCO; ++t) {
            const float4 a4 = ld4(&s_ab[0][16 * t + 4 * g]), b4 = ld4(&s_ab[1][16 * t + 4 * g]);
            float4 o;
            o.x = fmaf(a4.x, acc[t][0], b4.x); o.y = fmaf(a4.y, acc[t][1], b4.y);
            o.z = fmaf(a4.z, acc[t][2], b4.z); o.w = fmaf(a4.w, acc[t][3], b4.w);
            o.x = o.x > 0.f ? o.x : o.x * slope; o.y = o.y > 0.f ? o.y : o.y * slope;
            o.z = o.z > 0.f ? o.z : o.z * slope; o.w = o.w > 0.f ? o.w : o.w * slope;
            if (skip != nullptr) {                              // the ResNet join: lrelu(. + skip, join_slope), as add_lrelu computes it
                const float4 k4 = ld4(skip + (int64_t)r * Co + co_base + 16 * t + 4 * g);
                o.x = add_rn(o.x, k4.x); o.y = add_rn(o.y, k4.y); o.z = add_rn(o.z, k4.z); o.w = add_rn(o.w, k4.w);
                o.x = o.x > 0.f ? o.x : o.x * join_slope; o.y = o.y > 0.f ? o.y : o.y * join_slope;
                o.z = o.z > 0.f ? o.z : o.z * join_slope; o.w = o.w > 0.f ? o.w : o.w * join_slope;
            }
            st4(A + (int64_t)r * Co + co_base + 16 * t + 4 * g, o);
        }
    }
    fused_exit_reset(sync_ws, nblk, 2, bid);
}

// Workgroups of the forward kernel that can be resident at once on this device (the barrier needs every one of them).
static int mlp_small_capacity() {
    static int cap = -1;
    if (cap < 0) {
        int dev = 0, cus = 0, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mlp_small_fwd_kernel<4>, SM_BLOCK, 0) != hipSuccess) return 0;
        if (per_cu > 2) per_cu = 2;                                // stay well inside what the dispatcher really co-schedules
        cap = cus * per_cu;
    }
    return cap;
}

static bool mlp_small_shape_ok(int64_t M, int Ci, int Co) {
    return M >= 1 && M <= SM_MAX_ROWS && Ci >= 16 && Ci % 16 == 0 && Ci <= 1024 && Co >= SM_COLS && Co % SM_COLS == 0 && Co <= 1024;
}

// 16-channel tiles per workgroup: the widest of {4, 2, 1} that still spreads the layer over >= SM_MIN_BLOCKS workgroups (and fits)
static int mlp_small_tco(int64_t M, int Co, int cap) {
    const int64_t rb = cdiv(M, SM_ROWS);
    for (int tco = 4; tco >= 1; tco >>= 1) {
        const int64_t nblk = rb * (Co / (16 * tco));
        if (nblk > cap) return 0;                                  // narrower tiles only make more workgroups
        if (nblk >= SM_MIN_BLOCKS || tco == 1) return tco;
    }
    return 0;
}

}  // namespace crf

using namespace crf;

extern "C" size_t crfconv_gridsync_workspace(void) { return FW_TOTAL_WORDS * sizeof(unsigned); }

// index (in 32-bit words) of the sticky failure word inside a barrier workspace: non-zero after a launch whose grid
// barrier gave up (0x100 | phase).  The host reads it once per step / every few steps and raises.
extern "C" int crfconv_gridsync_fail_word(void) { return FW_FAIL * FW_LINE; }

// 1 when crfconv_mlp_small_forward applies: m <= 4096 rows, Ci a multiple of 16, Co a multiple of 64 (both <= 1024), and
// the (m / 64) x (Co / 64) workgroups fit the device at once.  Needs a GPU (occupancy query); 0 otherwise.
extern "C" int crfconv_mlp_small_supported(int64_t M, int Ci, int Co) {
    if (!mlp_small_shape_ok(M, Ci, Co)) return 0;
    return mlp_small_tco(M, Co, mlp_small_capacity()) > 0 ? 1 : 0;
}

extern "C" size_t crfconv_mlp_small_workspace(int64_t M, int Co) {
    if (M < 1 || Co < 1) return 0;
    return (size_t)cdiv(M, SM_ROWS) * (size_t)Co * 4 * sizeof(float);
}

// A = lrelu(BatchNorm_train(X W^T), slope); Y = X W^T is kept for the backward; coef [4][Co] = a | b | mean | rstd;
// running statistics updated when given.  sync_ws: crfconv_gridsync_workspace() bytes, ZERO before the first launch
// that uses it (the kernel leaves it zero); one such buffer per stream of concurrent launches.
static int mlp_small_forward_impl(const float* X, const float* W, int64_t M, int Ci, int Co, const float* gamma,
                                  const float* beta, float* run_mean, float* run_var, float momentum, float eps,
                                  float slope, const float* skip, float join_slope, float* Y, float* A, float* coef, void* ws,
                                  size_t ws_bytes, void* sync_ws, size_t sync_bytes, crf_stream_t stream) {
    CRF_REQUIRE(X && W && gamma && beta && Y && A && coef && ws && sync_ws, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(mlp_small_shape_ok(M, Ci, Co), CRF_ERR_UNSUPPORTED, "shape m=%lld Ci=%d Co=%d outside the one-launch kernel",
                (long long)M, Ci, Co);
    CRF_REQUIRE(ws_bytes >= crfconv_mlp_small_workspace(M, Co), CRF_ERR_WORKSPACE, "workspace too small");
    CRF_REQUIRE(sync_bytes >= crfconv_gridsync_workspace(), CRF_ERR_WORKSPACE, "barrier words: %zu bytes, need %zu", sync_bytes,
                crfconv_gridsync_workspace());
    const int tco = mlp_small_tco(M, Co, mlp_small_capacity());
    CRF_REQUIRE(tco > 0, CRF_ERR_UNSUPPORTED, "the workgroups of m=%lld Co=%d cannot all be resident (capacity %d)", (long long)M, Co,
                mlp_small_capacity());
    const dim3 grid((unsigned)cdiv(M, SM_ROWS), (unsigned)(Co / (16 * tco)));
#define SMF(T) hipLaunchKernelGGL(mlp_small_fwd_kernel<T>, grid, dim3(SM_BLOCK), 0, as_stream(stream), X, W, (int)M, Ci, Co, gamma, beta, \
                                  run_mean, run_var, momentum, eps, slope, Y, A, coef, reinterpret_cast<float*>(ws),               \
                                  reinterpret_cast<unsigned*>(sync_ws), skip, join_slope)
    if (tco == 4) SMF(4);
    else if (tco == 2) SMF(2);
    else SMF(1);
#undef SMF
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_mlp_small_forward(const float* X, const float* W, int64_t M, int Ci, int Co, const float* gamma,
                                         const float* beta, float* run_mean, float* run_var, float momentum, float eps,
                                         float slope, float* Y, float* A, float* coef, void* ws, size_t ws_bytes,
                                         void* sync_ws, size_t sync_bytes, crf_stream_t stream) {
    return mlp_small_forward_impl(X, W, M, Ci, Co, gamma, beta, run_mean, run_var, momentum, eps, slope, nullptr, 1.f, Y, A, coef, ws,
                                  ws_bytes, sync_ws, sync_bytes, stream);
}

// The same launch with the ResNet join folded in: A = lrelu(lrelu(BN_train(X W^T), slope) + skip, join_slope), skip [M, Co]
// (models/point_conv_big.py:84-88 with slope = 1: lin_out has no activation of its own).
extern "C" int crfconv_mlp_small_forward_join(const float* X, const float* W, int64_t M, int Ci, int Co, const float* gamma,
                                              const float* beta, float* run_mean, float* run_var, float momentum, float eps,
                                              float slope, const float* skip, float join_slope, float* Y, float* A, float* coef,
                                              void* ws, size_t ws_bytes, void* sync_ws, size_t sync_bytes, crf_stream_t stream) {
    CRF_REQUIRE(skip, CRF_ERR_ARG, "null pointer");
    return mlp_small_forward_impl(X, W, M, Ci, Co, gamma, beta, run_mean, run_var, momentum, eps, slope, skip, join_slope, Y, A, coef,
                                  ws, ws_bytes, sync_ws, sync_bytes, stream);
}
