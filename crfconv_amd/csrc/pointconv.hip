// PointConv (depth-wise point convolution with a per-edge weight MLP), gfx950.
//
// Reference semantics: models/point_conv_big.py:37-58.  The reference materialises
// [B, N, K, d] tensors for rel-pos, both MLP layers, the gathered features and their product;
// here nothing per-edge ever reaches HBM: every pass re-derives the weight MLP from the two
// 12-byte positions, and train-mode BatchNorm statistics come from dedicated reduction passes.
//
// Thread mapping: the d channels of a target point sit on L = d/4 adjacent lanes (one float4
// each); a 64-lane wavefront carries 64/L points and loops over their K neighbours.  Layer 1
// (3 -> d) is computed per lane for its own quad; layer 2 (d -> d) broadcasts h1 over the L lanes
// with shuffles against float4 rows of W2^T (registers for d <= 16, LDS above).
#include "common.hpp"
#include "outer_acc.hpp"
#include "gridsync.hpp"
#include "crf_matrices_body.hpp"
#include "reduce64_body.hpp"
#include <algorithm>

namespace crf {

#ifndef PARAMS_PB
#define PARAMS_PB 4
#endif
#ifndef UV_GROUP
#define UV_GROUP 4            // edges per trip of the narrow statistics / convolution pass (uvstats_kernel): 2 (round 2) 28.8, 8 27.5, 6 ~26, 4 24.9 us at d = 8
#endif
#ifndef PARAMS_PB_NARROW
#define PARAMS_PB_NARROW 4      // (d <= 8 with eight edges per trip, 240 registers: 51.5 vs 49.2 us -- not the trips)
#endif
constexpr int PBLOCK = 256;
constexpr int PWAVES = PBLOCK / WAVE;

// Wide layers live on the coarse levels (d = 64: 2560 points, d = 128: 1280 points at config 2): with a point on d/4 lanes a
// launch has a few hundred wavefronts, each one walking ALL K edges of its points through the d x d layer-2 product --
// one wavefront per SIMD running serially for 30-50 us (uvstats / bwd_dump / bwd_input at d = 128).  From d = 64 on, KS
// wavefronts of a workgroup therefore share the SAME points and split their edges (wavefront w of a group takes edge
// batches w, w + KS, ...); what a point accumulates is summed across the wavefronts through LDS (ks_sum) or by the block reductions
// that exist anyway.
#ifndef PC_KS_32
#define PC_KS_32 1
#endif
#ifndef PC_KS_64
#define PC_KS_64 4
#endif
#ifndef PC_KS_128
#define PC_KS_128 2          // swept with PC_KS_64 on the training step: (4, 4) 5.285, (2, 2) 5.249, (2, 4) 5.267, (4, 2) 5.235, (1, 2) 5.271 ms
#endif
// matrix-pipe forms of the wide layers (pointconv_wide.hip)
bool uvstats_mfma_ok(int K, int d);
int uvstats_mfma_launch(const float* x, const float* pos_src, const float* pos_tgt, const int32_t* idx32, int64_t m_tgt, int d,
                        const float* A1, const float* b1, const float* W2, float slope, const float* mean_rel3, float* shift, float* U,
                        float* V, float* partial, int64_t max_blocks, int64_t* nblk_out, unsigned* ticket, double* stats, hipStream_t st);

template <int D>
struct PC {
    static constexpr int L = D / 4;
    static constexpr int PPW = WAVE / L;
    static constexpr int KS = D >= 128 ? PC_KS_128 : (D >= 64 ? PC_KS_64 : (D >= 32 ? PC_KS_32 : 1));       // wavefronts sharing a point's edges (1, 2 or 4)
    static constexpr int PPB = PPW * PWAVES / KS;
#ifndef PC_W2_REGS_MAX_D
#define PC_W2_REGS_MAX_D 8       // layers up to this width keep W2^T in registers (0: every width reads it from LDS)
#endif
#ifndef PC_NARROW_WAVES
#define PC_NARROW_WAVES 1        // minimum waves per SIMD asked for the narrow statistics kernels
#endif
    static constexpr bool W2_IN_REGS = (D <= PC_W2_REGS_MAX_D);
    // W2^T rows in LDS: [D input channels][L + 1 float4] -- one float4 of padding per row, so that the transposing stage
    // (coalesced global reads of W2 rows, scattered LDS writes) spreads over eight banks instead of one
    static constexpr int W2LD = L + 1;
    // narrow layers (d <= 16) whose W2^T lives in LDS keep layer 1's rows {A1[c][0..2], b1[c]} there too (behind W2^T): 16 registers
    static constexpr bool A1_IN_LDS = !W2_IN_REGS && D <= 16;
    static constexpr int W2T_F4 = W2_IN_REGS ? 1 : D * W2LD + (A1_IN_LDS ? D : 0);      // float4 of the kernels' LDS weight array
    // Wide layers evaluate layer 2 for EB = 4 edges at a time: each W2^T row fetched from LDS then feeds 16 FMAs
    // per lane instead of 4, and h1 is exchanged through a per-wave LDS scratch (broadcast reads) instead of
    // one cross-lane shuffle per input channel -- the d x d product becomes VALU-bound instead of LDS-bound.
    static constexpr int EB = (D >= 32) ? 4 : 1;
    static constexpr int SCR_STRIDE = EB * L + 1;                 // float4 units per point (+1: bank spread)
    static constexpr int SCR_SIZE = EB > 1 ? PWAVES * PPW * SCR_STRIDE : 1;
};

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : slope * v; }

// Per-thread constants of the weight MLP for this lane's channel quad.
template <int D>
struct EdgeMLP {
    static constexpr int L = PC<D>::L;
    float4 a1[4];  // a1[c] = {A1[c][0], A1[c][1], A1[c][2], b1[c]} for the quad's 4 channels
    float4 w2t_reg[PC<D>::W2_IN_REGS ? D : 1];  // W2T[c'][quad] when held in registers
    const float4* w2t_lds;                      // [D][L + 1] float4 rows otherwise (PC<D>::W2LD)
    int lane, q;
    float slope;                                // LeakyReLU slope of layer 1 (0.1 dense, 0.01 sparse twin)

    __device__ __forceinline__ void init(const float* __restrict__ A1, const float* __restrict__ b1,
                                         const float* __restrict__ W2, float4* lds_w2t, int lane_, int q_, float slope_) {
        lane = lane_;
        q = q_;
        slope = slope_;
        if constexpr (PC<D>::A1_IN_LDS) {
            if ((int)threadIdx.x < D)
                lds_w2t[D * PC<D>::W2LD + threadIdx.x] = make_float4(A1[threadIdx.x * 3 + 0], A1[threadIdx.x * 3 + 1], A1[threadIdx.x * 3 + 2], b1[threadIdx.x]);
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ch = 4 * q + c;
                a1[c] = make_float4(A1[ch * 3 + 0], A1[ch * 3 + 1], A1[ch * 3 + 2], b1[ch]);
            }
        }
        if constexpr (PC<D>::W2_IN_REGS) {
#pragma unroll
            for (int cp = 0; cp < D; ++cp)  // W2T[cp][4q+c] = W2[4q+c][cp]
                w2t_reg[cp] = make_float4(W2[(4 * q + 0) * D + cp], W2[(4 * q + 1) * D + cp],
                                          W2[(4 * q + 2) * D + cp], W2[(4 * q + 3) * D + cp]);
            w2t_lds = nullptr;
        } else {
            float* s = reinterpret_cast<float*>(lds_w2t);
            // W2 is read the way it lies in memory (consecutive threads, consecutive addresses) and transposed on the way
            // into LDS: staged the other way round -- LDS order, i.e. a 4 D-byte stride between the lanes' global reads --
            // this prologue was ~15 us of every d = 128 kernel (64 KB through 64-line gathers)
            for (int t = threadIdx.x; t < D * D; t += PBLOCK) {
                const int c = t / D, cp = t % D;               // W2[c][cp] -> row cp (input channel), column c (output)
                s[cp * 4 * PC<D>::W2LD + c] = W2[t];
            }
            w2t_lds = lds_w2t;
        }
    }

    // pre-activation and activation of layer 1 for this lane's quad
    __device__ __forceinline__ void layer1(float rx, float ry, float rz, float4& pre, float4& h1) const {
        float4 r0, r1, r2, r3;
        if constexpr (PC<D>::A1_IN_LDS) {
            int ao = 0;
            asm volatile("" : "+v"(ao));                 // (read per edge, as W2^T: see layer2)
            const float4* ar = w2t_lds + D * PC<D>::W2LD + 4 * q + ao;
            r0 = ar[0]; r1 = ar[1]; r2 = ar[2]; r3 = ar[3];
        } else {
            r0 = a1[0]; r1 = a1[1]; r2 = a1[2]; r3 = a1[3];
        }
        pre.x = fmaf(r0.x, rx, fmaf(r0.y, ry, fmaf(r0.z, rz, r0.w)));
        pre.y = fmaf(r1.x, rx, fmaf(r1.y, ry, fmaf(r1.z, rz, r1.w)));
        pre.z = fmaf(r2.x, rx, fmaf(r2.y, ry, fmaf(r2.z, rz, r2.w)));
        pre.w = fmaf(r3.x, rx, fmaf(r3.y, ry, fmaf(r3.z, rz, r3.w)));
        h1 = make_float4(lrelu(pre.x, slope), lrelu(pre.y, slope), lrelu(pre.z, slope), lrelu(pre.w, slope));
    }

    // h2[quad] = sum_c' h1[c'] * W2[quad][c']
    __device__ __forceinline__ float4 layer2(float4 h1) const {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const int base = lane - q;
        static_for<L>([&](auto HQ) {
            constexpr int hq = decltype(HQ)::value;
            const float v0 = group_bcast<L, hq>(h1.x, base);
            const float v1 = group_bcast<L, hq>(h1.y, base);
            const float v2 = group_bcast<L, hq>(h1.z, base);
            const float v3 = group_bcast<L, hq>(h1.w, base);
            if constexpr (PC<D>::W2_IN_REGS) {
                acc = fma4(v0, w2t_reg[4 * hq + 0], acc);
                acc = fma4(v1, w2t_reg[4 * hq + 1], acc);
                acc = fma4(v2, w2t_reg[4 * hq + 2], acc);
                acc = fma4(v3, w2t_reg[4 * hq + 3], acc);
            } else {
                int wo = 0;
                if constexpr (D <= 16) asm volatile("" : "+v"(wo));    // (narrow layers: keep the reads inside the edge loop -- hoisted they are W2 in registers again)
                const float4* w = w2t_lds + wo;
                acc = fma4(v0, w[(4 * hq + 0) * PC<D>::W2LD + q], acc);
                acc = fma4(v1, w[(4 * hq + 1) * PC<D>::W2LD + q], acc);
                acc = fma4(v2, w[(4 * hq + 2) * PC<D>::W2LD + q], acc);
                acc = fma4(v3, w[(4 * hq + 3) * PC<D>::W2LD + q], acc);
            }
        });
        return acc;
    }

    // h2[e] = W2 h1[e] for EB edges of this lane's point at once (scratch: PC<D>::SCR_SIZE float4 in LDS)
    __device__ __forceinline__ void layer2_batch(const float4 (&h1)[PC<D>::EB], float4 (&h2)[PC<D>::EB],
                                                 float4* scratch) const {
        constexpr int EB = PC<D>::EB;
        if constexpr (EB == 1) {
            h2[0] = layer2(h1[0]);
        } else {
            const int wave = threadIdx.x >> 6, pl = lane / L;
            float4* mine = scratch + (wave * PC<D>::PPW + pl) * PC<D>::SCR_STRIDE;
#pragma unroll
            for (int e = 0; e < EB; ++e) {
                mine[e * L + q] = h1[e];
                h2[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            // same-wave LDS traffic is processed in order: the reads below see the writes above
#pragma unroll 2
            for (int cq = 0; cq < L; ++cq) {
                float4 hv[EB];
#pragma unroll
                for (int e = 0; e < EB; ++e) hv[e] = mine[e * L + cq];
                const float4 w0 = w2t_lds[(4 * cq + 0) * PC<D>::W2LD + q], w1 = w2t_lds[(4 * cq + 1) * PC<D>::W2LD + q];
                const float4 w2 = w2t_lds[(4 * cq + 2) * PC<D>::W2LD + q], w3 = w2t_lds[(4 * cq + 3) * PC<D>::W2LD + q];
#pragma unroll
                for (int e = 0; e < EB; ++e) {
                    h2[e] = fma4(hv[e].x, w0, h2[e]);
                    h2[e] = fma4(hv[e].y, w1, h2[e]);
                    h2[e] = fma4(hv[e].z, w2, h2[e]);
                    h2[e] = fma4(hv[e].w, w3, h2[e]);
                }
            }
        }
    }

    __device__ __forceinline__ float4 h2_of(float rx, float ry, float rz) const {
        float4 pre, h1;
        layer1(rx, ry, rz, pre, h1);
        return layer2(h1);
    }
};

struct Row {
    int64_t r;
    bool valid;
};

template <int D>
__device__ __forceinline__ Row my_row_at(int64_t m, unsigned block, int& lane, int& wave, int& q) {
    lane = threadIdx.x & 63;
    wave = threadIdx.x >> 6;
    q = lane % PC<D>::L;
    const int64_t row = (int64_t)block * PC<D>::PPB + (wave / PC<D>::KS) * PC<D>::PPW + lane / PC<D>::L;
    Row o;
    o.valid = row < m;
    o.r = o.valid ? row : m - 1;
    return o;
}
template <int D>
__device__ __forceinline__ Row my_row(int64_t m, int& lane, int& wave, int& q) {
    return my_row_at<D>(m, xcd_block_id(), lane, wave, q);
}

// Sum `v` over all lanes of the wave that share the same quad index q (xor over the point bits).
template <int D>
__device__ __forceinline__ float over_points(float v) {
#pragma unroll
    for (int o = PC<D>::L; o < WAVE; o <<= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

// Sum of a per-lane float4 over the KS wavefronts that share the same points (fixed order w = 0..KS-1); every wavefront
// returns the total.  s_ks: [PWAVES][WAVE] float4 in LDS.  All threads of the workgroup must call it.
template <int D>
__device__ __forceinline__ float4 ks_sum(float4 v, float4* s_ks, int lane, int wave) {
    if constexpr (PC<D>::KS == 1) {
        return v;
    } else {
        __syncthreads();                                   // s_ks may still be read from an earlier call
        s_ks[wave * WAVE + lane] = v;
        __syncthreads();
        const int w0 = wave - wave % PC<D>::KS;            // first wavefront of this group
        float4 t = s_ks[w0 * WAVE + lane];
#pragma unroll
        for (int w = 1; w < PC<D>::KS; ++w) {
            const float4 o = s_ks[(w0 + w) * WAVE + lane];
            t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
        }
        return t;
    }
}

// Block-level deterministic reduction of NV per-quad float4 values into partial[block][NV][D].
template <int D, int NV, bool WT = false>      // WT: rows stored write-through, for a last-workgroup sum in the same launch (gridsync.hpp)
__device__ __forceinline__ void block_reduce_store(const float4 (&v)[NV], float* sred /*[PWAVES][NV][D]*/,
                                                   float* __restrict__ partial, int lane, int wave, int q,
                                                   const unsigned bid = blockIdx.x, const unsigned nblk = gridDim.x) {
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        float4 t = v[n];
        t.x = over_points<D>(t.x);
        t.y = over_points<D>(t.y);
        t.z = over_points<D>(t.z);
        t.w = over_points<D>(t.w);
        if (lane < PC<D>::L) st4(sred + (wave * NV + n) * D + 4 * q, t);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < NV * D; t += PBLOCK) {
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < PWAVES; ++w) a += sred[w * NV * D + t];
        if constexpr (WT) st1_sc1(make_rsrc(partial, (int)nblk * NV * D * 4), ((int)bid * NV * D + t) * 4, a);
        else partial[(int64_t)bid * NV * D + t] = a;
    }
}

// ------------------------------------------------------------------ generic partial reduction
// out[slot] = sum_b partial[b][slot] in double, fixed order: one wavefront per slot, lane l sums
// blocks l, l+64, ... then a shuffle tree (same result every run).
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial,
                                                              int64_t nblk, int nslots,
                                                              double* __restrict__ out) {
    const int slot = blockIdx.x * (256 / WAVE) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (slot >= nslots) return;
    double a = 0.0;
    for (int64_t b = lane; b < nblk; b += WAVE) a += (double)partial[b * nslots + slot];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, WAVE);
    if (lane == 0) out[slot] = a;
}

__global__ __launch_bounds__(256) void reduce_partials_d_kernel(const double* __restrict__ partial,
                                                                int64_t nblk, int nslots,
                                                                double* __restrict__ out) {
    const int slot = blockIdx.x * (256 / WAVE) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (slot >= nslots) return;
    double a = 0.0;
    for (int64_t b = lane; b < nblk; b += WAVE) a += partial[b * nslots + slot];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, WAVE);
    if (lane == 0) out[slot] = a;
}

// (table and body: reduce64_body.hpp)
__global__ __launch_bounds__(256) void reduce_jobs_f64_kernel(const Reduce64Table t) { reduce_jobs_f64_body(t, blockIdx.x); }

// ------------------------------------------------------------------ rel-pos moments
__device__ __forceinline__ void moments_block(const float* __restrict__ pos_src, const float* __restrict__ pos_tgt,
                                              const int32_t* __restrict__ idx, int K, int64_t m_tgt, int64_t blk,
                                              float* __restrict__ partial_row) {
    __shared__ float sred[PWAVES][9];
    const int64_t i = blk * 256 + threadIdx.x;
    float a[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < m_tgt) {
        const float px = pos_tgt[3 * i], py = pos_tgt[3 * i + 1], pz = pos_tgt[3 * i + 2];
        for (int k = 0; k < K; ++k) {
            const int64_t j = idx[i * K + k];
            if (j < 0) continue;                       // "no neighbour" (padded variable-degree table)
            const float rx = px - pos_src[3 * j], ry = py - pos_src[3 * j + 1], rz = pz - pos_src[3 * j + 2];
            a[0] += rx; a[1] += ry; a[2] += rz;
            a[3] = fmaf(rx, rx, a[3]); a[4] = fmaf(rx, ry, a[4]); a[5] = fmaf(rx, rz, a[5]);
            a[6] = fmaf(ry, ry, a[6]); a[7] = fmaf(ry, rz, a[7]); a[8] = fmaf(rz, rz, a[8]);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int n = 0; n < 9; ++n) {
        const float t = wave_sum(a[n]);
        if (lane == 0) sred[wave][n] = t;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < PWAVES; ++w) t += sred[w][threadIdx.x];
        partial_row[threadIdx.x] = t;
    }
}
__global__ __launch_bounds__(256) void moments_kernel(const float* __restrict__ pos_src,
                                                      const float* __restrict__ pos_tgt,
                                                      const int32_t* __restrict__ idx, int K,
                                                      int64_t m_tgt, float* __restrict__ partial) {
    moments_block(pos_src, pos_tgt, idx, K, m_tgt, blockIdx.x, partial + (int64_t)blockIdx.x * 9);
}

// The moments of SEVERAL tables (a batch refresh recomputes them for every PointConv layer's table: nine pairs of launches one
// by one): the same per-workgroup partials and the same finishing order per table, two launches for all of them.
constexpr int MB_MAX = 16;
struct MomentsBatch {
    const float* pos_src[MB_MAX];
    const float* pos_tgt[MB_MAX];
    const int32_t* idx[MB_MAX];
    double* mean[MB_MAX];
    double* cov[MB_MAX];
    double* packed[MB_MAX];
    float* mean32[MB_MAX];
    double n_edges[MB_MAX];
    int K[MB_MAX], m_tgt[MB_MAX];
    int blk_base[MB_MAX + 1];
    int njobs;
};
__global__ __launch_bounds__(256) void moments_batched_kernel(const MomentsBatch t, float* __restrict__ partial) {
    int lo = 0, hi = t.njobs;                          // largest j with blk_base[j] <= blockIdx.x
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (t.blk_base[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
    }
    moments_block(t.pos_src[lo], t.pos_tgt[lo], t.idx[lo], t.K[lo], t.m_tgt[lo], (int)blockIdx.x - t.blk_base[lo],
                  partial + (int64_t)blockIdx.x * 9);
}

// ------------------------------------------------------------------ BatchNorm-2 statistics of h2
template <int D>
__global__ __launch_bounds__(PBLOCK) void stats_kernel(const float* __restrict__ pos_src,
                                                       const float* __restrict__ pos_tgt,
                                                       const int32_t* __restrict__ idx, int K,
                                                       int64_t m_tgt, const float* __restrict__ A1,
                                                       const float* __restrict__ b1,
                                                       const float* __restrict__ W2, float slope,
                                                       const float* __restrict__ mean_rel,
                                                       float* __restrict__ shift_out,
                                                       float* __restrict__ partial) {
    constexpr int EB = PC<D>::EB;
    __shared__ float4 s_w2t[PC<D>::W2T_F4];
    __shared__ float4 s_scr[PC<D>::SCR_SIZE];
    __shared__ float sred[PWAVES * 2 * D];
    int lane, wave, q;
    const Row rw = my_row<D>(m_tgt, lane, wave, q);
    EdgeMLP<D> mlp;
    mlp.init(A1, b1, W2, s_w2t, lane, q, slope);
    __syncthreads();
    const float4 shift = mlp.h2_of(mean_rel[0], mean_rel[1], mean_rel[2]);
    if (blockIdx.x == 0 && threadIdx.x < PC<D>::L) st4(shift_out + 4 * q, shift);

    const float px = pos_tgt[3 * rw.r], py = pos_tgt[3 * rw.r + 1], pz = pos_tgt[3 * rw.r + 2];
    const int32_t* irow = idx + rw.r * K;
    float4 acc[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    for (int k0 = (wave % PC<D>::KS) * EB; k0 < K; k0 += PC<D>::KS * EB) {
        float4 h1[EB], h2[EB];
        float live[EB];
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            const int jj = (k0 + e < K) ? irow[k0 + e] : -1;
            const int64_t j = jj < 0 ? 0 : jj;
            live[e] = (rw.valid && jj >= 0) ? 1.f : 0.f;
            float4 pre;
            mlp.layer1(px - pos_src[3 * j], py - pos_src[3 * j + 1], pz - pos_src[3 * j + 2], pre, h1[e]);
        }
        mlp.layer2_batch(h1, h2, s_scr);
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            const float4 dlt = make_float4((h2[e].x - shift.x) * live[e], (h2[e].y - shift.y) * live[e],
                                           (h2[e].z - shift.z) * live[e], (h2[e].w - shift.w) * live[e]);
            acc[0].x += dlt.x; acc[0].y += dlt.y; acc[0].z += dlt.z; acc[0].w += dlt.w;
            acc[1] = make_float4(fmaf(dlt.x, dlt.x, acc[1].x), fmaf(dlt.y, dlt.y, acc[1].y),
                                 fmaf(dlt.z, dlt.z, acc[1].z), fmaf(dlt.w, dlt.w, acc[1].w));
        }
    }
    block_reduce_store<D, 2>(acc, sred, partial, lane, wave, q);
}

// ------------------------------------------------------------------ forward
template <int D>
__global__ __launch_bounds__(PBLOCK) void forward_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ pos_src,
                                                         const float* __restrict__ pos_tgt,
                                                         const int32_t* __restrict__ idx, int K,
                                                         int64_t m_tgt, const float* __restrict__ A1,
                                                         const float* __restrict__ b1,
                                                         const float* __restrict__ W2, float slope,
                                                         const float* __restrict__ a2,
                                                         const float* __restrict__ b2,
                                                         float* __restrict__ out) {
    constexpr int EB = PC<D>::EB;
    __shared__ float4 s_w2t[PC<D>::W2T_F4];
    __shared__ float4 s_scr[PC<D>::SCR_SIZE];
    int lane, wave, q;
    const Row rw = my_row<D>(m_tgt, lane, wave, q);
    EdgeMLP<D> mlp;
    mlp.init(A1, b1, W2, s_w2t, lane, q, slope);
    __syncthreads();
    const float4 sa = ld4(a2 + 4 * q), sb = ld4(b2 + 4 * q);
    const float px = pos_tgt[3 * rw.r], py = pos_tgt[3 * rw.r + 1], pz = pos_tgt[3 * rw.r + 2];
    const int32_t* irow = idx + rw.r * K;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 2
    for (int k0 = (wave % PC<D>::KS) * EB; k0 < K; k0 += PC<D>::KS * EB) {
        float4 h1[EB], h2[EB], xj[EB];
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            const int jj = (k0 + e < K) ? irow[k0 + e] : -1;
            const int64_t j = jj < 0 ? 0 : jj;
            xj[e] = ld4(x + j * D + 4 * q);
            if (jj < 0) xj[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 pre;
            mlp.layer1(px - pos_src[3 * j], py - pos_src[3 * j + 1], pz - pos_src[3 * j + 2], pre, h1[e]);
        }
        mlp.layer2_batch(h1, h2, s_scr);
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            acc.x = fmaf(fmaf(sa.x, h2[e].x, sb.x), xj[e].x, acc.x);
            acc.y = fmaf(fmaf(sa.y, h2[e].y, sb.y), xj[e].y, acc.y);
            acc.z = fmaf(fmaf(sa.z, h2[e].z, sb.z), xj[e].z, acc.z);
            acc.w = fmaf(fmaf(sa.w, h2[e].w, sb.w), xj[e].w, acc.w);
        }
    }
    __shared__ float4 s_ks[PC<D>::KS > 1 ? PWAVES * WAVE : 1];
    acc = ks_sum<D>(acc, s_ks, lane, wave);
    if (rw.valid && (wave % PC<D>::KS == 0)) st4(out + rw.r * D + 4 * q, acc);
}

// ------------------------------------------------------------------ training forward: one edge pass
// BatchNorm-2 is affine per channel, so the convolution separates:
//   out_i[c] = sum_k (a2 h2_k + b2)[c] x_j[c] = a2[c] U_i[c] + (a2[c] shift[c] + b2[c]) V_i[c],
//   U_i = sum_k (h2_k - shift) * x_j,   V_i = sum_k x_j.
// In training a2 / b2 depend on the batch statistics of h2, which needed an edge pass of their own before the
// convolution pass; this kernel produces U, V AND the statistics partials in ONE pass over the edges, and the
// convolution finishes with an elementwise combine.  U and V also carry everything BatchNorm-2's backward
// needs (sum_e g_w = sum_i g_i V_i,  sum_e g_w (h2 - shift) = sum_i g_i U_i): no edge pass there either.
template <int D>
struct UvLds {
    float4 w2t[PC<D>::W2T_F4];
    float4 scr[PC<D>::SCR_SIZE];
    float red[PWAVES * 2 * D];
    float4 ks[PC<D>::KS > 1 ? PWAVES * WAVE : 1];
    double buf[4 * 256], tot[2 * D];
    int flag;
};
// (bid of n_own: this workgroup among the launch's workgroups that run this body -- all of them, or all but the riders of
// uvstats_hosting_kernel)
template <int D>
__device__ __forceinline__ void uvstats_body(const float* __restrict__ x,
                                                         const float* __restrict__ pos_src,
                                                         const float* __restrict__ pos_tgt,
                                                         const int32_t* __restrict__ idx, int K,
                                                         int64_t m_tgt, const float* __restrict__ A1,
                                                         const float* __restrict__ b1,
                                                         const float* __restrict__ W2, float slope,
                                                         const float* __restrict__ mean_rel,
                                                         float* __restrict__ shift_out,
                                                         float* __restrict__ U, float* __restrict__ V,
                                                         float* __restrict__ partial, unsigned* __restrict__ ticket,
                                                         double* __restrict__ stats, UvLds<D>& L, const unsigned bid, const unsigned n_own) {
    constexpr int EB = PC<D>::EB;
    float4* s_w2t = L.w2t;
    float4* s_scr = L.scr;
    float* sred = L.red;
    int lane, wave, q;
    const unsigned xcd = bid & 7u, base = n_own >> 3, rem = n_own & 7u;                 // xcd_block_id() over the n_own workgroups
    const Row rw = my_row_at<D>(m_tgt, xcd * base + (xcd < rem ? xcd : rem) + (bid >> 3), lane, wave, q);
    EdgeMLP<D> mlp;
    mlp.init(A1, b1, W2, s_w2t, lane, q, slope);
    __syncthreads();
    const float4 shift = mlp.h2_of(mean_rel[0], mean_rel[1], mean_rel[2]);
    if (bid == 0 && threadIdx.x < PC<D>::L) st4(shift_out + 4 * q, shift);

    const float px = pos_tgt[3 * rw.r], py = pos_tgt[3 * rw.r + 1], pz = pos_tgt[3 * rw.r + 2];
    const int32_t* irow = idx + rw.r * K;
    float4 acc[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    float4 u = make_float4(0.f, 0.f, 0.f, 0.f), v = u;
    if constexpr (EB == 1 && PC<D>::KS == 1) {
        // narrow layers (d <= 16): UG edges per trip -- their index entries, then ALL their feature / position rows, then the
        // arithmetic.  The two-edges-per-trip form below waited on memory 69 % of its cycles at four waves per SIMD
        // (SQ_WAIT_ANY, scratch/run_pcpmc.sh): sixteen dependent load rounds per point, two rows in flight each.
        constexpr int UG = UV_GROUP;
        for (int k0 = 0; k0 < K; k0 += UG) {
            int jj[UG];
#pragma unroll
            for (int e = 0; e < UG; ++e) jj[e] = (k0 + e < K) ? irow[k0 + e] : -1;
            float4 xj[UG];
            float rx[UG], ry[UG], rz[UG];
#pragma unroll
            for (int e = 0; e < UG; ++e) {
                const int64_t j = jj[e] < 0 ? 0 : jj[e];
                xj[e] = ld4(x + j * D + 4 * q);
                rx[e] = px - pos_src[3 * j]; ry[e] = py - pos_src[3 * j + 1]; rz[e] = pz - pos_src[3 * j + 2];
            }
#pragma unroll
            for (int e = 0; e < UG; ++e) {
                const float live = (rw.valid && jj[e] >= 0) ? 1.f : 0.f;
                const float4 xe = jj[e] < 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : xj[e];
                float4 pre, h1;
                mlp.layer1(rx[e], ry[e], rz[e], pre, h1);
                const float4 h2 = mlp.layer2(h1);
                const float4 dlt = make_float4((h2.x - shift.x) * live, (h2.y - shift.y) * live, (h2.z - shift.z) * live,
                                               (h2.w - shift.w) * live);
                acc[0].x += dlt.x; acc[0].y += dlt.y; acc[0].z += dlt.z; acc[0].w += dlt.w;
                acc[1] = make_float4(fmaf(dlt.x, dlt.x, acc[1].x), fmaf(dlt.y, dlt.y, acc[1].y),
                                     fmaf(dlt.z, dlt.z, acc[1].z), fmaf(dlt.w, dlt.w, acc[1].w));
                u = make_float4(fmaf(dlt.x, xe.x, u.x), fmaf(dlt.y, xe.y, u.y), fmaf(dlt.z, xe.z, u.z), fmaf(dlt.w, xe.w, u.w));
                v.x += xe.x; v.y += xe.y; v.z += xe.z; v.w += xe.w;
            }
        }
    } else
#pragma unroll 2
    for (int k0 = (wave % PC<D>::KS) * EB; k0 < K; k0 += PC<D>::KS * EB) {
        float4 h1[EB], h2[EB], xj[EB];
        float live[EB];
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            const int jj = (k0 + e < K) ? irow[k0 + e] : -1;
            const int64_t j = jj < 0 ? 0 : jj;
            live[e] = (rw.valid && jj >= 0) ? 1.f : 0.f;
            xj[e] = ld4(x + j * D + 4 * q);
            if (jj < 0) xj[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 pre;
            mlp.layer1(px - pos_src[3 * j], py - pos_src[3 * j + 1], pz - pos_src[3 * j + 2], pre, h1[e]);
        }
        mlp.layer2_batch(h1, h2, s_scr);
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            const float4 dlt = make_float4((h2[e].x - shift.x) * live[e], (h2[e].y - shift.y) * live[e],
                                           (h2[e].z - shift.z) * live[e], (h2[e].w - shift.w) * live[e]);
            acc[0].x += dlt.x; acc[0].y += dlt.y; acc[0].z += dlt.z; acc[0].w += dlt.w;
            acc[1] = make_float4(fmaf(dlt.x, dlt.x, acc[1].x), fmaf(dlt.y, dlt.y, acc[1].y),
                                 fmaf(dlt.z, dlt.z, acc[1].z), fmaf(dlt.w, dlt.w, acc[1].w));
            u = make_float4(fmaf(dlt.x, xj[e].x, u.x), fmaf(dlt.y, xj[e].y, u.y), fmaf(dlt.z, xj[e].z, u.z),
                            fmaf(dlt.w, xj[e].w, u.w));
            v.x += xj[e].x; v.y += xj[e].y; v.z += xj[e].z; v.w += xj[e].w;
        }
    }
    float4* s_ks = L.ks;
    u = ks_sum<D>(u, s_ks, lane, wave);
    v = ks_sum<D>(v, s_ks, lane, wave);
    if (rw.valid && (wave % PC<D>::KS == 0)) {
        st4(U + rw.r * D + 4 * q, u);
        st4(V + rw.r * D + 4 * q, v);
    }
    block_reduce_store<D, 2, true>(acc, sred, partial, lane, wave, q, bid, n_own);
    // with a ticket word the last workgroup to finish sums the rows into stats (reduce_partials_kernel's launch otherwise)
    // (the ticket group comes from bid, this workgroup's number among the n_own -- NOT from blockIdx.x: behind the riders of
    // uvstats_hosting_kernel the two differ, and groups counted on blockIdx.x never fill when n_own % LW_GROUPS != 0)
    if (ticket == nullptr || !last_workgroup_of(ticket, n_own, &L.flag, bid)) return;
    sum_partial_rows_f64<256>(make_rsrc(partial, (int)n_own * 2 * D * 4), (int)n_own, 2 * D, L.buf, L.tot);
    if (threadIdx.x < 2 * D) stats[threadIdx.x] = L.tot[threadIdx.x];
}
template <int D>
__global__ __launch_bounds__(PBLOCK, (D <= 16 ? PC_NARROW_WAVES : 1)) void uvstats_kernel(const float* __restrict__ x, const float* __restrict__ pos_src,
                                                         const float* __restrict__ pos_tgt, const int32_t* __restrict__ idx, int K,
                                                         int64_t m_tgt, const float* __restrict__ A1, const float* __restrict__ b1,
                                                         const float* __restrict__ W2, float slope, const float* __restrict__ mean_rel,
                                                         float* __restrict__ shift_out, float* __restrict__ U, float* __restrict__ V,
                                                         float* __restrict__ partial, unsigned* __restrict__ ticket,
                                                         double* __restrict__ stats) {
    __shared__ UvLds<D> L;
    uvstats_body<D>(x, pos_src, pos_tgt, idx, K, m_tgt, A1, b1, W2, slope, mean_rel, shift_out, U, V, partial, ticket, stats, L, blockIdx.x, gridDim.x);
}
// The same launch CARRYING the CRF layers' matrices (crf_matrices_body.hpp): Q = (I + c^T c)^-1, P = I - Q depend on parameters only,
// yet as a launch of their own (one workgroup per layer, a 27 us chain of 64 dependent pivots) they sat in the middle of the forward's
// launch chain.  Here their workgroups are the FIRST n_side of the grid of the network's first PointConv statistics pass (31 us; four
// workgroups per CU are resident, so only the first ones dispatched are certain to start at once), and nothing waits for them.  The two
// bodies SHARE their LDS (the union below): the riders must not cost the host's workgroups an occupancy step.
#define HOST_PAD(n) (((unsigned)(n) + 7u) & ~7u)
template <int D>
__global__ __launch_bounds__(PBLOCK) void uvstats_hosting_kernel(const float* __restrict__ x, const float* __restrict__ pos_src,
                                                                 const float* __restrict__ pos_tgt, const int32_t* __restrict__ idx, int K,
                                                                 int64_t m_tgt, const float* __restrict__ A1, const float* __restrict__ b1,
                                                                 const float* __restrict__ W2, float slope, const float* __restrict__ mean_rel,
                                                                 float* __restrict__ shift_out, float* __restrict__ U, float* __restrict__ V,
                                                                 float* __restrict__ partial, unsigned* __restrict__ ticket,
                                                                 double* __restrict__ stats, const CrfMatJobs side, const int n_side) {
    static_assert(PBLOCK == CMF_BLOCK, "the riders are workgroups of the host's size");
    __shared__ __attribute__((aligned(16))) union Both { UvLds<D> uv; char mats[CMF_LDS_BYTES]; } L;
    // the riders' slots are padded to a multiple of 8 (HOST_PAD(n_side) workgroups, the surplus ones leave at once): the host's
    // workgroup bid then sits on XCD blockIdx.x % 8 == bid % 8, which is what uvstats_body's XCD-contiguous row order assumes
    const unsigned n_pad = HOST_PAD(n_side);
    if (blockIdx.x < n_pad) {
        const int b = (int)blockIdx.x;
        if (b < n_side) crf_matrices_body(uni(side.c[b]), uni(side.H[b]), uni(side.Q[b]), uni(side.P[b]), L.mats);
        return;
    }
    uvstats_body<D>(x, pos_src, pos_tgt, idx, K, m_tgt, A1, b1, W2, slope, mean_rel, shift_out, U, V, partial, ticket, stats, L.uv,
                    blockIdx.x - n_pad, gridDim.x - n_pad);
}

// out = a2 U + (a2 shift + b2) V   over [m, d] rows (one thread per 4-channel quad).  BatchNorm-2's batch
// coefficients are folded from the statistics right here (a handful of flops per thread, identical in every
// thread of a channel): a2 = gamma rstd, b2 = beta - a2 mean.  Workgroup 0 also publishes a2 / b2 / aux2 = {mean,
// rstd} for the backward kernels and advances the running statistics.
__global__ __launch_bounds__(256) void uv_combine_kernel(const float* __restrict__ U, const float* __restrict__ V,
                                                         const double* __restrict__ stats, const float* __restrict__ shift,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         double n_edges, float* __restrict__ run_mean,
                                                         float* __restrict__ run_var, float momentum, float eps,
                                                         int64_t n4, int d4, float* __restrict__ a2_out,
                                                         float* __restrict__ b2_out, double* __restrict__ aux2,
                                                         float* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int d = 4 * d4;
    const int q = (int)(t % d4);
    float a[4], b[4], sh[4];
    const bool publish = blockIdx.x == 0 && threadIdx.x < d4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = 4 * q + e;
        const double m1 = stats[c] / n_edges;
        const double mean = (double)shift[c] + m1;
        double var = stats[d + c] / n_edges - m1 * m1;
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + (double)eps);
        const double aa = (double)gamma[c] * rstd;
        a[e] = (float)aa;
        b[e] = (float)((double)beta[c] - aa * mean);
        sh[e] = shift[c];
        if (publish) {
            a2_out[c] = a[e];
            b2_out[c] = b[e];
            aux2[c] = mean;
            aux2[d + c] = rstd;
            if (run_mean != nullptr) {
                const double unb = n_edges > 1.0 ? var * (n_edges / (n_edges - 1.0)) : var;
                run_mean[c] = (float)((1.0 - momentum) * run_mean[c] + momentum * mean);
                run_var[c] = (float)((1.0 - momentum) * run_var[c] + momentum * unb);
            }
        }
    }
    if (t >= n4) return;
    const float4 u = ld4(U + 4 * t), v = ld4(V + 4 * t);
    st4(out + 4 * t, make_float4(fmaf(a[0], u.x, fmaf(a[0], sh[0], b[0]) * v.x), fmaf(a[1], u.y, fmaf(a[1], sh[1], b[1]) * v.y),
                                 fmaf(a[2], u.z, fmaf(a[2], sh[2], b[2]) * v.z), fmaf(a[3], u.w, fmaf(a[3], sh[3], b[3]) * v.w)));
}

// partial[blk][0][d] = sum_i g_i V_i, partial[blk][1][d] = sum_i g_i U_i over the block's row slice
// BatchNorm-2 backward coefficients of channel c from the two sums (crfconv_pointconv_fold2_bwd's arithmetic)
__device__ __forceinline__ void fold2_bwd_coef(int c, int d, double sum_gw, double sum_raw, const float* __restrict__ shift,
                                               const double* __restrict__ aux2, const float* __restrict__ gamma, double n_edges,
                                               int use_batch, float* __restrict__ ca, float* __restrict__ cb, float* __restrict__ cc,
                                               float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const double mean = aux2[c], rstd = aux2[d + c], g = gamma[c];
    const double sum_gwh = rstd * (sum_raw - (mean - (double)shift[c]) * sum_gw);    // sum g_w * hhat
    dgamma[c] = (float)sum_gwh;
    dbeta[c] = (float)sum_gw;
    ca[c] = (float)(g * rstd);
    if (use_batch) {
        const double mgw = sum_gw / n_edges, mgh = sum_gwh / n_edges;
        cb[c] = (float)(-g * rstd * rstd * mgh);
        cc[c] = (float)(-g * rstd * mgw + g * rstd * rstd * mean * mgh);
    } else {
        cb[c] = 0.f;
        cc[c] = 0.f;
    }
}

// With a ticket word the LAST workgroup to finish adds the partial rows and derives the coefficients itself (gridsync.hpp:
// fold2_bwd_partials_kernel's launch disappears); ticket == nullptr leaves the rows for that kernel.
struct UvRed {                      // arguments of the BatchNorm-2 backward reduction of one PointConv layer (uv_bwd_reduce_body)
    const float* g; const float* U; const float* V;
    long long m;
    int d, nblk;
    float* partial;
    unsigned* ticket;
    const float* shift; const double* aux2; const float* gamma;
    double n_edges;
    int use_batch;
    float* ca; float* cb; float* cc; float* dgamma; float* dbeta;
};
// bid of r.nblk workgroups of 256 threads.  (2 d (256 / (d / 4)) = 2048 floats of LDS whatever d is.)
__device__ __forceinline__ void uv_bwd_reduce_body(const UvRed& r, const unsigned bid) {
    __shared__ float s_uv[2048];                     // [rows per trip][2][d]
    __shared__ double s_buf[4 * 256], s_tot[256];
    __shared__ int s_flag;
    const float* __restrict__ g = r.g; const float* __restrict__ U = r.U; const float* __restrict__ V = r.V;
    const int64_t m = r.m;
    const int d = r.d;
    const int d4 = d >> 2, rpi = 256 / d4;
    const int q = threadIdx.x % d4, rl = threadIdx.x / d4;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    const int64_t stride = (int64_t)r.nblk * rpi;
    auto add = [&](const float4 gg, const float4 uu, const float4 vv) {
        s1 = make_float4(fmaf(gg.x, vv.x, s1.x), fmaf(gg.y, vv.y, s1.y), fmaf(gg.z, vv.z, s1.z), fmaf(gg.w, vv.w, s1.w));
        s2 = make_float4(fmaf(gg.x, uu.x, s2.x), fmaf(gg.y, uu.y, s2.y), fmaf(gg.z, uu.z, s2.z), fmaf(gg.w, uu.w, s2.w));
    };
    int64_t row = (int64_t)bid * rpi + rl;
    for (; row + stride < m; row += 2 * stride) {        // two rows (six 16-byte loads) in flight per thread
        const int64_t o0 = row * d + 4 * q, o1 = (row + stride) * d + 4 * q;
        const float4 g0 = ld4(g + o0), u0 = ld4(U + o0), v0 = ld4(V + o0);
        const float4 g1 = ld4(g + o1), u1 = ld4(U + o1), v1 = ld4(V + o1);
        add(g0, u0, v0);
        add(g1, u1, v1);
    }
    for (; row < m; row += stride) {
        const int64_t o0 = row * d + 4 * q;
        add(ld4(g + o0), ld4(U + o0), ld4(V + o0));
    }
    st4(s_uv + (rl * 2 + 0) * d + 4 * q, s1);
    st4(s_uv + (rl * 2 + 1) * d + 4 * q, s2);
    __syncthreads();
    const __amdgpu_buffer_rsrc_t pr = make_rsrc(r.partial, r.nblk * 2 * d * 4);
    for (int t = threadIdx.x; t < 2 * d; t += 256) {
        float a = 0.f;
        for (int i = 0; i < rpi; ++i) a += s_uv[i * 2 * d + t];
        st1_sc1(pr, ((int)bid * 2 * d + t) * 4, a);
    }
    if (r.ticket == nullptr || !last_workgroup_of(r.ticket, (unsigned)r.nblk, &s_flag, bid)) return;
    sum_partial_rows_f64<256>(pr, r.nblk, 2 * d, s_buf, s_tot);
    if ((int)threadIdx.x < d)
        fold2_bwd_coef(threadIdx.x, d, s_tot[threadIdx.x], s_tot[d + threadIdx.x], r.shift, r.aux2, r.gamma, r.n_edges, r.use_batch, r.ca, r.cb, r.cc,
                       r.dgamma, r.dbeta);
}
__global__ __launch_bounds__(256) void uv_bwd_reduce_kernel(const UvRed r) { uv_bwd_reduce_body(r, blockIdx.x); }

// ------------------------------------------------------------------ backward pass 1: reductions
template <int D>
__global__ __launch_bounds__(PBLOCK) void bwd_reduce_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ gout,
                                                            const float* __restrict__ pos_src,
                                                            const float* __restrict__ pos_tgt,
                                                            const int32_t* __restrict__ idx, int K,
                                                            int64_t m_tgt, const float* __restrict__ A1,
                                                            const float* __restrict__ b1,
                                                            const float* __restrict__ W2, float slope,
                                                            const float* __restrict__ shift_p,
                                                            float* __restrict__ partial) {
    constexpr int EB = PC<D>::EB;
    __shared__ float4 s_w2t[PC<D>::W2T_F4];
    __shared__ float4 s_scr[PC<D>::SCR_SIZE];
    __shared__ float sred[PWAVES * 2 * D];
    int lane, wave, q;
    const Row rw = my_row<D>(m_tgt, lane, wave, q);
    EdgeMLP<D> mlp;
    mlp.init(A1, b1, W2, s_w2t, lane, q, slope);
    __syncthreads();
    const float4 shift = ld4(shift_p + 4 * q);
    const float px = pos_tgt[3 * rw.r], py = pos_tgt[3 * rw.r + 1], pz = pos_tgt[3 * rw.r + 2];
    const int32_t* irow = idx + rw.r * K;
    float4 g = ld4(gout + rw.r * D + 4 * q);
    if (!rw.valid) g = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 acc[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    for (int k0 = (wave % PC<D>::KS) * EB; k0 < K; k0 += PC<D>::KS * EB) {
        float4 h1[EB], h2[EB], xj[EB];
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            const int jj = (k0 + e < K) ? irow[k0 + e] : -1;
            const int64_t j = jj < 0 ? 0 : jj;
            xj[e] = ld4(x + j * D + 4 * q);
            if (jj < 0) xj[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 pre;
            mlp.layer1(px - pos_src[3 * j], py - pos_src[3 * j + 1], pz - pos_src[3 * j + 2], pre, h1[e]);
        }
        mlp.layer2_batch(h1, h2, s_scr);
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            const float4 gw = make_float4(g.x * xj[e].x, g.y * xj[e].y, g.z * xj[e].z, g.w * xj[e].w);
            acc[0].x += gw.x; acc[0].y += gw.y; acc[0].z += gw.z; acc[0].w += gw.w;
            acc[1] = make_float4(fmaf(gw.x, h2[e].x - shift.x, acc[1].x), fmaf(gw.y, h2[e].y - shift.y, acc[1].y),
                                 fmaf(gw.z, h2[e].z - shift.z, acc[1].z), fmaf(gw.w, h2[e].w - shift.w, acc[1].w));
        }
    }
    block_reduce_store<D, 2>(acc, sred, partial, lane, wave, q);
}

// ------------------------------------------------------------------ backward pass 2: parameters
// Block partials: float [dW2 D*D] and double [channel][dA1 x, y, z, db1].
template <int D>
__global__ __launch_bounds__(PBLOCK) void bwd_params_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ gout,
                                                            const float* __restrict__ pos_src,
                                                            const float* __restrict__ pos_tgt,
                                                            const int32_t* __restrict__ idx, int K,
                                                            int64_t m_tgt, const float* __restrict__ A1,
                                                            const float* __restrict__ b1,
                                                            const float* __restrict__ W2, float slope,
                                                            const float* __restrict__ ca,
                                                            const float* __restrict__ cb,
                                                            const float* __restrict__ cc,
                                                            float* __restrict__ partial,
                                                            double* __restrict__ partial_d) {
    constexpr int L = PC<D>::L;
    // dW2 = sum_e g_h2^T h1 -- a reduction over edges of d x d outer products: for d in {8, 16} on the MATRIX pipe (OuterAcc:
    // the wave's g_h2 / h1 rows of one edge per point through a per-wave LDS tile that IS the 16x16x4 fragment layout, four
    // MFMAs per edge round), which takes d FMAs + d/4 broadcasts per lane and edge off the vector ALU and 4 d accumulator
    // registers off the wave (d = 16: 256 -> fewer than 200 registers); other widths keep registers / LDS atomics.
    constexpr bool ACC_MFMA = (D == 8 || D == 16);
    constexpr bool ACC_REGS = (D <= 32) && !ACC_MFMA;          // dW2 accumulators in registers vs LDS atomics
    constexpr int NSLOT = D * D;
    __shared__ __attribute__((aligned(16))) float s_otile[ACC_MFMA ? 2 * PWAVES * 256 : 4];
    __shared__ float s_ored[ACC_MFMA ? PWAVES * D * D : 1];
    [[maybe_unused]] OuterAcc<ACC_MFMA ? D : 8, PBLOCK> oa;
    __shared__ double s_accd[PWAVES][4 * D];
    __shared__ float4 s_w2t[PC<D>::W2T_F4];
    constexpr bool W2_LDS = (D <= 64);            // d = 128: the 64 KB of rows stay in L1/L2 instead
    __shared__ float4 s_w2[W2_LDS ? D * L : 1];   // W2 rows as float4: s_w2[c * L + q'] = W2[c][4q'..]
    __shared__ float s_acc[NSLOT];                // block totals
    int lane, wave, q;
    const Row rw = my_row<D>(m_tgt, lane, wave, q);
    EdgeMLP<D> mlp;
    mlp.init(A1, b1, W2, s_w2t, lane, q, slope);
    {
        float* s = reinterpret_cast<float*>(s_w2);
        if constexpr (W2_LDS)
            for (int t = threadIdx.x; t < D * D; t += PBLOCK) s[t] = W2[t];
        for (int t = threadIdx.x; t < NSLOT; t += PBLOCK) s_acc[t] = 0.f;
    }
    __syncthreads();
    const float4 va = ld4(ca + 4 * q), vb = ld4(cb + 4 * q), vc = ld4(cc + 4 * q);
    const float px = pos_tgt[3 * rw.r], py = pos_tgt[3 * rw.r + 1], pz = pos_tgt[3 * rw.r + 2];
    const int32_t* irow = idx + rw.r * K;
    float4 g = ld4(gout + rw.r * D + 4 * q);

    float4 dw2[ACC_REGS ? D : 1];  // dw2[c'] = {dW2[4q+0][c'], dW2[4q+1][c'], dW2[4q+2][c'], dW2[4q+3][c']}
    if constexpr (ACC_REGS) {
#pragma unroll
        for (int c = 0; c < D; ++c) dw2[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // dA1 / db1 feed the analytic BatchNorm-1 backward on the host, where their large common
    // components cancel (scale / shift invariance): accumulate them in float64 end to end.
    // Within ONE point (the K edges of this loop) they are summed in float32 -- K terms of like magnitude, relative error
    // ~1e-7 per point, random across points -- and only the sums over points run in float64: the float64 FMAs of the
    // per-edge form were a third of this kernel's vector work.
    float da1[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};   // [axis][channel of the quad]
    float db[4] = {0, 0, 0, 0};
    const int base = lane - q;

    // PB edges per trip: their index entries first, then all their feature / position rows, then the arithmetic -- two
    // dependent memory phases per PB edges instead of two per edge (one or two wavefronts per SIMD hide nothing)
    constexpr int PB = (D <= 8) ? PARAMS_PB_NARROW : PARAMS_PB;
    for (int k0 = (wave % PC<D>::KS) * PB; k0 < K; k0 += PC<D>::KS * PB) {
        int jj_[PB];
#pragma unroll
        for (int u = 0; u < PB; ++u) jj_[u] = k0 + u < K ? irow[k0 + u] : -1;
        float4 xj_[PB];
        float rx_[PB], ry_[PB], rz_[PB];
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int64_t j = jj_[u] < 0 ? 0 : jj_[u];
            xj_[u] = ld4(x + j * D + 4 * q);
            rx_[u] = px - pos_src[3 * j]; ry_[u] = py - pos_src[3 * j + 1]; rz_[u] = pz - pos_src[3 * j + 2];
        }
#pragma unroll 1
    for (int u = 0; u < PB; ++u) {                  // rolled: the body exists once (unrolled it took 256 + 256 registers);
        const int jj = jj_[0];                      // slot 0 is the current edge, the slots rotate at the end
        const float live = (rw.valid && jj >= 0) ? 1.f : 0.f;
        const float4 xj = xj_[0];
        const float rx = rx_[0], ry = ry_[0], rz = rz_[0];
        float4 pre, h1;
        mlp.layer1(rx, ry, rz, pre, h1);
        const float4 h2 = mlp.layer2(h1);
        // g_h2 for this lane's quad (zero for padding rows and missing neighbours)
        float4 gh2;
        gh2.x = live * fmaf(va.x, g.x * xj.x, fmaf(vb.x, h2.x, vc.x));
        gh2.y = live * fmaf(va.y, g.y * xj.y, fmaf(vb.y, h2.y, vc.y));
        gh2.z = live * fmaf(va.z, g.z * xj.z, fmaf(vb.z, h2.z, vc.z));
        gh2.w = live * fmaf(va.w, g.w * xj.w, fmaf(vb.w, h2.w, vc.w));
        // dW2[quad][c'] += g_h2[quad] * h1[c'] ;  g_h1[quad'] = sum_c g_h2[c] W2[c][quad']
        if constexpr (ACC_MFMA) {
            float* ta = s_otile + wave * 512;
            oa.add_rows(gh2, h1, ta, ta + 256, lane);          // g_h2 is zero for padding rows and missing neighbours
        }
        float4 gh1 = make_float4(0.f, 0.f, 0.f, 0.f);
        static_for<L>([&](auto HQ) {
            constexpr int hq = decltype(HQ)::value;
            const float h0 = group_bcast<L, hq>(h1.x, base), h1b = group_bcast<L, hq>(h1.y, base);
            const float h2b = group_bcast<L, hq>(h1.z, base), h3 = group_bcast<L, hq>(h1.w, base);
            if constexpr (ACC_MFMA) {
                (void)h0; (void)h1b; (void)h2b; (void)h3;                  // (the outer product runs once per edge below)
            } else if constexpr (ACC_REGS) {
                dw2[4 * hq + 0] = fma4(h0, gh2, dw2[4 * hq + 0]);
                dw2[4 * hq + 1] = fma4(h1b, gh2, dw2[4 * hq + 1]);
                dw2[4 * hq + 2] = fma4(h2b, gh2, dw2[4 * hq + 2]);
                dw2[4 * hq + 3] = fma4(h3, gh2, dw2[4 * hq + 3]);
            } else {
                const float hv[4] = {h0, h1b, h2b, h3};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int cp = 4 * hq + c;
                    atomicAdd(&s_acc[(4 * q + 0) * D + cp], gh2.x * hv[c]);
                    atomicAdd(&s_acc[(4 * q + 1) * D + cp], gh2.y * hv[c]);
                    atomicAdd(&s_acc[(4 * q + 2) * D + cp], gh2.z * hv[c]);
                    atomicAdd(&s_acc[(4 * q + 3) * D + cp], gh2.w * hv[c]);
                }
            }
            const float g0 = group_bcast<L, hq>(gh2.x, base), g1 = group_bcast<L, hq>(gh2.y, base);
            const float g2 = group_bcast<L, hq>(gh2.z, base), g3 = group_bcast<L, hq>(gh2.w, base);
            if constexpr (W2_LDS) {
                gh1 = fma4(g0, s_w2[(4 * hq + 0) * L + q], gh1);
                gh1 = fma4(g1, s_w2[(4 * hq + 1) * L + q], gh1);
                gh1 = fma4(g2, s_w2[(4 * hq + 2) * L + q], gh1);
                gh1 = fma4(g3, s_w2[(4 * hq + 3) * L + q], gh1);
            } else {
                gh1 = fma4(g0, ld4(W2 + (4 * hq + 0) * D + 4 * q), gh1);
                gh1 = fma4(g1, ld4(W2 + (4 * hq + 1) * D + 4 * q), gh1);
                gh1 = fma4(g2, ld4(W2 + (4 * hq + 2) * D + 4 * q), gh1);
                gh1 = fma4(g3, ld4(W2 + (4 * hq + 3) * D + 4 * q), gh1);
            }
        });
        // through lrelu(0.1)
        const float4 gp = make_float4(gh1.x * (pre.x > 0.f ? 1.f : slope), gh1.y * (pre.y > 0.f ? 1.f : slope),
                                      gh1.z * (pre.z > 0.f ? 1.f : slope), gh1.w * (pre.w > 0.f ? 1.f : slope));
        const float gpd[4] = {gp.x, gp.y, gp.z, gp.w};
        const float rd[3] = {rx, ry, rz};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) da1[ax][c] = fmaf(rd[ax], gpd[c], da1[ax][c]);
            db[c] += gpd[c];
        }
#pragma unroll
        for (int v = 0; v + 1 < PB; ++v) {
            jj_[v] = jj_[v + 1]; xj_[v] = xj_[v + 1];
            rx_[v] = rx_[v + 1]; ry_[v] = ry_[v + 1]; rz_[v] = rz_[v + 1];
        }
    }
    }

    // wave-level sums over points, then one LDS add per wave (4 adders per slot, fixed slots)
    if constexpr (ACC_REGS) {
#pragma unroll
        for (int cp = 0; cp < D; ++cp) {
            float4 t = dw2[cp];
            t.x = over_points<D>(t.x); t.y = over_points<D>(t.y);
            t.z = over_points<D>(t.z); t.w = over_points<D>(t.w);
            if (lane < L) {
                atomicAdd(&s_acc[(4 * q + 0) * D + cp], t.x);
                atomicAdd(&s_acc[(4 * q + 1) * D + cp], t.y);
                atomicAdd(&s_acc[(4 * q + 2) * D + cp], t.z);
                atomicAdd(&s_acc[(4 * q + 3) * D + cp], t.w);
            }
        }
    }
    // float64 part: shuffle tree over the wave's points, per-wave LDS slots, fixed-order block sum
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int ax = 0; ax < 4; ++ax) {                       // ax == 3 -> db1
            double t = (double)(ax < 3 ? da1[ax < 3 ? ax : 0][c] : db[c]);
#pragma unroll
            for (int o = L; o < WAVE; o <<= 1) t += __shfl_xor(t, o, WAVE);
            if (lane < L) s_accd[wave][(4 * q + c) * 4 + ax] = t;
        }
    }
    __syncthreads();
    if constexpr (ACC_MFMA) {
        oa.store_partial(s_ored, partial, lane);               // block sum of the waves' accumulators -> partial[block][d * d]
    } else {
        for (int t = threadIdx.x; t < D * D; t += PBLOCK) partial[(int64_t)blockIdx.x * D * D + t] = s_acc[t];
    }
    for (int t = threadIdx.x; t < 4 * D; t += PBLOCK) {
        double a = 0.0;
#pragma unroll
        for (int w = 0; w < PWAVES; ++w) a += s_accd[w][t];
        partial_d[(int64_t)blockIdx.x * 4 * D + t] = a;        // [channel][x, y, z, bias]
    }
}

// ------------------------------------------------------------------ backward pass 2 for wide layers
// d >= 64 levels have few edges (<= 41k at config 2): write h1, g_h2 and rel per edge and let the
// host contract them with dense GEMMs (dW2 = g_h2^T h1 etc.) instead of reducing d*d sums in-kernel.
template <int D>
__device__ __forceinline__ void bwd_dump_body(const float* __restrict__ x,
                                                          const float* __restrict__ gout,
                                                          const float* __restrict__ pos_src,
                                                          const float* __restrict__ pos_tgt,
                                                          const int32_t* __restrict__ idx, int K,
                                                          int64_t m_tgt, const float* __restrict__ A1,
                                                          const float* __restrict__ b1,
                                                          const float* __restrict__ W2, float slope,
                                                          const float* __restrict__ ca,
                                                          const float* __restrict__ cb,
                                                          const float* __restrict__ cc,
                                                          float* __restrict__ h1_out,
                                                          float* __restrict__ gh2_out,
                                                          float* __restrict__ rel_out, unsigned block) {
    constexpr int EB = PC<D>::EB;
    __shared__ float4 s_w2t[PC<D>::W2T_F4];
    __shared__ float4 s_scr[PC<D>::SCR_SIZE];
    int lane, wave, q;
    const Row rw = my_row_at<D>(m_tgt, block, lane, wave, q);
    EdgeMLP<D> mlp;
    mlp.init(A1, b1, W2, s_w2t, lane, q, slope);
    __syncthreads();
    const float4 va = ld4(ca + 4 * q), vb = ld4(cb + 4 * q), vc = ld4(cc + 4 * q);
    const float px = pos_tgt[3 * rw.r], py = pos_tgt[3 * rw.r + 1], pz = pos_tgt[3 * rw.r + 2];
    const int32_t* irow = idx + rw.r * K;
    const float4 g = ld4(gout + rw.r * D + 4 * q);
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k0 = (wave % PC<D>::KS) * EB; k0 < K; k0 += PC<D>::KS * EB) {
        float4 h1[EB], h2[EB], xj[EB];
        float rx[EB], ry[EB], rz[EB];
        bool have[EB];
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            const int jj = (k0 + e < K) ? irow[k0 + e] : -1;
            have[e] = jj >= 0;
            const int64_t j = have[e] ? jj : 0;
            xj[e] = ld4(x + j * D + 4 * q);
            rx[e] = px - pos_src[3 * j]; ry[e] = py - pos_src[3 * j + 1]; rz[e] = pz - pos_src[3 * j + 2];
            float4 pre;
            mlp.layer1(rx[e], ry[e], rz[e], pre, h1[e]);
        }
        mlp.layer2_batch(h1, h2, s_scr);
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            if (k0 + e >= K || !rw.valid) continue;
            float4 gh2;
            gh2.x = fmaf(va.x, g.x * xj[e].x, fmaf(vb.x, h2[e].x, vc.x));
            gh2.y = fmaf(va.y, g.y * xj[e].y, fmaf(vb.y, h2[e].y, vc.y));
            gh2.z = fmaf(va.z, g.z * xj[e].z, fmaf(vb.z, h2[e].z, vc.z));
            gh2.w = fmaf(va.w, g.w * xj[e].w, fmaf(vb.w, h2[e].w, vc.w));
            const int64_t eid = rw.r * K + k0 + e;
            st4(h1_out + eid * D + 4 * q, have[e] ? h1[e] : zero);
            st4(gh2_out + eid * D + 4 * q, have[e] ? gh2 : zero);
            if (q == 0) {
                rel_out[3 * eid] = have[e] ? rx[e] : 0.f;
                rel_out[3 * eid + 1] = have[e] ? ry[e] : 0.f;
                rel_out[3 * eid + 2] = have[e] ? rz[e] : 0.f;
            }
        }
    }
}

template <int D>
__global__ __launch_bounds__(PBLOCK) void bwd_dump_kernel(const float* __restrict__ x, const float* __restrict__ gout,
                                                          const float* __restrict__ pos_src, const float* __restrict__ pos_tgt,
                                                          const int32_t* __restrict__ idx, int K, int64_t m_tgt,
                                                          const float* __restrict__ A1, const float* __restrict__ b1,
                                                          const float* __restrict__ W2, float slope, const float* __restrict__ ca,
                                                          const float* __restrict__ cb, const float* __restrict__ cc,
                                                          float* __restrict__ h1_out, float* __restrict__ gh2_out,
                                                          float* __restrict__ rel_out) {
    bwd_dump_body<D>(x, gout, pos_src, pos_tgt, idx, K, m_tgt, A1, b1, W2, slope, ca, cb, cc, h1_out, gh2_out, rel_out, xcd_block_id());
}

// The dump pass of SEVERAL layers of the same width in one launch (the two ResNet blocks of a coarse level: a few workgroups
// each, nothing on the backward chain waits for them -- crfconv_pointconv_wide_params_jobs).
constexpr int PJ_MAX = 8;
struct DumpJobs {
    const float* x[PJ_MAX]; const float* gout[PJ_MAX]; const float* pos_src[PJ_MAX]; const float* pos_tgt[PJ_MAX];
    const int32_t* idx[PJ_MAX];
    const float* A1[PJ_MAX]; const float* b1[PJ_MAX]; const float* W2[PJ_MAX]; const float* ca[PJ_MAX]; const float* cb[PJ_MAX];
    const float* cc[PJ_MAX];
    float* h1[PJ_MAX]; float* gh2[PJ_MAX]; float* rel[PJ_MAX];
    int K[PJ_MAX], m_tgt[PJ_MAX];
    float slope[PJ_MAX];
    int blk_base[PJ_MAX + 1];
    int njobs;
};
template <int D>
__global__ __launch_bounds__(PBLOCK) void bwd_dump_jobs_kernel(const DumpJobs t) {
    int j = 0;
    while (j + 1 < t.njobs && t.blk_base[j + 1] <= (int)blockIdx.x) ++j;
    bwd_dump_body<D>(t.x[j], t.gout[j], t.pos_src[j], t.pos_tgt[j], t.idx[j], t.K[j], t.m_tgt[j], t.A1[j], t.b1[j], t.W2[j], t.slope[j],
                     t.ca[j], t.cb[j], t.cc[j], t.h1[j], t.gh2[j], t.rel[j], blockIdx.x - (unsigned)t.blk_base[j]);
}

// ------------------------------------------------------------------ backward pass 2b for wide layers
// dA1 | db1 of the dumped path: gp = (g_h2 W2) * lrelu'(h1) per edge, then the four float64 column sums
// sum_e gp[e,c] * {rel_x, rel_y, rel_z, 1}.  gw = g_h2 W2 arrives from a dense GEMM; this is one streaming
// pass over gw / h1 / rel (the float64 elementwise + reduction launches it replaces cost 0.6 ms a step).
template <int D>
__device__ __forceinline__ void a1_reduce_body(const float* __restrict__ gw,
                                                        const float* __restrict__ h1,
                                                        const float* __restrict__ rel, int64_t E, float slope,
                                                        double* __restrict__ partial_d, unsigned block, unsigned nblock) {
    constexpr int L = D / 4, RPB = 256 / L;
    __shared__ double s_acc[256 * 16];
    const int q = threadIdx.x % L, rl = threadIdx.x / L;
    double acc[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[c][t] = 0.0;
    for (int64_t r = (int64_t)block * RPB + rl; r < E; r += (int64_t)nblock * RPB) {
        const float4 a = ld4(gw + r * D + 4 * q), h = ld4(h1 + r * D + 4 * q);
        const double rx = rel[3 * r], ry = rel[3 * r + 1], rz = rel[3 * r + 2];
        const float gp[4] = {a.x * (h.x > 0.f ? 1.f : slope), a.y * (h.y > 0.f ? 1.f : slope),
                             a.z * (h.z > 0.f ? 1.f : slope), a.w * (h.w > 0.f ? 1.f : slope)};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double v = gp[c];
            acc[c][0] += v * rx; acc[c][1] += v * ry; acc[c][2] += v * rz; acc[c][3] += v;
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t) s_acc[rl * (4 * D) + (4 * q + c) * 4 + t] = acc[c][t];
    __syncthreads();
    for (int t = threadIdx.x; t < 4 * D; t += 256) {
        double a = 0.0;
        for (int i = 0; i < RPB; ++i) a += s_acc[i * (4 * D) + t];
        partial_d[(int64_t)block * 4 * D + t] = a;        // [channel][x, y, z, bias]
    }
}

template <int D>
__global__ __launch_bounds__(256) void a1_reduce_kernel(const float* __restrict__ gw, const float* __restrict__ h1,
                                                        const float* __restrict__ rel, int64_t E, float slope,
                                                        double* __restrict__ partial_d) {
    a1_reduce_body<D>(gw, h1, rel, E, slope, partial_d, blockIdx.x, gridDim.x);
}
struct A1Jobs {
    const float* gw[PJ_MAX]; const float* h1[PJ_MAX]; const float* rel[PJ_MAX];
    double* partial_d[PJ_MAX];
    int E[PJ_MAX], nblk[PJ_MAX];
    float slope[PJ_MAX];
    int blk_base[PJ_MAX + 1];
    int njobs;
};
template <int D>
__global__ __launch_bounds__(256) void a1_reduce_jobs_kernel(const A1Jobs t) {
    int j = 0;
    while (j + 1 < t.njobs && t.blk_base[j + 1] <= (int)blockIdx.x) ++j;
    a1_reduce_body<D>(t.gw[j], t.h1[j], t.rel[j], t.E[j], t.slope[j], t.partial_d[j], blockIdx.x - (unsigned)t.blk_base[j], (unsigned)t.nblk[j]);
}

// ------------------------------------------------------------------ backward: input features
template <int D>
__global__ __launch_bounds__(PBLOCK) void bwd_input_kernel(const float* __restrict__ gout,
                                                           const float* __restrict__ pos_src,
                                                           const float* __restrict__ pos_tgt,
                                                           const int32_t* __restrict__ rev_ptr,
                                                           const int32_t* __restrict__ rev_eid, int K,
                                                           int64_t m_src, const float* __restrict__ A1,
                                                           const float* __restrict__ b1,
                                                           const float* __restrict__ W2, float slope,
                                                           const float* __restrict__ a2,
                                                           const float* __restrict__ b2,
                                                           float* __restrict__ dx, const UvRed red, const int n_own) {
    // round 5: the launch also HOSTS the layer's BatchNorm-2 backward reduction (the workgroups behind the first n_own; red.nblk = 0:
    // none): the two are independent -- dx needs the forward coefficients only -- and the reduction was a launch of its own in front
    // (ten ~8 us launches per step, most of them on grids of a few dozen workgroups)
    if ((int)blockIdx.x >= n_own) {
        uv_bwd_reduce_body(red, blockIdx.x - (unsigned)n_own);
        return;
    }
    constexpr int EB = PC<D>::EB;
    __shared__ float4 s_w2t[PC<D>::W2T_F4];
    __shared__ float4 s_scr[PC<D>::SCR_SIZE];
    int lane, wave, q;
    const unsigned nb = (unsigned)n_own, bb = blockIdx.x;                  // xcd_block_id() of the first n_own workgroups
    const unsigned xcd = bb & 7u, within = bb >> 3, base = nb >> 3, rem = nb & 7u;
    const Row rw = my_row_at<D>(m_src, xcd * base + (xcd < rem ? xcd : rem) + within, lane, wave, q);
    EdgeMLP<D> mlp;
    mlp.init(A1, b1, W2, s_w2t, lane, q, slope);
    __syncthreads();
    const float4 sa = ld4(a2 + 4 * q), sb = ld4(b2 + 4 * q);
    const float sx = pos_src[3 * rw.r], sy = pos_src[3 * rw.r + 1], sz = pos_src[3 * rw.r + 2];
    const int beg = rev_ptr[rw.r];
    const int deg = rw.valid ? rev_ptr[rw.r + 1] - beg : 0;
    int degmax = deg;  // uniform trip count: the MLP exchange needs every lane of a group active
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) degmax = max(degmax, __shfl_xor(degmax, o, WAVE));
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // UB edges per trip: their edge ids first, then all their position / gradient rows -- two dependent memory phases per
    // UB edges instead of two per edge (a wavefront iterates to the largest in-degree of its rows, ~35 trips of one edge)
    constexpr int UB = EB > 1 ? EB : 4;
    const int kshift = (K & (K - 1)) == 0 ? __ffs(K) - 1 : -1;
    for (int p0 = (wave % PC<D>::KS) * UB; p0 < degmax; p0 += PC<D>::KS * UB) {
        int eid[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) eid[u] = p0 + u < deg ? rev_eid[beg + p0 + u] : -1;
        float rxs[UB], rys[UB], rzs[UB];
        float4 gs[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            rxs[u] = rys[u] = rzs[u] = 0.f;
            gs[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (eid[u] >= 0) {
                const int64_t i = kshift >= 0 ? (eid[u] >> kshift) : (eid[u] / K);
                rxs[u] = pos_tgt[3 * i] - sx;
                rys[u] = pos_tgt[3 * i + 1] - sy;
                rzs[u] = pos_tgt[3 * i + 2] - sz;
                gs[u] = ld4(gout + i * D + 4 * q);
            }
        }
#pragma unroll
        for (int ub = 0; ub < UB; ub += EB) {
        float4 h1[EB], h2[EB], g[EB];
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            g[e] = gs[ub + e];
            float4 pre;
            mlp.layer1(rxs[ub + e], rys[ub + e], rzs[ub + e], pre, h1[e]);
        }
        mlp.layer2_batch(h1, h2, s_scr);
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            acc.x = fmaf(fmaf(sa.x, h2[e].x, sb.x), g[e].x, acc.x);  // g == 0 on inactive slots
            acc.y = fmaf(fmaf(sa.y, h2[e].y, sb.y), g[e].y, acc.y);
            acc.z = fmaf(fmaf(sa.z, h2[e].z, sb.z), g[e].z, acc.z);
            acc.w = fmaf(fmaf(sa.w, h2[e].w, sb.w), g[e].w, acc.w);
        }
        }
    }
    __shared__ float4 s_ks[PC<D>::KS > 1 ? PWAVES * WAVE : 1];
    acc = ks_sum<D>(acc, s_ks, lane, wave);
    if (rw.valid && (wave % PC<D>::KS == 0)) st4(dx + rw.r * D + 4 * q, acc);
}

static int check_pc(int64_t m, int K, int d) {
    CRF_REQUIRE(m > 0 && m < ((int64_t)1 << 31), CRF_ERR_ARG, "rows=%lld out of range", (long long)m);
    CRF_REQUIRE(K >= 1 && K <= 64, CRF_ERR_ARG, "K=%d out of range", K);
    CRF_REQUIRE(d == 4 || d == 8 || d == 16 || d == 32 || d == 64 || d == 128, CRF_ERR_UNSUPPORTED,
                "d=%d not in {4,8,16,32,64,128}", d);
    return CRF_OK;
}

template <int D>
constexpr int nblocks_of(int64_t m) { return (int)((m + PC<D>::PPB - 1) / PC<D>::PPB); }

#define DISPATCH_D(d, ...)                                        \
    switch (d) {                                                  \
        case 4: { constexpr int DD = 4; __VA_ARGS__; break; }     \
        case 8: { constexpr int DD = 8; __VA_ARGS__; break; }     \
        case 16: { constexpr int DD = 16; __VA_ARGS__; break; }   \
        case 32: { constexpr int DD = 32; __VA_ARGS__; break; }   \
        case 64: { constexpr int DD = 64; __VA_ARGS__; break; }   \
        default: { constexpr int DD = 128; __VA_ARGS__; break; }  \
    }

static int64_t blocks_for(int64_t m, int d) {
    const int ks = d >= 128 ? PC_KS_128 : (d >= 64 ? PC_KS_64 : (d >= 32 ? PC_KS_32 : 1));            // PC<D>::KS
    const int ppb = (WAVE / (d / 4)) * PWAVES / ks;
    return (m + ppb - 1) / ppb;
}

static int reduce_partials(const float* partial, int64_t nblk, int nslots, double* out, hipStream_t st) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)cdiv(nslots, 256 / WAVE)), dim3(256), 0, st, partial,
                       nblk, nslots, out);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

}  // namespace crf

using namespace crf;

extern "C" size_t crfconv_pointconv_workspace(int64_t m_tgt, int K, int d) {
    if (m_tgt <= 0 || d < 4) return 0;
    const size_t a = (size_t)cdiv(m_tgt, 256) * 9;                               // moments
    const size_t b = (size_t)blocks_for(m_tgt, d) * ((size_t)d * d + 8 * (size_t)d);  // params: float d*d + double 4d
    return sizeof(float) * (a > b ? a : b) + 1024;
}

extern "C" int crfconv_pointconv_moments(const float* pos_src, const float* pos_tgt, const int32_t* idx32,
                                         int K, int64_t m_tgt, double* out9, void* workspace,
                                         size_t workspace_bytes, crf_stream_t stream) {
    if (int rc = check_pc(m_tgt, K, 4)) return rc;
    CRF_REQUIRE(pos_src && pos_tgt && idx32 && out9 && workspace, CRF_ERR_ARG, "null pointer");
    const int64_t nblk = cdiv(m_tgt, 256);
    CRF_REQUIRE(workspace_bytes >= sizeof(float) * 9 * (size_t)nblk, CRF_ERR_WORKSPACE, "workspace too small");
    hipStream_t st = as_stream(stream);
    float* partial = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(moments_kernel, dim3((unsigned)nblk), dim3(256), 0, st, pos_src, pos_tgt, idx32, K,
                       m_tgt, partial);
    CRF_LAUNCH_CHECK();
    return reduce_partials(partial, nblk, 9, out9, st);
}

// The nine block-partial sums of moments_kernel -> mean [3], covariance [3, 3], packed {mean, cov} [12] (float64) and the mean in
// float32, in ONE single-workgroup launch: slot sums exactly as reduce_partials_kernel forms them (one wavefront per slot, lane l
// takes blocks l, l + 64, ..., shuffle tree), then mean = S1 / n, cov = S2 / n - mean mean^T with every operation rounded on its
// own (what the chain of framework ops this replaces computed: ~15 launches and, on a refresh, four copies per table).
__device__ __forceinline__ void moments_finish_block(const float* __restrict__ partial, int64_t nblk, double n_edges,
                                                     double* __restrict__ mean, double* __restrict__ cov,
                                                     double* __restrict__ packed, float* __restrict__ mean32) {
    __shared__ double s_sum[9];
    const int slot = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (slot < 9) {
        double a = 0.0;
        for (int64_t b = lane; b < nblk; b += WAVE) a += (double)partial[b * 9 + slot];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, WAVE);
        if (lane == 0) s_sum[slot] = a;
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t >= 12) return;
    double v;
    if (t < 3) {
        v = s_sum[t] / n_edges;
    } else {
        const int a = (t - 3) / 3, b = (t - 3) % 3;
        // second moments arrive as the upper triangle xx xy xz yy yz zz (index arithmetic, not a table: no private segment,
        // which a replayed hipGraph does not survive on ROCm 7.2 -- DESIGN.md 5b)
        const int lo = a < b ? a : b, hi = a < b ? b : a;
        const int tri = lo == 0 ? hi : (lo == 1 ? 2 + hi : 5);
        const double sec = s_sum[3 + tri] / n_edges;
        v = dadd_rn(sec, -dmul_rn(s_sum[a] / n_edges, s_sum[b] / n_edges));
    }
    packed[t] = v;
    if (t < 3) { mean[t] = v; mean32[t] = (float)v; }
    else cov[t - 3] = v;
}
__global__ __launch_bounds__(1024) void moments_finish_kernel(const float* __restrict__ partial, int64_t nblk, double n_edges,
                                                              double* __restrict__ mean, double* __restrict__ cov,
                                                              double* __restrict__ packed, float* __restrict__ mean32) {
    moments_finish_block(partial, nblk, n_edges, mean, cov, packed, mean32);
}
__global__ __launch_bounds__(1024) void moments_finish_batched_kernel(const MomentsBatch t, const float* __restrict__ partial) {
    const int j = blockIdx.x;
    moments_finish_block(partial + (int64_t)t.blk_base[j] * 9, t.blk_base[j + 1] - t.blk_base[j], t.n_edges[j], t.mean[j], t.cov[j],
                         t.packed[j], t.mean32[j]);
}

extern "C" int crfconv_pointconv_moments_packed(const float* pos_src, const float* pos_tgt, const int32_t* idx32, int K,
                                                int64_t m_tgt, double n_edges, double* mean, double* cov, double* packed,
                                                float* mean32, void* workspace, size_t workspace_bytes,
                                                crf_stream_t stream) {
    if (int rc = check_pc(m_tgt, K, 4)) return rc;
    CRF_REQUIRE(pos_src && pos_tgt && idx32 && mean && cov && packed && mean32 && workspace, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(n_edges > 0.0, CRF_ERR_ARG, "n_edges must be positive");
    const int64_t nblk = cdiv(m_tgt, 256);
    CRF_REQUIRE(workspace_bytes >= sizeof(float) * 9 * (size_t)nblk, CRF_ERR_WORKSPACE, "workspace too small");
    hipStream_t st = as_stream(stream);
    float* partial = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(moments_kernel, dim3((unsigned)nblk), dim3(256), 0, st, pos_src, pos_tgt, idx32, K, m_tgt, partial);
    CRF_LAUNCH_CHECK();
    hipLaunchKernelGGL(moments_finish_kernel, dim3(1), dim3(1024), 0, st, partial, nblk, n_edges, mean, cov, packed, mean32);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" size_t crfconv_pointconv_moments_batched_workspace(const crf_moments_job* jobs, int njobs) {
    if (!jobs || njobs < 1) return 0;
    int64_t nblk = 0;
    for (int j = 0; j < njobs; ++j) nblk += cdiv(jobs[j].m_tgt, 256);
    return sizeof(float) * 9 * (size_t)nblk;
}

// crfconv_pointconv_moments_packed for up to 16 tables in two launches: identical outputs (same partials, same order).
extern "C" int crfconv_pointconv_moments_batched(const crf_moments_job* jobs, int njobs, void* workspace, size_t workspace_bytes,
                                                 crf_stream_t stream) {
    CRF_REQUIRE(jobs && workspace, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(njobs >= 1 && njobs <= MB_MAX, CRF_ERR_ARG, "njobs=%d outside [1, %d]", njobs, MB_MAX);
    CRF_REQUIRE(workspace_bytes >= crfconv_pointconv_moments_batched_workspace(jobs, njobs), CRF_ERR_WORKSPACE, "workspace too small");
    MomentsBatch t;
    int64_t nblk = 0;
    for (int j = 0; j <= MB_MAX; ++j) {
        t.blk_base[j] = (int)nblk;
        if (j < njobs) {
            const crf_moments_job& jb = jobs[j];
            if (int rc = check_pc(jb.m_tgt, jb.K, 4)) return rc;
            CRF_REQUIRE(jb.pos_src && jb.pos_tgt && jb.idx32 && jb.mean && jb.cov && jb.packed && jb.mean32, CRF_ERR_ARG,
                        "job %d: null pointer", j);
            CRF_REQUIRE(jb.n_edges > 0.0 && jb.m_tgt < ((int64_t)1 << 31), CRF_ERR_ARG, "job %d: bad size", j);
            t.pos_src[j] = jb.pos_src; t.pos_tgt[j] = jb.pos_tgt; t.idx[j] = jb.idx32; t.mean[j] = jb.mean; t.cov[j] = jb.cov;
            t.packed[j] = jb.packed; t.mean32[j] = jb.mean32; t.n_edges[j] = jb.n_edges; t.K[j] = jb.K; t.m_tgt[j] = (int)jb.m_tgt;
            nblk += cdiv(jb.m_tgt, 256);
            CRF_REQUIRE(nblk < ((int64_t)1 << 31), CRF_ERR_UNSUPPORTED, "batch too large");
        } else if (j < MB_MAX) {
            t.pos_src[j] = nullptr; t.pos_tgt[j] = nullptr; t.idx[j] = nullptr; t.mean[j] = nullptr; t.cov[j] = nullptr;
            t.packed[j] = nullptr; t.mean32[j] = nullptr; t.n_edges[j] = 1.0; t.K[j] = 1; t.m_tgt[j] = 0;
        }
    }
    t.njobs = njobs;
    hipStream_t st = as_stream(stream);
    float* partial = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(moments_batched_kernel, dim3((unsigned)nblk), dim3(256), 0, st, t, partial);
    CRF_LAUNCH_CHECK();
    hipLaunchKernelGGL(moments_finish_batched_kernel, dim3((unsigned)njobs), dim3(1024), 0, st, t, (const float*)partial);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_pointconv_stats(const float* pos_src, const float* pos_tgt, const int32_t* idx32,
                                       int K, int64_t m_tgt, int d, const float* A1, const float* b1,
                                       const float* W2, float slope, const float* mean_rel3, float* shift,
                                       double* stats, void* workspace, size_t workspace_bytes,
                                       crf_stream_t stream) {
    if (int rc = check_pc(m_tgt, K, d)) return rc;
    CRF_REQUIRE(pos_src && pos_tgt && idx32 && A1 && b1 && W2 && mean_rel3 && shift && stats && workspace,
                CRF_ERR_ARG, "null pointer");
    const int64_t nblk = blocks_for(m_tgt, d);
    CRF_REQUIRE(workspace_bytes >= sizeof(float) * 2 * d * (size_t)nblk, CRF_ERR_WORKSPACE, "workspace too small");
    hipStream_t st = as_stream(stream);
    float* partial = reinterpret_cast<float*>(workspace);
    DISPATCH_D(d, {
        hipLaunchKernelGGL(stats_kernel<DD>, dim3((unsigned)nblk), dim3(PBLOCK), 0, st, pos_src, pos_tgt,
                           idx32, K, m_tgt, A1, b1, W2, slope, mean_rel3, shift, partial);
    });
    CRF_LAUNCH_CHECK();
    return reduce_partials(partial, nblk, 2 * d, stats, st);
}

namespace crf {
// fold2_bwd fed by the block partials of uv_bwd_reduce_kernel directly (partial [nblk][2d] float): the 16
// wavefronts of one workgroup sum the slots in the fixed order of reduce_partials_kernel, then the first d threads
// derive the coefficients -- one launch instead of two.
__global__ __launch_bounds__(1024) void fold2_bwd_partials_kernel(const float* __restrict__ partial, int nblk,
                                                                  const float* __restrict__ shift,
                                                                  const double* __restrict__ aux2,
                                                                  const float* __restrict__ gamma, double n_edges,
                                                                  int use_batch, int d, float* __restrict__ ca,
                                                                  float* __restrict__ cb, float* __restrict__ cc,
                                                                  float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ double s_red[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int slot = wave; slot < 2 * d; slot += 16) {
        double a = 0.0;
        for (int b = lane; b < nblk; b += WAVE) a += (double)partial[(int64_t)b * 2 * d + slot];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, WAVE);
        if (lane == 0) s_red[slot] = a;
    }
    __syncthreads();
    const int c = threadIdx.x;
    if (c >= d) return;
    fold2_bwd_coef(c, d, s_red[c], s_red[d + c], shift, aux2, gamma, n_edges, use_batch, ca, cb, cc, dgamma, dbeta);
}

}  // namespace crf

static int64_t uv_reduce_blocks(int64_t m, int d) {
    const int rpi = 256 / (d / 4);
    const int64_t nb = cdiv(m, (int64_t)rpi * 8);              // ~8 rows per thread
    return nb < 1 ? 1 : (nb > 512 ? 512 : nb);
}

extern "C" int crfconv_pointconv_forward_uv(const float* x, const float* pos_src, const float* pos_tgt,
                                            const int32_t* idx32, int K, int64_t m_tgt, int d, const float* A1,
                                            const float* b1, const float* W2, float slope, const float* mean_rel3,
                                            float* shift, double* stats, float* U, float* V, void* workspace,
                                            size_t workspace_bytes, unsigned* ticket, crf_stream_t stream) {
    if (int rc = check_pc(m_tgt, K, d)) return rc;
    CRF_REQUIRE(x && pos_src && pos_tgt && idx32 && A1 && b1 && W2 && mean_rel3 && shift && stats && U && V &&
                    workspace, CRF_ERR_ARG, "null pointer");
    int64_t nblk = blocks_for(m_tgt, d);
    CRF_REQUIRE(workspace_bytes >= sizeof(float) * 2 * d * (size_t)nblk, CRF_ERR_WORKSPACE, "workspace too small");
    hipStream_t st = as_stream(stream);
    float* partial = reinterpret_cast<float*>(workspace);
    if (uvstats_mfma_ok(K, d) && m_tgt < ((int64_t)1 << 27)) {
        // wide layers, K = 16: layer 2 of a point's sixteen edges on the matrix pipe (pointconv_wide.hip)
        const int64_t room = (int64_t)(workspace_bytes / (sizeof(float) * 2 * d));
        if (int rc = uvstats_mfma_launch(x, pos_src, pos_tgt, idx32, m_tgt, d, A1, b1, W2, slope, mean_rel3, shift, U, V, partial, room, &nblk, ticket, stats, st)) return rc;
        return ticket != nullptr ? CRF_OK : reduce_partials(partial, nblk, 2 * d, stats, st);
    }
    if (nblk * (d / 2) > 256 * 24) ticket = nullptr;    // the last workgroup's sum pays up to three rounds of eight 16-byte loads per thread
    DISPATCH_D(d, {
        hipLaunchKernelGGL(uvstats_kernel<DD>, dim3((unsigned)nblk), dim3(PBLOCK), 0, st, x, pos_src, pos_tgt, idx32, K,
                           m_tgt, A1, b1, W2, slope, mean_rel3, shift, U, V, partial, ticket, stats);
    });
    CRF_LAUNCH_CHECK();
    return ticket != nullptr ? CRF_OK : reduce_partials(partial, nblk, 2 * d, stats, st);
}

// crfconv_pointconv_forward_uv whose launch also CARRIES crfconv_crf_matrices_batched(c, H, n, Q, P) as its last n workgroups
// (uvstats_hosting_kernel: the riders come first in the grid): for the layer widths crfconv_pointconv_forward_uv_hosts() names -- d = 8, the first PointConv of the
// reference networks (models/point_conv_big.py:107).  Results of both are those of the two separate calls.
extern "C" int crfconv_pointconv_forward_uv_hosts(int K, int d) { return (d == 8 && !uvstats_mfma_ok(K, d)) ? 1 : 0; }
extern "C" int crfconv_pointconv_forward_uv_hosting(const float* x, const float* pos_src, const float* pos_tgt,
                                                    const int32_t* idx32, int K, int64_t m_tgt, int d, const float* A1,
                                                    const float* b1, const float* W2, float slope, const float* mean_rel3,
                                                    float* shift, double* stats, float* U, float* V, void* workspace,
                                                    size_t workspace_bytes, unsigned* ticket, const float* const* c, const int* H,
                                                    int n, float* const* Q, float* const* P, crf_stream_t stream) {
    if (int rc = check_pc(m_tgt, K, d)) return rc;
    CRF_REQUIRE(crfconv_pointconv_forward_uv_hosts(K, d) == 1, CRF_ERR_UNSUPPORTED, "forward_uv_hosting: d=%d K=%d is not a hosting width", d, K);
    CRF_REQUIRE(x && pos_src && pos_tgt && idx32 && A1 && b1 && W2 && mean_rel3 && shift && stats && U && V &&
                    workspace, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(c && H && Q && P && n >= 1 && n <= CM_MAX, CRF_ERR_ARG, "null pointer or n=%d outside [1, %d]", n, CM_MAX);
    CrfMatJobs side = {};
    for (int i = 0; i < n; ++i) {
        CRF_REQUIRE(c[i] && Q[i] && P[i] && H[i] >= 1 && H[i] <= 64, CRF_ERR_ARG, "rider %d: null pointer or H=%d outside [1, 64]", i, H[i]);
        side.c[i] = c[i]; side.Q[i] = Q[i]; side.P[i] = P[i]; side.H[i] = H[i];
    }
    const int64_t nblk = blocks_for(m_tgt, d);
    CRF_REQUIRE(workspace_bytes >= sizeof(float) * 2 * d * (size_t)nblk, CRF_ERR_WORKSPACE, "workspace too small");
    hipStream_t st = as_stream(stream);
    float* partial = reinterpret_cast<float*>(workspace);
    if (nblk * (d / 2) > 256 * 24) ticket = nullptr;
    hipLaunchKernelGGL(uvstats_hosting_kernel<8>, dim3((unsigned)nblk + HOST_PAD(n)), dim3(PBLOCK), 0, st, x, pos_src, pos_tgt, idx32, K, m_tgt, A1, b1,
                       W2, slope, mean_rel3, shift, U, V, partial, ticket, stats, side, n);
    CRF_LAUNCH_CHECK();
    return ticket != nullptr ? CRF_OK : reduce_partials(partial, nblk, 2 * d, stats, st);
}

extern "C" int crfconv_pointconv_combine(const float* U, const float* V, const double* stats, const float* shift,
                                         const float* gamma2, const float* beta2, double n_edges, float* run_mean,
                                         float* run_var, float momentum, float eps, int64_t m_tgt, int d, float* a2,
                                         float* b2, double* aux2, float* out, crf_stream_t stream) {
    if (int rc = check_pc(m_tgt, 1, d)) return rc;
    CRF_REQUIRE(U && V && stats && shift && gamma2 && beta2 && a2 && b2 && aux2 && out, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE((run_mean == nullptr) == (run_var == nullptr), CRF_ERR_ARG, "running statistics come as a pair");
    const int64_t n4 = m_tgt * (d / 4);
    hipLaunchKernelGGL(uv_combine_kernel, dim3((unsigned)cdiv(n4, 256)), dim3(256), 0, as_stream(stream), U, V, stats,
                       shift, gamma2, beta2, n_edges, run_mean, run_var, momentum, eps, n4, d / 4, a2, b2, aux2, out);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

static int uv_red_args(const float* gout, const float* U, const float* V, int64_t m_tgt, int d, const float* shift, const double* aux2,
                       const float* gamma2, double n_edges, int use_batch, float* ca, float* cb, float* cc, float* dgamma2, float* dbeta2,
                       void* workspace, size_t workspace_bytes, unsigned* ticket, UvRed& r) {
    if (int rc = check_pc(m_tgt, 1, d)) return rc;
    CRF_REQUIRE(gout && U && V && shift && aux2 && gamma2 && ca && cb && cc && dgamma2 && dbeta2 && workspace,
                CRF_ERR_ARG, "null pointer");
    const int64_t nblk = uv_reduce_blocks(m_tgt, d);
    CRF_REQUIRE(workspace_bytes >= sizeof(float) * 2 * d * (size_t)nblk, CRF_ERR_WORKSPACE, "workspace too small");
    r.g = gout; r.U = U; r.V = V; r.m = m_tgt; r.d = d; r.nblk = (int)nblk; r.partial = reinterpret_cast<float*>(workspace); r.ticket = ticket;
    r.shift = shift; r.aux2 = aux2; r.gamma = gamma2; r.n_edges = n_edges; r.use_batch = use_batch; r.ca = ca; r.cb = cb; r.cc = cc;
    r.dgamma = dgamma2; r.dbeta = dbeta2;
    return CRF_OK;
}

extern "C" int crfconv_pointconv_bwd_reduce_uv(const float* gout, const float* U, const float* V, int64_t m_tgt, int d,
                                               const float* shift, const double* aux2, const float* gamma2,
                                               double n_edges, int use_batch, float* ca, float* cb, float* cc,
                                               float* dgamma2, float* dbeta2, void* workspace, size_t workspace_bytes,
                                               unsigned* ticket, crf_stream_t stream) {
    UvRed r;
    if (int rc = uv_red_args(gout, U, V, m_tgt, d, shift, aux2, gamma2, n_edges, use_batch, ca, cb, cc, dgamma2, dbeta2, workspace,
                             workspace_bytes, ticket, r)) return rc;
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(uv_bwd_reduce_kernel, dim3((unsigned)r.nblk), dim3(256), 0, st, r);
    CRF_LAUNCH_CHECK();
    if (ticket != nullptr) return CRF_OK;             // the last workgroup has derived the coefficients
    hipLaunchKernelGGL(fold2_bwd_partials_kernel, dim3(1), dim3(1024), 0, st, r.partial, r.nblk, shift, aux2, gamma2,
                       n_edges, use_batch, d, ca, cb, cc, dgamma2, dbeta2);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_pointconv_forward(const float* x, const float* pos_src, const float* pos_tgt,
                                         const int32_t* idx32, int K, int64_t m_tgt, int d,
                                         const float* A1, const float* b1, const float* W2, float slope,
                                         const float* a2, const float* b2, float* out,
                                         crf_stream_t stream) {
    if (int rc = check_pc(m_tgt, K, d)) return rc;
    CRF_REQUIRE(x && pos_src && pos_tgt && idx32 && A1 && b1 && W2 && a2 && b2 && out, CRF_ERR_ARG,
                "null pointer");
    const int64_t nblk = blocks_for(m_tgt, d);
    DISPATCH_D(d, {
        hipLaunchKernelGGL(forward_kernel<DD>, dim3((unsigned)nblk), dim3(PBLOCK), 0, as_stream(stream), x,
                           pos_src, pos_tgt, idx32, K, m_tgt, A1, b1, W2, slope, a2, b2, out);
    });
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_pointconv_bwd_reduce(const float* x, const float* gout, const float* pos_src,
                                            const float* pos_tgt, const int32_t* idx32, int K,
                                            int64_t m_tgt, int d, const float* A1, const float* b1,
                                            const float* W2, float slope, const float* shift, double* red1,
                                            void* workspace, size_t workspace_bytes, crf_stream_t stream) {
    if (int rc = check_pc(m_tgt, K, d)) return rc;
    CRF_REQUIRE(x && gout && pos_src && pos_tgt && idx32 && A1 && b1 && W2 && shift && red1 && workspace,
                CRF_ERR_ARG, "null pointer");
    const int64_t nblk = blocks_for(m_tgt, d);
    CRF_REQUIRE(workspace_bytes >= sizeof(float) * 2 * d * (size_t)nblk, CRF_ERR_WORKSPACE, "workspace too small");
    hipStream_t st = as_stream(stream);
    float* partial = reinterpret_cast<float*>(workspace);
    DISPATCH_D(d, {
        hipLaunchKernelGGL(bwd_reduce_kernel<DD>, dim3((unsigned)nblk), dim3(PBLOCK), 0, st, x, gout, pos_src,
                           pos_tgt, idx32, K, m_tgt, A1, b1, W2, slope, shift, partial);
    });
    CRF_LAUNCH_CHECK();
    return reduce_partials(partial, nblk, 2 * d, red1, st);
}

extern "C" int crfconv_pointconv_bwd_params(const float* x, const float* gout, const float* pos_src,
                                            const float* pos_tgt, const int32_t* idx32, int K,
                                            int64_t m_tgt, int d, const float* A1, const float* b1,
                                            const float* W2, float slope, const float* ca, const float* cb,
                                            const float* cc, double* dW2, double* dA1b1, void* workspace,
                                            size_t workspace_bytes, crf_stream_t stream) {
    if (int rc = check_pc(m_tgt, K, d)) return rc;
    CRF_REQUIRE(x && gout && pos_src && pos_tgt && idx32 && A1 && b1 && W2 && ca && cb && cc && workspace, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE((dW2 == nullptr) == (dA1b1 == nullptr), CRF_ERR_ARG, "dW2 and dA1b1: both or neither (neither = the partial slabs only)");
    CRF_REQUIRE(d <= 16, CRF_ERR_UNSUPPORTED, "in-kernel parameter reduction covers d <= 16; use crfconv_pointconv_bwd_dump for d=%d", d);
    const int64_t nblk = blocks_for(m_tgt, d);
    const size_t fbytes = (sizeof(float) * (size_t)d * d * (size_t)nblk + 255) & ~(size_t)255;
    const size_t dbytes = sizeof(double) * 4 * (size_t)d * (size_t)nblk;
    CRF_REQUIRE(workspace_bytes >= fbytes + dbytes + 256, CRF_ERR_WORKSPACE, "workspace too small");
    hipStream_t st = as_stream(stream);
    char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    float* partial = reinterpret_cast<float*>(base);
    double* partial_d = reinterpret_cast<double*>(base + fbytes);
    switch (d) {
        case 4: hipLaunchKernelGGL(bwd_params_kernel<4>, dim3((unsigned)nblk), dim3(PBLOCK), 0, st, x, gout, pos_src, pos_tgt, idx32, K, m_tgt, A1, b1, W2, slope, ca, cb, cc, partial, partial_d); break;
        case 8: hipLaunchKernelGGL(bwd_params_kernel<8>, dim3((unsigned)nblk), dim3(PBLOCK), 0, st, x, gout, pos_src, pos_tgt, idx32, K, m_tgt, A1, b1, W2, slope, ca, cb, cc, partial, partial_d); break;
        default: hipLaunchKernelGGL(bwd_params_kernel<16>, dim3((unsigned)nblk), dim3(PBLOCK), 0, st, x, gout, pos_src, pos_tgt, idx32, K, m_tgt, A1, b1, W2, slope, ca, cb, cc, partial, partial_d); break;
    }
    CRF_LAUNCH_CHECK();
    if (dW2 == nullptr) return CRF_OK;                   // the caller sums the slabs later (crfconv_pointconv_bwd_params_slabs)
    if (int rc = reduce_partials(partial, nblk, d * d, dW2, st)) return rc;
    hipLaunchKernelGGL(reduce_partials_d_kernel, dim3((unsigned)cdiv(4 * d, 256 / WAVE)), dim3(256), 0, st, partial_d,
                       nblk, 4 * d, dA1b1);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

// Where crfconv_pointconv_bwd_params left its partial slabs inside `workspace`: float [nblk][d*d] (dW2) and double [nblk][4d]
// (dA1 | db1) -- for a caller that passed dW2 = dA1b1 = NULL and finishes the sums with crfconv_reduce_jobs_f64.
extern "C" int crfconv_pointconv_bwd_params_slabs(void* workspace, int64_t m_tgt, int d, const float** slab_w2, const double** slab_a1,
                                                  int64_t* nblk_out) {
    CRF_REQUIRE(workspace && slab_w2 && slab_a1 && nblk_out && m_tgt > 0 && d >= 4 && d <= 16, CRF_ERR_ARG, "bad argument");
    const int64_t nblk = blocks_for(m_tgt, d);
    const size_t fbytes = (sizeof(float) * (size_t)d * d * (size_t)nblk + 255) & ~(size_t)255;
    char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    *slab_w2 = reinterpret_cast<const float*>(base);
    *slab_a1 = reinterpret_cast<const double*>(base + fbytes);
    *nblk_out = nblk;
    return CRF_OK;
}

extern "C" int crfconv_reduce_jobs_f64(const crf_reduce64_job* jobs, int njobs, crf_stream_t stream) {
    CRF_REQUIRE(jobs || njobs == 0, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(njobs >= 0, CRF_ERR_ARG, "njobs=%d < 0", njobs);
    hipStream_t st = as_stream(stream);
    for (int j0 = 0; j0 < njobs; j0 += R64_MAX) {
        Reduce64Table t;
        const int n = njobs - j0 < R64_MAX ? njobs - j0 : R64_MAX;
        int64_t waves = 0;
        for (int j = 0; j <= R64_MAX; ++j) {
            t.wave_base[j] = (int)waves;
            if (j < n) {
                const crf_reduce64_job& jb = jobs[j0 + j];
                CRF_REQUIRE(jb.partial && jb.out && jb.nblk > 0 && jb.nblk < ((int64_t)1 << 31) && jb.nslots > 0, CRF_ERR_ARG,
                            "job %d is malformed", j0 + j);
                t.partial[j] = jb.partial; t.out[j] = jb.out; t.is_float[j] = jb.is_float; t.nblk[j] = (int)jb.nblk; t.nslots[j] = jb.nslots;
                waves += jb.nslots;
                CRF_REQUIRE(waves < ((int64_t)1 << 30), CRF_ERR_UNSUPPORTED, "too many slots in one batch");
            } else if (j < R64_MAX) {
                t.partial[j] = nullptr; t.out[j] = nullptr; t.is_float[j] = 0; t.nblk[j] = 0; t.nslots[j] = 0;
            }
        }
        t.njobs = n;
        hipLaunchKernelGGL(reduce_jobs_f64_kernel, dim3((unsigned)cdiv(waves, 256 / WAVE)), dim3(256), 0, st, t);
        CRF_LAUNCH_CHECK();
    }
    return CRF_OK;
}

static int bwd_input_launch(const float* gout, const float* pos_src, const float* pos_tgt, const int32_t* rev_ptr, const int32_t* rev_eid,
                            int K, int64_t m_src, int d, const float* A1, const float* b1, const float* W2, float slope, const float* a2,
                            const float* b2, float* dx, const UvRed& red, crf_stream_t stream) {
    if (int rc = check_pc(m_src, K, d)) return rc;
    CRF_REQUIRE(gout && pos_src && pos_tgt && rev_ptr && rev_eid && A1 && b1 && W2 && a2 && b2 && dx,
                CRF_ERR_ARG, "null pointer");
    const int64_t nblk = blocks_for(m_src, d);
    DISPATCH_D(d, {
        hipLaunchKernelGGL(bwd_input_kernel<DD>, dim3((unsigned)(nblk + red.nblk)), dim3(PBLOCK), 0, as_stream(stream), gout,
                           pos_src, pos_tgt, rev_ptr, rev_eid, K, m_src, A1, b1, W2, slope, a2, b2, dx, red, (int)nblk);
    });
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_pointconv_bwd_input(const float* gout, const float* pos_src, const float* pos_tgt,
                                           const int32_t* rev_ptr, const int32_t* rev_eid, int K,
                                           int64_t m_src, int d, const float* A1, const float* b1,
                                           const float* W2, float slope, const float* a2, const float* b2, float* dx,
                                           crf_stream_t stream) {
    UvRed none{};
    none.nblk = 0;
    return bwd_input_launch(gout, pos_src, pos_tgt, rev_ptr, rev_eid, K, m_src, d, A1, b1, W2, slope, a2, b2, dx, none, stream);
}

// crfconv_pointconv_bwd_input and crfconv_pointconv_bwd_reduce_uv (ticketed) of one layer in ONE launch: the input gradient needs the
// forward coefficients only, the reduction's outputs (ca, cb, cc: the parameter pass's; dgamma2, dbeta2) are not read before the
// next launch.  Same results as the two calls.
extern "C" int crfconv_pointconv_bwd_input_reduce(const float* gout, const float* pos_src, const float* pos_tgt, const int32_t* rev_ptr,
                                                  const int32_t* rev_eid, int K, int64_t m_src, int64_t m_tgt, int d, const float* A1,
                                                  const float* b1, const float* W2, float slope, const float* a2, const float* b2, float* dx,
                                                  const float* U, const float* V, const float* shift, const double* aux2,
                                                  const float* gamma2, double n_edges, int use_batch, float* ca, float* cb, float* cc,
                                                  float* dgamma2, float* dbeta2, void* workspace, size_t workspace_bytes, unsigned* ticket,
                                                  crf_stream_t stream) {
    CRF_REQUIRE(ticket, CRF_ERR_ARG, "null pointer");
    UvRed r;
    if (int rc = uv_red_args(gout, U, V, m_tgt, d, shift, aux2, gamma2, n_edges, use_batch, ca, cb, cc, dgamma2, dbeta2, workspace,
                             workspace_bytes, ticket, r)) return rc;
    return bwd_input_launch(gout, pos_src, pos_tgt, rev_ptr, rev_eid, K, m_src, d, A1, b1, W2, slope, a2, b2, dx, r, stream);
}

extern "C" int crfconv_pointconv_bwd_dump(const float* x, const float* gout, const float* pos_src,
                                          const float* pos_tgt, const int32_t* idx32, int K, int64_t m_tgt,
                                          int d, const float* A1, const float* b1, const float* W2, float slope,
                                          const float* ca, const float* cb, const float* cc, float* h1,
                                          float* gh2, float* rel, crf_stream_t stream) {
    if (int rc = check_pc(m_tgt, K, d)) return rc;
    CRF_REQUIRE(x && gout && pos_src && pos_tgt && idx32 && A1 && b1 && W2 && ca && cb && cc && h1 && gh2 && rel,
                CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(m_tgt * K < ((int64_t)1 << 31), CRF_ERR_ARG, "too many edges");
    const int64_t nblk = blocks_for(m_tgt, d);
    DISPATCH_D(d, {
        hipLaunchKernelGGL(bwd_dump_kernel<DD>, dim3((unsigned)nblk), dim3(PBLOCK), 0, as_stream(stream), x, gout,
                           pos_src, pos_tgt, idx32, K, m_tgt, A1, b1, W2, slope, ca, cb, cc, h1, gh2, rel);
    });
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

// crfconv_pointconv_bwd_dump for several layers: one launch per width present among the jobs (the two ResNet blocks of a level
// share one), workgroups of the jobs laid end to end.  Identical h1 / gh2 / rel.  jobs: host array.
extern "C" int crfconv_pointconv_bwd_dump_jobs(const crf_pc_dump_job* jobs, int njobs, crf_stream_t stream) {
    CRF_REQUIRE(jobs || njobs == 0, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(njobs >= 0, CRF_ERR_ARG, "njobs=%d < 0", njobs);
    hipStream_t st = as_stream(stream);
    for (int j = 0; j < njobs; ++j) {
        const crf_pc_dump_job& jb = jobs[j];
        if (int rc = check_pc(jb.m_tgt, jb.K, jb.d)) return rc;
        CRF_REQUIRE(jb.x && jb.gout && jb.pos_src && jb.pos_tgt && jb.idx32 && jb.A1 && jb.b1 && jb.W2 && jb.ca && jb.cb && jb.cc && jb.h1 &&
                        jb.gh2 && jb.rel, CRF_ERR_ARG, "job %d: null pointer", j);
        CRF_REQUIRE(jb.m_tgt * jb.K < ((int64_t)1 << 31), CRF_ERR_ARG, "job %d: too many edges", j);
    }
    static const int widths[6] = {4, 8, 16, 32, 64, 128};
    for (int w = 0; w < 6; ++w) {
        const int d = widths[w];
        int j = 0;
        while (j < njobs) {
            DumpJobs t;
            int n = 0;
            int64_t blocks = 0;
            for (; j < njobs && n < PJ_MAX; ++j) {
                const crf_pc_dump_job& jb = jobs[j];
                if (jb.d != d) continue;
                t.x[n] = jb.x; t.gout[n] = jb.gout; t.pos_src[n] = jb.pos_src; t.pos_tgt[n] = jb.pos_tgt; t.idx[n] = jb.idx32;
                t.A1[n] = jb.A1; t.b1[n] = jb.b1; t.W2[n] = jb.W2; t.ca[n] = jb.ca; t.cb[n] = jb.cb; t.cc[n] = jb.cc;
                t.h1[n] = jb.h1; t.gh2[n] = jb.gh2; t.rel[n] = jb.rel; t.K[n] = jb.K; t.m_tgt[n] = (int)jb.m_tgt; t.slope[n] = jb.slope;
                t.blk_base[n] = (int)blocks;
                blocks += blocks_for(jb.m_tgt, d);
                ++n;
            }
            if (n == 0) break;
            for (int k = n; k <= PJ_MAX; ++k) t.blk_base[k] = (int)blocks;
            for (int k = n; k < PJ_MAX; ++k) {
                t.x[k] = t.gout[k] = t.pos_src[k] = t.pos_tgt[k] = t.A1[k] = t.b1[k] = t.W2[k] = t.ca[k] = t.cb[k] = t.cc[k] = nullptr;
                t.idx[k] = nullptr; t.h1[k] = t.gh2[k] = t.rel[k] = nullptr; t.K[k] = 1; t.m_tgt[k] = 0; t.slope[k] = 1.f;
            }
            t.njobs = n;
            DISPATCH_D(d, { hipLaunchKernelGGL(bwd_dump_jobs_kernel<DD>, dim3((unsigned)blocks), dim3(PBLOCK), 0, st, t); });
            CRF_LAUNCH_CHECK();
        }
    }
    return CRF_OK;
}

// crfconv_pointconv_bwd_a1 (slabs only: the sums are the caller's, crfconv_reduce_jobs_f64) for several layers, one launch per
// width.  Each job's workspace as for the one-layer call; slabs at its 256-byte-aligned start.
extern "C" int crfconv_pointconv_bwd_a1_jobs(const crf_pc_a1_job* jobs, int njobs, crf_stream_t stream) {
    CRF_REQUIRE(jobs || njobs == 0, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(njobs >= 0, CRF_ERR_ARG, "njobs=%d < 0", njobs);
    hipStream_t st = as_stream(stream);
    for (int j = 0; j < njobs; ++j) {
        const crf_pc_a1_job& jb = jobs[j];
        CRF_REQUIRE(jb.gw && jb.h1 && jb.rel && jb.workspace, CRF_ERR_ARG, "job %d: null pointer", j);
        CRF_REQUIRE(jb.d == 8 || jb.d == 16 || jb.d == 32 || jb.d == 64 || jb.d == 128, CRF_ERR_UNSUPPORTED, "job %d: d=%d", j, jb.d);
        CRF_REQUIRE(jb.n_edges > 0 && jb.n_edges < ((int64_t)1 << 31), CRF_ERR_ARG, "job %d: n_edges out of range", j);
        CRF_REQUIRE(jb.workspace_bytes >= crfconv_pointconv_bwd_a1_workspace(jb.n_edges, jb.d), CRF_ERR_WORKSPACE, "job %d: workspace too small", j);
    }
    static const int widths[5] = {8, 16, 32, 64, 128};
    for (int w = 0; w < 5; ++w) {
        const int d = widths[w];
        int j = 0;
        while (j < njobs) {
            A1Jobs t;
            int n = 0;
            int64_t blocks = 0;
            for (; j < njobs && n < PJ_MAX; ++j) {
                const crf_pc_a1_job& jb = jobs[j];
                if (jb.d != d) continue;
                const int64_t nblk = crfconv_pointconv_bwd_a1_nblk(jb.n_edges, d);
                t.gw[n] = jb.gw; t.h1[n] = jb.h1; t.rel[n] = jb.rel;
                t.partial_d[n] = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(jb.workspace) + 255) & ~(uintptr_t)255);
                t.E[n] = (int)jb.n_edges; t.nblk[n] = (int)nblk; t.slope[n] = jb.slope;
                t.blk_base[n] = (int)blocks;
                blocks += nblk;
                ++n;
            }
            if (n == 0) break;
            for (int k = n; k <= PJ_MAX; ++k) t.blk_base[k] = (int)blocks;
            for (int k = n; k < PJ_MAX; ++k) { t.gw[k] = t.h1[k] = t.rel[k] = nullptr; t.partial_d[k] = nullptr; t.E[k] = 0; t.nblk[k] = 1; t.slope[k] = 1.f; }
            t.njobs = n;
            switch (d) {
                case 8: hipLaunchKernelGGL(a1_reduce_jobs_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, st, t); break;
                case 16: hipLaunchKernelGGL(a1_reduce_jobs_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, st, t); break;
                case 32: hipLaunchKernelGGL(a1_reduce_jobs_kernel<32>, dim3((unsigned)blocks), dim3(256), 0, st, t); break;
                case 64: hipLaunchKernelGGL(a1_reduce_jobs_kernel<64>, dim3((unsigned)blocks), dim3(256), 0, st, t); break;
                default: hipLaunchKernelGGL(a1_reduce_jobs_kernel<128>, dim3((unsigned)blocks), dim3(256), 0, st, t); break;
            }
            CRF_LAUNCH_CHECK();
        }
    }
    return CRF_OK;
}

extern "C" size_t crfconv_pointconv_bwd_a1_workspace(int64_t n_edges, int d) {
    if (n_edges <= 0 || d < 8) return 0;
    const int64_t nblk = std::min<int64_t>(1024, cdiv(n_edges, 256 / (d / 4) * 8));
    return sizeof(double) * 4 * (size_t)d * (size_t)nblk + 256;
}

extern "C" int crfconv_pointconv_bwd_a1(const float* gw, const float* h1, const float* rel, int64_t n_edges, int d,
                                        float slope, double* dA1b1, void* workspace, size_t workspace_bytes,
                                        crf_stream_t stream) {
    CRF_REQUIRE(gw && h1 && rel && workspace, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(d == 8 || d == 16 || d == 32 || d == 64 || d == 128, CRF_ERR_UNSUPPORTED, "d=%d not in {8, 16, 32, 64, 128}", d);
    CRF_REQUIRE(n_edges > 0 && n_edges < ((int64_t)1 << 31), CRF_ERR_ARG, "n_edges=%lld out of range", (long long)n_edges);
    CRF_REQUIRE(workspace_bytes >= crfconv_pointconv_bwd_a1_workspace(n_edges, d), CRF_ERR_WORKSPACE, "workspace too small");
    const int64_t nblk = std::min<int64_t>(1024, cdiv(n_edges, 256 / (d / 4) * 8));
    double* partial_d = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    hipStream_t st = as_stream(stream);
    switch (d) {
        case 8: hipLaunchKernelGGL(a1_reduce_kernel<8>, dim3((unsigned)nblk), dim3(256), 0, st, gw, h1, rel, n_edges, slope, partial_d); break;
        case 16: hipLaunchKernelGGL(a1_reduce_kernel<16>, dim3((unsigned)nblk), dim3(256), 0, st, gw, h1, rel, n_edges, slope, partial_d); break;
        case 32: hipLaunchKernelGGL(a1_reduce_kernel<32>, dim3((unsigned)nblk), dim3(256), 0, st, gw, h1, rel, n_edges, slope, partial_d); break;
        case 64: hipLaunchKernelGGL(a1_reduce_kernel<64>, dim3((unsigned)nblk), dim3(256), 0, st, gw, h1, rel, n_edges, slope, partial_d); break;
        default: hipLaunchKernelGGL(a1_reduce_kernel<128>, dim3((unsigned)nblk), dim3(256), 0, st, gw, h1, rel, n_edges, slope, partial_d); break;
    }
    CRF_LAUNCH_CHECK();
    if (dA1b1 == nullptr) return CRF_OK;                 // slabs only: double [nblk][4d] at the 256-byte-aligned start of `workspace`,
                                                         // nblk = crfconv_pointconv_bwd_a1_nblk; summed later by crfconv_reduce_jobs_f64
    hipLaunchKernelGGL(reduce_partials_d_kernel, dim3((unsigned)cdiv(4 * d, 256 / WAVE)), dim3(256), 0, st, partial_d,
                       nblk, 4 * d, dA1b1);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int64_t crfconv_pointconv_bwd_a1_nblk(int64_t n_edges, int d) {
    if (n_edges <= 0 || d < 8) return 0;
    return std::min<int64_t>(1024, cdiv(n_edges, 256 / (d / 4) * 8));
}

// ====================================================================== BatchNorm folding (tiny, one block)
// The weight MLP's two BatchNorms are folded into per-channel affine coefficients.  BN-1 sits directly on
// Linear(3 -> d) of rel = p_i - p_j, so its batch statistics are ANALYTIC in the first two moments of rel:
//   mean1[c] = w_c . mu,   var1[c] = w_c^T Sigma w_c      (mu [3], Sigma [3,3] from crfconv_pointconv_moments)
// and so is its backward.  These kernels replace ~60 tiny framework launches per convolution.
namespace crf {

// mom = {mu[3], Sigma[9]} float64.  aux1 [3, d] float64 out = {a = gamma*rstd, mean1, var1}.
__device__ __forceinline__ void fold1_body(const float* __restrict__ W1, const float* __restrict__ gamma,
                                           const float* __restrict__ beta, const double* __restrict__ mom,
                                           double n_edges, float* __restrict__ run_mean,
                                           float* __restrict__ run_var, float momentum, float eps,
                                           int use_batch, int d, float* __restrict__ A1,
                                           float* __restrict__ b1, double* __restrict__ aux1) {
    const int c = threadIdx.x;
    if (c >= d) return;
    const double w[3] = {W1[3 * c], W1[3 * c + 1], W1[3 * c + 2]};
    double mean, var;
    if (use_batch) {
        mean = w[0] * mom[0] + w[1] * mom[1] + w[2] * mom[2];
        var = 0.0;
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) var += w[a] * mom[3 + 3 * a + b] * w[b];
        if (var < 0.0) var = 0.0;
        if (run_mean != nullptr) {
            const double unb = n_edges > 1.0 ? var * (n_edges / (n_edges - 1.0)) : var;
            run_mean[c] = (float)((1.0 - momentum) * run_mean[c] + momentum * mean);
            run_var[c] = (float)((1.0 - momentum) * run_var[c] + momentum * unb);
        }
    } else {
        mean = run_mean[c];
        var = run_var[c];
    }
    const double a = (double)gamma[c] / sqrt(var + (double)eps);
    A1[3 * c] = (float)(a * w[0]);
    A1[3 * c + 1] = (float)(a * w[1]);
    A1[3 * c + 2] = (float)(a * w[2]);
    b1[c] = (float)((double)beta[c] - a * mean);
    aux1[c] = a;
    aux1[d + c] = mean;
    aux1[2 * d + c] = var;
}

__global__ __launch_bounds__(128) void fold1_kernel(const float* __restrict__ W1, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, const double* __restrict__ mom,
                                                    double n_edges, float* __restrict__ run_mean,
                                                    float* __restrict__ run_var, float momentum, float eps,
                                                    int use_batch, int d, float* __restrict__ A1,
                                                    float* __restrict__ b1, double* __restrict__ aux1) {
    fold1_body(W1, gamma, beta, mom, n_edges, run_mean, run_var, momentum, eps, use_batch, d, A1, b1, aux1);
}

// fold1 of ALL PointConv layers of a network in one launch (a workgroup per layer): its inputs -- the weight MLP's first
// layer and the rel-pos moments of the layer's table -- are known before the forward pass starts.
constexpr int F1F_MAX = 24;
struct Fold1Table {
    crf_fold1_job job[F1F_MAX];
};
__global__ __launch_bounds__(128) void fold1_batched_kernel(const Fold1Table t) {
    const crf_fold1_job& j = t.job[blockIdx.x];
    fold1_body(j.W1, j.gamma1, j.beta1, j.mom, j.n_edges, j.run_mean, j.run_var, j.momentum, j.eps, j.use_batch, j.d, j.A1, j.b1,
               j.aux1);
}

// dA1b1 [d, 4] float64 = {dA1[c][0..2], db1[c]}  ->  dW1 [d,3], dgamma1, dbeta1
__device__ __forceinline__ void fold1_bwd_body(const float* __restrict__ W1, const float* __restrict__ gamma,
                                               const double* __restrict__ mom, const double* __restrict__ aux1,
                                               const double* __restrict__ dA1b1, float eps, int use_batch,
                                               int d, float* __restrict__ dW1, float* __restrict__ dgamma,
                                               float* __restrict__ dbeta, const double* __restrict__ dW2_f64,
                                               float* __restrict__ dW2_f32) {
    if (dW2_f64 != nullptr)                 // the float64 accumulator of bwd_params -> the float32 gradient (no cast launch)
        for (int i = threadIdx.x; i < d * d; i += 128) dW2_f32[i] = (float)dW2_f64[i];
    const int c = threadIdx.x;
    if (c >= d) return;
    const double w[3] = {W1[3 * c], W1[3 * c + 1], W1[3 * c + 2]};
    const double a = aux1[c], mean = aux1[d + c], var = aux1[2 * d + c];
    const double r = 1.0 / sqrt(var + (double)eps);
    const double dA[3] = {dA1b1[4 * c], dA1b1[4 * c + 1], dA1b1[4 * c + 2]};
    const double db = dA1b1[4 * c + 3];
    const double da = dA[0] * w[0] + dA[1] * w[1] + dA[2] * w[2] - db * mean;   // A = a w, b = beta - a mean
    dgamma[c] = (float)(da * r);
    dbeta[c] = (float)db;
    double dw[3] = {a * dA[0], a * dA[1], a * dA[2]};
    if (use_batch) {
        const double dv = -0.5 * da * (double)gamma[c] * r * r * r;             // a = gamma (var + eps)^-1/2
        for (int k = 0; k < 3; ++k) {
            double sw = 0.0;
            for (int b = 0; b < 3; ++b) sw += mom[3 + 3 * k + b] * w[b];         // (Sigma w)_k, Sigma symmetric
            dw[k] += -db * a * mom[k] + dv * 2.0 * sw;                           // through mean1 and var1
        }
    }
    dW1[3 * c] = (float)dw[0];
    dW1[3 * c + 1] = (float)dw[1];
    dW1[3 * c + 2] = (float)dw[2];
}

__global__ __launch_bounds__(128) void fold1_bwd_kernel(const float* __restrict__ W1, const float* __restrict__ gamma,
                                                        const double* __restrict__ mom, const double* __restrict__ aux1,
                                                        const double* __restrict__ dA1b1, float eps, int use_batch,
                                                        int d, float* __restrict__ dW1, float* __restrict__ dgamma,
                                                        float* __restrict__ dbeta, const double* __restrict__ dW2_f64,
                                                        float* __restrict__ dW2_f32) {
    fold1_bwd_body(W1, gamma, mom, aux1, dA1b1, eps, use_batch, d, dW1, dgamma, dbeta, dW2_f64, dW2_f32);
}

// The same for ALL PointConv layers of a backward pass in one launch (a workgroup per layer): nothing in the pass waits for
// dW1 / dgamma1 / dbeta1, so the caller queues the jobs and runs them once at the end (ops.deferred_weight_grads).
constexpr int F1_MAX = 24;
struct Fold1BwdTable {
    crf_fold1_bwd_job job[F1_MAX];
};
__global__ __launch_bounds__(128) void fold1_bwd_batched_kernel(const Fold1BwdTable t) {
    const crf_fold1_bwd_job& j = t.job[blockIdx.x];
    fold1_bwd_body(j.W1, j.gamma1, j.mom, j.aux1, j.dA1b1, j.eps, j.use_batch, j.d, j.dW1, j.dgamma1, j.dbeta1, j.dW2_f64,
                   j.dW2_f32);
}

// stats [2, d] float64 = {sum(h2 - shift), sum (h2 - shift)^2}  ->  a2, b2; aux2 [2, d] = {mean2, rstd2}
__global__ __launch_bounds__(128) void fold2_kernel(const double* __restrict__ stats, const float* __restrict__ shift,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    double n_edges, float* __restrict__ run_mean,
                                                    float* __restrict__ run_var, float momentum, float eps,
                                                    int use_batch, int d, float* __restrict__ a2,
                                                    float* __restrict__ b2, double* __restrict__ aux2) {
    const int c = threadIdx.x;
    if (c >= d) return;
    double mean, var;
    if (use_batch) {
        const double m1 = stats[c] / n_edges;
        mean = (double)shift[c] + m1;
        var = stats[d + c] / n_edges - m1 * m1;
        if (var < 0.0) var = 0.0;
        if (run_mean != nullptr) {
            const double unb = n_edges > 1.0 ? var * (n_edges / (n_edges - 1.0)) : var;
            run_mean[c] = (float)((1.0 - momentum) * run_mean[c] + momentum * mean);
            run_var[c] = (float)((1.0 - momentum) * run_var[c] + momentum * unb);
        }
    } else {
        mean = run_mean[c];
        var = run_var[c];
    }
    const double rstd = 1.0 / sqrt(var + (double)eps);
    const double a = (double)gamma[c] * rstd;
    a2[c] = (float)a;
    b2[c] = (float)((double)beta[c] - a * mean);
    aux2[c] = mean;
    aux2[d + c] = rstd;
}

// red [2, d] float64 = {sum g_w, sum g_w (h2 - shift)}  ->  dgamma2, dbeta2 and the coefficients of
// g_h2 = ca g_w + cb h2 + cc  (BatchNorm backward; eval: ca = gamma rstd, cb = cc = 0)
__global__ __launch_bounds__(128) void fold2_bwd_kernel(const double* __restrict__ red, const float* __restrict__ shift,
                                                        const double* __restrict__ aux2, const float* __restrict__ gamma,
                                                        double n_edges, int use_batch, int d, float* __restrict__ ca,
                                                        float* __restrict__ cb, float* __restrict__ cc,
                                                        float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = threadIdx.x;
    if (c >= d) return;
    fold2_bwd_coef(c, d, red[c], red[d + c], shift, aux2, gamma, n_edges, use_batch, ca, cb, cc, dgamma, dbeta);
}

}  // namespace crf

extern "C" int crfconv_pointconv_fold1(const float* W1, const float* gamma1, const float* beta1, const double* mom,
                                       double n_edges, float* run_mean, float* run_var, float momentum, float eps,
                                       int use_batch, int d, float* A1, float* b1, double* aux1, crf_stream_t stream) {
    CRF_REQUIRE(W1 && gamma1 && beta1 && mom && A1 && b1 && aux1, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(d >= 1 && d <= 128, CRF_ERR_UNSUPPORTED, "d=%d outside [1, 128]", d);
    CRF_REQUIRE(use_batch || (run_mean && run_var), CRF_ERR_ARG, "eval mode needs running statistics");
    hipLaunchKernelGGL(fold1_kernel, dim3(1), dim3(128), 0, as_stream(stream), W1, gamma1, beta1, mom, n_edges, run_mean,
                       run_var, momentum, eps, use_batch, d, A1, b1, aux1);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_pointconv_fold1_batched(const crf_fold1_job* jobs, int njobs, crf_stream_t stream) {
    CRF_REQUIRE(jobs || njobs == 0, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(njobs >= 0, CRF_ERR_ARG, "njobs=%d < 0", njobs);
    for (int j = 0; j < njobs; ++j) {
        const crf_fold1_job& b = jobs[j];
        CRF_REQUIRE(b.W1 && b.gamma1 && b.beta1 && b.mom && b.A1 && b.b1 && b.aux1, CRF_ERR_ARG, "job %d: null pointer", j);
        CRF_REQUIRE(b.d >= 1 && b.d <= 128, CRF_ERR_UNSUPPORTED, "job %d: d=%d outside [1, 128]", j, b.d);
        CRF_REQUIRE(b.use_batch || (b.run_mean && b.run_var), CRF_ERR_ARG, "job %d: eval mode needs running statistics", j);
    }
    for (int j0 = 0; j0 < njobs; j0 += F1F_MAX) {
        Fold1Table t;
        const int n = njobs - j0 < F1F_MAX ? njobs - j0 : F1F_MAX;
        for (int j = 0; j < F1F_MAX; ++j) t.job[j] = jobs[j0 + (j < n ? j : 0)];
        hipLaunchKernelGGL(fold1_batched_kernel, dim3((unsigned)n), dim3(128), 0, as_stream(stream), t);
        CRF_LAUNCH_CHECK();
    }
    return CRF_OK;
}

extern "C" int crfconv_pointconv_fold1_bwd(const float* W1, const float* gamma1, const double* mom, const double* aux1,
                                           const double* dA1b1, float eps, int use_batch, int d, float* dW1,
                                           float* dgamma1, float* dbeta1, const double* dW2_f64, float* dW2_f32,
                                           crf_stream_t stream) {
    CRF_REQUIRE(W1 && gamma1 && mom && aux1 && dA1b1 && dW1 && dgamma1 && dbeta1, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE((dW2_f64 == nullptr) == (dW2_f32 == nullptr), CRF_ERR_ARG, "dW2_f64 / dW2_f32: both or neither");
    CRF_REQUIRE(d >= 1 && d <= 128, CRF_ERR_UNSUPPORTED, "d=%d outside [1, 128]", d);
    hipLaunchKernelGGL(fold1_bwd_kernel, dim3(1), dim3(128), 0, as_stream(stream), W1, gamma1, mom, aux1, dA1b1, eps,
                       use_batch, d, dW1, dgamma1, dbeta1, dW2_f64, dW2_f32);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_pointconv_fold1_bwd_batched(const crf_fold1_bwd_job* jobs, int njobs, crf_stream_t stream) {
    CRF_REQUIRE(jobs || njobs == 0, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(njobs >= 0, CRF_ERR_ARG, "njobs=%d < 0", njobs);
    for (int j = 0; j < njobs; ++j) {
        const crf_fold1_bwd_job& b = jobs[j];
        CRF_REQUIRE(b.W1 && b.gamma1 && b.mom && b.aux1 && b.dA1b1 && b.dW1 && b.dgamma1 && b.dbeta1, CRF_ERR_ARG,
                    "job %d: null pointer", j);
        CRF_REQUIRE((b.dW2_f64 == nullptr) == (b.dW2_f32 == nullptr), CRF_ERR_ARG, "job %d: dW2_f64 / dW2_f32: both or neither", j);
        CRF_REQUIRE(b.d >= 1 && b.d <= 128, CRF_ERR_UNSUPPORTED, "job %d: d=%d outside [1, 128]", j, b.d);
    }
    for (int j0 = 0; j0 < njobs; j0 += F1_MAX) {
        Fold1BwdTable t;
        const int n = njobs - j0 < F1_MAX ? njobs - j0 : F1_MAX;
        for (int j = 0; j < n; ++j) t.job[j] = jobs[j0 + j];
        for (int j = n; j < F1_MAX; ++j) t.job[j] = jobs[j0];
        hipLaunchKernelGGL(fold1_bwd_batched_kernel, dim3((unsigned)n), dim3(128), 0, as_stream(stream), t);
        CRF_LAUNCH_CHECK();
    }
    return CRF_OK;
}

extern "C" int crfconv_pointconv_fold2(const double* stats, const float* shift, const float* gamma2, const float* beta2,
                                       double n_edges, float* run_mean, float* run_var, float momentum, float eps,
                                       int use_batch, int d, float* a2, float* b2, double* aux2, crf_stream_t stream) {
    CRF_REQUIRE(gamma2 && beta2 && a2 && b2 && aux2 && (!use_batch || (stats && shift)), CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(d >= 1 && d <= 128, CRF_ERR_UNSUPPORTED, "d=%d outside [1, 128]", d);
    CRF_REQUIRE(use_batch || (run_mean && run_var), CRF_ERR_ARG, "eval mode needs running statistics");
    hipLaunchKernelGGL(fold2_kernel, dim3(1), dim3(128), 0, as_stream(stream), stats, shift, gamma2, beta2, n_edges,
                       run_mean, run_var, momentum, eps, use_batch, d, a2, b2, aux2);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_pointconv_fold2_bwd(const double* red, const float* shift, const double* aux2,
                                           const float* gamma2, double n_edges, int use_batch, int d, float* ca,
                                           float* cb, float* cc, float* dgamma2, float* dbeta2, crf_stream_t stream) {
    CRF_REQUIRE(red && shift && aux2 && gamma2 && ca && cb && cc && dgamma2 && dbeta2, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(d >= 1 && d <= 128, CRF_ERR_UNSUPPORTED, "d=%d outside [1, 128]", d);
    hipLaunchKernelGGL(fold2_bwd_kernel, dim3(1), dim3(128), 0, as_stream(stream), red, shift, aux2, gamma2, n_edges,
                       use_batch, d, ca, cb, cc, dgamma2, dbeta2);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}
