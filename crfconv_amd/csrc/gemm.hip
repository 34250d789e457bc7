// Small dense products of the per-point layers on the coarse levels:  C [M, N] = A [M, K] · B (+ bias [N]) (+ addend [M, N])
//
// The shapes the row-streaming kernels of linear.hip do not take: the coarse-level forward (models/common.py:30,35 -- MLP.lin
// at 640 .. 10 240 rows, up to 512 channels) and every dX = gY · W of the MLP / ResNet-block backward (autograd of the same
// lines), plus the per-edge g_h1 = g_h2 · W2 of the wide PointConv layers (models/point_conv_big.py:45-47).  M is small
// (10^2 .. 10^5), K and N are 32 .. 512: a few hundred MFLOP each, bounded by dependent memory round trips and by how many
// wavefronts the grid gives the 1024 SIMDs, not by the matrix pipe.
//
// Both operands go through LDS, fetched with whole-cache-line loads (8 lanes per 128-byte row segment of a 32-wide K chunk;
// 64-byte row pieces read straight into the fragment layout measured 2 TB/s: 46 us for 2560 x 256 x 512), chunks c + 1 and
// c + 2 in registers while chunk c is on the matrix pipe.  Four wavefronts per workgroup, WR x WC, each WM x WN tiles of
// v_mfma_f32_16x16x4_f32 (exact f32 products, f32 accumulation), D[i = n][j = row]:
//   A operand lane l = (rr = l & 15, g = l >> 4):  B-matrix element [k = kc + 4 g + e][n = 16 tn + rr]
//   B operand lane l:                              A-matrix element [row = 16 tm + rr][k = kc + 4 g + e]
//   lane l ends with C[row = 16 tm + rr][n = 16 tn + 4 g + 0..3]: one 16-byte store.
// The B tile keeps the layout its storage order gives for free ([n][k] for B = W [N, K], [k][n] for B = W [K, N]): no
// transposing scalar traffic.  The tile shape is picked per problem so that the grid covers the chip.  Summation order
// over k is fixed by the shape alone: results are bitwise reproducible.
//
// Forms of the one tile routine (gemm_tile): plain (crfconv_gemm: bias / addend epilogue, any widths); STATS (crfconv_gemm_stats:
// BatchNorm statistic records of C from the epilogue); PRO (crfconv_mlp_small_backward: the A operand is the BatchNorm-backward
// gradient gY, formed from (gA, Y) and row-tile sums while it is loaded); jobs (crfconv_gemm_jobs: several independent products in
// one launch).  bn_bwd_tile_sums_kernel, the PRO form's first launch, lives here too.
#include "common.hpp"
#include "gridsync.hpp"

#include <cstdlib>

namespace crf {

using gf32x4 = __attribute__((ext_vector_type(4))) float;

#ifndef GM_BK_
#define GM_BK_ 32
#endif
#ifndef GM_PF_
#define GM_PF_ 2
#endif
#ifndef GM_WAIT_SLEEP_
#define GM_WAIT_SLEEP_ 4       // s_sleep argument (64 clocks each) between two polls of an in-launch wait
#endif
constexpr int GM_BLOCK = 256, GM_BK = GM_BK_;      // k columns per chunk (a multiple of 16)
constexpr int GM_PF = GM_PF_;                      // operand chunks in flight per workgroup (register sets; even)
static_assert(GM_PF >= 2 && GM_PF % 2 == 0, "an even number of register sets (the LDS buffers alternate)");

// four consecutive floats of which the first `valid` exist (VEC: widths are multiples of 4, so it is all or nothing and the
// address is 16-byte aligned; else element by element -- odd widths such as the 13-class logits)
template <bool VEC>
__device__ __forceinline__ float4 gm_ld4(const float* __restrict__ p, int valid) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (VEC) {
        if (valid > 0) v = *reinterpret_cast<const float4*>(p);
    } else {
        if (valid > 0) v.x = p[0];
        if (valid > 1) v.y = p[1];
        if (valid > 2) v.z = p[2];
        if (valid > 3) v.w = p[3];
    }
    return v;
}

// PRO form (crfconv_mlp_small_backward): the A operand is not read but FORMED while it is loaded -- gY, the gradient in front of a
// train-mode BatchNorm + LeakyReLU, from (gA, Y) and per-channel coefficients: g1 = gA * lrelu'(a y + b), yh = (y - mean) rstd,
// gY = a (g1 - sum g1 / M - yh sum g1 yh / M).  The two channel means come finished from the tile-sum launch (its last workgroup per
// column slab adds the row-tile partials in float64, round 5: until then EVERY workgroup of the product re-summed all of them --
// 98 KB of partials per workgroup at 256 channels, 2 560 workgroups for a 10 240-row layer); the workgroups of the first column
// slab also store the gY tiles they form (the weight gradient needs them).
constexpr int GM_PRO_MAXK = 512;
struct GemmPro {
    const float* Y;            // [M, K] pre-BatchNorm activations
    const float* coef;         // [4][K]: a | b | mean | rstd
    const float* fin;          // [2][K]: sum g1 / M | sum g1 yh / M (zeros for an eval-mode BatchNorm), left by the tile-sum launch
    float slope;
    float* gY;                 // [M, K] out
    // one-launch form (mlp_small_bwd_jobs_kernel): `fin` is written by tile-sum workgroups of the SAME launch.  sync: the job slot's wait
    // block (gridsync.hpp WL_*: replicas of the finished-slab count -- wait for `need` --, exit tickets of the `nprod` product workgroups,
    // the last of which zeroes the block); fail: the sticky word of gridsync.hpp.  sync == nullptr: `fin` was left by an earlier launch.
    unsigned* sync = nullptr;
    unsigned need = 0, nprod = 0;
    unsigned* fail = nullptr;
};
constexpr int GM_PRO_LDS = 6 * GM_PRO_MAXK + 4;     // a | b | mean | rstd | sum g1 / M | sum g1 yh / M | one flag word

template <int WM, int WN, int WR, int WC, bool BNK, bool PRO>
constexpr int gemm_lds_floats() {
    constexpr int BM = 16 * WM * WR, BN = 16 * WN * WC;
    return (PRO ? GM_PRO_LDS : 0) + 2 * BM * (GM_BK + 4) + 2 * (BNK ? BN * (GM_BK + 4) : GM_BK * (BN + 4));
}

// STATS form (crfconv_gemm_stats): the epilogue also leaves BatchNorm statistic records of the tile it holds -- one
// {shift, rows, sum (v - shift), sum (v - shift)^2} tuple per 16-row group and output channel, the layout
// crfconv_bn_coef_from_records combines (Chan, float64) -- so the statistics pass over Y never runs.
template <int WM, int WN, int WR, int WC, bool BNK, bool VEC, bool PRO, bool STATS>
__device__ __forceinline__ void gemm_tile(const float* __restrict__ A, const float* __restrict__ B,
                                          const float* __restrict__ bias, const float* __restrict__ addend,
                                          int M, int N, int K, float* __restrict__ C, const GemmPro& pro,
                                          float* __restrict__ stat_rec, const unsigned bx, const unsigned by,
                                          float* __restrict__ lds /*gemm_lds_floats<...>() floats, 16-byte aligned: the kernel's ONE buffer*/,
                                          const unsigned gridx = 0 /*PRO with pro.sync: row tiles of the job*/) {
    static_assert(WR * WC * WAVE == GM_BLOCK, "four wavefronts");
    static_assert(!PRO || (VEC && !BNK), "the prologue form is the dX product of aligned widths");
    constexpr int BM = 16 * WM * WR, BN = 16 * WN * WC;
    constexpr int LDA = GM_BK + 4;                      // [BM][LDA]: 16-byte fragment reads along k, 8 lanes cover the 32 banks
    constexpr int LDN = GM_BK + 4;                      // BNK: [BN][LDN], read like A
    constexpr int LDK = BN + 4;                         // else: [GM_BK][LDK], scalar reads of rows 4 g + e: banks 16 g apart
    constexpr int TA = BM * LDA, TB = BNK ? BN * LDN : GM_BK * LDK;
    constexpr int NA4 = BM * GM_BK / 4, NB4 = BN * GM_BK / 4;       // float4 per tile
    constexpr int PA = (NA4 + GM_BLOCK - 1) / GM_BLOCK, PB = (NB4 + GM_BLOCK - 1) / GM_BLOCK;
    // a kernel that serves several tile classes (the job forms) owns one LDS buffer of the largest: static arrays here would add up
    float* const sPro = lds;                            // PRO: a | b | mean | rstd | sum g1 / M | sum g1 yh / M
    float (*sA)[TA] = reinterpret_cast<float (*)[TA]>(lds + (PRO ? GM_PRO_LDS : 0));
    float (*sB)[TB] = reinterpret_cast<float (*)[TB]>(lds + (PRO ? GM_PRO_LDS : 0) + 2 * TA);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rr = lane & 15, g = lane >> 4;
    const int wr = wave / WC, wc = wave - wr * WC;
    const int m0 = bx * BM, n0 = by * BN;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    struct Regs { float4 a[PA], b[PB], y[PRO ? PA : 1]; int kc; };
    auto fetch = [&](int kc, Regs& r) {
        r.kc = kc;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int q = threadIdx.x + GM_BLOCK * i;
            r.a[i] = zero4;
            if constexpr (PRO) r.y[i] = zero4;
            if (NA4 % GM_BLOCK != 0 && q >= NA4) continue;
            const int row = m0 + q / (GM_BK / 4), k = kc + 4 * (q % (GM_BK / 4));
            if (row < M) r.a[i] = gm_ld4<VEC>(A + (int64_t)row * K + k, K - k);
            if constexpr (PRO) {
                if (row < M) r.y[i] = gm_ld4<VEC>(pro.Y + (int64_t)row * K + k, K - k);
            }
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int q = threadIdx.x + GM_BLOCK * i;
            r.b[i] = zero4;
            if (NB4 % GM_BLOCK != 0 && q >= NB4) continue;
            if constexpr (BNK) {
                const int n = n0 + q / (GM_BK / 4), k = kc + 4 * (q % (GM_BK / 4));
                if (n < N) r.b[i] = gm_ld4<VEC>(B + (int64_t)n * K + k, K - k);
            } else {
                const int k = kc + q / (BN / 4), n = n0 + 4 * (q % (BN / 4));
                if (k < K) r.b[i] = gm_ld4<VEC>(B + (int64_t)k * N + n, N - n);
            }
        }
    };
    auto park = [&](int buf, const Regs& r) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int q = threadIdx.x + GM_BLOCK * i;
            if (NA4 % GM_BLOCK != 0 && q >= NA4) continue;
            float4 av = r.a[i];
            if constexpr (PRO) {
                const int row = m0 + q / (GM_BK / 4), k = r.kc + 4 * (q % (GM_BK / 4));
                if (row < M && k < K) {                  // (rows / channels past the end stay zero)
                    const float4 ca = *reinterpret_cast<const float4*>(sPro + k), cb = *reinterpret_cast<const float4*>(sPro + GM_PRO_MAXK + k);
                    const float4 mu = *reinterpret_cast<const float4*>(sPro + 2 * GM_PRO_MAXK + k), rs = *reinterpret_cast<const float4*>(sPro + 3 * GM_PRO_MAXK + k);
                    const float4 c2 = *reinterpret_cast<const float4*>(sPro + 4 * GM_PRO_MAXK + k), c3 = *reinterpret_cast<const float4*>(sPro + 5 * GM_PRO_MAXK + k);
                    const float4 yv = r.y[i];
                    float4 g1 = av;
                    g1.x *= fmaf(ca.x, yv.x, cb.x) > 0.f ? 1.f : pro.slope;
                    g1.y *= fmaf(ca.y, yv.y, cb.y) > 0.f ? 1.f : pro.slope;
                    g1.z *= fmaf(ca.z, yv.z, cb.z) > 0.f ? 1.f : pro.slope;
                    g1.w *= fmaf(ca.w, yv.w, cb.w) > 0.f ? 1.f : pro.slope;
                    av.x = ca.x * (g1.x - c2.x - (yv.x - mu.x) * rs.x * c3.x);
                    av.y = ca.y * (g1.y - c2.y - (yv.y - mu.y) * rs.y * c3.y);
                    av.z = ca.z * (g1.z - c2.z - (yv.z - mu.z) * rs.z * c3.z);
                    av.w = ca.w * (g1.w - c2.w - (yv.w - mu.w) * rs.w * c3.w);
                    if (by == 0) *reinterpret_cast<float4*>(pro.gY + (int64_t)row * K + k) = av;
                }
            }
            *reinterpret_cast<float4*>(sA[buf] + (q / (GM_BK / 4)) * LDA + 4 * (q % (GM_BK / 4))) = av;
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int q = threadIdx.x + GM_BLOCK * i;
            if (NB4 % GM_BLOCK != 0 && q >= NB4) continue;
            if constexpr (BNK) *reinterpret_cast<float4*>(sB[buf] + (q / (GM_BK / 4)) * LDN + 4 * (q % (GM_BK / 4))) = r.b[i];
            else *reinterpret_cast<float4*>(sB[buf] + (q / (BN / 4)) * LDK + 4 * (q % (BN / 4))) = r.b[i];
        }
    };

    gf32x4 acc[WM][WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int n = n0 + 16 * (wc * WN + j) + 4 * g;
        float4 bv = zero4;
        if (bias != nullptr) bv = gm_ld4<VEC>(bias + n, N - n);
#pragma unroll
        for (int i = 0; i < WM; ++i) acc[i][j] = gf32x4{bv.x, bv.y, bv.z, bv.w};
    }
    auto compute = [&](int buf) {
        const float* ta = sA[buf] + (16 * wr * WM + rr) * LDA + 4 * g;
        const float* tb = BNK ? sB[buf] + (16 * wc * WN + rr) * LDN + 4 * g : sB[buf] + 4 * g * LDK + 16 * wc * WN + rr;
#pragma unroll
        for (int s = 0; s < GM_BK / 16; ++s) {
            float av[WM][4], wv[WN][4];
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                const float4 a4 = *reinterpret_cast<const float4*>(ta + 16 * i * LDA + 16 * s);
                av[i][0] = a4.x; av[i][1] = a4.y; av[i][2] = a4.z; av[i][3] = a4.w;
            }
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                if constexpr (BNK) {
                    const float4 w4 = *reinterpret_cast<const float4*>(tb + 16 * j * LDN + 16 * s);
                    wv[j][0] = w4.x; wv[j][1] = w4.y; wv[j][2] = w4.z; wv[j][3] = w4.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) wv[j][e] = tb[(16 * s + e) * LDK + 16 * j];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j][e], av[i][e], acc[i][j], 0, 0, 0);
        }
    };

    const int nchunk = (K + GM_BK - 1) / GM_BK;
    // GM_PF register sets: chunks c + 1 .. c + GM_PF - 1 sit in registers (or are on their way) while chunk c is on the matrix
    // pipe, chunk c + GM_PF is requested as soon as the set of chunk c has been parked.  A 512-input product is sixteen dependent
    // LDS hand-overs; with two sets (rounds 1-4) each of them also was a memory round trip.
    Regs rs[GM_PF];
#pragma unroll
    for (int d = 0; d < GM_PF; ++d)
        if (d == 0 || d < nchunk) fetch(GM_BK * d, rs[d]);
    if constexpr (PRO) {                                 // behind the first operand loads' issue: channel coefficients into LDS
        for (int k = threadIdx.x; k < K; k += GM_BLOCK) {
            sPro[k] = pro.coef[k];
            sPro[GM_PRO_MAXK + k] = pro.coef[K + k];
            sPro[2 * GM_PRO_MAXK + k] = pro.coef[2 * K + k];
            sPro[3 * GM_PRO_MAXK + k] = pro.coef[3 * K + k];
        }
        if (pro.sync == nullptr) {
            for (int k = threadIdx.x; k < K; k += GM_BLOCK) {
                sPro[4 * GM_PRO_MAXK + k] = pro.fin[k];
                sPro[5 * GM_PRO_MAXK + k] = pro.fin[K + k];
            }
        } else {
            // the two channel means come from tile-sum workgroups of this launch (lower workgroup indices: dispatched before this one, they
            // wait for nobody): thread 0 polls its replica of the job's slab count, bounded like the grid barrier's spin
            int* const s_ok = reinterpret_cast<int*>(sPro + 6 * GM_PRO_MAXK);
            if (threadIdx.x == 0) {
                int ok = 1;
                unsigned spins = 0;
                const unsigned* rep = pro.sync + ((bx + by * 7u) % (unsigned)WL_REPL) * FW_LINE;
                while (__hip_atomic_load(rep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < pro.need) {
                    __builtin_amdgcn_s_sleep(GM_WAIT_SLEEP_);
                    if (++spins > FW_SPIN_LIMIT) {
                        __hip_atomic_store(pro.fail, 0x300u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = 0;
                        break;
                    }
                }
                *s_ok = ok;
            }
            __syncthreads();
            const bool ok = *s_ok != 0;
            const __amdgpu_buffer_rsrc_t fr = make_rsrc(pro.fin, 2 * K * 4);         // written write-through on another CU: read past L2
            for (int k = threadIdx.x; k < K; k += GM_BLOCK) {
                const float c2 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(fr, 4 * k, 0, 16));
                const float c3 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(fr, 4 * (K + k), 0, 16));
                sPro[4 * GM_PRO_MAXK + k] = ok ? c2 : __builtin_nanf("");            // a spin that gave up poisons this workgroup's tiles
                sPro[5 * GM_PRO_MAXK + k] = c3;
            }
        }
        __syncthreads();
    }
    park(0, rs[0]);
    __syncthreads();
    // iteration c = cb + d: chunk c sits in LDS buffer d & 1 (GM_PF is even), its register set d is free again and takes chunk
    // c + GM_PF; the set of chunk c + 1 is parked behind the products of chunk c
    for (int cb = 0; cb < nchunk; cb += GM_PF) {
#pragma unroll
        for (int d = 0; d < GM_PF; ++d) {
            const int c = cb + d;
            if (c >= nchunk) break;
            if (c + GM_PF < nchunk) fetch(GM_BK * (c + GM_PF), rs[d]);
            compute(d & 1);
            if (c + 1 < nchunk) {
                park((d + 1) & 1, rs[(d + 1) % GM_PF]);   // that buffer's readers passed the barrier that ended chunk c - 1
                __syncthreads();
            }
        }
    }
    if constexpr (STATS) {
        static_assert(!STATS || (VEC && WM == 1), "statistic records: one 16-row group per wavefront, aligned widths");
        // lane (rr, g) holds rows rr of the group and channels 4 g .. 4 g + 3 of each of its WN tiles: fold the 16 row lanes
        const int row0 = m0 + 16 * wr * WM;                 // first row of this wavefront's group
        const int nrows = row0 < M ? (M - row0 < 16 ? M - row0 : 16) : 0;
        const bool rvalid = rr < nrows;
        const int rec = (int)bx * WR + wr;          // record = 16-row group
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int n = n0 + 16 * (wc * WN + j) + 4 * g;
            float sv[4], s1[4], s2[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sv[e] = __shfl(acc[0][j][e], lane & 0x30, WAVE);            // the group's first row: a sample as shift
                const float d = rvalid ? acc[0][j][e] - sv[e] : 0.f;
                s1[e] = d;
                s2[e] = d * d;
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) {
                    s1[e] += __shfl_xor(s1[e], o, WAVE);
                    s2[e] += __shfl_xor(s2[e], o, WAVE);
                }
            }
            if (rr == 0 && n < N) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    *reinterpret_cast<float4*>(stat_rec + ((int64_t)rec * N + n + e) * 4) = make_float4(sv[e], (float)nrows, s1[e], s2[e]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int row = m0 + 16 * (wr * WM + i) + rr;
        if (row >= M) continue;
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int n = n0 + 16 * (wc * WN + j) + 4 * g;
            if (n >= N) continue;
            float4 o = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            if (addend != nullptr) {
                const float4 a4 = gm_ld4<VEC>(addend + (int64_t)row * N + n, N - n);
                o.x += a4.x; o.y += a4.y; o.z += a4.z; o.w += a4.w;
            }
            float* cp = C + (int64_t)row * N + n;
            if constexpr (VEC) {
                *reinterpret_cast<float4*>(cp) = o;
            } else {
                cp[0] = o.x;
                if (n + 1 < N) cp[1] = o.y;
                if (n + 2 < N) cp[2] = o.z;
                if (n + 3 < N) cp[3] = o.w;
            }
        }
    }
    if constexpr (PRO) {
        // exit tickets in two levels (a burst on one word is served one by one, ~7 ns each); the job's last product workgroup out zeroes
        // the replicas for the next launch -- every other one has left its wait by then
        if (pro.sync != nullptr && threadIdx.x == 0) {
            const unsigned local = bx + by * gridx, g = local % (unsigned)WL_GROUPS;
            const unsigned n_in_group = pro.nprod / WL_GROUPS + (g < pro.nprod % WL_GROUPS ? 1u : 0u);
            unsigned* gt = pro.sync + (WL_REPL + g) * FW_LINE;
            if (__hip_atomic_fetch_add(gt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == n_in_group) {
                __hip_atomic_store(gt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned* top = pro.sync + (WL_REPL + WL_GROUPS) * FW_LINE;
                const unsigned groups = pro.nprod < (unsigned)WL_GROUPS ? pro.nprod : (unsigned)WL_GROUPS;
                if (__hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == groups) {
                    __hip_atomic_store(top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    for (int r = 0; r < WL_REPL; ++r) __hip_atomic_store(pro.sync + r * FW_LINE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
}

template <int WM, int WN, int WR, int WC, bool BNK, bool VEC, bool PRO = false, bool STATS = false>
__global__ __launch_bounds__(GM_BLOCK) void gemm_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                        const float* __restrict__ bias, const float* __restrict__ addend,
                                                        int M, int N, int K, float* __restrict__ C, const GemmPro pro = GemmPro(),
                                                        float* __restrict__ stat_rec = nullptr) {
    __shared__ __attribute__((aligned(16))) float lds[gemm_lds_floats<WM, WN, WR, WC, BNK, PRO>()];
    gemm_tile<WM, WN, WR, WC, BNK, VEC, PRO, STATS>(A, B, bias, addend, M, N, K, C, pro, stat_rec, blockIdx.x, blockIdx.y, lds);
}

// SEVERAL products C_j = A_j B_j (B [K, N], aligned widths, 32 x 32 tiles) in one launch: the tiles of the jobs are laid end to
// end, a workgroup finds its job by a scan over the (<= 8) prefix entries -- crfconv_gemm_jobs, the g_h1 = g_h2 W2 products of all
// wide PointConv layers of a backward pass.
constexpr int GJ_MAX = 8;
struct GemmJobs {
    const float* A[GJ_MAX]; const float* B[GJ_MAX]; float* C[GJ_MAX];
    int M[GJ_MAX], N[GJ_MAX], K[GJ_MAX], tiles_x[GJ_MAX];
    int tile_base[GJ_MAX + 1];
    int njobs;
};
__global__ __launch_bounds__(GM_BLOCK) void gemm_jobs_kernel(const GemmJobs t) {
    int j = 0;
    while (j + 1 < t.njobs && t.tile_base[j + 1] <= (int)blockIdx.x) ++j;
    const unsigned local = blockIdx.x - (unsigned)t.tile_base[j];
    __shared__ __attribute__((aligned(16))) float lds[gemm_lds_floats<1, 1, 2, 2, false, false>()];
    gemm_tile<1, 1, 2, 2, false, true, false, false>(t.A[j], t.B[j], nullptr, nullptr, t.M[j], t.N[j], t.K[j], t.C[j], GemmPro(), nullptr,
                                                     local % (unsigned)t.tiles_x[j], local / (unsigned)t.tiles_x[j], lds);
}

// INDEPENDENT coarse-level MLP blocks in one launch (round 5): the STATS product of each job (crfconv_gemm_stats_jobs) and the PRO
// product of each job's backward (crfconv_mlp_small_backward_jobs).  A coarse launch is a dependent chain of memory round trips on
// a grid that covers a fraction of the chip; two blocks whose inputs are both ready -- unary_nn[i] / pairwise_nn[i] of a CRF layer
// (models/continuous_crf_conv_big.py:56-60), shortcut / lin_in of a strided ResNet block (models/point_conv_big.py:79-88) -- run side
// by side for the price of the longer one.  Same tiles, same summation order as the one-job launches.
constexpr int GG_MAX = 4;
// wide[j]: the job's tiles are 32 rows x 64 columns (a wavefront owns 16 x 32: the A fragment feeds two products, half the
// workgroups pay the load -> LDS -> store chain) -- the flop-heavy layers (GM_WIDE_MIN_N_ columns and GM_WIDE_MIN_TILES_ 32 x 32 tiles).
#ifndef GM_WIDE_MIN_N_
#define GM_WIDE_MIN_N_ 128
#endif
#ifndef GM_WIDE_MIN_TILES_
#define GM_WIDE_MIN_TILES_ 1024
#endif
static inline bool gm_wide(int64_t M, int N) { return N >= GM_WIDE_MIN_N_ && ((M + 31) / 32) * (int64_t)((N + 31) / 32) >= GM_WIDE_MIN_TILES_; }
struct GemmStatsJobs {
    const float* A[GG_MAX]; const float* B[GG_MAX]; float* C[GG_MAX]; float* rec[GG_MAX];
    int M[GG_MAX], N[GG_MAX], K[GG_MAX], tiles_x[GG_MAX], wide[GG_MAX];
    int tile_base[GG_MAX + 1];
    int njobs;
};
__global__ __launch_bounds__(GM_BLOCK) void gemm_stats_jobs_kernel(const GemmStatsJobs t) {
    int j = 0;
    while (j + 1 < t.njobs && t.tile_base[j + 1] <= (int)blockIdx.x) ++j;
    const unsigned local = blockIdx.x - (unsigned)t.tile_base[j];
    const unsigned tx = (unsigned)uni(t.tiles_x[j]);
    const unsigned bx = (unsigned)uni((int)(local % tx)), by = (unsigned)uni((int)(local / tx));
    __shared__ __attribute__((aligned(16))) float lds[gemm_lds_floats<1, 2, 2, 2, true, false>()];
    if (uni(t.wide[j]))
        gemm_tile<1, 2, 2, 2, true, true, false, true>(uni(t.A[j]), uni(t.B[j]), nullptr, nullptr, uni(t.M[j]), uni(t.N[j]), uni(t.K[j]), uni(t.C[j]), GemmPro(),
                                                       uni(t.rec[j]), bx, by, lds);
    else
        gemm_tile<1, 1, 2, 2, true, true, false, true>(uni(t.A[j]), uni(t.B[j]), nullptr, nullptr, uni(t.M[j]), uni(t.N[j]), uni(t.K[j]), uni(t.C[j]), GemmPro(),
                                                       uni(t.rec[j]), bx, by, lds);
}
struct GemmProJobs {
    const float* A[GG_MAX]; const float* B[GG_MAX]; const float* addend[GG_MAX]; float* C[GG_MAX];
    GemmPro pro[GG_MAX];
    int M[GG_MAX], N[GG_MAX], K[GG_MAX], tiles_x[GG_MAX], wide[GG_MAX];
    int tile_base[GG_MAX + 1];
    int njobs;
};
__global__ __launch_bounds__(GM_BLOCK) void gemm_pro_jobs_kernel(const GemmProJobs t) {
    int j = 0;
    while (j + 1 < t.njobs && t.tile_base[j + 1] <= (int)blockIdx.x) ++j;
    const unsigned local = blockIdx.x - (unsigned)t.tile_base[j];
    GemmPro pro;                                        // the job's entry, pinned into scalar registers
    pro.Y = uni(t.pro[j].Y); pro.coef = uni(t.pro[j].coef); pro.fin = uni(t.pro[j].fin); pro.slope = uni(t.pro[j].slope); pro.gY = uni(t.pro[j].gY);
    const unsigned tx = (unsigned)uni(t.tiles_x[j]);
    const unsigned bx = (unsigned)uni((int)(local % tx)), by = (unsigned)uni((int)(local / tx));
    __shared__ __attribute__((aligned(16))) float lds[gemm_lds_floats<1, 2, 2, 2, false, true>()];
    if (uni(t.wide[j]))
        gemm_tile<1, 2, 2, 2, false, true, true, false>(uni(t.A[j]), uni(t.B[j]), nullptr, uni(t.addend[j]), uni(t.M[j]), uni(t.N[j]), uni(t.K[j]), uni(t.C[j]),
                                                        pro, nullptr, bx, by, lds, tx);
    else
        gemm_tile<1, 1, 2, 2, false, true, true, false>(uni(t.A[j]), uni(t.B[j]), nullptr, uni(t.addend[j]), uni(t.M[j]), uni(t.N[j]), uni(t.K[j]), uni(t.C[j]),
                                                        pro, nullptr, bx, by, lds, tx);
}

}  // namespace crf

namespace crf {

// Row-tile partials of the two channel sums of a BatchNorm backward: partial[tile][0][c] = sum g1, [1][c] = sum g1 yh over the tile's
// BT_ROWS rows (g1 = gA lrelu'(a y + b), yh = (y - mean) rstd).  A workgroup = one tile x 64 channels: 16 lanes x 16 bytes per row
// (whole 256-byte row segments), 16 rows per pass; float32 inside a thread's eight rows, float64 across the 16 row threads.
constexpr int BT_ROWS = 128, BT_CH = 64, BT_NR = BT_ROWS / 16;
// what the LAST workgroup of a column slab leaves for the product launch and the caller
struct TileSumFin {
    unsigned* ticket;          // this slab's ticket word (zero; left zero)
    int ntile;                 // workgroups (row tiles) of the slab
    int training;
    float inv_m;
    float* fin;                // [2][K]
    float* dgamma;             // [K]
    float* dbeta;              // [K]
    unsigned* done = nullptr;  // one-launch form: the job's slab count (GemmPro::sync[0]); `fin` then goes out write-through
};
struct TileSumLds { float red[16][2][64]; int last; };
typedef unsigned int bt_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void bn_bwd_tile_sums_body(const float* __restrict__ gA, const float* __restrict__ Y,
                                                       const float* __restrict__ coef, int M, int K, int tile_rows,
                                                       float slope, double* __restrict__ partial, const int bx, const int by,
                                                       const TileSumFin f, TileSumLds& L) {
    float (&s_red)[16][2][64] = L.red;
    int& s_last = L.last;
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = by * BT_CH + 4 * cq;
    const int row0 = bx * tile_rows;
    const int row_end = row0 + tile_rows < M ? row0 + tile_rows : M;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    if (c < K) {
        const float4 ca = *reinterpret_cast<const float4*>(coef + c), cb = *reinterpret_cast<const float4*>(coef + K + c);
        const float4 mu = *reinterpret_cast<const float4*>(coef + 2 * K + c), rs = *reinterpret_cast<const float4*>(coef + 3 * K + c);
        for (int base = row0; base < row_end; base += BT_ROWS) {      // one trip at <= 3072 rows (tiles of 128 rows)
            float4 g[BT_NR], y[BT_NR];                       // every load of a trip in flight at once: one round trip
#pragma unroll
            for (int u = 0; u < BT_NR; ++u) {
                const int r = base + rl + 16 * u;
                const bool in = r < row_end;
                g[u] = in ? *reinterpret_cast<const float4*>(gA + (int64_t)r * K + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                y[u] = in ? *reinterpret_cast<const float4*>(Y + (int64_t)r * K + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < BT_NR; ++u) {                // (rows past the end hold g = 0: they add nothing)
                float4 v = g[u];
                v.x *= fmaf(ca.x, y[u].x, cb.x) > 0.f ? 1.f : slope;
                v.y *= fmaf(ca.y, y[u].y, cb.y) > 0.f ? 1.f : slope;
                v.z *= fmaf(ca.z, y[u].z, cb.z) > 0.f ? 1.f : slope;
                v.w *= fmaf(ca.w, y[u].w, cb.w) > 0.f ? 1.f : slope;
                s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
                s2.x = fmaf(v.x, (y[u].x - mu.x) * rs.x, s2.x); s2.y = fmaf(v.y, (y[u].y - mu.y) * rs.y, s2.y);
                s2.z = fmaf(v.z, (y[u].z - mu.z) * rs.z, s2.z); s2.w = fmaf(v.w, (y[u].w - mu.w) * rs.w, s2.w);
            }
        }
    }
    *reinterpret_cast<float4*>(&s_red[rl][0][4 * cq]) = s1;
    *reinterpret_cast<float4*>(&s_red[rl][1][4 * cq]) = s2;
    __syncthreads();
    // partial [ntile][2][K] doubles, stored write-through: the slab's last workgroup (another CU, maybe another XCD) reads them
    const __amdgpu_buffer_rsrc_t pr = make_rsrc(partial, f.ntile * 2 * K * 8);
    const int which = threadIdx.x / BT_CH, ch = threadIdx.x - which * BT_CH;
    const bool mine = threadIdx.x < 2 * BT_CH && by * BT_CH + ch < K;
    if (mine) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += (double)s_red[w][which][ch];           // fixed order
        const unsigned long long bits = (unsigned long long)__double_as_longlong(t);
        const bt_u32x2 u = {(unsigned)bits, (unsigned)(bits >> 32)};
        __builtin_amdgcn_raw_buffer_store_b64(u, pr, ((bx * 2 + which) * K + by * BT_CH + ch) * 8, 0, 16);
    }
    // "the last workgroup finishes" (gridsync.hpp), per column slab: its row tiles draw tickets on the slab's word
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(f.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old + 1 == (unsigned)f.ntile;
        if (last) __hip_atomic_store(f.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    if (mine) {
    // the row-tile partials of this thread's (sum, channel) in tile order, all loads of a 24-tile round in flight (one round at <= 3072
    // rows): the order -- hence every bit -- of the sums the product launch used to form per workgroup
    const int cg = by * BT_CH + ch;
    double tot = 0.0;
    for (int t0 = 0; t0 < f.ntile; t0 += 24) {
        double v[24];
#pragma unroll
        for (int u = 0; u < 24; ++u) {
            const int tt = t0 + u < f.ntile ? t0 + u : f.ntile - 1;
            const bt_u32x2 w2 = __builtin_amdgcn_raw_buffer_load_b64(pr, ((tt * 2 + which) * K + cg) * 8, 0, 16);
            const double d = __longlong_as_double((long long)(((unsigned long long)w2.y << 32) | w2.x));
            v[u] = t0 + u < f.ntile ? d : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 24; ++u) tot += v[u];
    }
    (which == 0 ? f.dbeta : f.dgamma)[cg] = (float)tot;
    const float mean = f.training ? (float)(tot * (double)f.inv_m) : 0.f;
    if (f.done == nullptr) f.fin[which * K + cg] = mean;
    else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(mean), make_rsrc(f.fin, 2 * K * 4), 4 * (which * K + cg), 0, 16);
    }
    if (f.done != nullptr) {                             // the product workgroups of this launch wait for the job's slabs
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x < WL_REPL) __hip_atomic_fetch_add(f.done + threadIdx.x * FW_LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

struct TileSumJobs {
    const float* gA[GG_MAX]; const float* Y[GG_MAX]; const float* coef[GG_MAX]; double* partial[GG_MAX];
    float* fin[GG_MAX]; float* dgamma[GG_MAX]; float* dbeta[GG_MAX];
    int M[GG_MAX], K[GG_MAX], tile_rows[GG_MAX], ntile[GG_MAX], training[GG_MAX];
    float slope[GG_MAX], inv_m[GG_MAX];
    unsigned* ticket;          // BT_SLABS lines per job
    int blk_base[GG_MAX + 1];
    int njobs;
};
constexpr int BT_SLABS = GM_PRO_MAXK / BT_CH;      // column slabs of a job at most (8): job j, slab s draws on ticket line 8 j + s
__device__ __forceinline__ void bn_bwd_tile_sums_job(const TileSumJobs& t, const int blk, TileSumLds& L, unsigned* const* done = nullptr) {
    int j = 0;
    while (j + 1 < t.njobs && t.blk_base[j + 1] <= blk) ++j;
    const int local = blk - t.blk_base[j];
    const int ntile = uni(t.ntile[j]);
    const int bx = local % ntile, by = local / ntile;
    TileSumFin f;
    f.ticket = uni(t.ticket) + (j * BT_SLABS + by) * FW_LINE; f.ntile = ntile; f.training = uni(t.training[j]); f.inv_m = uni(t.inv_m[j]);
    f.fin = uni(t.fin[j]); f.dgamma = uni(t.dgamma[j]); f.dbeta = uni(t.dbeta[j]);
    f.done = done != nullptr ? uni(done[j]) : nullptr;
    bn_bwd_tile_sums_body(uni(t.gA[j]), uni(t.Y[j]), uni(t.coef[j]), uni(t.M[j]), uni(t.K[j]), uni(t.tile_rows[j]), uni(t.slope[j]), uni(t.partial[j]),
                          bx, by, f, L);
}
__global__ __launch_bounds__(256) void bn_bwd_tile_sums_jobs_kernel(const TileSumJobs t) {
    __shared__ TileSumLds L;
    bn_bwd_tile_sums_job(t, (int)blockIdx.x, L);
}

// BOTH launches of crfconv_mlp_small_backward_jobs as ONE (round 6): workgroups 0 .. nsum - 1 are the tile-sum workgroups, the rest the
// product's.  A product workgroup requests its first operand chunks, then waits until its job's column slabs have left the two channel
// means (GemmPro::sync) -- the tile-sum workgroups have the lower indices, are dispatched first and wait for nobody, so the wait ends
// whatever part of the grid is resident.  What it saves is the launch boundary (drain, cache write-back, dispatch: 4-5 us of stream time
// on launches whose own work is 2-6 us, 21 times per training step).  Same arithmetic in the same order: results are bit-identical to
// the two launches'.
struct SmallBwdSync { unsigned* done[GG_MAX]; int nsum; };
__global__ __launch_bounds__(GM_BLOCK) void mlp_small_bwd_jobs_kernel(const TileSumJobs ts, const GemmProJobs t, const SmallBwdSync sy) {
    __shared__ __attribute__((aligned(16))) float lds[gemm_lds_floats<1, 2, 2, 2, false, true>()];
    static_assert(sizeof(TileSumLds) <= sizeof(lds), "the tile-sum workgroups use the product's buffer");
    if ((int)blockIdx.x < sy.nsum) {
        bn_bwd_tile_sums_job(ts, (int)blockIdx.x, *reinterpret_cast<TileSumLds*>(lds), sy.done);
        return;
    }
    const int blk = (int)blockIdx.x - sy.nsum;
    int j = 0;
    while (j + 1 < t.njobs && t.tile_base[j + 1] <= blk) ++j;
    const unsigned local = (unsigned)(blk - t.tile_base[j]);
    GemmPro pro;
    pro.Y = uni(t.pro[j].Y); pro.coef = uni(t.pro[j].coef); pro.fin = uni(t.pro[j].fin); pro.slope = uni(t.pro[j].slope); pro.gY = uni(t.pro[j].gY);
    pro.sync = uni(t.pro[j].sync); pro.need = (unsigned)uni((int)t.pro[j].need); pro.nprod = (unsigned)uni((int)t.pro[j].nprod); pro.fail = uni(t.pro[j].fail);
    const unsigned tx = (unsigned)uni(t.tiles_x[j]);
    const unsigned bx = (unsigned)uni((int)(local % tx)), by = (unsigned)uni((int)(local / tx));
    if (uni(t.wide[j]))
        gemm_tile<1, 2, 2, 2, false, true, true, false>(uni(t.A[j]), uni(t.B[j]), nullptr, uni(t.addend[j]), uni(t.M[j]), uni(t.N[j]), uni(t.K[j]), uni(t.C[j]),
                                                        pro, nullptr, bx, by, lds, tx);
    else
        gemm_tile<1, 1, 2, 2, false, true, true, false>(uni(t.A[j]), uni(t.B[j]), nullptr, uni(t.addend[j]), uni(t.M[j]), uni(t.N[j]), uni(t.K[j]), uni(t.C[j]),
                                                        pro, nullptr, bx, by, lds, tx);
}
}  // namespace crf

extern "C" int crfconv_mlp_small_backward_supported(int64_t M, int Ci, int Co) {
    return (M >= 1 && M < (int64_t)1 << 24 && Ci >= 4 && Co >= 4 && Ci % 4 == 0 && Co % 4 == 0 && Co <= crf::GM_PRO_MAXK) ? 1 : 0;
}

// at most BT_MAXTILES row tiles of 128 rows or proportionally more.  The slab's last workgroup sums them (24 loads in flight per
// round), so the cap trades rounds of that sum against trips of every tile's row loop: 24 tiles was the cap while every workgroup of
// the PRODUCT re-summed them; with the finished means 48 measured 3.836 ms against 3.868 (96 and 192: the same) -- DESIGN 9, T1.
static void bt_plan(int64_t M, int& ntile, int& tile_rows) {
    constexpr int BT_MAXTILES = 48;
    int64_t rows = crf::BT_ROWS;
    if ((M + rows - 1) / rows > BT_MAXTILES) rows = (((M + BT_MAXTILES - 1) / BT_MAXTILES + 15) / 16) * 16;
    tile_rows = (int)rows;
    ntile = (int)((M + rows - 1) / rows);
}

// partials [ntile][2][Co] doubles | the finished means [2][Co] floats (each 256-byte aligned)
static size_t bt_fin_offset(int Co, int ntile) { return (sizeof(double) * 2 * (size_t)Co * (size_t)ntile + 255) & ~(size_t)255; }
extern "C" size_t crfconv_mlp_small_backward_workspace(int64_t M, int Co) {
    if (M < 1 || Co < 1) return 0;
    int ntile, tile_rows;
    bt_plan(M, ntile, tile_rows);
    return bt_fin_offset(Co, ntile) + sizeof(float) * 2 * (size_t)Co + 512;
}

// Backward of one coarse-level MLP block A = lrelu(BN(X W^T), slope) behind its one-launch forward (crfconv_mlp_small_forward),
// in TWO launches: row-tile partials of the channel sums, then dX [M, Ci] = gY W (+ addend) with gY [M, Co] formed in the
// product's operand load (and stored for the weight gradient), dgamma / dbeta on the way.  Same results as
// crfconv_bn_backward followed by crfconv_gemm up to summation order.  coef: the [4][Co] block of the forward.
extern "C" int crfconv_mlp_small_backward(const float* gA, const float* Y, const float* coef, const float* W, const float* addend,
                                          int64_t M, int Ci, int Co, int training, float slope, float* gY, float* dX, float* dgamma,
                                          float* dbeta, void* workspace, size_t workspace_bytes, unsigned* ticket, void* stream) {
    crf_mlp_bwd_job job;
    job.gA = gA; job.Y = Y; job.coef = coef; job.W = W; job.addend = addend; job.M = M; job.Ci = Ci; job.Co = Co; job.training = training;
    job.slope = slope; job.gY = gY; job.dX = dX; job.dgamma = dgamma; job.dbeta = dbeta; job.workspace = workspace; job.workspace_bytes = workspace_bytes;
    return crfconv_mlp_small_backward_jobs(&job, 1, ticket, stream);
}

// The backward of up to 4 INDEPENDENT coarse-level MLP blocks (crf_mlp_bwd_job: the arguments of crfconv_mlp_small_backward per
// block) in TWO launches for all of them: the row-tile sums of every block, then every block's dX product.  Results per block are
// bit-identical to crfconv_mlp_small_backward's.
static int mlp_small_backward_jobs_impl(const crf_mlp_bwd_job* jobs, int njobs, unsigned* ticket, unsigned* sync_ws, void* stream);
extern "C" int crfconv_mlp_small_backward_jobs(const crf_mlp_bwd_job* jobs, int njobs, unsigned* ticket, void* stream) {
    return mlp_small_backward_jobs_impl(jobs, njobs, ticket, nullptr, stream);
}
// The same as ONE launch (mlp_small_bwd_jobs_kernel): the product's workgroups wait inside the launch for the tile-sum workgroups of
// their job.  sync_ws: the barrier words of crfconv_gridsync_workspace() (zero, left zero; its sticky failure word reports a wait that
// gave up -- that launch's dX / gY are NaN).  Bit-identical results.
extern "C" int crfconv_mlp_small_backward_jobs_one_launch(const crf_mlp_bwd_job* jobs, int njobs, unsigned* ticket, unsigned* sync_ws, void* stream) {
    CRF_REQUIRE(sync_ws != nullptr, CRF_ERR_ARG, "null barrier workspace");
    return mlp_small_backward_jobs_impl(jobs, njobs, ticket, sync_ws, stream);
}
static int mlp_small_backward_jobs_impl(const crf_mlp_bwd_job* jobs, int njobs, unsigned* ticket, unsigned* sync_ws, void* stream) {
    CRF_REQUIRE(jobs && ticket && njobs >= 1 && njobs <= crf::GG_MAX, CRF_ERR_ARG, "1 .. %d jobs (got %d) and the ticket words", crf::GG_MAX, njobs);
    crf::TileSumJobs ts;
    crf::GemmProJobs gp;
    crf::SmallBwdSync sy;
    for (int j = 0; j < crf::GG_MAX; ++j) sy.done[j] = nullptr;
    sy.nsum = 0;
    int64_t blocks = 0, tiles = 0;
    for (int j = 0; j <= crf::GG_MAX; ++j) {
        ts.blk_base[j] = (int)blocks;
        gp.tile_base[j] = (int)tiles;
        if (j >= crf::GG_MAX) break;
        if (j >= njobs) {
            ts.gA[j] = nullptr; ts.Y[j] = nullptr; ts.coef[j] = nullptr; ts.partial[j] = nullptr; ts.M[j] = 0; ts.K[j] = 4; ts.tile_rows[j] = crf::BT_ROWS;
            ts.ntile[j] = 1; ts.slope[j] = 1.f; ts.fin[j] = nullptr; ts.dgamma[j] = nullptr; ts.dbeta[j] = nullptr; ts.training[j] = 0; ts.inv_m[j] = 0.f;
            gp.A[j] = nullptr; gp.B[j] = nullptr; gp.addend[j] = nullptr; gp.C[j] = nullptr; gp.pro[j] = crf::GemmPro(); gp.M[j] = 0; gp.N[j] = 4; gp.K[j] = 4;
            gp.tiles_x[j] = 1; gp.wide[j] = 0;
            continue;
        }
        const crf_mlp_bwd_job& b = jobs[j];
        CRF_REQUIRE(b.gA && b.Y && b.coef && b.W && b.gY && b.dX && b.dgamma && b.dbeta && b.workspace, CRF_ERR_ARG, "job %d: null pointer", j);
        CRF_REQUIRE(crfconv_mlp_small_backward_supported(b.M, b.Ci, b.Co), CRF_ERR_UNSUPPORTED,
                    "job %d: mlp_small_backward %lld x %d -> %d: widths must be multiples of 4, Co <= %d", j, (long long)b.M, b.Ci, b.Co, crf::GM_PRO_MAXK);
        CRF_REQUIRE(b.workspace_bytes >= crfconv_mlp_small_backward_workspace(b.M, b.Co), CRF_ERR_WORKSPACE, "job %d: workspace too small", j);
        double* partial = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(b.workspace) + 255) & ~(uintptr_t)255);
        int ntile, tile_rows;
        bt_plan(b.M, ntile, tile_rows);
        ts.gA[j] = b.gA; ts.Y[j] = b.Y; ts.coef[j] = b.coef; ts.partial[j] = partial; ts.M[j] = (int)b.M; ts.K[j] = b.Co; ts.tile_rows[j] = tile_rows;
        float* fin = reinterpret_cast<float*>(reinterpret_cast<char*>(partial) + bt_fin_offset(b.Co, ntile));
        ts.ntile[j] = ntile; ts.slope[j] = b.slope; ts.fin[j] = fin; ts.dgamma[j] = b.dgamma; ts.dbeta[j] = b.dbeta; ts.training[j] = b.training;
        ts.inv_m[j] = (float)(1.0 / (double)b.M);
        blocks += (int64_t)ntile * ((b.Co + crf::BT_CH - 1) / crf::BT_CH);
        crf::GemmPro pro;
        pro.Y = b.Y; pro.coef = b.coef; pro.fin = fin; pro.slope = b.slope; pro.gY = b.gY;
        gp.A[j] = b.gA; gp.B[j] = b.W; gp.addend[j] = b.addend; gp.C[j] = b.dX; gp.pro[j] = pro; gp.M[j] = (int)b.M; gp.N[j] = b.Ci; gp.K[j] = b.Co;
        gp.tiles_x[j] = (int)((b.M + 31) / 32);
        gp.wide[j] = crf::gm_wide(b.M, b.Ci) ? 1 : 0;
        const int64_t jt = (int64_t)gp.tiles_x[j] * ((b.Ci + (gp.wide[j] ? 63 : 31)) / (gp.wide[j] ? 64 : 32));
        tiles += jt;
        CRF_REQUIRE(tiles + blocks < ((int64_t)1 << 31), CRF_ERR_UNSUPPORTED, "too many tiles in one batch");
        if (sync_ws != nullptr) {                       // the job slot's wait block behind the barrier words
            gp.pro[j].sync = sync_ws + (crf::FW_WAIT + j * crf::WL_LINES) * crf::FW_LINE;
            gp.pro[j].need = (unsigned)((b.Co + crf::BT_CH - 1) / crf::BT_CH);
            gp.pro[j].nprod = (unsigned)jt;
            gp.pro[j].fail = sync_ws + crf::FW_FAIL * crf::FW_LINE;
            sy.done[j] = gp.pro[j].sync;
        }
    }
    ts.njobs = njobs;
    ts.ticket = ticket;
    gp.njobs = njobs;
    hipStream_t st = crf::as_stream(stream);
    if (sync_ws != nullptr) {
        sy.nsum = (int)blocks;
        hipLaunchKernelGGL(crf::mlp_small_bwd_jobs_kernel, dim3((unsigned)(blocks + tiles)), dim3(crf::GM_BLOCK), 0, st, ts, gp, sy);
        CRF_LAUNCH_CHECK();
        return CRF_OK;
    }
    hipLaunchKernelGGL(crf::bn_bwd_tile_sums_jobs_kernel, dim3((unsigned)blocks), dim3(256), 0, st, ts);
    CRF_LAUNCH_CHECK();
    hipLaunchKernelGGL(crf::gemm_pro_jobs_kernel, dim3((unsigned)tiles), dim3(crf::GM_BLOCK), 0, st, gp);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" size_t crfconv_gemm_stat_records(int64_t M) { return M < 1 ? 0 : (size_t)(2 * ((M + 31) / 32)); }

// C [M, N] = A [M, K] B^T (B [N, K]: the F.linear weight) with BatchNorm statistic records of C from the epilogue: stat_rec
// float [crfconv_gemm_stat_records(M)][N][4] = {shift, rows, sum (v - shift), sum (v - shift)^2} per 16-row group and channel,
// to be combined by crfconv_bn_coef_from_records.  N, K multiples of 4.
extern "C" int crfconv_gemm_stats(const float* A, const float* B, int64_t M, int N, int K, float* C, float* stat_rec, void* stream) {
    crf_gemm_stats_job job;
    job.A = A; job.B = B; job.M = M; job.N = N; job.K = K; job.C = C; job.stat_rec = stat_rec;
    return crfconv_gemm_stats_jobs(&job, 1, stream);
}

// crfconv_gemm_stats for up to 4 independent products in ONE launch: C_j = A_j B_j^T with the statistic records of each C_j.
extern "C" int crfconv_gemm_stats_jobs(const crf_gemm_stats_job* jobs, int njobs, void* stream) {
    CRF_REQUIRE(jobs && njobs >= 1 && njobs <= crf::GG_MAX, CRF_ERR_ARG, "1 .. %d jobs (got %d)", crf::GG_MAX, njobs);
    crf::GemmStatsJobs t;
    int64_t tiles = 0;
    for (int j = 0; j <= crf::GG_MAX; ++j) {
        t.tile_base[j] = (int)tiles;
        if (j >= crf::GG_MAX) break;
        if (j >= njobs) {
            t.A[j] = nullptr; t.B[j] = nullptr; t.C[j] = nullptr; t.rec[j] = nullptr; t.M[j] = 0; t.N[j] = 4; t.K[j] = 4; t.tiles_x[j] = 1; t.wide[j] = 0;
            continue;
        }
        const crf_gemm_stats_job& b = jobs[j];
        CRF_REQUIRE(b.A && b.B && b.C && b.stat_rec, CRF_ERR_ARG, "job %d: null pointer", j);
        CRF_REQUIRE(b.M >= 1 && b.M < ((int64_t)1 << 31) && b.N >= 4 && b.K >= 4 && b.N % 4 == 0 && b.K % 4 == 0, CRF_ERR_UNSUPPORTED,
                    "job %d: gemm_stats %lld x %d x %d: N and K must be multiples of 4", j, (long long)b.M, b.N, b.K);
        t.A[j] = b.A; t.B[j] = b.B; t.C[j] = b.C; t.rec[j] = b.stat_rec; t.M[j] = (int)b.M; t.N[j] = b.N; t.K[j] = b.K;
        t.tiles_x[j] = (int)((b.M + 31) / 32);
        t.wide[j] = crf::gm_wide(b.M, b.N) ? 1 : 0;
        tiles += (int64_t)t.tiles_x[j] * ((b.N + (t.wide[j] ? 63 : 31)) / (t.wide[j] ? 64 : 32));
        CRF_REQUIRE(tiles < ((int64_t)1 << 31), CRF_ERR_UNSUPPORTED, "too many tiles in one batch");
    }
    t.njobs = njobs;
    hipLaunchKernelGGL(crf::gemm_stats_jobs_kernel, dim3((unsigned)tiles), dim3(crf::GM_BLOCK), 0, crf::as_stream(stream), t);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

// C_j [M_j, N_j] = A_j [M_j, K_j] B_j [K_j, N_j] for up to 8 products per launch (jobs: host array); N, K multiples of 4.  Same tiles
// and summation order as crfconv_gemm on each job.
extern "C" int crfconv_gemm_jobs(const crf_gemm_job* jobs, int njobs, void* stream) {
    CRF_REQUIRE(jobs || njobs == 0, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(njobs >= 0, CRF_ERR_ARG, "njobs=%d < 0", njobs);
    hipStream_t st = crf::as_stream(stream);
    for (int j0 = 0; j0 < njobs; j0 += crf::GJ_MAX) {
        crf::GemmJobs t;
        const int n = njobs - j0 < crf::GJ_MAX ? njobs - j0 : crf::GJ_MAX;
        int64_t tiles = 0;
        for (int j = 0; j <= crf::GJ_MAX; ++j) {
            t.tile_base[j] = (int)tiles;
            if (j < n) {
                const crf_gemm_job& jb = jobs[j0 + j];
                CRF_REQUIRE(jb.A && jb.B && jb.C, CRF_ERR_ARG, "job %d: null pointer", j0 + j);
                CRF_REQUIRE(jb.M >= 1 && jb.M < ((int64_t)1 << 31) && jb.N >= 4 && jb.K >= 4 && jb.N % 4 == 0 && jb.K % 4 == 0, CRF_ERR_UNSUPPORTED,
                            "job %d: %lld x %d x %d (N, K multiples of 4)", j0 + j, (long long)jb.M, jb.N, jb.K);
                t.A[j] = jb.A; t.B[j] = jb.B; t.C[j] = jb.C; t.M[j] = (int)jb.M; t.N[j] = jb.N; t.K[j] = jb.K;
                t.tiles_x[j] = (int)((jb.M + 31) / 32);
                tiles += (int64_t)t.tiles_x[j] * ((jb.N + 31) / 32);
                CRF_REQUIRE(tiles < ((int64_t)1 << 31), CRF_ERR_UNSUPPORTED, "too many tiles in one batch");
            } else if (j < crf::GJ_MAX) {
                t.A[j] = nullptr; t.B[j] = nullptr; t.C[j] = nullptr; t.M[j] = 0; t.N[j] = 4; t.K[j] = 4; t.tiles_x[j] = 1;
            }
        }
        t.njobs = n;
        hipLaunchKernelGGL(crf::gemm_jobs_kernel, dim3((unsigned)tiles), dim3(crf::GM_BLOCK), 0, st, t);
        CRF_LAUNCH_CHECK();
    }
    return CRF_OK;
}

extern "C" int crfconv_gemm_supported(int64_t M, int N, int K) {
    return (M >= 1 && M < (int64_t)1 << 31 && N >= 1 && K >= 1 && N < (1 << 24) && K < (1 << 24)) ? 1 : 0;
}

// C [M, N] = A [M, K] · B + bias + addend.  b_is_nk != 0: B is [N, K] row-major (the F.linear weight: C = A Bᵀ),
// else [K, N] row-major (C = A B: dX = gY · W with W [Co, Ci]).  bias [N] and addend [M, N] may be NULL; addend may alias C.
extern "C" int crfconv_gemm(const float* A, const float* B, const float* bias, const float* addend, int64_t M, int N, int K,
                            int b_is_nk, float* C, void* stream) {
    CRF_REQUIRE(A != nullptr && B != nullptr && C != nullptr, CRF_ERR_ARG, "null operand");
    CRF_REQUIRE(crfconv_gemm_supported(M, N, K), CRF_ERR_UNSUPPORTED, "gemm %lld x %d x %d: shape out of range", (long long)M, N, K);
    // tile shapes (rows x columns per workgroup): 64 x 64, 32 x 64, 64 x 32, 32 x 32 -- the largest that still gives the
    // grid `min_blocks` workgroups.  Measured on the shapes of the training step (scratch/gemm_bench.py, graph replays): 32 x 32 is the
    // fastest or within 0.5 us of it from 640 x 64 to 163 840 x 32 -- these launches are latency-bound, many short wavefronts win
    constexpr int min_blocks = 4096;
    auto blocks = [&](int bm, int bn) { return ((M + bm - 1) / bm) * (int64_t)((N + bn - 1) / bn); };
    int shape = 3;
    if (N > 32 && blocks(64, 64) >= min_blocks) shape = 0;
    else if (N > 32 && blocks(32, 64) >= min_blocks) shape = 1;
    else if (blocks(64, 32) >= min_blocks) shape = 2;
    hipStream_t st = crf::as_stream(stream);
    const dim3 blk(crf::GM_BLOCK);
    const bool vec = N % 4 == 0 && K % 4 == 0;           // 16-byte accesses; odd widths (13-class logits) go element by element
#define GM2(WM, WN, WR, WC, NK, V) \
    hipLaunchKernelGGL((crf::gemm_kernel<WM, WN, WR, WC, NK, V>), grid, blk, 0, st, A, B, bias, addend, (int)M, N, K, C)
#define GM(WM, WN, WR, WC)                                                                                                    \
    do {                                                                                                                      \
        const dim3 grid((unsigned)((M + 16 * WM * WR - 1) / (16 * WM * WR)), (unsigned)((N + 16 * WN * WC - 1) / (16 * WN * WC))); \
        if (b_is_nk) { if (vec) GM2(WM, WN, WR, WC, true, true); else GM2(WM, WN, WR, WC, true, false); }                    \
        else { if (vec) GM2(WM, WN, WR, WC, false, true); else GM2(WM, WN, WR, WC, false, false); }                           \
    } while (0)
    switch (shape) {
        case 0: GM(2, 2, 2, 2); break;       // 64 x 64
        case 1: GM(1, 2, 2, 2); break;       // 32 x 64
        case 2: GM(1, 2, 4, 1); break;       // 64 x 32
        default: GM(1, 1, 2, 2); break;      // 32 x 32
    }
#undef GM
#undef GM2
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}
