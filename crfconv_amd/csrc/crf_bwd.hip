// Backward of the CRF mean-field loop + similarity for the fast shapes (K in {16, 32}, k0 = 1): T + 1 launches.
//
// Autograd of models/continuous_crf_conv_big.py:49-54 (similarity) and :63-72 (loop).  With x_t = z Q + (A x_{t-1}) P,
// A = the row-stochastic softmax weights s over the table's columns 1..K-1, and G_t = dL/dx_t (G_T given):
//
//   G_{t-1} = A^T (G_t P^T) = (A^T G_t) P^T         the only sequential part: T sparse products with A^T
//   ds_ik   = sum_t <G_t[i] P^T, x_{t-1}[j(i,k)]>   dP = sum_t (A x_{t-1})^T G_t      dQ = z^T sum_t G_t
//   dz      = G_0 + (sum_t G_t) Q^T                 dy  = softmax / squared-distance backward of ds
//
//   bwd_rev_kernel<CHAIN>  x (T-1)   G_{t-1} = (A^T G_t) P^T for t = T .. 2.  The first one reads the reverse edge list
//                          (rev_eid -> s[e]) and leaves {e, s[e]} records in CSR order for the later walks (8 B per edge,
//                          one coalesced load instead of edge id -> weight gather).
//   bwd_edge_all_kernel    per point, index row / weight row / ds in registers over all T gather passes:
//                          gm_t = G_t P^T, m_t = sum_k s_k x_{t-1}[j_k], ds_k += <gm_t, x_{t-1}[j_k]>; dP (m_t^T G_t) and
//                          dQ (z^T sum G_t) on the otherwise idle matrix pipe (OuterAcc); (sum_t G_t) Q^T stored for the
//                          last launch; then w = -2 s (ds - <s, ds>), dy_self = sum_k w_k (y_i - y_j) from the same
//                          registers.  Needs G_T .. G_1 only, so it runs BEFORE the last reverse walk.
//   bwd_rev_kernel<FINAL>            ONE reverse walk for both remaining scatters: G_0 = (A^T G_1) P^T -> dz, and
//                          dy[j] = dy_self[j] + sum_{e in rev(j)} w[e] (y_j - y_i); a few extra workgroups of the same
//                          launch finish dP / dQ from the block partials (ticketed two-level sum, fixed order).
//
// The reverse walks are LOAD-BALANCED: a wavefront owns R consecutive source rows = one contiguous range of the reverse
// edge list.  Its lanes take that range edge by edge (64 records per coalesced load, no lane idles on a short row, no
// row-length-dependent tail rounds: three dependent memory phases per wave whatever the in-degrees are), move the
// gathered rows into a per-wave LDS tile, and only then do the rows' lane groups sum their own segment of the tile, in
// ascending edge order (fixed summation order: results are bitwise reproducible).  In-degrees of a kNN graph spread from
// 0 to ~40 around K; the row-per-lane-group form (round 2) kept every wave alive for the longest row's chain of
// record -> gather rounds (waves waited 62-66 % of their cycles, profiles/r2b_meanfield_pmc.md).
//
// Round 6, what the walks cost beyond their streaming traffic (diagnostic builds -DEXP_FREE / -DEXP_NOLDS below, level 0:
// chain / first / final walk 12.4 / 15.7 / 27.2 us; gathers made free 11.0 / 14.9 / 24.4; tile removed 9.1 / 13.9 / 20.7;
// both 6.2 / 9.3 / 11.8): the LDS tile and the gathers each hide behind the other, so the tile had to shrink first.
//   * PAIR: the gathering lanes multiply by the weights and add the two (L = 4: two pairs of) CONSECUTIVE edges they hold
//     before the tile: half the tile traffic, no weight tile in the chain walks, one tile phase per chunk
//     (chain 12.4 -> 10.0, final 26.9 -> 24.0 us).
//   * the final walk with two lane groups per row instead of four (16 rows and five load rounds per wavefront, 25 %
//     of the slots padding instead of 50 %: 24.2 -> 20.9 us) -- a vector-memory instruction costs the same whether a
//     quarter, half or all of its lanes carry a request (measured on step_fast_kernel with lanes masked by EXEC or
//     addressed out of range: -2 ... -13 % of the gather time for half the lanes; whole instructions out of range: free).
#include "crf_common.hpp"

#ifndef CRF_GATHER_SRD
#define CRF_GATHER_SRD 1      // row gathers of the reverse walks through a buffer resource with 32-bit byte offsets (0: 64-bit pointers)
#endif

namespace crf {

struct __attribute__((aligned(8))) RevEdge {
    int e;        // edge id i * K + k: target row i = e >> log2(K)
    float s;      // its softmax weight s[e]
};

__device__ __forceinline__ unsigned xcd_block_id_of(unsigned b, unsigned nb) {
    const unsigned xcd = b & 7u, within = b >> 3;
    const unsigned base = nb >> 3, rem = nb & 7u;
    return xcd * base + (xcd < rem ? xcd : rem) + within;
}

// Geometry of a reverse walk: L lanes per row (one float4 each), EP lane groups per row in the tile reduction (a row's
// entries alternate between them by position in the row), R rows per wavefront, MAXRR coalesced record loads (64
// edges each) issued before the first wait, TR of them per LDS tile.
#ifndef REV_HEAD
#define REV_HEAD 125           // load rounds of a wavefront cover this many per cent of R average rows (16 reverse edges each)
#endif
template <int H, int EPV>
struct Rev {
    static constexpr int NW = H >= 32 ? 1 : BLOCK / WAVE;      // wavefronts per workgroup (wide rows: the tile of ONE wave fills the LDS budget)
    static constexpr int L = H / 4, EP = EPV, R = WAVE / (L * EP), RPB = R * NW;
    static_assert(R >= 1 && R * L * EP == WAVE, "row groups must tile the wavefront");
    // an average row has K = 16 reverse edges (self column included): REV_HEAD - 100 = 25 % head room, then a second chunk
    // (R = 8: 3 rounds for 2 -- half the slots padding; R = 16: 5 for 4; 10 % head room at R = 32 overflows too often: 10.4 -> 11.2 us)
    static constexpr int MAXRR = (R * 16 * REV_HEAD / 100 + 63) / 64;
    static constexpr int CAP = 64 * MAXRR;
};

#ifndef REV_PAD_OOB
#define REV_PAD_OOB 1          // padding lanes of the reverse walks address out of range (no line lookup) instead of re-reading the last record
#endif
#ifndef REV_PAIR
#define REV_PAIR 1             // pairs of consecutive edges summed by the gathering lanes before the tile ...
#endif
#ifndef REV_PAIR_MAXL
#define REV_PAIR_MAXL 4        // ... in the final walk for rows of up to 4 lanes only (H = 32 / 64: chain walks 11.0 -> 9.5 / 9.4 -> 7.7 us, final walk 14.5 -> 15.5 / 11.6 -> 13.6)
#endif
#ifndef REV_EP_CHAIN
#define REV_EP_CHAIN 2
#endif
#ifndef REV_EP_FINAL
#define REV_EP_FINAL 2
#endif
#ifndef REV_EP_FINAL16
#define REV_EP_FINAL16 2       // final walk of H = 16 (one lane group per row: 97 KB of LDS per workgroup, 17.3 -> 20.7 us)
#endif
#ifndef REV_EP_CHAIN16
#define REV_EP_CHAIN16 1       // chain walks of H >= 16 and the final walks of H >= 32: one lane group per row (H = 32: chain 9.6 -> 8.0, first 11.2 -> 9.9 us; H = 64 final 11.6 -> 10.6; H = 16 chain 7.2 -> 6.9)
#endif
#ifndef REV_TR_CHAIN
#define REV_TR_CHAIN 3         // load rounds per LDS tile, chain walks WITHOUT pair sums (H = 8: 192 entries x 32 B = 6 KiB per wave)
#endif
#ifndef REV_TR_FINAL
#define REV_TR_FINAL 2         // final walk without pair sums (with them: one tile of all the chunk's rounds)
#endif

template <int L, int EP>
__device__ __forceinline__ float4 fold_halves(float4 a) {
#pragma unroll
    for (int o = L; o < L * EP; o <<= 1) {
        if constexpr (L == 2) {              // o = 2 -> the other pair of the DPP quad; o = 4 -> next quad (shuffle)
            if (o == 2) {
                a.x += quad_xor2(a.x); a.y += quad_xor2(a.y); a.z += quad_xor2(a.z); a.w += quad_xor2(a.w);
                continue;
            }
        }
        a.x += __shfl_xor(a.x, o, WAVE); a.y += __shfl_xor(a.y, o, WAVE);
        a.z += __shfl_xor(a.z, o, WAVE); a.w += __shfl_xor(a.w, o, WAVE);
    }
    return a;
}

enum { REV_CHAIN = 0, REV_FINAL = 1 };

struct RevArgs {
    const float* Gin;            // G_t rows gathered by the walk
    const RevEdge* rec_in;       // !FIRST: records left by the first walk
    const int32_t* rev_ptr;
    const int32_t* rev_eid;      // FIRST
    const float* s;              // FIRST
    RevEdge* rec_out;            // FIRST: may be NULL (nobody walks again)
    const float* P;
    int K;                       // 16 or 32 (e >> log2 K = target row)
    int64_t m;
    float* Gprev;                // CHAIN: (A^T Gin) P^T
    // FINAL
    const float* w;              // [m, K] distance-gradient weights by edge id
    const float* y;
    const float* dy_self;
    const float* dzq;            // (sum_t G_t) Q^T
    float* dz;
    float* dy;
    SmallJob j0, j1;             // dP / dQ block partials (n_reduce blocks in front of the walk's own)
    int nslots;
    float* scratch;
    unsigned* ticket;
    int n_reduce;
};

// PAD_OOB: the lanes of a load round past the wave's range (a quarter to a third of the slots: MAXRR rounds carry 25 % head
// room over the AVERAGE in-degree) address their record, weight and rows OUT OF RANGE of a buffer resource: the load
// returns zeros without a request to the vector cache.  (Re-reading the range's last record instead -- the first
// branch-free form -- is an L1 hit.  Measured the SAME to 0.1 us in all three walks: what a load instruction costs does not
// depend on how many of its lanes ask for something.  Kept: padding slots hold zeros whatever a table's last record is.)
// Needs the edge tables below 4 GiB (the launcher checks; else PAD_OOB = false).
constexpr int OOB_OFF = (int)0xFFFFFFF0u;
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

#ifndef REV_WPE
#define REV_WPE 1              // waves per SIMD asked of the register allocator (narrow rows)
#endif
template <int H, int EPV, int TR, int MODE, bool FIRST, int KSH, bool PAD_OOB>
__global__ __launch_bounds__((Rev<H, EPV>::NW * WAVE), (H <= 8 ? REV_WPE : 1)) void bwd_rev_kernel(const RevArgs a) {
    using RV = Rev<H, EPV>;
    constexpr int L = RV::L, EP = RV::EP, R = RV::R, MAXRR = RV::MAXRR, CAP = RV::CAP;
    constexpr bool FINAL = MODE == REV_FINAL;
    // PAIR: a slot (the L lanes of a row piece) gathers L CONSECUTIVE edges per load round; fifteen times out of sixteen two
    // neighbours in that run belong to the same row, so the slot multiplies by the weights itself and hands ONE partial
    // row per PAIR of edges to the tile (half the LDS traffic of the walk, no weight tile in the chain walks).  A pair
    // that straddles a row start keeps its halves apart: the first in the tile, the second in the starting row's side
    // slot.  One tile phase per chunk.
    constexpr bool PAIR = REV_PAIR && L >= 2 && (L <= REV_PAIR_MAXL || !FINAL);
    constexpr int TRR = PAIR ? MAXRR : (TR < MAXRR ? TR : MAXRR), TE = 64 * TRR;      // tile entries (edges)
    constexpr int TEP = TE / 2;                                             // PAIR: tile entries (pairs); then R side slots
    constexpr int NW = RV::NW;
    __shared__ float4 sM[H * L];                                           // P^T
    __shared__ float4 s_rows[NW][(FINAL ? 2 : 1) * (PAIR ? TEP + R : TE) * L];      // gathered G (and y) rows, entry-major
    __shared__ float s_wt[NW][PAIR ? (FINAL ? TEP + R : 1) : (FINAL ? 2 : 1) * TE];  // s (and w) per entry; PAIR: sums of w
    constexpr int FLW = (CAP + 8 + 511) / 512 * 64;                        // (whole 64-lane stores of 8 bytes)
    __shared__ uint2 s_flag[PAIR ? NW : 1][PAIR ? FLW : 1];                // PAIR: per position of the chunk, 1 + local row that starts there
    if constexpr (FINAL) {
        if ((int)blockIdx.x < a.n_reduce) {                                // dP / dQ block partials -> dP, dQ
            reduce_small_body(a.j0, a.j1, a.nslots, a.scratch, a.ticket, blockIdx.x, (unsigned)a.n_reduce);
            return;
        }
    }
    // P^T: one element per thread is fetched now and parked in LDS after the walk (a load + wait + LDS write in front
    // of the walk's own loads would put a whole memory round trip ahead of them)
    constexpr bool EARLY_M = H * H <= NW * WAVE;
    [[maybe_unused]] float pm = 0.f;
    if constexpr (EARLY_M) {
        const int t = threadIdx.x < H * H ? threadIdx.x : 0;
        pm = a.P[(t % H) * H + t / H];                                      // sM[h][c] = P[c][h]
    }
    const unsigned nred = FINAL ? (unsigned)a.n_reduce : 0u;
    const unsigned bid = xcd_block_id_of(blockIdx.x - nred, gridDim.x - nred);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int q = lane % L, el = (lane / L) % EP, slot = lane / L;
    const int64_t m = a.m;
    const int64_t row0 = ((int64_t)bid * NW + wave) * R;
    int64_t row = row0 + lane / (L * EP);
    const bool valid = row < m;
    if (!valid) row = m - 1;
    const int beg = a.rev_ptr[row], end1 = a.rev_ptr[row + 1];              // (row + 1 <= m: in range; no branch around a load)
    const int end = valid ? end1 : beg;
    // the wave's range of the reverse edge list (uniform: scalar loads)
    const int64_t r0c = row0 < m ? row0 : m, r1c = row0 + R < m ? row0 + R : m;
    const int B = a.rev_ptr[r0c], Eend = a.rev_ptr[r1c];
    float4* trow = s_rows[wave];
    float* twt = s_wt[wave];
    [[maybe_unused]] float4 yj = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (FINAL) yj = ld4(a.y + row * H + 4 * q);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);                          // sum s G[i]
    [[maybe_unused]] float4 acc2 = make_float4(0.f, 0.f, 0.f, 0.f);        // sum w (y_j - y_i); PAIR: sum w y_i, with
    [[maybe_unused]] float accw = 0.f;                                     //   accw = sum w
#if CRF_GATHER_SRD
    const __amdgpu_buffer_rsrc_t rG = make_rsrc(a.Gin, (int)(m * H * 4));
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rY = make_rsrc(FINAL ? a.y : a.Gin, (int)(m * H * 4));
#endif
    // PAD_OOB: the edge tables as buffer resources ([m K] edges: 4 bytes per id / weight, 8 per record)
    const unsigned n_edge = (unsigned)m << KSH;
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rE = make_rsrc(FIRST ? (const void*)a.rev_eid : (const void*)a.rec_in, (int)(n_edge * (FIRST ? 4u : 8u)));
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rS = make_rsrc(FIRST ? a.s : a.Gin, (int)(n_edge * 4u));
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rW = make_rsrc(FINAL ? a.w : a.Gin, (int)(n_edge * 4u));

    auto chunk = [&](const int c0) {
        // Phases 1 and 2 are BRANCH-FREE: every round is issued, rounds (lanes) past the wave's range re-read its last
        // record (an L1 hit) -- with a uniform branch per round the compiler drains the memory counter at every block
        // boundary and the loads of a phase run one after the other instead of together.
        // ---- phase 1: records of up to CAP edges, 64 per load, one edge per lane
        int e[MAXRR];
        float sv[MAXRR];
        [[maybe_unused]] float wv[MAXRR];
#pragma unroll
        for (int rr = 0; rr < MAXRR; ++rr) {
            const int p = c0 + 64 * rr + lane;
            [[maybe_unused]] const int pl = p < Eend ? p : Eend - 1, pc = pl > 0 ? pl : 0;
            if constexpr (PAD_OOB) {
                if constexpr (FIRST) {
                    e[rr] = (int)__builtin_amdgcn_raw_buffer_load_b32(rE, p < Eend ? p * 4 : OOB_OFF, 0, 0);
                } else {
                    const u32x2_t t = __builtin_amdgcn_raw_buffer_load_b64(rE, p < Eend ? p * 8 : OOB_OFF, 0, 0);
                    e[rr] = (int)t.x; sv[rr] = __uint_as_float(t.y);
                }
            } else if constexpr (FIRST) {
                e[rr] = a.rev_eid[pc];
            } else {
                const RevEdge t = a.rec_in[pc];
                e[rr] = t.e; sv[rr] = t.s;
            }
        }
        if constexpr (FIRST) {
#pragma unroll
            for (int rr = 0; rr < MAXRR; ++rr) {
#ifdef EXP_FREE
                e[rr] = (c0 + 64 * rr + lane < Eend ? c0 + 64 * rr + lane : 0);
#endif
                if constexpr (PAD_OOB) sv[rr] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rS, c0 + 64 * rr + lane < Eend ? e[rr] * 4 : OOB_OFF, 0, 0));
                else sv[rr] = a.s[e[rr]];
            }
        }
        if constexpr (FINAL) {
#pragma unroll
            for (int rr = 0; rr < MAXRR; ++rr) {
#ifdef EXP_FREE
                e[rr] = (c0 + 64 * rr + lane < Eend ? c0 + 64 * rr + lane : 0);
#endif
                if constexpr (PAD_OOB) wv[rr] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rW, c0 + 64 * rr + lane < Eend ? e[rr] * 4 : OOB_OFF, 0, 0));
                else wv[rr] = a.w[e[rr]];
            }
        }
        [[maybe_unused]] int fl[MAXRR][L >= 2 ? L / 2 : 1];
        if constexpr (PAIR) {
            // row starts of the chunk as bytes in LDS (needs rev_ptr only: runs while the records are in flight)
            unsigned char* fb = reinterpret_cast<unsigned char*>(s_flag[wave]);
#pragma unroll
            for (int i = 0; i < FLW; i += WAVE) s_flag[wave][i + lane] = make_uint2(0u, 0u);
            __builtin_amdgcn_wave_barrier();
            const int rel = beg - c0;
            if (lane % (L * EP) == 0 && beg < end && rel >= 0 && rel < CAP) fb[rel] = (unsigned char)(1 + lane / (L * EP));
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int rr = 0; rr < MAXRR; ++rr)
#pragma unroll
                for (int h = 0; h < L / 2; ++h) fl[rr][h] = fb[64 * rr + slot * L + 2 * h + 1];
#pragma unroll
            for (int rr = 0; rr < MAXRR; ++rr) {                           // (lanes past the range: weight 0 whatever they loaded)
                const bool in = c0 + 64 * rr + lane < Eend;
                sv[rr] = in ? sv[rr] : 0.f;
                if constexpr (FINAL) wv[rr] = in ? wv[rr] : 0.f;
            }
        }
        // ---- phase 2: row gathers, L sub-rounds per load round (slot k of sub-round sb takes the edge lane k L + sb loaded)
        float4 g[MAXRR][L];
        [[maybe_unused]] float4 gy[FINAL ? MAXRR : 1][FINAL ? L : 1];
#pragma unroll
        for (int rr = 0; rr < MAXRR; ++rr) {
            static_for<L>([&](auto SB) {
                constexpr int sb = decltype(SB)::value;
                const int ee = __float_as_int(group_bcast<L, sb>(__int_as_float(e[rr]), lane - q));
#if CRF_GATHER_SRD
                int boff = (ee >> KSH) * (4 * H) + 16 * q;            // 32-bit byte offset on a buffer resource (see crf.hip)
                if constexpr (PAD_OOB) boff = c0 + 64 * rr + (lane - q) + sb < Eend ? boff : OOB_OFF;      // (this slot's edge: lane slot L + sb of the round)
#ifdef EXP_FREE      // timing experiment only (wrong results): every gather reads the lane's own row / consecutive words
                boff = (int)row0 * (4 * H) + 16 * lane;
#endif
                g[rr][sb] = ld4_buf(rG, boff);
                if constexpr (FINAL) gy[rr][sb] = ld4_buf(rY, boff);
#else
                const int64_t i = ee >> KSH;
                g[rr][sb] = ld4(a.Gin + i * H + 4 * q);
                if constexpr (FINAL) gy[rr][sb] = ld4(a.y + i * H + 4 * q);
#endif
            });
        }
        if constexpr (FIRST) {
            if (a.rec_out != nullptr) {
#pragma unroll
                for (int rr = 0; rr < MAXRR; ++rr) {
                    const int p = c0 + 64 * rr + lane;
                    if (p < Eend) {
                        RevEdge t; t.e = e[rr]; t.s = sv[rr];
                        a.rec_out[p] = t;
                    }
                }
            }
        }
#ifdef EXP_NOLDS     // timing experiment only (wrong results): no tile, every lane sums what it gathered
#pragma unroll
        for (int rr = 0; rr < MAXRR; ++rr) {
            static_for<L>([&](auto SB) {
                constexpr int sb = decltype(SB)::value;
                acc = fma4(sv[rr], g[rr][sb], acc);
                if constexpr (FINAL) acc2 = fma4(wv[rr], sub4(yj, gy[rr][sb]), acc2);
            });
        }
        if (true) return;
#endif
        if constexpr (PAIR) {
            constexpr int YO = (TEP + R) * L;                              // FINAL: the y table behind the G table
            auto mul4 = [](float f, float4 v) { return make_float4(f * v.x, f * v.y, f * v.z, f * v.w); };
            auto add4 = [](float4 u, float4 v) { return make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w); };
#pragma unroll
            for (int rr = 0; rr < MAXRR; ++rr) {
                static_for<L / 2>([&](auto HP) {                           // the slot's L consecutive edges of the round, pair by pair
                    constexpr int h = decltype(HP)::value;
                    const float s0 = group_bcast<L, 2 * h>(sv[rr], lane - q), s1 = group_bcast<L, 2 * h + 1>(sv[rr], lane - q);
                    const float4 lo = mul4(s0, g[rr][2 * h]), hi = mul4(s1, g[rr][2 * h + 1]);
                    const int pi = 32 * rr + slot * (L / 2) + h, side = TEP + fl[rr][h] - 1;
                    const bool split = fl[rr][h] != 0;
                    trow[pi * L + q] = split ? lo : add4(lo, hi);
                    if (split) trow[side * L + q] = hi;
                    if constexpr (FINAL) {
                        const float w0 = group_bcast<L, 2 * h>(wv[rr], lane - q), w1 = group_bcast<L, 2 * h + 1>(wv[rr], lane - q);
                        const float4 ylo = mul4(w0, gy[rr][2 * h]), yhi = mul4(w1, gy[rr][2 * h + 1]);
                        trow[YO + pi * L + q] = split ? ylo : add4(ylo, yhi);
                        twt[pi] = split ? w0 : w0 + w1;
                        if (split) {
                            trow[YO + side * L + q] = yhi;
                            twt[side] = w1;
                        }
                    }
                });
            }
            __builtin_amdgcn_wave_barrier();                               // LDS operations of one wave complete in order
            const int w1e = c0 + CAP < Eend ? c0 + CAP : Eend;
            const int lo = beg > c0 ? beg : c0, hi = end < w1e ? end : w1e;
            const int rl = lo - c0, rh = hi - c0;
            if (lo < hi && (rl & 1) && el == 0) {                          // the row starts on a pair's second edge: its side slot
                const int sl = TEP + lane / (L * EP);
                const float4 c = trow[sl * L + q];
                acc = make_float4(acc.x + c.x, acc.y + c.y, acc.z + c.z, acc.w + c.w);
                if constexpr (FINAL) {
                    const float4 cy = trow[YO + sl * L + q];
                    acc2 = make_float4(acc2.x + cy.x, acc2.y + cy.y, acc2.z + cy.z, acc2.w + cy.w);
                    accw += twt[sl];
                }
            }
            const int pb = lo < hi ? (rh + 1) >> 1 : 0;                    // pairs [ceil(rl / 2), ceil(rh / 2)) hold the row's other edges
            for (int pi = ((rl + 1) >> 1) + el; pi < pb; pi += 4 * EP) {
                float4 c[4];
                [[maybe_unused]] float4 cy[4];
                [[maybe_unused]] float cw[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int pp = pi + u * EP;
                    const bool ok = pp < pb;
                    const int ent = ok ? pp : pi;
                    c[u] = trow[ent * L + q];
                    if (!ok) c[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if constexpr (FINAL) {
                        cy[u] = trow[YO + ent * L + q];
                        cw[u] = ok ? twt[ent] : 0.f;
                        if (!ok) cy[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc = make_float4(acc.x + c[u].x, acc.y + c[u].y, acc.z + c[u].z, acc.w + c[u].w);
                    if constexpr (FINAL) {
                        acc2 = make_float4(acc2.x + cy[u].x, acc2.y + cy[u].y, acc2.z + cy[u].z, acc2.w + cy[u].w);
                        accw += cw[u];
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            return;
        }
        // ---- phase 3: through the LDS tile, TRR load rounds at a time; rows sum their own entries in edge order
#pragma unroll
        for (int t0 = 0; t0 < MAXRR; t0 += TRR) {
            {       // (no uniform test around a tile group either: the compiler would sink the group's gathers into it)
#pragma unroll
                for (int rr = t0; rr < t0 + TRR && rr < MAXRR; ++rr) {
                    {
                        const int eb = 64 * (rr - t0);
                        twt[eb + lane] = sv[rr];
                        if constexpr (FINAL) twt[TE + eb + lane] = wv[rr];
                        static_for<L>([&](auto SB) {
                            constexpr int sb = decltype(SB)::value;
                            const int ent = eb + slot * L + sb;
                            trow[ent * L + q] = g[rr][sb];
                            if constexpr (FINAL) trow[(TE + ent) * L + q] = gy[rr][sb];
                        });
                    }
                }
                __builtin_amdgcn_wave_barrier();                           // LDS operations of one wave complete in order
                const int w0 = c0 + 64 * t0;
                const int wcap = (t0 + TRR < MAXRR ? w0 + TE : c0 + CAP);      // the chunk's last tile may be a short one
                const int w1 = wcap < Eend ? wcap : Eend;
                const int lo = beg > w0 ? beg : w0, hi = end < w1 ? end : w1;
                int p = lo + ((el - (lo - beg)) & (EP - 1));              // first entry >= lo of this lane group's parity
                for (; p < hi; p += 4 * EP) {
                    float4 c[4];
                    float cs[4];
                    [[maybe_unused]] float4 cy[4];
                    [[maybe_unused]] float cw[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int pp = p + u * EP;
                        const bool ok = pp < hi;
                        const int ent = ok ? pp - w0 : lo - w0;
                        c[u] = trow[ent * L + q];
                        cs[u] = ok ? twt[ent] : 0.f;
                        if constexpr (FINAL) {
                            cy[u] = trow[(TE + ent) * L + q];
                            cw[u] = ok ? twt[TE + ent] : 0.f;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        acc = fma4(cs[u], c[u], acc);
                        if constexpr (FINAL) acc2 = fma4(cw[u], sub4(yj, cy[u]), acc2);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    };
    // the first chunk is straight-line code (a loop pre-header would make the compiler wait for the rows' rev_ptr pair
    // before the record loads are even issued); further chunks only for ranges beyond CAP edges
    // (unconditional, also for a wave without edges: its loads re-read record max(Eend - 1, 0) and no row sums anything)
    chunk(B);
    for (int c0 = B + CAP; c0 < Eend; c0 += CAP) chunk(c0);
    acc = fold_halves<L, EP>(acc);
    if constexpr (FINAL) acc2 = fold_halves<L, EP>(acc2);
    if constexpr (FINAL && PAIR) {                                         // sum w (y_j - y_i) = y_j sum w - sum w y_i
        accw = fold_halves<L, EP>(make_float4(accw, 0.f, 0.f, 0.f)).x;
        acc2 = make_float4(yj.x * accw - acc2.x, yj.y * accw - acc2.y, yj.z * accw - acc2.z, yj.w * accw - acc2.w);
    }
    if constexpr (EARLY_M) {
        if (threadIdx.x < H * H) reinterpret_cast<float*>(sM)[threadIdx.x] = pm;
    } else {
        load_matrix<H, NW * WAVE>(sM, a.P, true);
    }
    __syncthreads();                                                       // P^T staged
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (!FINAL) {
        const float4 gp = matvec_acc<H>(acc, sM, lane, q, zero);
        if (valid && el == 0) st4(a.Gprev + row * H + 4 * q, gp);
    } else {
        float4 add = zero, ds = zero;
        if (el == 0) {
            add = ld4(a.dzq + row * H + 4 * q);
            ds = ld4(a.dy_self + row * H + 4 * q);
        }
        const float4 g0 = matvec_acc<H>(acc, sM, lane, q, add);            // G_0 + (sum_t G_t) Q^T
        if (valid && el == 0) {
            st4(a.dz + row * H + 4 * q, g0);
            st4(a.dy + row * H + 4 * q, make_float4(acc2.x + ds.x, acc2.y + ds.y, acc2.z + ds.z, acc2.w + ds.w));
        }
    }
}

// Per point, all T steps with the index row, the weight row and ds in registers (see the file header).
#ifndef EDGE_LB
#define EDGE_LB 1          // minimum waves per SIMD asked of the register allocator
#endif
template <int H, int K, bool U16>
__global__ __launch_bounds__(BLOCK, (H <= 16 && K == 16) ? EDGE_LB : 1) void bwd_edge_all_kernel(const float* __restrict__ gout, const float* __restrict__ Gs,
                                                             const float* __restrict__ xs, const float* __restrict__ z,
                                                             const float* __restrict__ y, const float* __restrict__ s,
                                                             const int32_t* __restrict__ idx,
                                                             const uint16_t* __restrict__ idx16, int n_tgt, int n_src,
                                                             const float* __restrict__ P, const float* __restrict__ Q,
                                                             int T, float* __restrict__ mts, float* __restrict__ Gcopy,
                                                             float* __restrict__ sumG_out, float* __restrict__ dzq,
                                                             float* __restrict__ dp_partial, float* __restrict__ dq_partial,
                                                             float* __restrict__ w, float* __restrict__ dy_self, int64_t m) {
    constexpr int L = Geo<H>::L;
    constexpr bool INK = H == 8 || H == 16;          // dP, dQ accumulated here (OuterAcc); else m_t / sum G are stored
    __shared__ float4 sPT[MatStage<H>::F4], sQT[MatStage<H>::F4];
    __shared__ __attribute__((aligned(16))) float s_tile[INK ? 2 * (BLOCK / WAVE) * 256 : 4];
    __shared__ float s_red[INK ? (BLOCK / WAVE) * H * H : 1];
    int lane, q;
    bool valid;
    const int64_t r = my_point<H>(m, lane, q, valid);
    // neighbour rows are addressed as  uniform base (SGPR pair) + 32-bit byte offset: one VGPR per neighbour instead of
    // the index AND a 64-bit address (the kernel sits at 200+ registers; the host checks m H 4 < 2^32)
    unsigned boff[K];
    float sw[K], dd[K];
    {
        int j[K];
        load_index_row_t<K, U16>(idx, idx16, r, n_tgt, n_src, j);
#pragma unroll
        for (int k = 0; k < K; ++k) boff[k] = ((unsigned)j[k] * H + 4 * q) * 4u;
#ifdef EXP_FREE
#pragma unroll
        for (int k = 0; k < K; ++k) boff[k] = ((unsigned)r * H + 4 * q) * 4u;
#endif
    }
    MatStage<H> mp, mq;                              // P^T, Q^T: fetched now, parked behind the first gathers' issue
    mp.fetch(P, true);
    mq.fetch(Q, true);
    const int tab_bytes = (int)(m * H * 4);          // one [m, H] table (the host checks m H 4 < 2^31)
    load_row<K, float4>(s + r * K, sw);
#pragma unroll
    for (int k = 0; k < K; ++k) dd[k] = 0.f;
    const int64_t step = m * H;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 sg = zero;
    [[maybe_unused]] OuterAcc<INK ? H : 8> oa, oq;
#ifndef EDGE_NB
#define EDGE_NB 15         // neighbour rows in flight per lane (8 -> 15: 25.7 -> 23.7 us at level 0)
#endif
    for (int i = 0; i < T; ++i) {                   // i-th entry of Gs is step t = T - i (entry 0 = gout)
        const int t = T - i;
        const float* __restrict__ xprev = t >= 2 ? xs + (int64_t)(t - 2) * step : z;
        float4 gi = ld4((i == 0 ? gout : Gs + (int64_t)i * step) + r * H + 4 * q);
        float4 msg = zero;
        float4 gmi = zero;
        const __amdgpu_buffer_rsrc_t xr = make_rsrc(xprev, tab_bytes);     // uniform base in SGPRs, 32-bit offsets in VGPRs
#pragma unroll
        for (int k0 = 1; k0 < K; k0 += EDGE_NB) {
            float4 nb[EDGE_NB];
#pragma unroll
            for (int k = k0; k < k0 + EDGE_NB && k < K; ++k) nb[k - k0] = ld4_buf(xr, (int)boff[k]);
            if (k0 == 1) {
                if (i == 0) {                        // P^T, Q^T staged: the wait sits BEHIND the first gathers' issue
                    mp.park(sPT);
                    mq.park(sQT);
                    __syncthreads();
                }
                gmi = matvec_acc<H>(gi, sPT, lane, q, zero);               // G_t P^T while the first gathers fly
            }
#pragma unroll
            for (int k = k0; k < k0 + EDGE_NB && k < K; ++k) {
                msg = fma4(sw[k], nb[k - k0], msg);
                dd[k] += group_sum<L>(dot4(gmi, nb[k - k0]));
            }
            __builtin_amdgcn_sched_barrier(0);       // EDGE_NB rows in flight, not K - 1: the batches keep the kernel at 3-4 waves per SIMD
        }
        if (!valid) { msg = zero; gi = zero; }
        sg = make_float4(sg.x + gi.x, sg.y + gi.y, sg.z + gi.z, sg.w + gi.w);
        if constexpr (INK) {
            float* ta = s_tile + (threadIdx.x >> 6) * 512;
            oa.add_rows(msg, gi, ta, ta + 256, lane);
        } else {
            if (valid) st4(mts + (int64_t)i * step + r * H + 4 * q, msg);
            if (valid && i == 0) st4(Gcopy + r * H + 4 * q, gi);            // the stacked [T, m, H] operand of dP = m^T G
        }
    }
    // (sum_t G_t) Q^T for the last reverse walk; dQ = z^T sum_t G_t
    const float4 dzqv = matvec_acc<H>(sg, sQT, lane, q, zero);
    if (valid) st4(dzq + r * H + 4 * q, dzqv);
    if constexpr (INK) {
        float4 zi = ld4(z + r * H + 4 * q);
        if (!valid) zi = zero;
        float* ta = s_tile + (threadIdx.x >> 6) * 512;
        oq.add_rows(zi, sg, ta, ta + 256, lane);
    } else {
        if (valid) st4(sumG_out + r * H + 4 * q, sg);
    }
    // softmax / distance backward from the registers
    const float4 yi = ld4(y + r * H + 4 * q);
    float dotv = 0.f;
#pragma unroll
    for (int k = 1; k < K; ++k) dotv = fmaf(sw[k], dd[k], dotv);
    float4 acc = zero;
    dd[0] = 0.f;
    const __amdgpu_buffer_rsrc_t yr = make_rsrc(y, tab_bytes);
#pragma unroll
    for (int k0 = 1; k0 < K; k0 += 8) {
        float4 nb[8];
#pragma unroll
        for (int k = k0; k < k0 + 8 && k < K; ++k) nb[k - k0] = ld4_buf(yr, (int)boff[k]);
#pragma unroll
        for (int k = k0; k < k0 + 8 && k < K; ++k) {
            const float wk = -2.0f * sw[k] * (dd[k] - dotv);
            acc = fma4(wk, sub4(yi, nb[k - k0]), acc);
            dd[k] = wk;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    store_rows_coalesced<H, K>(dd, w, lane, q, m);
    if (valid) st4(dy_self + r * H + 4 * q, acc);
    if constexpr (INK) {
        oa.store_partial(s_red, dp_partial, lane);
        __syncthreads();
        oq.store_partial(s_red, dq_partial, lane);
    }
}

}  // namespace crf

using namespace crf;

extern "C" int crfconv_meanfield_backward_supported(int H, int K, int k0) {
    return (k0 == 1 && (K == 16 || K == 32) && (H == 4 || H == 8 || H == 16 || H == 32 || H == 64)) ? 1 : 0;
}

// 1 when dP and dQ come out of the backward launches themselves (H in {8, 16}); 0 when the caller finishes them from
// mts / Gs / sumG with crfconv_linear_wgrad
extern "C" int crfconv_meanfield_backward_param_grads_inside(int H) { return (H == 8 || H == 16) ? 1 : 0; }

static size_t bwd_ws_layout(int64_t m, int H, int K, size_t* off_dp, size_t* off_dq, size_t* off_scratch) {
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t o = up((size_t)m * K * sizeof(RevEdge));
    size_t nb_edge = 0;
    DISPATCH_H(H, { nb_edge = (size_t)cdiv(m, Geo<HH>::PPB); });
    *off_dp = o; o = up(o + nb_edge * H * H * 4);
    *off_dq = o; o = up(o + nb_edge * H * H * 4);
    *off_scratch = o; o = up(o + 2 * RS_CHUNKS * (size_t)H * H * 4);
    return o;
}

extern "C" size_t crfconv_meanfield_backward_workspace(int64_t m, int H, int K) {
    size_t a, b, c;
    return bwd_ws_layout(m, H, K, &a, &b, &c);
}

template <int HH, int MODE, bool FIRST>
static void launch_rev(const RevArgs& ra, int64_t m, hipStream_t st) {
    constexpr int EPV = (HH == 4) ? 4 : (MODE == REV_FINAL && HH <= 16 ? (HH == 16 ? REV_EP_FINAL16 : REV_EP_FINAL) : (HH <= 8 ? REV_EP_CHAIN : REV_EP_CHAIN16));
    // LDS tile of a wave: 64 TR entries x 4 H bytes (x 2 + in the final walk): 6-9 KiB, so that four workgroups fit a CU
    // (the pair-sum form ignores TR: 32 MAXRR pair entries + R side slots per table)
    constexpr int TR = MODE == REV_FINAL ? (HH <= 8 ? REV_TR_FINAL : 1) : (HH <= 8 ? REV_TR_CHAIN : (HH == 16 ? 2 : 8));
    const unsigned nb = (unsigned)cdiv(m, Rev<HH, EPV>::RPB) + (MODE == REV_FINAL ? (unsigned)ra.n_reduce : 0u);
    const bool oob = REV_PAD_OOB && CRF_GATHER_SRD && m * ra.K * 8 < (int64_t)0xFFFFFFF0u;      // the record table as one buffer resource
    if (ra.K == 16) {
        if (oob) hipLaunchKernelGGL((bwd_rev_kernel<HH, EPV, TR, MODE, FIRST, 4, true>), dim3(nb), dim3(Rev<HH, EPV>::NW * WAVE), 0, st, ra);
        else hipLaunchKernelGGL((bwd_rev_kernel<HH, EPV, TR, MODE, FIRST, 4, false>), dim3(nb), dim3(Rev<HH, EPV>::NW * WAVE), 0, st, ra);
    } else {
        if (oob) hipLaunchKernelGGL((bwd_rev_kernel<HH, EPV, TR, MODE, FIRST, 5, true>), dim3(nb), dim3(Rev<HH, EPV>::NW * WAVE), 0, st, ra);
        else hipLaunchKernelGGL((bwd_rev_kernel<HH, EPV, TR, MODE, FIRST, 5, false>), dim3(nb), dim3(Rev<HH, EPV>::NW * WAVE), 0, st, ra);
    }
}

template <int HH, int KK, typename... A>
static void launch_edge_all(dim3 grid, hipStream_t st, const float* gout, const float* Gs, const float* xs, const float* z,
                            const float* y, const float* s, const int32_t* idx32, const uint16_t* idx16, A... rest) {
    if (idx16) hipLaunchKernelGGL((bwd_edge_all_kernel<HH, KK, true>), grid, dim3(BLOCK), 0, st, gout, Gs, xs, z, y, s, idx32, idx16, rest...);
    else hipLaunchKernelGGL((bwd_edge_all_kernel<HH, KK, false>), grid, dim3(BLOCK), 0, st, gout, Gs, xs, z, y, s, idx32, idx16, rest...);
}

extern "C" int crfconv_meanfield_backward(const float* gout, const float* z, const float* y, const float* s,
                                          const float* xs, const int32_t* idx32, const uint16_t* idx16, int n_tgt,
                                          int n_src, const int32_t* rev_ptr, const int32_t* rev_eid, int K, int k0,
                                          int64_t m, int H, const float* Q, const float* P, int T, float* Gs,
                                          float* dzq, float* mts, float* sumG, float* dz, float* w, float* dy_self,
                                          float* dy, float* dP, float* dQ, void* ws, size_t ws_bytes,
                                          unsigned* ticket, crf_stream_t stream) {
    if (int rc = check_common(m, H, K, k0)) return rc;
    CRF_REQUIRE(crfconv_meanfield_backward_supported(H, K, k0) == 1 && T >= 1, CRF_ERR_UNSUPPORTED,
                "restructured mean-field backward: K=%d k0=%d T=%d not supported", K, k0, T);
    const bool inside = crfconv_meanfield_backward_param_grads_inside(H) == 1;
    CRF_REQUIRE(m * H * 4 < ((int64_t)1 << 31), CRF_ERR_UNSUPPORTED, "m=%lld x H=%d: row tables of 2 GiB and more are not addressable "
                "by the kernels' 32-bit byte offsets", (long long)m, H);
    CRF_REQUIRE(gout && z && y && s && xs && idx32 && rev_ptr && rev_eid && Q && P && Gs && dzq && dz && w && dy_self &&
                dy && ws, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(inside ? (dP && dQ && ticket) : (mts && sumG), CRF_ERR_ARG,
                "H=%d: %s", H, inside ? "dP, dQ and ticket are required" : "mts and sumG are required");
    size_t off_dp, off_dq, off_scratch;
    const size_t need = bwd_ws_layout(m, H, K, &off_dp, &off_dq, &off_scratch);
    CRF_REQUIRE(ws_bytes >= need && (reinterpret_cast<uintptr_t>(ws) & 15) == 0, CRF_ERR_ARG,
                "workspace of %zu bytes (16-byte aligned), need %zu", ws_bytes, need);
    CRF_REQUIRE(idx16 == nullptr || (n_tgt > 0 && n_src > 0 && n_src <= 65536 && m % n_tgt == 0), CRF_ERR_ARG,
                "u16 table needs n_src <= 65536 and m a multiple of n_tgt (n_tgt=%d n_src=%d)", n_tgt, n_src);
    CRF_REQUIRE(m * H * 4 < ((int64_t)1 << 31), CRF_ERR_ARG, "m H = %lld rows x channels exceed 32-bit byte offsets", (long long)(m * H));
    hipStream_t st = as_stream(stream);
    char* wsb = static_cast<char*>(ws);
    RevEdge* rec = reinterpret_cast<RevEdge*>(wsb);
    float* dp_partial = reinterpret_cast<float*>(wsb + off_dp);
    float* dq_partial = reinterpret_cast<float*>(wsb + off_dq);
    float* scratch = reinterpret_cast<float*>(wsb + off_scratch);
    const int64_t step = m * H;
    DISPATCH_H(H, {
        const dim3 grid((unsigned)cdiv(m, Geo<HH>::PPB));
        RevArgs ra{};
        ra.rev_ptr = rev_ptr; ra.rev_eid = rev_eid; ra.s = s; ra.P = P; ra.K = K; ra.m = m;
        ra.rec_in = rec;
        // entry i of Gs belongs to step t = T - i; G_T = gout (entry 0 is written by edge-all only when mts is given)
        for (int i = 0; i + 1 < T; ++i) {
            ra.Gin = i == 0 ? gout : Gs + i * step;
            ra.Gprev = Gs + (i + 1) * step;
            ra.rec_out = rec;
            if (i == 0) launch_rev<HH, REV_CHAIN, true>(ra, m, st);
            else launch_rev<HH, REV_CHAIN, false>(ra, m, st);
            CRF_LAUNCH_CHECK();
        }
        float* gcopy = mts != nullptr ? Gs : nullptr;
        if (K == 16) launch_edge_all<HH, 16>(grid, st, gout, Gs, xs, z, y, s, idx32, idx16, n_tgt, n_src, P, Q, T, mts, gcopy, sumG, dzq, dp_partial, dq_partial, w, dy_self, m);
        else launch_edge_all<HH, 32>(grid, st, gout, Gs, xs, z, y, s, idx32, idx16, n_tgt, n_src, P, Q, T, mts, gcopy, sumG, dzq, dp_partial, dq_partial, w, dy_self, m);
        CRF_LAUNCH_CHECK();
        ra.Gin = T == 1 ? gout : Gs + (int64_t)(T - 1) * step;
        ra.Gprev = nullptr; ra.rec_out = nullptr;
        ra.w = w; ra.y = y; ra.dy_self = dy_self; ra.dzq = dzq; ra.dz = dz; ra.dy = dy;
        ra.n_reduce = 0;
        if (inside) {
            ra.j0 = SmallJob{dp_partial, dP, (int)grid.x};
            ra.j1 = SmallJob{dq_partial, dQ, (int)grid.x};
            ra.nslots = HH * HH; ra.scratch = scratch; ra.ticket = ticket; ra.n_reduce = RS_CHUNKS;
        }
        if (T == 1) launch_rev<HH, REV_FINAL, true>(ra, m, st);
        else launch_rev<HH, REV_FINAL, false>(ra, m, st);
        CRF_LAUNCH_CHECK();
    });
    return CRF_OK;
}
