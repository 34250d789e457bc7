// Mean-field forward as ONE launch with block-resident rows (round 6).
//
// The per-step kernels of crf.hip are bound by what a launch boundary forces them to move: every step re-reads the index row,
// the weight row and z of every point (96 of its 192 bytes per point at H = 8, K = 16), and every neighbour row goes through the
// vector cache, whose tag lookup serves ~one 128-byte line per clock and CU whatever fraction of the line is wanted (a 32-byte
// row = a quarter; profiles/r6_meanfield_ta.md: 11 300 line accesses per CU in a 15 000-cycle step launch).  This kernel removes
// both for SPATIALLY SORTED clouds (the device collate emits Morton order):
//
//   * one workgroup per CU owns PB CONSECUTIVE rows for the whole forward.  Index rows, soft-max weights and z Q of its points stay
//     in REGISTERS across the T steps, split over a point's lanes (lane q holds columns KL q .. and hands a column on by a DPP
//     broadcast when its turn comes); nothing but x_{t-1} is read again.
//   * the block's own rows of y | z (step 1) and of x_{t-1} (later steps) sit in LDS.  In Morton order ~80 % of a point's
//     neighbours are rows of its own block of 640 (measured on the 4 cm synthetic cloud: 0.80 at 640 rows): they are ds_read_b128
//     from LDS (no tag lookups).  The rest are buffer loads as before.
//   * ONE register per neighbour says where its row is: a >= 0: byte address inside the LDS row buffer; a < 0: bit 31 + byte offset in
//     the row table.  Both accesses are branch-free for every neighbour: the LDS address is max(a, 0) (row 0 of the buffer holds
//     zeros), the buffer offset is a ^ 0x80000000 (for an in-block neighbour: past the table's end -- the load returns 0 and makes
//     no memory request).
//   * step 1 (no barrier in front of it) adds the two results per neighbour (x + 0 is x): s and x_1 are those of
//     sim_step_fast_kernel bit for bit.  Steps 2 .. T split their sum: the IN-BLOCK part of step t + 1 needs nothing but the
//     block's own x_t, so it runs while the block's stores drain and the grid barrier completes (arrive -> in-block sums ->
//     wait); only the out-of-block fifth is left behind the barrier.  The message is thereby added in a different order
//     (in-block columns first): x_t, t >= 2, equals the per-step launches' to rounding (1e-6 relative), not bit for bit.
//   * rows are published write-through (sc1) and remote rows read with sc1 loads (another CU's stores never refresh this CU's L1).
//
// Measured (profiles/r6_block_stamps.md): the first form of this file (merged access in every step, full barrier) spent per step
// 3.0 us computing, 0.3-3 us draining stores and 2.1-2.3 us in the barrier -- the split hides the first two behind the third.
// Co-residency: one workgroup per CU at most (the host checks the grid against the device's CU count; every workgroup fits a CU
// alone); a barrier that cannot complete gives up after ~1 s with the sticky failure word (ops.check_gridsync).
//
// Reference semantics: models/continuous_crf_conv_big.py:49-54 (similarity), :63-72 (loop).
#include "crf_common.hpp"

namespace crf {

// Packed float32 arithmetic (v_pk_add_f32 / v_pk_fma_f32: two components per instruction at the rate of one): the kernel's phases are
// bound by the NUMBER of vector instructions (profiles/r6_block_stamps.md), and component-wise add / subtract / multiply-add
// give the same bits either way.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 lo2(float4 v) { return f32x2{v.x, v.y}; }
__device__ __forceinline__ f32x2 hi2(float4 v) { return f32x2{v.z, v.w}; }
__device__ __forceinline__ float4 cat2(f32x2 l, f32x2 h) { return make_float4(l.x, l.y, h.x, h.y); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return cat2(lo2(a) + lo2(b), hi2(a) + hi2(b)); }
__device__ __forceinline__ float4 psub4(float4 a, float4 b) { return cat2(lo2(a) - lo2(b), hi2(a) - hi2(b)); }
__device__ __forceinline__ float4 pfma4(float w, float4 x, float4 acc) {
    const f32x2 ww = {w, w};
    return cat2(__builtin_elementwise_fma(ww, lo2(x), lo2(acc)), __builtin_elementwise_fma(ww, hi2(x), hi2(acc)));
}

// A workgroup of NW wavefronts owns PB = HP * (NW / 2) * PPW rows: HP "half passes" -- every wavefront carries HP / 2 points per lane
// group, and when HP is odd the FIRST half of the wavefronts one more.  (8 wavefronts, HP = 5) = 640 rows with five wavefront-passes on
// every SIMD (a workgroup's wavefronts w and w + 4 share a SIMD); ten wavefronts of two passes each left two SIMDs with six and the
// other two waiting for them (profiles/r6_block_stamps.md).
template <int H, int NW, int HP>
struct BlkGeo {
    static_assert(NW % 2 == 0, "half passes");
    static constexpr int L = H / 4, PPW = WAVE / L, NT = NW * WAVE, PPL = (HP + 1) / 2, PB = HP * (NW / 2) * PPW, RB = 4 * H;
    static constexpr bool HALF = (HP & 1) != 0;
};

__device__ __forceinline__ float4 lds4(const float4* buf, int byte_addr) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(buf) + byte_addr);
}

// This lane's share of an index row: the K neighbours of a point are split over its L lanes, lane q holds columns KL q .. KL q + KL - 1
// (KL = K / L) and hands a column to the others by a DPP broadcast when its turn comes -- the index row, the weights and hence the
// registers a point occupies across the steps are NOT replicated over its lanes.
template <int K, int L, bool U16>
__device__ __forceinline__ void load_index_share(const int32_t* __restrict__ idx32, const uint16_t* __restrict__ idx16, int r,
                                                 int n_tgt, int n_src, int q, int (&jh)[K / L]) {
    constexpr int KL = K / L;
    if constexpr (U16) {
        static_assert(KL % 4 == 0, "a lane's share is whole 8-byte pieces");
        const int cb = (int)((unsigned)r / (unsigned)n_tgt) * n_src;
        if constexpr (KL % 8 == 0) {
            const uint4* p = reinterpret_cast<const uint4*>(idx16 + (int64_t)r * K + KL * q);
#pragma unroll
            for (int c = 0; c < KL / 8; ++c) {
                const uint4 v = p[c];
                const unsigned wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    jh[8 * c + 2 * e] = cb + (int)(wv[e] & 0xffffu);
                    jh[8 * c + 2 * e + 1] = cb + (int)(wv[e] >> 16);
                }
            }
        } else {
            const uint2 v = *reinterpret_cast<const uint2*>(idx16 + (int64_t)r * K + KL * q);
            jh[0] = cb + (int)(v.x & 0xffffu); jh[1] = cb + (int)(v.x >> 16);
            jh[2] = cb + (int)(v.y & 0xffffu); jh[3] = cb + (int)(v.y >> 16);
        }
    } else {
        const int4* p = reinterpret_cast<const int4*>(idx32 + (int64_t)r * K + KL * q);
#pragma unroll
        for (int c = 0; c < KL / 4; ++c) {
            const int4 v = p[c];
            jh[4 * c] = v.x; jh[4 * c + 1] = v.y; jh[4 * c + 2] = v.z; jh[4 * c + 3] = v.w;
        }
    }
}
template <int L, int HQ>
__device__ __forceinline__ int group_bcast_i(int v, int base_lane) {
    return __float_as_int(group_bcast<L, HQ>(__int_as_float(v), base_lane));
}

// STAMP: diagnostic build only (crfconv_meanfield_forward_block_stamps): thread 0 of every workgroup writes 100 MHz s_memrealtime
// stamps of its phases to dbg[block][..]; no output depends on them.
#define BLK_STAMP(slot)                                                                           \
    do {                                                                                          \
        if constexpr (STAMP) {                                                                    \
            const unsigned long long t_ = __builtin_amdgcn_s_memrealtime();                       \
            if (threadIdx.x == 0) dbg[(size_t)blockIdx.x * 64 + (slot)] = t_;                     \
        }                                                                                         \
    } while (0)

// The grid barrier of gridsync.hpp (fused_grid_sync: same words, same counting) in two halves, so that work that needs nothing from
// other workgroups can sit between them.  blk_arrive: call after every wavefront has drained its write-through stores AND the
// workgroup has synchronised; blk_wait returns false (for the whole workgroup) when the spin gave up.
__device__ __forceinline__ void blk_arrive(unsigned* ws, unsigned phase, unsigned n_in_group, unsigned n_groups) {
    if (threadIdx.x == 0) {
        const unsigned g = blockIdx.x & 7u;
        const unsigned old = __hip_atomic_fetch_add(ws + (FW_CNT + g) * FW_LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == n_in_group * phase) {
            const unsigned o2 = __hip_atomic_fetch_add(ws + FW_TOP * FW_LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (o2 + 1 == n_groups * phase)
                for (unsigned g2 = 0; g2 < n_groups; ++g2)
                    __hip_atomic_store(ws + (FW_GEN + g2) * FW_LINE, phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
__device__ __forceinline__ bool blk_wait(unsigned* ws, unsigned phase, int* s_ok) {
    if (threadIdx.x == 0) {
        unsigned* gen = ws + (FW_GEN + (blockIdx.x & 7u)) * FW_LINE;
        int ok = 1;
        unsigned spins = 0;
        for (;;) {
            if (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= phase) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > FW_SPIN_LIMIT) {
                __hip_atomic_store(ws + FW_FAIL * FW_LINE, 0x200u | phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        *s_ok = ok;
    }
    __syncthreads();
    return *s_ok != 0;
}

template <int H, int K, int NW, int HP, bool U16, bool STAMP = false>
__global__ __launch_bounds__(NW * WAVE) void mf_block_kernel(const float* __restrict__ y, const float* __restrict__ z,
                                                             const int32_t* __restrict__ idx, const uint16_t* __restrict__ idx16,
                                                             int n_tgt, int n_src, const float* __restrict__ Q,
                                                             const float* __restrict__ P, float* __restrict__ s, float* xs,
                                                             int64_t m64, int T, unsigned* ws, unsigned long long* dbg) {
    BLK_STAMP(0);
    using G = BlkGeo<H, NW, HP>;
    constexpr int L = G::L, PPW = G::PPW, NT = G::NT, PB = G::PB, RB = G::RB, PPL = G::PPL, CPR = K / 4, NCH = PPW * CPR, KL = K / L;
    constexpr int TAG = (int)0x80000000u;
    static_assert(L == 2 || L == 4, "neighbour columns travel between a point's lanes as DPP quad permutes");
    static_assert(NCH % WAVE == 0, "weight rows leave as whole 1 KiB stores");
    __shared__ float4 bufA[(PB + 1) * L];            // row 0: zeros (where the LDS read of an out-of-block neighbour goes); own row r at r + 1
    __shared__ float4 bufB[(PB + 1) * L];            // z (step 1 only)
    __shared__ float4 sQ[MatStage<H, NT>::F4];
    __shared__ float4 sP[MatStage<H, NT>::F4];
    __shared__ float4 tile[PPL][NW][NCH];            // a wavefront's weight rows on their way out as 1 KiB-contiguous stores
    __shared__ int s_ok;
    const int m = (int)m64;
    const int lane = threadIdx.x & 63, q = lane % L, gl = lane - q, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned bid = xcd_block_id();
    const int base = (int)bid * PB;
    // the last pass of an odd number of half passes belongs to the first half of the wavefronts only (wave-uniform)
    auto act = [&](const int p) { return !(G::HALF && p == PPL - 1) || wave < NW / 2; };
    if (threadIdx.x < L) {
        bufA[threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
        bufB[threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    // ---------------------------------------------------------------- own rows -> registers + LDS; where every neighbour's row is
    int a[PPL][KL];
    int rl[PPL], own[PPL];
    bool valid[PPL];
    float4 yi[PPL], zi[PPL];
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
        int jh[KL];
        rl[p] = act(p) ? (p * NW + wave) * PPW + lane / L : 0;
        const int r0 = base + rl[p];
        valid[p] = act(p) && r0 < m;
        const int r = valid[p] ? r0 : m - 1;
        own[p] = r * RB + 16 * q;
        load_index_share<K, L, U16>(idx, idx16, r, n_tgt, n_src, q, jh);
        yi[p] = ld4(y + (int64_t)r * H + 4 * q);
        zi[p] = ld4(z + (int64_t)r * H + 4 * q);
#pragma unroll
        for (int i = 0; i < KL; ++i) {
            const int jl = jh[i] - base;
            a[p][i] = (unsigned)jl < (unsigned)PB ? (jl + 1) * RB : (TAG | (jh[i] * RB));
        }
    }
    MatStage<H, NT> mq, mp;
    mq.fetch(Q, false);
    mp.fetch(P, false);
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
        if (act(p)) {
            bufA[(rl[p] + 1) * L + q] = yi[p];
            bufB[(rl[p] + 1) * L + q] = zi[p];
        }
    }
    mq.park(sQ);
    mp.park(sP);
    __syncthreads();
    BLK_STAMP(1);

    const int step_bytes = m * RB;
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(y, step_bytes), rz = make_rsrc(z, step_bytes);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(xs, step_bytes * (T > 0 ? T : 1));

    // column (lq, i) of point p: this lane's piece of the row from LDS and from the row table, added (one of the two is zero)
#define BLK_BOTH(buf, LOADG)                                                              \
    [&](auto LQ_, auto I_) {                                                              \
        const int av = group_bcast_i<L, decltype(LQ_)::value>(a[p][decltype(I_)::value], gl) + 16 * q;   \
        const int la = av > 0 ? av : 0, go = av ^ TAG;                                    \
        (void)go;                                                                         \
        return add4(lds4(buf, la), LOADG);                                                \
    }

    // ---------------------------------------------------------------- similarity + step 1 (bit-identical to sim_step_fast_kernel)
    float wh[PPL][KL];
    float4 zq[PPL], o[PPL];
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
        if (!act(p)) continue;
        float d[K];
        float dmin = 3.4e38f;
        static_for<L>([&](auto LQ) {
            constexpr int lq = decltype(LQ)::value;
            float4 nb[KL];
            auto row = BLK_BOTH(bufA, ld4_buf(ry, go));
            static_for<KL>([&](auto I) {
                if constexpr (KL * lq + decltype(I)::value >= 1) nb[decltype(I)::value] = row(LQ, I);
            });
            static_for<KL>([&](auto I) {
                constexpr int i = decltype(I)::value, k = KL * lq + i;
                if constexpr (k >= 1) {
                    const float4 df = psub4(yi[p], nb[i]);
                    d[k] = group_sum<L>(dot4(df, df));
                    dmin = fminf(dmin, d[k]);
                }
            });
        });
        // this lane's share of the soft-max: exp of its own KL columns only; the denominator is still added in column order
        // (lane 0's columns, handed on, lane 1's, ...): the weights are those of sim_step_fast_kernel bit for bit
        d[0] = dmin;
        float eh[KL];
#pragma unroll
        for (int i = 0; i < KL; ++i) {
            float mine = d[i];
            static_for<L - 1>([&](auto LQ) {
                constexpr int lq = decltype(LQ)::value + 1;
                mine = q == lq ? d[KL * lq + i] : mine;
            });
            eh[i] = __expf(dmin - mine);
        }
        eh[0] = q == 0 ? 0.f : eh[0];
        float acc = 0.f;
        static_for<L>([&](auto LQ) {
            constexpr int lq = decltype(LQ)::value;
            if constexpr (lq > 0) acc = group_bcast<L, lq - 1>(acc, gl);
#pragma unroll
            for (int i = 0; i < KL; ++i) acc += eh[i];          // (meaningful in lane lq; the others add along)
        });
        const float inv = 1.0f / group_bcast<L, L - 1>(acc, gl);
#pragma unroll
        for (int i = 0; i < KL; ++i) wh[p][i] = eh[i] * inv;
        if (s != nullptr) {                                      // parked in the tile; stored BEHIND the x_1 rows (see below)
            float4* mine = tile[p][wave];
            const int pl = lane / L;
#pragma unroll
            for (int c = 0; c < KL / 4; ++c)
                mine[pl * CPR + q * (KL / 4) + c] = make_float4(wh[p][4 * c], wh[p][4 * c + 1], wh[p][4 * c + 2], wh[p][4 * c + 3]);
        }
        if (T > 0) {
            float4 msg = make_float4(0.f, 0.f, 0.f, 0.f);
            static_for<L>([&](auto LQ) {
                constexpr int lq = decltype(LQ)::value;
                float4 nb[KL];
                auto row = BLK_BOTH(bufB, ld4_buf(rz, go));
                static_for<KL>([&](auto I) {
                    if constexpr (KL * lq + decltype(I)::value >= 1) nb[decltype(I)::value] = row(LQ, I);
                });
                static_for<KL>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    if constexpr (KL * lq + i >= 1) msg = pfma4(group_bcast<L, lq>(wh[p][i], gl), nb[i], msg);
                });
            });
            zq[p] = matvec_acc<H>(zi[p], sQ, lane, q, make_float4(0.f, 0.f, 0.f, 0.f));
            o[p] = matvec_acc<H>(msg, sP, lane, q, zq[p]);
        }
        __builtin_amdgcn_sched_barrier(0);             // one point's rows at a time: the next pass's gathers are NOT hoisted over this one's arithmetic (registers)
    }
    BLK_STAMP(2);
    // x_1 first (what the other workgroups wait for), the weight rows behind it: the barrier's drain below leaves them in flight
    if (T > 0) {
#pragma unroll
        for (int p = 0; p < PPL; ++p) {
            if (valid[p]) {
                if (T > 1) st4_sc1(rx, own[p], o[p]);
                else st4(xs + (int64_t)(base + rl[p]) * H + 4 * q, o[p]);
            }
        }
    }
    // the weight rows leave from the tile as 1 KiB-contiguous stores.  With more steps to come they are issued BEHIND the first barrier's
    // arrival (41 KB per CU: their issue alone stalls a wavefront for microseconds while the store path is busy, and nobody waits for them)
    auto store_weights = [&]() {
        __builtin_amdgcn_wave_barrier();                          // LDS operations of one wave complete in order
#pragma unroll
        for (int p = 0; p < PPL; ++p) {
            if (!act(p)) continue;
            const float4* mine = tile[p][wave];
            const int row0 = base + (p * NW + wave) * PPW;
#pragma unroll
            for (int c = lane; c < NCH; c += WAVE)
                if (row0 + c / CPR < m) st4(s + (int64_t)row0 * K + 4 * c, mine[c]);
        }
    };
    if (T <= 1 && s != nullptr) store_weights();
    if (T <= 1) return;

    unsigned n_in_group, n_groups;
    grid_sync_groups(gridDim.x, blockIdx.x, n_in_group, n_groups);
    __syncthreads();                                  // every wavefront is done with its y / z gathers: bufA becomes x_1
#pragma unroll
    for (int p = 0; p < PPL; ++p)
        if (act(p)) bufA[(rl[p] + 1) * L + q] = o[p];
    __syncthreads();
    BLK_STAMP(3);

    // ---------------------------------------------------------------- steps 2 .. T
    // sum over the columns whose rows are in the block (LDS; the others read the zero row) / outside it (buffer loads; the others
    // get an offset past the table's end), in column order each
    auto inblock = [&](auto PP, int q16) {
        constexpr int p = decltype(PP)::value;
        float4 msg = make_float4(0.f, 0.f, 0.f, 0.f);
        static_for<L>([&](auto LQ) {
            constexpr int lq = decltype(LQ)::value;
            float4 nb[KL];
            static_for<KL>([&](auto I) {
                constexpr int i = decltype(I)::value;
                if constexpr (KL * lq + i >= 1) {
                    const int av = group_bcast_i<L, lq>(a[p][i], gl) + q16;
                    nb[i] = lds4(bufA, av > 0 ? av : 0);
                }
            });
            static_for<KL>([&](auto I) {
                constexpr int i = decltype(I)::value;
                if constexpr (KL * lq + i >= 1) msg = pfma4(group_bcast<L, lq>(wh[p][i], gl), nb[i], msg);
            });
            // the next share's reads stay behind this share's sums (its addresses now "depend" on them): one share of rows is the register budget
            asm volatile("" : "+v"(msg.x), "+v"(msg.y), "+v"(msg.z), "+v"(msg.w), "+v"(q16));
        });
        return msg;
    };
    for (int t = 1; t < T; ++t) {
        // (the address decode below is the same in every step: without this opaque copy the compiler hoists all 2 x 15 x PPL decoded
        // addresses out of the loop into registers -- 205 VGPRs, i.e. spills at three wavefronts per SIMD)
        int q16 = 16 * q;
        asm volatile("" : "+v"(q16));
#pragma unroll
        for (int p = 0; p < PPL; ++p) {
#pragma unroll
            for (int i = 0; i < KL; ++i) {          // (the lane-to-lane broadcasts of a and wh are loop-invariant too)
                asm volatile("" : "+v"(a[p][i]));
                asm volatile("" : "+v"(wh[p][i]));
            }
        }
        float4 msgin[PPL];
        msgin[0] = inblock(std::integral_constant<int, 0>{}, q16);
        BLK_STAMP(8 * t + 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the x_t rows have left
        __syncthreads();
        BLK_STAMP(8 * t + 4);
        blk_arrive(ws, (unsigned)t, n_in_group, n_groups);
        if (t == 1 && s != nullptr) store_weights();
        static_for<PPL - 1>([&](auto PP) {
            constexpr int p1 = decltype(PP)::value + 1;
            if (act(p1)) msgin[p1] = inblock(std::integral_constant<int, p1>{}, q16);
        });
        BLK_STAMP(8 * t + 5);
        if (!blk_wait(ws, (unsigned)t, &s_ok)) return;
        BLK_STAMP(8 * t + 1);
        const int sbase = (t - 1) * step_bytes;
#pragma unroll
        for (int p = 0; p < PPL; ++p) {
            if (!act(p)) continue;
            float4 msg = msgin[p];
            static_for<L>([&](auto LQ) {
                constexpr int lq = decltype(LQ)::value;
                float4 nb[KL];
                static_for<KL>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    if constexpr (KL * lq + i >= 1) {
                        const int av = group_bcast_i<L, lq>(a[p][i], gl) + q16;
                        nb[i] = ld4_sc1(rx, av ^ TAG, sbase);
                    }
                });
                static_for<KL>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    if constexpr (KL * lq + i >= 1) msg = pfma4(group_bcast<L, lq>(wh[p][i], gl), nb[i], msg);
                });
            });
            // (no tie between the shares here: all K - 1 loads of a pass in flight -- this part sits behind the barrier, its round trips are the step's critical path)
            asm volatile("" : "+v"(msg.x), "+v"(msg.y), "+v"(msg.z), "+v"(msg.w), "+v"(q16));
            o[p] = matvec_acc<H>(msg, sP, lane, q, zq[p]);
            if (valid[p]) {
                if (t + 1 < T) st4_sc1(rx, own[p] + sbase + step_bytes, o[p]);       // the last step's rows are read by later launches only
                else st4(xs + (int64_t)t * m * H + (int64_t)(base + rl[p]) * H + 4 * q, o[p]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        BLK_STAMP(8 * t + 2);
        if (t + 1 < T) {                              // (blk_wait's barrier: every wavefront is done with x_{t-1} in bufA)
#pragma unroll
            for (int p = 0; p < PPL; ++p)
                if (act(p)) bufA[(rl[p] + 1) * L + q] = o[p];
            __syncthreads();
        }
    }
    if constexpr (STAMP) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        BLK_STAMP(7);
    }
    fused_exit_reset(ws, gridDim.x, T, blockIdx.x);
#undef BLK_BOTH
}

// count[0] += entries of columns k0 .. K-1 whose source row shares the block of `rows` consecutive rows with its target (one thread per row)
__global__ __launch_bounds__(256) void block_locality_kernel(const int32_t* __restrict__ idx, int m, int K, int k0, int rows,
                                                             unsigned long long* __restrict__ count) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    int c = 0;
    if (i < m) {
        const int b = i / rows;
        for (int k = k0; k < K; ++k) {
            const int j = idx[(int64_t)i * K + k];
            c += (j >= 0 && j / rows == b) ? 1 : 0;
        }
    }
    c = (int)wave_sum((float)c);                                  // <= 64 * 64: exact in float
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, (unsigned long long)c);
}

}  // namespace crf

using namespace crf;

namespace {

struct BlkPlan {
    int nw, ppl, pb, nblk;         // (ppl: half passes)
};
// compiled shapes (wavefronts per workgroup, points per lane group), by rows per workgroup.  (10, 2): 144 VGPRs at 3 wavefronts per SIMD.
// Measured and dropped: (4, 5) -- one wavefront per SIMD, five passes each, the only 640-row shape whose wavefronts divide evenly over the
// four SIMDs: 28.7 us against 23.6 (a lone wavefront issues a vector instruction every ~8 cycles); two workgroups of 320 rows per CU
// (10 wavefronts each at 96 VGPRs): not co-resident -- the second workgroup's ten wavefronts do not fit the SIMDs the first left uneven.
constexpr int BLK_SHAPES[][2] = {{4, 4}, {8, 4}, {8, 5}, {12, 4}};      // (wavefronts, half passes): 256 | 512 | 640 | 768 rows

// The smallest compiled block that covers m rows with no more workgroups than the device has CUs, or nblk = 0 when none does.
template <int H>
BlkPlan plan_for(int64_t m, int cus) {
    constexpr int PPW = WAVE / (H / 4);
    for (const auto& sh : BLK_SHAPES) {
        const int pb = sh[1] * (sh[0] / 2) * PPW;
        const int64_t nblk = cdiv(m, pb);
        if (nblk <= cus) return {sh[0], sh[1], pb, (int)nblk};
    }
    return {0, 0, 0, 0};
}

int device_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        cus = n;
    }
    return cus;
}

template <int H, int K, int NW, int HP>
int launch_block(const float* z, const float* y, const int32_t* idx32, const uint16_t* idx16, int n_tgt, int n_src, int64_t m,
                 const float* Q, const float* P, int T, float* s, float* xs, unsigned* ws, int nblk, hipStream_t st) {
    if (idx16) hipLaunchKernelGGL((mf_block_kernel<H, K, NW, HP, true>), dim3((unsigned)nblk), dim3(NW * WAVE), 0, st, y, z, idx32, idx16, n_tgt, n_src, Q, P, s, xs, m, T, ws, (unsigned long long*)nullptr);
    else hipLaunchKernelGGL((mf_block_kernel<H, K, NW, HP, false>), dim3((unsigned)nblk), dim3(NW * WAVE), 0, st, y, z, idx32, idx16, n_tgt, n_src, Q, P, s, xs, m, T, ws, (unsigned long long*)nullptr);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

}  // namespace

/* Rows per workgroup of crfconv_meanfield_forward_block for m rows on the current device, 0 when the shape is not covered
 * (H = 8, K = 16, k0 = 1, T >= 1; at most one block of <= 768 rows per CU; row tables below 2 GiB).  A grid of at most one
 * workgroup per CU is always co-resident (every workgroup fits a CU alone).  Whether the form PAYS depends on the point order:
 * crfconv_block_locality(). */
extern "C" int crfconv_meanfield_forward_block_rows(int64_t m, int H, int K, int k0, int T) {
    if (H != 8 || K != 16 || k0 != 1 || T < 1 || m <= 0 || m * H * 4 * (int64_t)T >= ((int64_t)1 << 31) - 4096) return 0;
    const int cus = device_cus();
    return cus > 0 ? plan_for<8>(m, cus).pb : 0;
}

/* crfconv_meanfield_forward_u16 as ONE launch with block-resident rows (this file's header).  ws: the grid-barrier words
 * (crfconv_gridsync_workspace() bytes, zero before the first launch, left zero). */
extern "C" int crfconv_meanfield_forward_block(const float* z, const float* y, const int32_t* idx32, const uint16_t* idx16,
                                               int n_tgt, int n_src, int K, int k0, int64_t m, int H, const float* Q,
                                               const float* P, int T, float* s, float* xs, void* ws, crf_stream_t stream) {
    if (int rc = check_common(m, H, K, k0)) return rc;
    CRF_REQUIRE(z && y && idx32 && Q && P && xs && ws, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(s || T == 1, CRF_ERR_ARG, "s == NULL needs T == 1");
    CRF_REQUIRE(idx16 == nullptr || (n_tgt > 0 && n_src > 0 && n_src <= 65536 && m % n_tgt == 0), CRF_ERR_ARG,
                "u16 table needs n_src <= 65536 and m a multiple of n_tgt (n_tgt=%d n_src=%d)", n_tgt, n_src);
    CRF_REQUIRE(crfconv_meanfield_forward_block_rows(m, H, K, k0, T) > 0, CRF_ERR_UNSUPPORTED,
                "block-resident mean field: shape m=%lld H=%d K=%d k0=%d T=%d is not covered", (long long)m, H, K, k0, T);
    const BlkPlan pl = plan_for<8>(m, device_cus());
    hipStream_t st = as_stream(stream);
    unsigned* w = reinterpret_cast<unsigned*>(ws);
#define BLK_CASE(NW_, PPL_) if (pl.nw == NW_ && pl.ppl == PPL_) return launch_block<8, 16, NW_, PPL_>(z, y, idx32, idx16, n_tgt, n_src, m, Q, P, T, s, xs, w, pl.nblk, st)
    BLK_CASE(4, 4);
    BLK_CASE(8, 4);
    BLK_CASE(8, 5);
    BLK_CASE(12, 4);
#undef BLK_CASE
    CRF_REQUIRE(false, CRF_ERR_UNSUPPORTED, "block-resident mean field: no kernel for %d wavefronts x %d passes", pl.nw, pl.ppl);
}

/* Diagnostic twin (scratch/mf_block_stamps.py): 100 MHz phase stamps per workgroup in dbg [blocks][64] (u64); 640 rows per workgroup
 * (shape 0: 8 wavefronts x 5 half passes -- the shipped shape; 1: 10 wavefronts x 2 passes), uint16 tables only. */
extern "C" int crfconv_meanfield_forward_block_stamps(const float* z, const float* y, const int32_t* idx32, const uint16_t* idx16,
                                                      int n_tgt, int n_src, int64_t m, const float* Q, const float* P, int T, float* s,
                                                      float* xs, void* ws, int shape, unsigned long long* dbg, crf_stream_t stream) {
    CRF_REQUIRE(z && y && idx32 && idx16 && Q && P && s && xs && ws && dbg && T >= 1, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(cdiv(m, 640) <= device_cus() && m * 32 * (int64_t)T < ((int64_t)1 << 31) - 4096, CRF_ERR_UNSUPPORTED, "too many rows");
    if (shape == 1)
        hipLaunchKernelGGL((mf_block_kernel<8, 16, 10, 4, true, true>), dim3((unsigned)cdiv(m, 640)), dim3(640), 0, as_stream(stream), y, z, idx32, idx16,
                           n_tgt, n_src, Q, P, s, xs, m, T, reinterpret_cast<unsigned*>(ws), dbg);
    else
        hipLaunchKernelGGL((mf_block_kernel<8, 16, 8, 5, true, true>), dim3((unsigned)cdiv(m, 640)), dim3(512), 0, as_stream(stream), y, z, idx32, idx16,
                           n_tgt, n_src, Q, P, s, xs, m, T, reinterpret_cast<unsigned*>(ws), dbg);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_block_locality(const int32_t* idx32, int64_t m, int K, int k0, int rows, unsigned long long* count, crf_stream_t stream) {
    CRF_REQUIRE(idx32 && count && m > 0 && m < ((int64_t)1 << 31) && K >= 1 && K <= 64 && k0 >= 0 && k0 < K && rows >= 1, CRF_ERR_ARG,
                "block_locality: bad argument (m=%lld K=%d k0=%d rows=%d)", (long long)m, K, k0, rows);
    hipStream_t st = as_stream(stream);
    CRF_HIP(hipMemsetAsync(count, 0, sizeof(unsigned long long), st));
    hipLaunchKernelGGL(block_locality_kernel, dim3((unsigned)cdiv(m, 256)), dim3(256), 0, st, idx32, (int)m, K, k0, rows, count);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}
