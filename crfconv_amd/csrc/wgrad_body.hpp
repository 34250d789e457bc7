// The weight-gradient partial pass  partial[slice][co][ci] = sum_{rows of the slice} G[r][co] X[r][ci]  as a device routine with its job
// table, shared by the translation units that run it: linear.hip (the pass's own launches) and gemm.hip (round 5: the pending
// passes of a backward ride as SIDE jobs on the coarse-level launches of the chain, whose own workgroups cover a fraction of the chip).
#pragma once
#include "common.hpp"

namespace crf {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int WG_BLOCK = 256;
constexpr int WG_WAVES = WG_BLOCK / WAVE;
#ifndef WG_RED_BUFS_
#define WG_RED_BUFS_ 2       // wave-sized LDS buffers of the cross-wave sum (1, 2 or 4 = one round)
#endif
constexpr int WG_RED_BUFS = WG_RED_BUFS_;

// TCO x TCI tiles of 16x16 per block (output slab 16*TCO x 16*TCI at (co0, ci0) = blockIdx.y / z).
template <int TCO, int TCI>
__device__ __forceinline__ void wgrad_body(const float* __restrict__ G, const float* __restrict__ X, int64_t M, int Co, int Ci,
                                           int rows_per_block, float* __restrict__ partial /*[nblk][Co][Ci]*/,
                                           float* __restrict__ partial_b /*[nblk][Co] or null*/, int bx, int by, int bz,
                                           float* __restrict__ s_red_ /*[WG_RED_BUFS][TCO TCI 256]*/, float* __restrict__ s_b_ /*[waves][16 TCO]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int co_base = by * 16 * TCO, ci_base = bz * 16 * TCI;
    const int kk = lane >> 4, cc = lane & 15;
    f32x4 acc[TCO][TCI];
#pragma unroll
    for (int a = 0; a < TCO; ++a)
#pragma unroll
        for (int b = 0; b < TCI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[TCO];
#pragma unroll
    for (int a = 0; a < TCO; ++a) bsum[a] = 0.f;

    const int64_t row_begin = (int64_t)bx * rows_per_block;
    const int64_t row_end = row_begin + rows_per_block < M ? row_begin + rows_per_block : M;
    // waves interleave 16-row groups (4 k-steps) inside the block's slice; all operand loads of a group are issued
    // before its MFMAs, so each lane keeps 4 (TCO + TCI) dword loads in flight
#ifndef WG_PREFETCH_
#define WG_PREFETCH_ 0
#endif
    struct Frag { float av[4][TCO], bv[4][TCI]; };
    auto load = [&](int64_t r0, Frag& f) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t r = r0 + 4 * u + kk;
            const bool rv = r < row_end;
#pragma unroll
            for (int a = 0; a < TCO; ++a) {
                const int co = co_base + 16 * a + cc;
                f.av[u][a] = (rv && co < Co) ? G[r * Co + co] : 0.f;
            }
#pragma unroll
            for (int b = 0; b < TCI; ++b) {
                const int ci = ci_base + 16 * b + cc;
                f.bv[u][b] = (rv && ci < Ci) ? X[r * Ci + ci] : 0.f;
            }
        }
    };
    auto compute = [&](const Frag& f) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int a = 0; a < TCO; ++a) {
                bsum[a] += f.av[u][a];
#pragma unroll
                for (int b = 0; b < TCI; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.av[u][a], f.bv[u][b], acc[a][b], 0, 0, 0);
            }
    };
    if constexpr (WG_PREFETCH_ != 0) {
        Frag cur, nxt;                                  // the next group's operands on their way while this group is on the matrix pipe
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
    // LDS of the caller (a kernel that serves several tile classes owns ONE buffer of the largest class's size).  The four waves' tiles
    // go through WG_RED_BUFS wave-sized buffers in rounds (round 6; one buffer per wave until then: 64 KB for the <4, 4> class, two
    // workgroups = two waves per SIMD on a CU whatever the job's class): the waves of a round park, every thread adds them to its slots,
    // ((s0 + s1) + s2) + s3 -- the order of the one-round sum, every bit.
    constexpr int NSLOT = TCO * TCI * 256, PER = NSLOT / WG_BLOCK;
    static_assert(NSLOT % WG_BLOCK == 0 && WG_WAVES % WG_RED_BUFS == 0, "slots per thread; whole rounds");
    float (*s_red)[NSLOT] = reinterpret_cast<float (*)[NSLOT]>(s_red_);
    float (*s_b)[TCO * 16] = reinterpret_cast<float (*)[TCO * 16]>(s_b_);
    auto park = [&]() {
#pragma unroll
        for (int a = 0; a < TCO; ++a)
#pragma unroll
            for (int b = 0; b < TCI; ++b)
#pragma unroll
                for (int g = 0; g < 4; ++g) s_red[wave % WG_RED_BUFS][(a * TCI + b) * 256 + (4 * kk + g) * 16 + cc] = acc[a][b][g];
    };
#pragma unroll
    for (int a = 0; a < TCO; ++a) {
        // bias: lanes with the same cc over the 4 k-groups
        float t = bsum[a];
        t += __shfl_xor(t, 16, WAVE);
        t += __shfl_xor(t, 32, WAVE);
        if (kk == 0) s_b[wave][a * 16 + cc] = t;
    }
    // rounds of WG_RED_BUFS waves: park, every thread adds the parked tiles to its running slots in wave order
    float v[PER];
#pragma unroll
    for (int rd = 0; rd < WG_WAVES / WG_RED_BUFS; ++rd) {
        if (rd > 0) __syncthreads();                    // the previous round's readers are done with the buffers
        if (wave / WG_RED_BUFS == rd) park();
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int t = threadIdx.x + WG_BLOCK * i;
#pragma unroll
            for (int b = 0; b < WG_RED_BUFS; ++b) v[i] = (rd == 0 && b == 0) ? s_red[0][t] : v[i] + s_red[b][t];
        }
    }
    const int64_t pb = (int64_t)bx;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int t = threadIdx.x + WG_BLOCK * i;
        const int tile = t >> 8, a = tile / TCI, b = tile % TCI, r = (t >> 4) & 15, j = t & 15;
        const int co = co_base + 16 * a + r, ci = ci_base + 16 * b + j;
        if (co < Co && ci < Ci) partial[(pb * Co + co) * Ci + ci] = v[i];
    }
    if (partial_b != nullptr && bz == 0) {
        for (int t = threadIdx.x; t < TCO * 16; t += WG_BLOCK) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < WG_WAVES; ++w) v += s_b[w][t];
            const int co = co_base + t;
            if (co < Co) partial_b[pb * Co + co] = v;
        }
    }
}

// The partial passes of SEVERAL layers in one launch (crfconv_linear_wgrad_partial_jobs): the weight gradients of the coarse
// levels are ~9 us launches of a few workgroups each and nothing on the backward chain waits for them, so they are queued and
// issued together once the chain is done -- the jobs' workgroups (laid end to end, job found by a binary search over the
// prefix) run side by side.  Same partial slabs as one wgrad_kernel launch per job.
constexpr int WJ_MAX = 48;          // (48 x 61 bytes of table in the kernel arguments; config 2 queues 35 passes: one launch)
struct WgJobTable {
    const float* G[WJ_MAX];
    const float* X[WJ_MAX];
    float* partial[WJ_MAX];
    float* partial_b[WJ_MAX];
    int M[WJ_MAX], Co[WJ_MAX], Ci[WJ_MAX], rows_per_block[WJ_MAX], nblk[WJ_MAX], gy[WJ_MAX];
    int blk_base[WJ_MAX + 1];
    int njobs;
    unsigned char cls[WJ_MAX];                         // 10 TCO + TCI (wgrad_jobs_any_kernel)
};
// One workgroup of a job table: looks its job up (binary search over the prefix), dispatches on the job's tile class; the caller owns
// ONE LDS buffer of the largest class (s_red: WG_RED_BUFS * 4 * 4 * 256 floats, s_b: WG_WAVES * 4 * 16).
__device__ __forceinline__ void wgrad_any_run(const WgJobTable& t, int blk, float* s_red, float* s_b) {
    int lo = 0, hi = t.njobs;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (t.blk_base[mid] <= blk) lo = mid; else hi = mid;
    }
    const int local = blk - t.blk_base[lo];
    const int bx = local % t.nblk[lo], rest = local / t.nblk[lo];
    const int by = rest % t.gy[lo], bz = rest / t.gy[lo];
#define WB(TA, TB) wgrad_body<TA, TB>(t.G[lo], t.X[lo], t.M[lo], t.Co[lo], t.Ci[lo], t.rows_per_block[lo], t.partial[lo], t.partial_b[lo], bx, by, bz, s_red, s_b)
    switch (t.cls[lo]) {
        case 44: WB(4, 4); break;
        case 22: WB(2, 2); break;
        case 42: WB(4, 2); break;
        case 24: WB(2, 4); break;
        case 14: WB(1, 4); break;
        case 41: WB(4, 1); break;
        case 12: WB(1, 2); break;
        case 21: WB(2, 1); break;
        default: WB(1, 1); break;
    }
#undef WB
}

struct WgPlan {
    int tco, tci, gy, gz, nblk, rows_per_block;
};
#ifndef WG_LONG_ROWS_
#define WG_LONG_ROWS_ 256     // swept on the training step, one box, two runs each: 64 (round 3's plan) 4.413 ms, 128 4.398, 192 4.395, 256 4.386, 320 4.397, 384 4.409, 512 4.411
#endif


static WgPlan wg_plan(int64_t M, int Co, int Ci) {
    WgPlan p;
    const int t_co = (Co + 15) / 16, t_ci = (Ci + 15) / 16;
    p.tco = t_co >= 4 ? 4 : (t_co >= 2 ? 2 : 1);
    p.tci = t_ci >= 4 ? 4 : (t_ci >= 2 ? 2 : 1);
    p.gy = (t_co + p.tco - 1) / p.tco;
    p.gz = (t_ci + p.tci - 1) / p.tci;
    // ~512 workgroups over the chip = row-slices x (gy x gz output slabs); at least 32 slices, at least 64 rows
    // (one 16-row group per wave) each: the partial slabs the second kernel sums stay a small fraction of the
    // operand bytes
    int64_t slices = 512 / ((int64_t)p.gy * p.gz);
    if (slices < 32) slices = 32;
    int64_t rows = (M + slices - 1) / slices;
    if (rows < 64) rows = 64;
    rows = (rows + 63) / 64 * 64;
#if WG_LONG_ROWS_ > 64
    // Longer slices for the layers whose partial passes share one launch (M <= 65536: crfconv_linear_wgrad_partial_jobs, ~22
    // wavefronts per SIMD of work): every slice writes a Co x Ci slab that the reduce launch reads again -- 92 MB written and
    // re-read per step with the 64 ... 192-row slices of the plan above; past 256 rows the launch's longest workgroups cost more
    // than the slabs save
    if (M <= 65536) {
        int64_t want = WG_LONG_ROWS_;
        if (M < 2 * want) want = ((M + 1) / 2 + 63) / 64 * 64;
        if (rows < want) rows = want;
    }
#endif
    p.rows_per_block = (int)rows;
    p.nblk = (int)((M + rows - 1) / rows);
    return p;
}


// host: the table of n <= WJ_MAX jobs (validated by the caller) and the number of workgroups they need
static inline void wg_fill_table(const crf_wgrad_job* jobs, int n, WgJobTable& t, int64_t& blocks) {
    blocks = 0;
    for (int k = 0; k < n; ++k) {
        const crf_wgrad_job& jb = jobs[k];
        const WgPlan p = wg_plan(jb.M, jb.Co, jb.Ci);
        float* partial = reinterpret_cast<float*>(jb.workspace);
        t.G[k] = jb.G; t.X[k] = jb.X; t.partial[k] = partial;
        t.partial_b[k] = jb.want_bias ? partial + (size_t)p.nblk * jb.Co * jb.Ci : nullptr;
        t.M[k] = (int)jb.M; t.Co[k] = jb.Co; t.Ci[k] = jb.Ci; t.rows_per_block[k] = p.rows_per_block; t.nblk[k] = p.nblk; t.gy[k] = p.gy;
        t.cls[k] = (unsigned char)(10 * p.tco + p.tci);
        t.blk_base[k] = (int)blocks;
        blocks += (int64_t)p.nblk * p.gy * p.gz;
    }
    for (int k = n; k <= WJ_MAX; ++k) t.blk_base[k] = (int)blocks;
    for (int k = n; k < WJ_MAX; ++k) {
        t.G[k] = nullptr; t.X[k] = nullptr; t.partial[k] = nullptr; t.partial_b[k] = nullptr;
        t.M[k] = 0; t.Co[k] = 1; t.Ci[k] = 1; t.rows_per_block[k] = 64; t.nblk[k] = 1; t.gy[k] = 1; t.cls[k] = 11;
    }
    t.njobs = n;
}

}  // namespace crf
