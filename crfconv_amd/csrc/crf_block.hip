// Mean-field forward as ONE launch with block-resident rows (round 6).
//
// The per-step kernels of crf.hip are bound by what a launch boundary forces them to move: every step re-reads the index row,
// the weight row and z of every point (96 of its 192 bytes per point at H = 8, K = 16), and every neighbour row goes through the
// texture path, which delivers ~one 128-byte line per clock and CU whatever fraction of the line is wanted (a 32-byte row = a
// quarter).  This kernel removes both for SPATIALLY SORTED clouds (the device collate emits Morton order):
//
//   * one workgroup per CU owns PB CONSECUTIVE rows for the whole forward.  Index rows, soft-max weights and z Q of its points stay
//     in REGISTERS across the T steps (a lane pair carries PPL points); nothing but x_{t-1} is read again.
//   * the block's own rows of y | z (step 1) and of x_{t-1} (later steps) sit in LDS.  In Morton order ~80 % of a point's
//     neighbours are rows of its own block of 640 (measured on the 4 cm synthetic cloud: 0.80 at 640 rows, 0.84 at 1024):
//     they are ds_read_b128 from LDS (256 B/clk/CU, no tag lookups).  The rest are buffer loads as before.
//   * both paths are issued for every neighbour, branch-free: the LDS read of an out-of-block neighbour goes to a row of zeros,
//     the buffer load of an in-block one gets an offset past the resource's end (returns 0, no memory request); the two
//     results are OR-ed (x | +0.0 is x, bit for bit).
//   * between steps: the grid barrier of gridsync.hpp.  Rows are published write-through (sc1) and remote rows are read with
//     sc1 loads (another CU's stores never refresh this CU's L1); with four fifths of the gathers in LDS the slow L1-bypassing
//     path (2.4x the L1 path) carries a fifth of what sank the first one-launch form.
//
// Results are those of crf.hip's kernels bit for bit (same operations in the same order per point; tests/test_gpu_model.py).
// Co-residency: one workgroup per CU at most (the host checks the grid against the device's CU count and the occupancy query);
// a barrier that cannot complete gives up after ~1 s with the sticky failure word (ops.check_gridsync).
//
// Reference semantics: models/continuous_crf_conv_big.py:49-54 (similarity), :63-72 (loop).
#include "crf_common.hpp"

namespace crf {

__device__ __forceinline__ float4 or4(float4 a, float4 b) {
    return make_float4(__uint_as_float(__float_as_uint(a.x) | __float_as_uint(b.x)), __uint_as_float(__float_as_uint(a.y) | __float_as_uint(b.y)),
                       __uint_as_float(__float_as_uint(a.z) | __float_as_uint(b.z)), __uint_as_float(__float_as_uint(a.w) | __float_as_uint(b.w)));
}

constexpr int BLK_OOB = 0x7ffffff0;          // a byte offset past the end of every row table (tables are < 2 GiB: the launcher checks)

template <int H, int NW, int PPL>
struct BlkGeo {
    static constexpr int L = H / 4, PPW = WAVE / L, NT = NW * WAVE, PB = PPL * NW * PPW, RB = 4 * H;
    static constexpr int LDS_BYTES = 2 * (PB + 1) * L * 16;
};

// lds: byte address of this lane's 16-byte piece inside a row buffer; glob: byte offset for the buffer load
template <int H, int PB>
__device__ __forceinline__ void blk_addr(int j, int base, int q, int& lds, int& glob) {
    constexpr int L = H / 4, RB = 4 * H;
    const int jl = j - base;
    const bool in = (unsigned)jl < (unsigned)PB;
    lds = ((in ? jl : PB) * L + q) * 16;
    glob = in ? BLK_OOB : j * RB + 16 * q;
}
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

template <int H, int K, int NW, int PPL, bool U16>
__global__ __launch_bounds__(NW * WAVE) void mf_block_kernel(const float* __restrict__ y, const float* __restrict__ z,
                                                             const int32_t* __restrict__ idx, const uint16_t* __restrict__ idx16,
                                                             int n_tgt, int n_src, const float* __restrict__ Q,
                                                             const float* __restrict__ P, float* __restrict__ s, float* xs,
                                                             int64_t m64, int T, unsigned* ws) {
    using G = BlkGeo<H, NW, PPL>;
    constexpr int L = G::L, PPW = G::PPW, NT = G::NT, PB = G::PB, CPR = K / 4, NCH = PPW * CPR, KL = K / L;
    static_assert(L == 2 || L == 4, "neighbour columns travel between a point's lanes as DPP quad permutes");
    __shared__ float4 bufA[(PB + 1) * L];            // row PB: zeros (where the LDS read of an out-of-block neighbour goes)
    __shared__ float4 bufB[(PB + 1) * L];
    __shared__ float4 sQ[MatStage<H, NT>::F4];
    __shared__ float4 sP[MatStage<H, NT>::F4];
    __shared__ float4 tile[NW][NCH];                 // a wavefront's weight rows on their way out as 1 KiB-contiguous stores
    __shared__ int s_ok;
    const int m = (int)m64;
    const int lane = threadIdx.x & 63, q = lane % L, gl = lane - q, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned bid = xcd_block_id();
    const int base = (int)bid * PB;
    if (threadIdx.x < L) {
        bufA[PB * L + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
        bufB[PB * L + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    // ---------------------------------------------------------------- own rows -> registers + LDS
    int jh[PPL][KL];
    int rl[PPL], own[PPL];
    bool valid[PPL];
    float4 yi[PPL], zi[PPL];
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
        rl[p] = (p * NW + wave) * PPW + lane / L;
        const int r0 = base + rl[p];
        valid[p] = r0 < m;
        const int r = valid[p] ? r0 : m - 1;
        own[p] = r * G::RB + 16 * q;
        load_index_share<K, L, U16>(idx, idx16, r, n_tgt, n_src, q, jh[p]);
        yi[p] = ld4(y + (int64_t)r * H + 4 * q);
        zi[p] = ld4(z + (int64_t)r * H + 4 * q);
    }
    MatStage<H, NT> mq, mp;
    mq.fetch(Q, false);
    mp.fetch(P, false);
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
        bufA[rl[p] * L + q] = yi[p];
        bufB[rl[p] * L + q] = zi[p];
    }
    mq.park(sQ);
    mp.park(sP);
    __syncthreads();

    const int step_bytes = m * G::RB;
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(y, step_bytes), rz = make_rsrc(z, step_bytes);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(xs, step_bytes * (T > 0 ? T : 1));

    // one share of neighbour rows (columns KL lq .. KL lq + KL - 1, minus column 0) from an LDS buffer + a row table
#define BLK_GATHER(lq, buf, LOADG)                                                        \
    float4 nb[KL];                                                                        \
    static_for<KL>([&](auto I) {                                                          \
        constexpr int i = decltype(I)::value;                                             \
        if constexpr (KL * lq + i >= 1) {                                                 \
            int la, go;                                                                   \
            blk_addr<H, PB>(group_bcast_i<L, lq>(jh[p][i], gl), base, q, la, go);         \
            nb[i] = or4(lds4(buf, la), LOADG);                                            \
        }                                                                                 \
    })

    // ---------------------------------------------------------------- similarity + step 1
    float wh[PPL][KL];
    float4 zq[PPL], o[PPL];
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
        float d[K];
        float dmin = 3.4e38f;
        static_for<L>([&](auto LQ) {
            constexpr int lq = decltype(LQ)::value;
            BLK_GATHER(lq, bufA, ld4_buf(ry, go));
            static_for<KL>([&](auto I) {
                constexpr int i = decltype(I)::value, k = KL * lq + i;
                if constexpr (k >= 1) {
                    const float4 df = sub4(yi[p], nb[i]);
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
        if (s != nullptr) {
            float4* mine = tile[wave];
            const int pl = lane / L;
#pragma unroll
            for (int c = 0; c < KL / 4; ++c)
                mine[pl * CPR + q * (KL / 4) + c] = make_float4(wh[p][4 * c], wh[p][4 * c + 1], wh[p][4 * c + 2], wh[p][4 * c + 3]);
            __builtin_amdgcn_wave_barrier();
            const int row0 = base + (p * NW + wave) * PPW;
#pragma unroll
            for (int c = lane; c < NCH; c += WAVE)
                if (row0 + c / CPR < m) st4(s + (int64_t)row0 * K + 4 * c, mine[c]);
            __builtin_amdgcn_wave_barrier();
        }
        if (T > 0) {
            float4 msg = make_float4(0.f, 0.f, 0.f, 0.f);
            static_for<L>([&](auto LQ) {
                constexpr int lq = decltype(LQ)::value;
                BLK_GATHER(lq, bufB, ld4_buf(rz, go));
                static_for<KL>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    if constexpr (KL * lq + i >= 1) msg = fma4(group_bcast<L, lq>(wh[p][i], gl), nb[i], msg);
                });
            });
            zq[p] = matvec_acc<H>(zi[p], sQ, lane, q, make_float4(0.f, 0.f, 0.f, 0.f));
            o[p] = matvec_acc<H>(msg, sP, lane, q, zq[p]);
            if (valid[p]) {
                if (T > 1) st4_sc1(rx, own[p], o[p]);
                else st4(xs + (int64_t)(base + rl[p]) * H + 4 * q, o[p]);
            }
        }
    }
    if (T <= 1) return;

    unsigned n_in_group, n_groups;
    grid_sync_groups(gridDim.x, blockIdx.x, n_in_group, n_groups);
    __syncthreads();                                  // every wavefront is done with its y / z gathers: bufA becomes x_1
#pragma unroll
    for (int p = 0; p < PPL; ++p) bufA[rl[p] * L + q] = o[p];

    // ---------------------------------------------------------------- steps 2 .. T
    for (int t = 1; t < T; ++t) {
        if (!fused_grid_sync<false>(ws, (unsigned)t, n_in_group, n_groups, &s_ok, nullptr, blockIdx.x)) return;
        const float4* cur = (t & 1) ? bufA : bufB;
        float4* nxt = (t & 1) ? bufB : bufA;
        const int sbase = (t - 1) * step_bytes;
#pragma unroll
        for (int p = 0; p < PPL; ++p) {
            float4 msg = make_float4(0.f, 0.f, 0.f, 0.f);
            static_for<L>([&](auto LQ) {
                constexpr int lq = decltype(LQ)::value;
                BLK_GATHER(lq, cur, ld4_sc1(rx, go, sbase));
                static_for<KL>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    if constexpr (KL * lq + i >= 1) msg = fma4(group_bcast<L, lq>(wh[p][i], gl), nb[i], msg);
                });
            });
            o[p] = matvec_acc<H>(msg, sP, lane, q, zq[p]);
            if (valid[p]) {
                if (t + 1 < T) st4_sc1(rx, own[p] + sbase + step_bytes, o[p]);       // the last step's rows are read by later launches only
                else st4(xs + (int64_t)t * m * H + (int64_t)(base + rl[p]) * H + 4 * q, o[p]);
            }
            nxt[rl[p] * L + q] = o[p];
        }
    }
    fused_exit_reset(ws, gridDim.x, T, blockIdx.x);
#undef BLK_GATHER
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
    int nw, pb, nblk;
};
constexpr int BLK_PPL = 2;                          // points per lane group (the register budget: 158 VGPRs at 3 wavefronts per SIMD)
constexpr int BLK_NWS[] = {4, 8, 10, 12};           // wavefronts per workgroup of the compiled shapes (more than 12 = 4 per SIMD = 128 VGPRs: spills)

// PB = PPL * NW * PPW rows per workgroup with at most one workgroup per CU: the smallest compiled block that covers m rows with no
// more workgroups than the device has CUs, or nblk = 0 when none does.
template <int H>
BlkPlan plan_for(int64_t m, int cus) {
    constexpr int PPW = WAVE / (H / 4);
    for (int nw : BLK_NWS) {
        const int pb = nw * BLK_PPL * PPW;
        const int64_t nblk = cdiv(m, pb);
        if (nblk <= cus) return {nw, pb, (int)nblk};
    }
    return {0, 0, 0};
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

template <int H, int K, int NW>
int launch_block(const float* z, const float* y, const int32_t* idx32, const uint16_t* idx16, int n_tgt, int n_src, int64_t m,
                 const float* Q, const float* P, int T, float* s, float* xs, unsigned* ws, int nblk, hipStream_t st) {
    if (idx16) hipLaunchKernelGGL((mf_block_kernel<H, K, NW, BLK_PPL, true>), dim3((unsigned)nblk), dim3(NW * WAVE), 0, st, y, z, idx32, idx16, n_tgt, n_src, Q, P, s, xs, m, T, ws);
    else hipLaunchKernelGGL((mf_block_kernel<H, K, NW, BLK_PPL, false>), dim3((unsigned)nblk), dim3(NW * WAVE), 0, st, y, z, idx32, idx16, n_tgt, n_src, Q, P, s, xs, m, T, ws);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

}  // namespace

/* Rows per workgroup of crfconv_meanfield_forward_block for m rows on the current device, 0 when the shape is not covered
 * (H = 8, K = 16, k0 = 1, T >= 1; at most one block of <= 768 rows per CU; row tables below 2 GiB).  A grid of at most one
 * workgroup per CU is always co-resident (every workgroup fits a CU alone: 67-80 KB of LDS, <= 768 threads).  Whether the form
 * PAYS depends on the point order: crfconv_block_locality(). */
extern "C" int crfconv_meanfield_forward_block_rows(int64_t m, int H, int K, int k0, int T) {
    if (H != 8 || K != 16 || k0 != 1 || T < 1 || m <= 0 || m * H * 4 * (int64_t)T >= ((int64_t)1 << 31) - 64) return 0;
    const int cus = device_cus();
    return cus > 0 ? plan_for<8>(m, cus).pb : 0;
}

/* crfconv_meanfield_forward_u16 as ONE launch with block-resident rows (this file's header).  ws: the grid-barrier words
 * (crfconv_gridsync_workspace() bytes, zero before the first launch, left zero).  Same outputs, bit for bit. */
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
    switch (pl.nw) {
        case 4: return launch_block<8, 16, 4>(z, y, idx32, idx16, n_tgt, n_src, m, Q, P, T, s, xs, w, pl.nblk, st);
        case 8: return launch_block<8, 16, 8>(z, y, idx32, idx16, n_tgt, n_src, m, Q, P, T, s, xs, w, pl.nblk, st);
        case 10: return launch_block<8, 16, 10>(z, y, idx32, idx16, n_tgt, n_src, m, Q, P, T, s, xs, w, pl.nblk, st);
        default: return launch_block<8, 16, 12>(z, y, idx32, idx16, n_tgt, n_src, m, Q, P, T, s, xs, w, pl.nblk, st);
    }
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
