// Helpers shared by the mean-field translation units (crf.hip: forward + generic backward; crf_bwd.hip: the
// restructured backward for K in {16, 32}, k0 = 1).  Thread mapping and layout: see the header of crf.hip.
#pragma once
#include "common.hpp"
#include "gridsync.hpp"
#include "outer_acc.hpp"

#include <cstdlib>

namespace crf {

constexpr int BLOCK = 256;

// v (float4 per lane, quad q of the point's H-vector)  ->  acc + v_full * Mat, where
// sM holds Mat [H][H] row-major as float4 rows: sM[h * L + q] = Mat[h][4q .. 4q+3].
template <int H>
__device__ __forceinline__ float4 matvec_acc(float4 v, const float4* sM, int lane, int q, float4 acc) {
    constexpr int L = H / 4;
    const int base = lane - q;
    static_for<L>([&](auto HQ) {                     // broadcasts inside a DPP quad for H <= 16 (group_bcast), shuffles above
        constexpr int hq = decltype(HQ)::value;
        const float v0 = group_bcast<L, hq>(v.x, base);
        const float v1 = group_bcast<L, hq>(v.y, base);
        const float v2 = group_bcast<L, hq>(v.z, base);
        const float v3 = group_bcast<L, hq>(v.w, base);
        acc = fma4(v0, sM[(4 * hq + 0) * L + q], acc);
        acc = fma4(v1, sM[(4 * hq + 1) * L + q], acc);
        acc = fma4(v2, sM[(4 * hq + 2) * L + q], acc);
        acc = fma4(v3, sM[(4 * hq + 3) * L + q], acc);
    });
    return acc;
}

template <int H, int NT = BLOCK>
__device__ __forceinline__ void load_matrix(float4* sM, const float* __restrict__ Mat, bool transpose) {
    // sM[h][c] = transpose ? Mat[c][h] : Mat[h][c]
    float* s = reinterpret_cast<float*>(sM);
    for (int t = threadIdx.x; t < H * H; t += NT) {
        const int h = t / H, c = t % H;
        s[t] = transpose ? Mat[c * H + h] : Mat[t];
    }
}

template <int H, int NT = BLOCK>
struct Geo {
    static constexpr int L = H / 4, PPW = WAVE / L, PPB = PPW * (NT / WAVE);
};

template <int H, int NT = BLOCK>
__device__ __forceinline__ int64_t my_point(int64_t m, int& lane, int& q, bool& valid) {
    lane = threadIdx.x & 63;
    q = lane % Geo<H>::L;
    const int64_t row = (int64_t)xcd_block_id() * Geo<H, NT>::PPB + (threadIdx.x >> 6) * Geo<H>::PPW + lane / Geo<H>::L;
    valid = row < m;
    return valid ? row : m - 1;
}

__device__ __forceinline__ float4 sub4(float4 a, float4 b) {
    return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
}

// K-wide row of 32-bit values (K % 4 == 0) as K/4 aligned dwordx4 loads.
template <int K, typename T4, typename T>
__device__ __forceinline__ void load_row(const T* __restrict__ p, T (&out)[K]) {
#pragma unroll
    for (int c = 0; c < K / 4; ++c) {
        const T4 v = reinterpret_cast<const T4*>(p)[c];
        out[4 * c + 0] = v.x; out[4 * c + 1] = v.y; out[4 * c + 2] = v.z; out[4 * c + 3] = v.w;
    }
}

// Index row in either layout: int32 global rows, or uint16 per-cloud local ids (half the bytes; valid when
// every cloud has <= 65536 source points) decoded as  cloud * n_src + id  with cloud = row / n_tgt.
// The layout is a TEMPLATE argument of the streaming kernels: as a run-time branch the two load paths are separate
// basic blocks, the compiler drains the memory counter at their join, and every later load of the kernel (weight row,
// own rows, matrices) is issued only after the index row has ARRIVED -- one extra memory round trip per launch.
template <int K, bool U16>
__device__ __forceinline__ void load_index_row_t(const int32_t* __restrict__ idx32, const uint16_t* __restrict__ idx16,
                                                 int64_t r, int n_tgt, int n_src, int (&j)[K]) {
    if constexpr (U16) {
        const int base = (int)((unsigned)r / (unsigned)n_tgt) * n_src;       // rows < 2^31 (check_common): a 32-bit division, not the
        const uint4* p = reinterpret_cast<const uint4*>(idx16 + r * K);       // ~130-instruction 64-bit one in front of the first load
#pragma unroll
        for (int c = 0; c < K / 8; ++c) {
            const uint4 v = p[c];
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                j[8 * c + 2 * e] = base + (int)(w[e] & 0xffffu);
                j[8 * c + 2 * e + 1] = base + (int)(w[e] >> 16);
            }
        }
    } else {
        load_row<K, int4>(idx32 + r * K, j);
    }
}
// The same row as BYTE OFFSETS of this lane's 16-byte piece of each neighbour row ([*, H] float table: row j starts at
// 4 H j, the lane's piece at + 16 q): what a buffer load through a resource descriptor takes as its 32-bit address operand.
// One VGPR and one 32-bit shift-add per neighbour, where a generic 64-bit pointer costs two VGPRs and a 64-bit multiply-add
// chain (the forward kernels issue 15-30 such gathers per point).  Needs rows * 4 H < 2^31 (the launcher checks).
template <int K, bool U16, int H>
__device__ __forceinline__ void load_index_offsets_t(const int32_t* __restrict__ idx32, const uint16_t* __restrict__ idx16,
                                                     int64_t r, int n_tgt, int n_src, int q, int (&off)[K]) {
    constexpr int RB = 4 * H;                                                  // bytes per row
    if constexpr (U16) {
        const int base = (int)((unsigned)r / (unsigned)n_tgt) * n_src * RB + 16 * q;
        const uint4* p = reinterpret_cast<const uint4*>(idx16 + r * K);
#pragma unroll
        for (int c = 0; c < K / 8; ++c) {
            const uint4 v = p[c];
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                off[8 * c + 2 * e] = base + (int)(w[e] & 0xffffu) * RB;
                off[8 * c + 2 * e + 1] = base + (int)(w[e] >> 16) * RB;
            }
        }
    } else {
        int j[K];
        load_row<K, int4>(idx32 + r * K, j);
#pragma unroll
        for (int k = 0; k < K; ++k) off[k] = j[k] * RB + 16 * q;
    }
}

template <int K>
__device__ __forceinline__ void load_index_row(const int32_t* __restrict__ idx32, const uint16_t* __restrict__ idx16,
                                               int64_t r, int n_tgt, int n_src, int (&j)[K]) {
    if (idx16 != nullptr) load_index_row_t<K, true>(idx32, idx16, r, n_tgt, n_src, j);
    else load_index_row_t<K, false>(idx32, idx16, r, n_tgt, n_src, j);
}

// An H x H matrix on its way into LDS in two halves: fetch() issues the global loads into registers (branch-free, no
// wait), park() writes them to LDS -- called late, behind the kernel's own row loads and gathers.  load_matrix() in
// one go is  load -> wait -> LDS write  at the top of a kernel: a whole memory round trip in front of everything else.
template <int H, int NT = BLOCK>
struct MatStage {
    static constexpr int N = (H * H + NT - 1) / NT;
    static constexpr int F4 = N * NT / 4;         // float4 slots of the LDS array park() writes (>= H * H / 4: EVERY thread stores, see park)
    float v[N];
    __device__ __forceinline__ void fetch(const float* __restrict__ Mat, bool transpose) {
#pragma unroll
        for (int u = 0; u < N; ++u) {
            const int t0 = (int)threadIdx.x + u * NT, t = t0 < H * H ? t0 : 0;
            v[u] = transpose ? Mat[(t % H) * H + t / H] : Mat[t];           // sM[h][c] = transpose ? Mat[c][h] : Mat[h][c]
        }
    }
    // unconditional stores (threads past H * H write into the array's padding): under a condition the compiler sinks the
    // global loads of fetch() into the conditional block, behind the gathers, and drains the memory counter there
    __device__ __forceinline__ void park(float4* sM) const {
        float* s = reinterpret_cast<float*>(sM);
#pragma unroll
        for (int u = 0; u < N; ++u) s[(int)threadIdx.x + u * NT] = v[u];
    }
};

// K-wide rows of the wave's PPW points (PPW * K contiguous floats in memory) written as 1 KiB-contiguous stores.
// Stored straight from the owning lanes, each store instruction touches 64/L rows with 16 bytes each (measured on
// the level-0 first kernel: +2.1 us); routed through a per-wave LDS tile every instruction writes consecutive bytes.
// One call per kernel (the tile is a single static array per instantiation).
template <int H, int K, int NT = BLOCK>
__device__ __forceinline__ void store_rows_coalesced(const float (&d)[K], float* __restrict__ dst, int lane, int q,
                                                     int64_t m) {
    constexpr int L = Geo<H>::L, PPW = Geo<H>::PPW, CPR = K / 4, NCH = PPW * CPR;   // 16-byte chunks per row / wave
    __shared__ float4 tile[NT / WAVE][NCH];
    float4* mine = tile[threadIdx.x >> 6];
    const int p = lane / L;
#pragma unroll
    for (int c = 0; c < CPR; ++c)
        if ((c % L) == q) mine[p * CPR + c] = make_float4(d[4 * c], d[4 * c + 1], d[4 * c + 2], d[4 * c + 3]);
    __builtin_amdgcn_wave_barrier();     // LDS operations of one wave complete in order
    const int64_t row0 = (int64_t)xcd_block_id() * Geo<H, NT>::PPB + (threadIdx.x >> 6) * PPW;
#pragma unroll
    for (int c = lane; c < NCH; c += WAVE)
        if (row0 + c / CPR < m) st4(dst + row0 * K + 4 * c, mine[c]);
}

// One source row per L lanes; walks the row's incoming edges (ascending edge id, fixed order).
// Gprev[j] = add[j] + sum_{e in rev(j)} s[e] gm[e / K]     (s is 0 on columns < k0)
// EP edge-lanes per row walk the row's incoming edges EP at a time (edge p = beg + lane-group, += EP) and fold their
// partial sums by xor-shuffles (fixed tree): in-degrees of a kNN graph spread from 0 to ~40, and with one lane
// group per row a wavefront iterates to the LARGEST in-degree of its 64/L rows.
#ifndef SCAT_EP
#define SCAT_EP 2          // edge lanes per row (L <= 4); swept with SCAT_UB on the level-0 backward: (4,4) 104.8, (2,8) 98.5, (2,4) 106, (4,8) 106, (1,16) 107 us
#endif
#ifndef SCAT_UB
#define SCAT_UB 8          // records / rows in flight per lane and round
#endif
template <int H>
struct Scat {
    static constexpr int L = H / 4, EP = (L <= 4) ? SCAT_EP : 1, RPW = WAVE / (L * EP), RPB = RPW * (BLOCK / WAVE);
};

template <int H>
__device__ __forceinline__ float4 fold_edge_lanes(float4 a) {
    constexpr int L = Scat<H>::L;
#pragma unroll
    for (int o = L; o < L * Scat<H>::EP; o <<= 1) {
        a.x += __shfl_xor(a.x, o, WAVE); a.y += __shfl_xor(a.y, o, WAVE);
        a.z += __shfl_xor(a.z, o, WAVE); a.w += __shfl_xor(a.w, o, WAVE);
    }
    return a;
}

// out[job][slot] = sum_b partial[job][b][slot] for two jobs of few slots (H * H <= 256) and MANY slabs (one per
// workgroup of the producing kernel): RS_CHUNKS workgroups each sum a contiguous range of slabs (eight loads in flight
// per lane), publish the chunk sums write-through, take a ticket; the last one adds the chunk sums in chunk order --
// one launch, bitwise reproducible.  ticket: one zero word, reset by the last workgroup.
constexpr int RS_CHUNKS = 128;
struct SmallJob {
    const float* partial;
    float* out;
    int nblk;
};

__device__ __forceinline__ void reduce_small_body(SmallJob j0, SmallJob j1, int nslots, float* scratch,
                                                  unsigned* ticket, unsigned bid, unsigned nblk) {
    __shared__ float s_part[256];
    __shared__ int s_last;
    const int groups = 256 / nslots, grp = threadIdx.x / nslots, slot = threadIdx.x % nslots;   // nslots in {64, 256}
    const __amdgpu_buffer_rsrc_t sr = make_rsrc(scratch, 2 * RS_CHUNKS * nslots * 4);
    for (int job = 0; job < 2; ++job) {
        const SmallJob jb = job == 0 ? j0 : j1;
        const int per = (jb.nblk + RS_CHUNKS - 1) / RS_CHUNKS;
        const int lo = bid * per, hi = lo + per < jb.nblk ? lo + per : jb.nblk;
        float acc = 0.f;
        for (int b = lo + grp; b < hi; b += 8 * groups) {        // eight slabs in flight per lane, added in slab order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = b + u * groups < hi ? jb.partial[(size_t)(b + u * groups) * nslots + slot] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        __syncthreads();
        s_part[threadIdx.x] = acc;
        __syncthreads();
        if (grp == 0) {
            float v = s_part[slot];
            for (int g2 = 1; g2 < groups; ++g2) v += s_part[g2 * nslots + slot];
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), sr, ((job * RS_CHUNKS + (int)bid) * nslots + slot) * 4, 0, 16);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old + 1 == nblk;
        if (s_last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    // the last workgroup: chunk sums in chunk order, `groups` interleaved sub-sums of 16 loads in flight each
    for (int job = 0; job < 2; ++job) {
        float* out = job == 0 ? j0.out : j1.out;
        float acc = 0.f;
        for (int c = grp; c < RS_CHUNKS; c += 16 * groups) {
            float u[16];
#pragma unroll
            for (int k = 0; k < 16; ++k)
                u[k] = c + k * groups < RS_CHUNKS
                           ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(sr, ((job * RS_CHUNKS + c + k * groups) * nslots + slot) * 4, 0, 16))
                           : 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) acc += u[k];
        }
        __syncthreads();
        s_part[threadIdx.x] = acc;
        __syncthreads();
        if (grp == 0) {
            float v = s_part[slot];
            for (int g2 = 1; g2 < groups; ++g2) v += s_part[g2 * nslots + slot];
            out[slot] = v;
        }
    }
}

inline int check_common(int64_t m, int H, int K, int k0) {
    CRF_REQUIRE(m > 0 && m < (int64_t)1 << 31, CRF_ERR_ARG, "rows m=%lld out of range", (long long)m);
    CRF_REQUIRE(H == 4 || H == 8 || H == 16 || H == 32 || H == 64, CRF_ERR_UNSUPPORTED,
                "hidden channels H=%d not in {4,8,16,32,64}", H);
    CRF_REQUIRE(K >= 1 && K <= 64 && k0 >= 0 && k0 < K, CRF_ERR_ARG, "K=%d k0=%d invalid", K, k0);
    CRF_REQUIRE(m * K < (int64_t)1 << 31, CRF_ERR_ARG, "edge ids exceed int32 (m=%lld K=%d)", (long long)m, K);
    return CRF_OK;
}

inline int kshift_of(int K) {
    for (int sft = 0; sft < 7; ++sft)
        if ((1 << sft) == K) return sft;
    return -1;
}

#define DISPATCH_H(H, ...)                                      \
    switch (H) {                                                \
        case 4: { constexpr int HH = 4; __VA_ARGS__; break; }   \
        case 8: { constexpr int HH = 8; __VA_ARGS__; break; }   \
        case 16: { constexpr int HH = 16; __VA_ARGS__; break; } \
        case 32: { constexpr int HH = 32; __VA_ARGS__; break; } \
        default: { constexpr int HH = 64; __VA_ARGS__; break; } \
    }

}  // namespace crf
