// Device-wide STABLE radix sort of (uint64 key, uint32 value) pairs -- this library's own, for the two places that sorted
// through rocPRIM until round 3: the voxel-key sort of the grid subsampling (grid_subsample.hip, utils/cpp_wrappers:
// grid_subsampling.cpp:53-56 keys) and the distance sort of the possibility crop (evaluate.hip,
// datasets/semantic3d_dataset.py:433 KDTree.query order).  No scratch (private-segment) memory, so -- unlike rocPRIM's
// onesweep kernel on ROCm 7.2 -- its launches can sit in a captured hipGraph.
//
// LSD, 8-bit digits, three launches per pass over CHUNKS of RSORT_CHUNK consecutive items, one wavefront per chunk:
//   histogram   counts[digit][chunk]                      (LDS atomics per wavefront, one coalesced read of the keys)
//   scan        exclusive scan of counts in digit-major order (scan.hpp) = first output slot of every (digit, chunk)
//   scatter     the wavefront re-reads its chunk IN ORDER, 64 items a round; lanes with equal digits find each other with
//               eight ballots (one per digit bit: mask = AND_b (bit_b ? ballot_b : ~ballot_b)), rank = popcount of the lower
//               lanes of the mask, and a 256-entry running count per wavefront (LDS) carries the rounds -- stable by
//               construction, no sorting network, no atomics on the output side.
// HBM bytes per pass: 2 x 8 (keys: histogram + scatter read) + 4 (values) + 12 (pairs written) = 32 B per item.
#pragma once
#include "common.hpp"
#include "scan.hpp"

namespace crf {

constexpr int RSORT_CHUNK = 1024, RSORT_WPB = 4;          // items per wavefront chunk; wavefronts per workgroup

template <int UNUSED = 0>      // (a template: the header is included by two translation units)
__global__ __launch_bounds__(64 * RSORT_WPB) void rsort_hist_kernel(const unsigned long long* __restrict__ keys, int64_t n,
                                                                    int shift, int64_t nchunks, int32_t* __restrict__ counts) {
    __shared__ int s_h[RSORT_WPB][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t chunk = (int64_t)blockIdx.x * RSORT_WPB + wave;
    int* h = s_h[wave];
#pragma unroll
    for (int u = 0; u < 4; ++u) h[4 * lane + u] = 0;
    __builtin_amdgcn_wave_barrier();
    if (chunk < nchunks) {
        const int64_t lo = chunk * RSORT_CHUNK;
#pragma unroll 4
        for (int r = 0; r < RSORT_CHUNK / 64; ++r) {
            const int64_t i = lo + 64 * r + lane;
            if (i < n) atomicAdd(&h[(int)((keys[i] >> shift) & 255ull)], 1);
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (chunk < nchunks) {
#pragma unroll
        for (int u = 0; u < 4; ++u) counts[(int64_t)(64 * u + lane) * nchunks + chunk] = h[64 * u + lane];
    }
}

template <int UNUSED = 0>
__global__ __launch_bounds__(64 * RSORT_WPB) void rsort_scatter_kernel(const unsigned long long* __restrict__ keys,
                                                                       const uint32_t* __restrict__ vals, int64_t n, int shift,
                                                                       int64_t nchunks, const int32_t* __restrict__ bases,
                                                                       unsigned long long* __restrict__ keys_out,
                                                                       uint32_t* __restrict__ vals_out) {
    __shared__ int s_run[RSORT_WPB][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t chunk = (int64_t)blockIdx.x * RSORT_WPB + wave;
    if (chunk >= nchunks) return;                          // whole wavefronts leave: the ballots below see full waves only
    int* run = s_run[wave];
#pragma unroll
    for (int u = 0; u < 4; ++u) run[64 * u + lane] = bases[(int64_t)(64 * u + lane) * nchunks + chunk];
    __builtin_amdgcn_wave_barrier();
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int64_t lo = chunk * RSORT_CHUNK;
    for (int r = 0; r < RSORT_CHUNK / 64; ++r) {
        const int64_t i = lo + 64 * r + lane;
        const bool valid = i < n;
        if (__ballot(valid) == 0ull) break;                // uniform
        const unsigned long long k = valid ? keys[i] : 0ull;
        const uint32_t v = valid ? vals[i] : 0u;
        const unsigned d = (unsigned)((k >> shift) & 255ull);
        unsigned long long mask = __ballot(valid);         // lanes past the end match nobody
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bal = __ballot((d >> b) & 1u);
            mask &= ((d >> b) & 1u) ? bal : ~bal;
        }
        const int base = run[d];                           // (every lane of a group reads the same word)
        const int rank = __popcll(mask & lt);
        __builtin_amdgcn_wave_barrier();                   // LDS operations of a wavefront complete in order: reads, then the update
        if (valid && rank == 0) run[d] = base + __popcll(mask);
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            keys_out[base + rank] = k;
            vals_out[base + rank] = v;
        }
    }
}

inline int64_t rsort_chunks(int64_t n) { return cdiv(n, (int64_t)RSORT_CHUNK); }
// scratch: counts [256 nchunks] + bases [256 nchunks] + scan block sums
inline size_t rsort_workspace(int64_t n) {
    const int64_t cells = 256 * rsort_chunks(n);
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    return 2 * up(sizeof(int32_t) * (size_t)cells) + up(sizeof(int32_t) * scan_block_sums(cells)) + 256;
}
// Sorts n pairs by bits [begin_bit, end_bit) of the key (both multiples of 8), ascending, stable.  Ping-pongs between the
// (a) and (b) buffers; returns 0 when the sorted pairs end in (keys_a, vals_a), 1 when in (keys_b, vals_b).  n < 2^31.
inline int rsort_pairs_u64(unsigned long long* keys_a, uint32_t* vals_a, unsigned long long* keys_b, uint32_t* vals_b, int64_t n,
                           int begin_bit, int end_bit, void* workspace, hipStream_t st) {
    const int64_t nchunks = rsort_chunks(n), cells = 256 * nchunks;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    char* ws = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    int32_t* counts = reinterpret_cast<int32_t*>(ws);
    int32_t* bases = reinterpret_cast<int32_t*>(ws + up(sizeof(int32_t) * (size_t)cells));
    int32_t* sums = reinterpret_cast<int32_t*>(ws + 2 * up(sizeof(int32_t) * (size_t)cells));
    const dim3 grid((unsigned)cdiv(nchunks, (int64_t)RSORT_WPB)), blk(64 * RSORT_WPB);
    int where = 0;
    for (int shift = begin_bit; shift < end_bit; shift += 8) {
        unsigned long long* ki = where ? keys_b : keys_a;
        unsigned long long* ko = where ? keys_a : keys_b;
        uint32_t* vi = where ? vals_b : vals_a;
        uint32_t* vo = where ? vals_a : vals_b;
        hipLaunchKernelGGL(rsort_hist_kernel<0>, grid, blk, 0, st, ki, n, shift, nchunks, counts);
        exclusive_scan_i32(counts, bases, cells, sums, st);
        hipLaunchKernelGGL(rsort_scatter_kernel<0>, grid, blk, 0, st, ki, vi, n, shift, nchunks, bases, ko, vo);
        where ^= 1;
    }
    return where;
}

}  // namespace crf
