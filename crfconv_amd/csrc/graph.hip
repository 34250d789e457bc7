// Neighbour-table plumbing: int64 per-cloud tables -> int32 global rows, and the reverse
// (source-major) CSR every backward scatter walks.  Integer work, HBM-bound; counting sort + scan.hpp's exclusive scan
// (no library sort, no scratch memory), flat coalesced kernels.
#include "common.hpp"

#include "scan.hpp"

namespace crf {

__global__ __launch_bounds__(256) void narrow_kernel(const int64_t* __restrict__ idx64,
                                                     int64_t total, int64_t per_cloud,
                                                     int64_t n_src, int32_t* __restrict__ idx32,
                                                     uint16_t* __restrict__ idx16,
                                                     int32_t* __restrict__ bad) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    int64_t v = idx64[t];
    const int64_t b = t / per_cloud;
    if (v < 0 || v >= n_src) {
        atomicAdd(bad, 1);
        v = v < 0 ? 0 : n_src - 1;
    }
    idx32[t] = (int32_t)(b * n_src + v);
    if (idx16 != nullptr) idx16[t] = (uint16_t)v;      // per-cloud local id (n_src <= 65536)
}

// Same narrowing, one thread per ROW, with columns sort_from .. K-1 of the row re-ordered by ascending source id.
// Every consumer of a table reduces over its columns (sums, maxima), so the column order is free; ascending ids
// make the k-th gather of adjacent (spatially sorted) target rows land on adjacent source rows -- same or
// neighbouring cache lines (measured on the level-0 mean-field kernels: -4..-7 %).  Column 0 (the query itself in
// a self-query kNN table, which the CRF layer drops by POSITION, continuous_crf_conv_big.py:45-47) stays put.
template <int KT>
__device__ __forceinline__ void narrow_sorted_row(const int64_t* __restrict__ idx64, int64_t row, int64_t rows_per_cloud, int K,
                                                  int sort_from, int64_t n_src, int32_t* __restrict__ idx32,
                                                  uint16_t* __restrict__ idx16, int32_t* __restrict__ bad) {
    const int64_t b = row / rows_per_cloud;
    int v[KT];
    int nbad = 0;
    const bool full = K == KT;                          // the usual case (K = 16 / 32 / 64): whole-row 16-byte accesses
    if (full) {
        const longlong2* __restrict__ src = reinterpret_cast<const longlong2*>(idx64 + row * KT);      // KT * 8 bytes: 16-byte aligned
#pragma unroll
        for (int k = 0; k < KT; k += 2) {
            const longlong2 x2 = src[k / 2];
            int64_t x = x2.x, y = x2.y;
            if (x < 0 || x >= n_src) { ++nbad; x = x < 0 ? 0 : n_src - 1; }
            if (y < 0 || y >= n_src) { ++nbad; y = y < 0 ? 0 : n_src - 1; }
            v[k] = (int)x;
            v[k + 1] = (int)y;
        }
    } else {
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            int64_t x = k < K ? idx64[row * K + k] : n_src;        // padding sorts behind every real id
            if (k < K && (x < 0 || x >= n_src)) { ++nbad; x = x < 0 ? 0 : n_src - 1; }
            v[k] = (int)x;
        }
    }
    if (nbad) atomicAdd(bad, nbad);
    // odd-even transposition network on the registers (KT compile-time: no scratch)
#pragma unroll
    for (int pass = 0; pass < KT; ++pass) {
#pragma unroll
        for (int k = (pass & 1); k + 1 < KT; k += 2) {
            const bool in = k >= sort_from;
            const int lo = min(v[k], v[k + 1]), hi = max(v[k], v[k + 1]);
            v[k] = in ? lo : v[k];
            v[k + 1] = in ? hi : v[k + 1];
        }
    }
    if (full) {
        const int base = (int)(b * n_src);
        int4* __restrict__ d32 = reinterpret_cast<int4*>(idx32 + row * KT);
#pragma unroll
        for (int k = 0; k < KT; k += 4) d32[k / 4] = make_int4(base + v[k], base + v[k + 1], base + v[k + 2], base + v[k + 3]);
        if (idx16 != nullptr) {
            uint4* __restrict__ d16 = reinterpret_cast<uint4*>(idx16 + row * KT);
#pragma unroll
            for (int k = 0; k < KT; k += 8)
                d16[k / 8] = make_uint4((unsigned)v[k] | ((unsigned)v[k + 1] << 16), (unsigned)v[k + 2] | ((unsigned)v[k + 3] << 16),
                                        (unsigned)v[k + 4] | ((unsigned)v[k + 5] << 16), (unsigned)v[k + 6] | ((unsigned)v[k + 7] << 16));
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < KT; ++k)
        if (k < K) {
            idx32[row * K + k] = (int32_t)(b * n_src + v[k]);
            if (idx16 != nullptr) idx16[row * K + k] = (uint16_t)v[k];
        }
}
template <int KT>
__global__ __launch_bounds__(256) void narrow_sorted_kernel(const int64_t* __restrict__ idx64, int64_t rows,
                                                            int64_t rows_per_cloud, int K, int sort_from,
                                                            int64_t n_src, int32_t* __restrict__ idx32,
                                                            uint16_t* __restrict__ idx16, int32_t* __restrict__ bad) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    narrow_sorted_row<KT>(idx64, row, rows_per_cloud, K, sort_from, n_src, idx32, idx16, bad);
}

// The same for SEVERAL tables in one launch (a batch refresh narrows every table of the batch: 13 for PointConvBig's five
// levels, most of them a few thousand rows = one launch of ~5 us each).  A workgroup's 256 rows belong to one table, found by
// a binary search over the chunk prefix; tables that keep their column order (K = 1 up-sampling tables, sort_from >= K - 1)
// run the same body with nothing to exchange -- identical idx32 / idx16 / bad counts to the one-table entry points.
constexpr int NB_MAX = 32;
struct NarrowBatch {
    const int64_t* idx64[NB_MAX];
    int32_t* idx32[NB_MAX];
    uint16_t* idx16[NB_MAX];
    int32_t* bad[NB_MAX];
    int rows[NB_MAX], rows_per_cloud[NB_MAX], K[NB_MAX], sort_from[NB_MAX], n_src[NB_MAX];
    int chunk_base[NB_MAX + 1];
    int njobs;
};
template <int KT>
__global__ __launch_bounds__(256) void narrow_batched_kernel(const NarrowBatch t) {
    int lo = 0, hi = t.njobs;                          // largest j with chunk_base[j] <= blockIdx.x
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (t.chunk_base[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
    }
    const int row = ((int)blockIdx.x - t.chunk_base[lo]) * 256 + (int)threadIdx.x;
    if (row >= t.rows[lo]) return;
    narrow_sorted_row<KT>(t.idx64[lo], row, t.rows_per_cloud[lo], t.K[lo], t.sort_from[lo], t.n_src[lo], t.idx32[lo], t.idx16[lo],
                          t.bad[lo]);
}

// edge ids 0..E-1 and sort keys: the source row, or m_src for "no neighbour" entries (< 0), which
// therefore sort behind every real row and fall outside rev_ptr[0 .. m_src].
static size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

// ------------------------------------------------------------------ reverse CSR by counting sort (no library sort)
// keys (source rows, m_src = "missing") are bounded and a kNN-style graph has ~K edges per row: histogram -> exclusive
// scan -> fill (cursor atomics: arbitrary order inside a row) -> per-row rank sort of the edge ids, which restores the
// ascending (= stable) order every consumer's fixed summation order relies on.  Seven small launches, none of which needs
// scratch memory: rocPRIM's onesweep radix sort (eleven launches per table, private segment of 80 B) FAULTED when it was
// replayed from a captured hipGraph with eager launches in between (ROCm 7.2; data.CollateGraph), and was the largest
// single cost of the per-batch table refresh.
__global__ __launch_bounds__(256) void rev_count_kernel(const int32_t* __restrict__ idx, int64_t E, int64_t m_src,
                                                        int32_t* __restrict__ cnt) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const int32_t v = idx[e];
    atomicAdd(&cnt[(v < 0 || v >= m_src) ? m_src : v], 1);
}

__global__ __launch_bounds__(256) void rev_fill_kernel(const int32_t* __restrict__ idx, int64_t E, int64_t m_src,
                                                       const int32_t* __restrict__ ptrs, int32_t* __restrict__ cursor,
                                                       int32_t* __restrict__ tmp_eid) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const int32_t v = idx[e];
    if (v < 0 || v >= m_src) return;
    tmp_eid[ptrs[v] + atomicAdd(&cursor[v], 1)] = (int32_t)e;
}

// Hub rows (more than HUB_LEN entries: duplicate / degenerate clouds, where thousands of targets name one source): the all-pairs
// ranking is quadratic (a 40 960-entry row = 26 M tile comparisons on one wavefront, ~0.1 s), so such a row is sorted by a
// wavefront-wide LSD radix sort instead -- 8-bit digits, ceil(key_bits / 8) stable passes ping-ponging between the slot array
// `a` (where the fill left the ids) and the output row `b`, linear in the row length.  `hist`: 256 LDS ints of this wavefront.
constexpr int HUB_LEN = 512;
__device__ __forceinline__ void hub_sort_row(int32_t* __restrict__ a, int32_t* __restrict__ b, int len, int key_bits,
                                             int* __restrict__ hist, int lane) {
    const int passes = key_bits <= 8 ? 1 : (key_bits <= 16 ? 2 : (key_bits <= 24 ? 3 : 4));
    const unsigned long long lt = (1ull << lane) - 1ull;
    int32_t* src = a;
    int32_t* dst = b;
    for (int pass = 0; pass < passes; ++pass) {
        const int shift = 8 * pass;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < 4; ++u) hist[4 * lane + u] = 0;
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < len; i += WAVE) atomicAdd(&hist[((unsigned)src[i] >> shift) & 255u], 1);
        __builtin_amdgcn_wave_barrier();
        {   // exclusive scan of the 256 bins: four consecutive bins per lane + a wavefront scan of the lane totals
            const int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
            const int tot = h0 + h1 + h2 + h3;
            int incl = tot;
#pragma unroll
            for (int d = 1; d < WAVE; d <<= 1) {
                const int up = __shfl_up(incl, d, WAVE);
                if (lane >= d) incl += up;
            }
            const int ex = incl - tot;
            __builtin_amdgcn_wave_barrier();
            hist[4 * lane] = ex; hist[4 * lane + 1] = ex + h0; hist[4 * lane + 2] = ex + h0 + h1; hist[4 * lane + 3] = ex + h0 + h1 + h2;
        }
        __builtin_amdgcn_wave_barrier();
        for (int i0 = 0; i0 < len; i0 += WAVE) {              // in order: the pass is stable
            const int i = i0 + lane;
            const bool valid = i < len;
            const int32_t k = valid ? src[i] : 0;
            const int d = valid ? (int)(((unsigned)k >> shift) & 255u) : 256;
            unsigned long long todo = __ballot(valid);
            int slot = 0;
            while (todo != 0ull) {                            // one round per distinct digit of the 64 keys
                const int leader = __ffsll((long long)todo) - 1;
                const int ld = __shfl(d, leader, WAVE);
                const unsigned long long m = __ballot(valid && d == ld);
                const int base = hist[ld];                    // every lane reads (one address), the group's lanes use it
                if (valid && d == ld) slot = base + __popcll(m & lt);
                __builtin_amdgcn_wave_barrier();              // LDS operations of a wavefront complete in order: read, then update
                if (lane == leader) hist[ld] = base + __popcll(m);
                __builtin_amdgcn_wave_barrier();
                todo &= ~m;
            }
            if (valid) dst[slot] = k;
        }
        __threadfence();                                      // this wavefront re-reads what it wrote: past L1
        int32_t* t = src; src = dst; dst = t;
    }
    if (src != b) {                                           // an even number of passes ended in `a`
        for (int i = lane; i < len; i += WAVE) b[i] = src[i];
    }
}

// one wavefront per source row: rank of every edge id among the row's ids (ids are distinct) = its sorted position
__global__ __launch_bounds__(256) void rev_sort_rows_kernel(const int32_t* __restrict__ ptrs, int32_t* __restrict__ tmp_eid,
                                                            int64_t m_src, int key_bits, int32_t* __restrict__ rev_eid) {
    __shared__ int s_hist[4][256];
    const int lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= m_src) return;
    const int beg = ptrs[j], len = ptrs[j + 1] - beg;
    if (len > HUB_LEN) {
        hub_sort_row(tmp_eid + beg, rev_eid + beg, len, key_bits, s_hist[threadIdx.x >> 6], lane);
    } else if (len <= 64) {
        const int32_t v = lane < len ? tmp_eid[beg + lane] : 0x7fffffff;
        int rank = 0;
        for (int k = 0; k < len; ++k) rank += __shfl(v, k, WAVE) < v ? 1 : 0;
        if (lane < len) rev_eid[beg + rank] = v;
    } else {
        // long rows (hubs: up_idx tables when many fine points share a coarse one, degenerate clouds): 64 ids of the row per
        // lane-chunk, ranked against the row in tiles of 64 -- one coalesced load per tile and 64 lane broadcasts, no
        // dependent global load per comparison (a row of 40960 ids: ~0.1 s instead of ~1e9 dependent loads)
        for (int i0 = 0; i0 < len; i0 += 64) {
            const int i = i0 + lane;
            const int32_t v = i < len ? tmp_eid[beg + i] : 0x7fffffff;
            int rank = 0;
            for (int t0 = 0; t0 < len; t0 += 64) {
                const int32_t u = t0 + lane < len ? tmp_eid[beg + t0 + lane] : 0x7fffffff;
                const int nt = len - t0 < 64 ? len - t0 : 64;
                for (int k = 0; k < nt; ++k) rank += __shfl(u, k, WAVE) < v ? 1 : 0;
            }
            if (i < len) rev_eid[beg + rank] = v;
        }
    }
}

// ------------------------------------------------------------------ the same for SEVERAL tables in one set of launches
// A batch refresh rebuilds the reverse CSR of every table of the batch (14 for PointConvBig's five levels): as separate
// builds that is ~100 launches of a few microseconds each.  Batched: the tables' row ranges (m_src + 1 counters each) and
// edge ranges are laid end to end -- ONE histogram, ONE scan, ONE fill, ONE rank pass; rows and edges find their table by
// a binary search over <= RB_MAX prefix entries.
constexpr int RB_MAX = 32, RB_EPB = 1024;           // edges per workgroup of the count / fill passes
struct RevBatch {
    const int32_t* idx[RB_MAX];
    int32_t* rev_ptr[RB_MAX];
    int32_t* rev_eid[RB_MAX];
    int m_src[RB_MAX];
    int row_base[RB_MAX + 1];                      // prefix of (m_src + 1)
    int edge_base[RB_MAX + 1];                     // prefix of E
    int chunk_base[RB_MAX + 1];                    // prefix of ceil(E / RB_EPB)
    int rowchunk_base[RB_MAX + 1];                 // prefix of ceil((m_src + 1) / RB_RR): wavefronts of the rank pass
    int njobs;
};
__device__ __forceinline__ int rb_find(const int* prefix, int n, int v) {      // largest j with prefix[j] <= v
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (prefix[mid] <= v) lo = mid; else hi = mid;
    }
    return lo;
}
// Count and fill work through an LDS WINDOW of RB_WIN source rows around the workgroup's own place in the table: points are
// Morton-ordered, so the 1024 edges of a workgroup (64 targets x K) name a few hundred distinct, nearby sources -- the
// per-edge atomics are LDS atomics, and only one global atomic per DISTINCT source of the window leaves the CU (the batch's
// 5.4 M device-scope atomics were 74 us (count) + 99 us (fill) of the 1.49 ms collate graph).  Sources outside the window
// (cloud boundaries, Morton seams) take the direct global atomic.
constexpr int RB_WIN = 2048;
__device__ __forceinline__ int rb_window_base(const int32_t* __restrict__ idx, int e0, int E, int m_src) {
    int mid = e0 + RB_EPB / 2;
    if (mid >= E) mid = E - 1;
    int v = idx[mid];                                  // some neighbour of the workgroup's middle target (uniform load)
    if (v < 0 || v >= m_src) v = m_src;
    const int base = v - RB_WIN / 2;
    return base < 0 ? 0 : base;
}
__global__ __launch_bounds__(256) void revb_count_kernel(const RevBatch t, int32_t* __restrict__ cnt) {
    __shared__ int s_cnt[RB_WIN];
    const int j = rb_find(t.chunk_base, t.njobs, (int)blockIdx.x);
    const int E = t.edge_base[j + 1] - t.edge_base[j], m_src = t.m_src[j];
    const int e0 = ((int)blockIdx.x - t.chunk_base[j]) * RB_EPB;
    const int32_t* __restrict__ idx = t.idx[j];
    for (int w = threadIdx.x; w < RB_WIN; w += 256) s_cnt[w] = 0;
    const int base = rb_window_base(idx, e0, E, m_src);
    int32_t* __restrict__ c = cnt + t.row_base[j];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < RB_EPB / 256; ++u) {
        const int e = e0 + u * 256 + (int)threadIdx.x;
        if (e < E) {
            int v = idx[e];
            if (v < 0 || v >= m_src) v = m_src;
            const int w = v - base;
            if (w >= 0 && w < RB_WIN) atomicAdd(&s_cnt[w], 1);
            else atomicAdd(&c[v], 1);
        }
    }
    __syncthreads();
    for (int w = threadIdx.x; w < RB_WIN; w += 256) {
        const int n = s_cnt[w];
        if (n > 0) atomicAdd(&c[base + w], n);           // base + w <= m_src by construction of the window test
    }
}
__global__ __launch_bounds__(256) void revb_fill_kernel(const RevBatch t, const int32_t* __restrict__ ptrs,
                                                        int32_t* __restrict__ cursor, int32_t* __restrict__ tmp) {
    __shared__ int s_cnt[RB_WIN];                       // edges of this workgroup per window row, then its running local cursor
    __shared__ int s_slot[RB_WIN];                      // first global slot of the workgroup's run in that row
    const int j = rb_find(t.chunk_base, t.njobs, (int)blockIdx.x);
    const int E = t.edge_base[j + 1] - t.edge_base[j], m_src = t.m_src[j];
    const int e0 = ((int)blockIdx.x - t.chunk_base[j]) * RB_EPB;
    const int32_t* __restrict__ idx = t.idx[j];
    for (int w = threadIdx.x; w < RB_WIN; w += 256) s_cnt[w] = 0;
    const int base = rb_window_base(idx, e0, E, m_src);
    const int rb = t.row_base[j];
    __syncthreads();
    int vv[RB_EPB / 256];
#pragma unroll
    for (int u = 0; u < RB_EPB / 256; ++u) {
        const int e = e0 + u * 256 + (int)threadIdx.x;
        int v = -1;
        if (e < E) {
            v = idx[e];
            if (v < 0 || v >= m_src) v = -1;            // no source: not part of any reverse row
        }
        vv[u] = v;
        const int w = v - base;
        if (v >= 0 && w >= 0 && w < RB_WIN) atomicAdd(&s_cnt[w], 1);
    }
    __syncthreads();
    for (int w = threadIdx.x; w < RB_WIN; w += 256) {
        const int n = s_cnt[w];
        if (n > 0) {
            const int r = rb + base + w;
            s_slot[w] = ptrs[r] + atomicAdd(&cursor[r], n);      // ONE returning global atomic per distinct source
            s_cnt[w] = 0;
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < RB_EPB / 256; ++u) {
        const int v = vv[u];
        if (v < 0) continue;
        const int e = e0 + u * 256 + (int)threadIdx.x;
        const int w = v - base;
        if (w >= 0 && w < RB_WIN) tmp[s_slot[w] + atomicAdd(&s_cnt[w], 1)] = e;     // global slot, LOCAL edge id
        else tmp[ptrs[rb + v] + atomicAdd(&cursor[rb + v], 1)] = e;
    }
}
// Rank pass: a wavefront takes RB_RR consecutive rows of one table = ONE contiguous run of the slot array (about 64
// entries at K = 16): one coalesced load into its LDS tile, then every lane ranks its entry inside its own row (the number
// of smaller edge ids in the row's segment of the tile, LDS broadcast reads).  Two dependent memory round trips per
// wavefront for four rows -- one wavefront PER ROW measured 181 us for the batch's 0.5 M rows, all of it latency.  Runs
// longer than the tile (hub rows) fall back to the tiled all-pairs form of rev_sort_rows_kernel, row by row.
constexpr int RB_RR = 4, RB_TILE = 256;
__global__ __launch_bounds__(256) void revb_rows_kernel(const RevBatch t, const int32_t* __restrict__ ptrs,
                                                        int32_t* __restrict__ tmp) {
    __shared__ int s_tile[4][RB_TILE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = (int)blockIdx.x * 4 + wave;
    if (gw >= t.rowchunk_base[t.njobs]) return;
    const int j = rb_find(t.rowchunk_base, t.njobs, gw);
    const int v0 = (gw - t.rowchunk_base[j]) * RB_RR;
    const int m_src = t.m_src[j], eb = t.edge_base[j];
    int nrows = m_src + 1 - v0;                          // rows of the concatenation this wavefront owns (incl. the bucket row)
    if (nrows > RB_RR) nrows = RB_RR;
    int nreal = m_src - v0;                              // ... of which real source rows
    if (nreal > RB_RR) nreal = RB_RR;
    const int r0 = t.row_base[j] + v0;
    const int p = lane <= nreal ? ptrs[r0 + lane] : 0;     // bounds of the real rows (ptrs[r0 + nreal]: the bucket row's, always there)
    if (lane < nrows) t.rev_ptr[j][v0 + lane] = p - eb;
    if (nreal <= 0) return;
    int bnd[RB_RR + 1];
#pragma unroll
    for (int q = 0; q <= RB_RR; ++q) bnd[q] = __shfl(p, q < nreal ? q : nreal, WAVE);
    const int beg = bnd[0], n = bnd[RB_RR] - bnd[0];
    int32_t* __restrict__ out = t.rev_eid[j] - eb;        // indexed by GLOBAL slot
    if (n <= RB_TILE) {
        int* tile = s_tile[wave];
        for (int i = lane; i < n; i += 64) tile[i] = tmp[beg + i];
        __builtin_amdgcn_wave_barrier();                 // LDS operations of one wavefront complete in order
        for (int i = lane; i < n; i += 64) {
            const int gslot = beg + i;
            int rb_ = bnd[0], re_ = bnd[1];
#pragma unroll
            for (int q = 1; q < RB_RR; ++q)
                if (gslot >= bnd[q]) { rb_ = bnd[q]; re_ = bnd[q + 1]; }
            const int x = tile[i];
            int rank = 0;
            for (int k = rb_ - beg; k < re_ - beg; ++k) rank += tile[k] < x ? 1 : 0;
            out[rb_ + rank] = x;
        }
    } else {
        int key_bits = 1;                                  // local edge ids are < E of this table
        while (key_bits < 31 && (1 << key_bits) < t.edge_base[j + 1] - eb) ++key_bits;
        for (int q = 0; q < nreal; ++q) {
            const int b = bnd[q], len = bnd[q + 1] - b;
            if (len > HUB_LEN) {                           // a hub: radix sort, linear in the row length
                static_assert(RB_TILE >= 256, "the tile doubles as the 256-bin histogram");
                hub_sort_row(tmp + b, out + b, len, key_bits, s_tile[wave], lane);
                continue;
            }
            for (int i0 = 0; i0 < len; i0 += 64) {
                const int i = i0 + lane;
                const int32_t x = i < len ? tmp[b + i] : 0x7fffffff;
                int rank = 0;
                for (int t0 = 0; t0 < len; t0 += 64) {
                    const int32_t u = t0 + lane < len ? tmp[b + t0 + lane] : 0x7fffffff;
                    const int nt = len - t0 < 64 ? len - t0 : 64;
                    for (int k = 0; k < nt; ++k) rank += __shfl(u, k, WAVE) < x ? 1 : 0;
                }
                if (i < len) out[b + rank] = x;
            }
        }
    }
}

__global__ __launch_bounds__(256) void zero_i32_kernel(int32_t* __restrict__ p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = 0;
}

// Any number of device-to-device copies in ONE launch (MultiScaleData.load_: the ~25 tensors of a freshly collated batch into
// the static buffers a captured training step reads).  A workgroup moves one 64 KB chunk; chunk_begin = prefix sum of the jobs'
// chunk counts.  16-byte accesses when both pointers are 16-byte aligned, bytes otherwise and for the tail.
constexpr int CJ_MAX = 96, CJ_CHUNK = 65536;
struct CopyJobTable {
    const char* src[CJ_MAX];
    char* dst[CJ_MAX];
    long long nbytes[CJ_MAX];
    int chunk_begin[CJ_MAX + 1];
    int njobs;
};
__global__ __launch_bounds__(256) void copy_jobs_kernel(const CopyJobTable t) {
    const int g = blockIdx.x;
    int lo = 0, hi = t.njobs;                          // largest j with chunk_begin[j] <= g
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (t.chunk_begin[mid] <= g) lo = mid; else hi = mid;
    }
    const long long off = (long long)(g - t.chunk_begin[lo]) * CJ_CHUNK;
    long long n = t.nbytes[lo] - off;
    if (n > CJ_CHUNK) n = CJ_CHUNK;
    const char* s = t.src[lo] + off;
    char* d = t.dst[lo] + off;
    long long done = 0;
    if ((((uintptr_t)s | (uintptr_t)d) & 15) == 0) {
        const long long n16 = n >> 4;
        for (long long i = threadIdx.x; i < n16; i += 256) reinterpret_cast<uint4*>(d)[i] = reinterpret_cast<const uint4*>(s)[i];
        done = n16 << 4;
    }
    for (long long i = done + threadIdx.x; i < n; i += 256) d[i] = s[i];
}

}  // namespace crf

using namespace crf;

extern "C" int crfconv_copy_jobs(const crf_copy_job* jobs, int njobs, crf_stream_t stream) {
    CRF_REQUIRE(jobs || njobs == 0, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(njobs >= 0, CRF_ERR_ARG, "njobs=%d < 0", njobs);
    hipStream_t st = as_stream(stream);
    for (int j0 = 0; j0 < njobs; j0 += CJ_MAX) {
        CopyJobTable t;
        const int n = njobs - j0 < CJ_MAX ? njobs - j0 : CJ_MAX;
        long long total = 0;
        for (int j = 0; j < CJ_MAX; ++j) {
            t.chunk_begin[j] = (int)total;
            if (j < n) {
                const crf_copy_job& jb = jobs[j0 + j];
                CRF_REQUIRE(jb.nbytes >= 0 && (jb.nbytes == 0 || (jb.src && jb.dst)), CRF_ERR_ARG, "job %d is malformed", j0 + j);
                t.src[j] = reinterpret_cast<const char*>(jb.src);
                t.dst[j] = reinterpret_cast<char*>(jb.dst);
                t.nbytes[j] = jb.nbytes;
                total += (jb.nbytes + CJ_CHUNK - 1) / CJ_CHUNK;
                CRF_REQUIRE(total < ((long long)1 << 30), CRF_ERR_ARG, "too many bytes in one batch");
            } else {
                t.src[j] = nullptr; t.dst[j] = nullptr; t.nbytes[j] = 0;
            }
        }
        t.chunk_begin[CJ_MAX] = (int)total;
        t.njobs = n;
        if (total == 0) continue;
        hipLaunchKernelGGL(copy_jobs_kernel, dim3((unsigned)total), dim3(256), 0, st, t);
        CRF_LAUNCH_CHECK();
    }
    return CRF_OK;
}

extern "C" int crfconv_index_narrow(const int64_t* idx64, int64_t B, int64_t n_tgt, int K,
                                    int64_t n_src, int32_t* idx32, uint16_t* idx16, int32_t* bad_count,
                                    crf_stream_t stream) {
    CRF_REQUIRE(idx64 && idx32 && bad_count, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(idx16 == nullptr || n_src <= 65536, CRF_ERR_ARG, "uint16 table needs n_src <= 65536");
    CRF_REQUIRE(B > 0 && n_tgt > 0 && K > 0 && n_src > 0, CRF_ERR_ARG, "empty table");
    CRF_REQUIRE(B * n_src < ((int64_t)1 << 31) && B * n_tgt * K < ((int64_t)1 << 31), CRF_ERR_UNSUPPORTED,
                "table too large for int32 rows / edge ids (B=%lld n_src=%lld n_tgt=%lld K=%d)",
                (long long)B, (long long)n_src, (long long)n_tgt, K);
    const int64_t total = B * n_tgt * K;
    hipLaunchKernelGGL(narrow_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, as_stream(stream),
                       idx64, total, n_tgt * K, n_src, idx32, idx16, bad_count);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_index_narrow_sorted(const int64_t* idx64, int64_t B, int64_t n_tgt, int K,
                                           int64_t n_src, int sort_from, int32_t* idx32, uint16_t* idx16,
                                           int32_t* bad_count, crf_stream_t stream) {
    CRF_REQUIRE(sort_from >= 0, CRF_ERR_ARG, "sort_from=%d < 0", sort_from);
    if (sort_from >= K - 1 || K > 64)       // nothing to re-order (or wider than the register network)
        return crfconv_index_narrow(idx64, B, n_tgt, K, n_src, idx32, idx16, bad_count, stream);
    CRF_REQUIRE(idx64 && idx32 && bad_count, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(idx16 == nullptr || n_src <= 65536, CRF_ERR_ARG, "uint16 table needs n_src <= 65536");
    CRF_REQUIRE(B > 0 && n_tgt > 0 && K > 0 && n_src > 0, CRF_ERR_ARG, "empty table");
    CRF_REQUIRE(B * n_src < ((int64_t)1 << 31) && B * n_tgt * K < ((int64_t)1 << 31), CRF_ERR_UNSUPPORTED,
                "table too large for int32 rows / edge ids (B=%lld n_src=%lld n_tgt=%lld K=%d)",
                (long long)B, (long long)n_src, (long long)n_tgt, K);
    const int64_t rows = B * n_tgt;
    const dim3 grid((unsigned)cdiv(rows, 256)), blk(256);
    hipStream_t st = as_stream(stream);
    if (K <= 16)
        hipLaunchKernelGGL(narrow_sorted_kernel<16>, grid, blk, 0, st, idx64, rows, n_tgt, K, sort_from, n_src, idx32, idx16, bad_count);
    else if (K <= 32)
        hipLaunchKernelGGL(narrow_sorted_kernel<32>, grid, blk, 0, st, idx64, rows, n_tgt, K, sort_from, n_src, idx32, idx16, bad_count);
    else
        hipLaunchKernelGGL(narrow_sorted_kernel<64>, grid, blk, 0, st, idx64, rows, n_tgt, K, sort_from, n_src, idx32, idx16, bad_count);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

// crfconv_index_narrow_sorted for up to 32 tables in ONE launch (jobs: host array).  K <= 64 for every job.
extern "C" int crfconv_index_narrow_batched(const crf_narrow_job* jobs, int njobs, crf_stream_t stream) {
    CRF_REQUIRE(jobs, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(njobs >= 1 && njobs <= NB_MAX, CRF_ERR_ARG, "njobs=%d outside [1, %d]", njobs, NB_MAX);
    NarrowBatch t;
    int64_t chunks = 0;
    int kmax = 1;
    for (int j = 0; j <= NB_MAX; ++j) {
        t.chunk_base[j] = (int)chunks;
        if (j < njobs) {
            const crf_narrow_job& jb = jobs[j];
            CRF_REQUIRE(jb.idx64 && jb.idx32 && jb.bad_count, CRF_ERR_ARG, "job %d: null pointer", j);
            CRF_REQUIRE(jb.idx16 == nullptr || jb.n_src <= 65536, CRF_ERR_ARG, "job %d: uint16 table needs n_src <= 65536", j);
            CRF_REQUIRE(jb.B > 0 && jb.n_tgt > 0 && jb.K > 0 && jb.K <= 64 && jb.n_src > 0 && jb.sort_from >= 0, CRF_ERR_ARG,
                        "job %d: bad shape (B=%lld n_tgt=%lld K=%d n_src=%lld sort_from=%d)", j, (long long)jb.B,
                        (long long)jb.n_tgt, jb.K, (long long)jb.n_src, jb.sort_from);
            CRF_REQUIRE(jb.B * jb.n_src < ((int64_t)1 << 31) && jb.B * jb.n_tgt * jb.K < ((int64_t)1 << 31), CRF_ERR_UNSUPPORTED,
                        "job %d: table too large for int32 rows / edge ids", j);
            t.idx64[j] = jb.idx64; t.idx32[j] = jb.idx32; t.idx16[j] = jb.idx16; t.bad[j] = jb.bad_count;
            t.rows[j] = (int)(jb.B * jb.n_tgt); t.rows_per_cloud[j] = (int)jb.n_tgt; t.K[j] = jb.K;
            t.sort_from[j] = jb.sort_from; t.n_src[j] = (int)jb.n_src;
            chunks += cdiv(jb.B * jb.n_tgt, 256);
            CRF_REQUIRE(chunks < ((int64_t)1 << 31), CRF_ERR_UNSUPPORTED, "batch too large");
            if (jb.K > kmax) kmax = jb.K;
        } else if (j < NB_MAX) {
            t.idx64[j] = nullptr; t.idx32[j] = nullptr; t.idx16[j] = nullptr; t.bad[j] = nullptr;
            t.rows[j] = 0; t.rows_per_cloud[j] = 1; t.K[j] = 1; t.sort_from[j] = 0; t.n_src[j] = 1;
        }
    }
    t.njobs = njobs;
    const dim3 grid((unsigned)chunks), blk(256);
    hipStream_t st = as_stream(stream);
    if (kmax <= 16) hipLaunchKernelGGL(narrow_batched_kernel<16>, grid, blk, 0, st, t);
    else if (kmax <= 32) hipLaunchKernelGGL(narrow_batched_kernel<32>, grid, blk, 0, st, t);
    else hipLaunchKernelGGL(narrow_batched_kernel<64>, grid, blk, 0, st, t);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" size_t crfconv_reverse_csr_workspace(int64_t E, int64_t m_src) {
    if (E <= 0 || m_src <= 0) return 0;
    // [cnt m_src + 1][cursor m_src + 1][block sums][tmp_eid E]
    const size_t nb = (size_t)cdiv(m_src + 1, SCAN_EPB);
    return 2 * align_up(sizeof(int32_t) * (size_t)(m_src + 1)) + align_up(sizeof(int32_t) * nb) +
           align_up(sizeof(int32_t) * (size_t)E) + 256;
}

extern "C" int crfconv_reverse_csr(const int32_t* idx32, int64_t E, int64_t m_src, int32_t* rev_ptr,
                                   int32_t* rev_eid, void* workspace, size_t workspace_bytes,
                                   crf_stream_t stream) {
    CRF_REQUIRE(idx32 && rev_ptr && rev_eid && workspace, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(E > 0 && m_src > 0 && E < ((int64_t)1 << 31) && m_src < ((int64_t)1 << 31), CRF_ERR_ARG,
                "E=%lld m_src=%lld out of range", (long long)E, (long long)m_src);
    const size_t need = crfconv_reverse_csr_workspace(E, m_src);
    CRF_REQUIRE(workspace_bytes >= need, CRF_ERR_WORKSPACE, "reverse_csr workspace %zu < %zu",
                workspace_bytes, need);
    hipStream_t st = as_stream(stream);
    char* ws = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const int64_t n = m_src + 1, nb = cdiv(n, SCAN_EPB);
    const size_t seg = align_up(sizeof(int32_t) * (size_t)n);
    int32_t* cnt = reinterpret_cast<int32_t*>(ws);
    int32_t* cursor = reinterpret_cast<int32_t*>(ws + seg);
    int32_t* sums = reinterpret_cast<int32_t*>(ws + 2 * seg);
    int32_t* tmp_eid = reinterpret_cast<int32_t*>(ws + 2 * seg + align_up(sizeof(int32_t) * (size_t)nb));
    const dim3 blk(256), egrid((unsigned)cdiv(E, 256));
    int64_t zg = cdiv((int64_t)(2 * seg / 4), 256);
    if (zg > 1024) zg = 1024;
    hipLaunchKernelGGL(zero_i32_kernel, dim3((unsigned)zg), blk, 0, st, cnt, (int64_t)(2 * seg / 4));          // cnt + cursor
    hipLaunchKernelGGL(rev_count_kernel, egrid, blk, 0, st, idx32, E, m_src, cnt);
    exclusive_scan_i32(cnt, rev_ptr, n, sums, st);
    hipLaunchKernelGGL(rev_fill_kernel, egrid, blk, 0, st, idx32, E, m_src, rev_ptr, cursor, tmp_eid);
    int key_bits = 1;                                     // edge ids are < E
    while (key_bits < 31 && ((int64_t)1 << key_bits) < E) ++key_bits;
    hipLaunchKernelGGL(rev_sort_rows_kernel, dim3((unsigned)cdiv(m_src, 4)), blk, 0, st, rev_ptr, tmp_eid, m_src, key_bits, rev_eid);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" size_t crfconv_reverse_csr_batched_workspace(const crf_rev_job* jobs, int njobs) {
    if (!jobs || njobs < 1) return 0;
    int64_t rows = 0, edges = 0;
    for (int j = 0; j < njobs; ++j) { rows += jobs[j].m_src + 1; edges += jobs[j].E; }
    return 3 * align_up(sizeof(int32_t) * (size_t)(rows + 1)) + align_up(sizeof(int32_t) * scan_block_sums(rows + 1)) +
           align_up(sizeof(int32_t) * (size_t)edges) + 256;
}

// rev_ptr [m_src + 1] / rev_eid [E] of every job, as crfconv_reverse_csr builds them one table at a time (same contents),
// in five launches for the whole batch.  At most 32 tables per call; rows and edges of the batch must stay below 2^31.
extern "C" int crfconv_reverse_csr_batched(const crf_rev_job* jobs, int njobs, void* workspace, size_t workspace_bytes,
                                           crf_stream_t stream) {
    CRF_REQUIRE(jobs && workspace, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(njobs >= 1 && njobs <= RB_MAX, CRF_ERR_ARG, "njobs=%d outside [1, %d]", njobs, RB_MAX);
    CRF_REQUIRE(workspace_bytes >= crfconv_reverse_csr_batched_workspace(jobs, njobs), CRF_ERR_WORKSPACE, "workspace too small");
    RevBatch t;
    int64_t rows = 0, edges = 0, chunks = 0, rowchunks = 0;
    for (int j = 0; j <= RB_MAX; ++j) {
        if (j <= njobs) { t.row_base[j] = (int)rows; t.edge_base[j] = (int)edges; t.chunk_base[j] = (int)chunks; t.rowchunk_base[j] = (int)rowchunks; }
        else { t.row_base[j] = t.row_base[njobs]; t.edge_base[j] = t.edge_base[njobs]; t.chunk_base[j] = t.chunk_base[njobs]; t.rowchunk_base[j] = t.rowchunk_base[njobs]; }
        if (j < njobs) {
            const crf_rev_job& jb = jobs[j];
            CRF_REQUIRE(jb.idx32 && jb.rev_ptr && jb.rev_eid && jb.E > 0 && jb.m_src > 0, CRF_ERR_ARG, "job %d is malformed", j);
            t.idx[j] = jb.idx32; t.rev_ptr[j] = jb.rev_ptr; t.rev_eid[j] = jb.rev_eid; t.m_src[j] = (int)jb.m_src;
            rows += jb.m_src + 1; edges += jb.E; chunks += cdiv(jb.E, RB_EPB); rowchunks += cdiv(jb.m_src + 1, RB_RR);
            CRF_REQUIRE(rows < ((int64_t)1 << 31) - 1 && edges < ((int64_t)1 << 31), CRF_ERR_UNSUPPORTED, "batch too large for int32 offsets");
        } else if (j < RB_MAX) {
            t.idx[j] = nullptr; t.rev_ptr[j] = nullptr; t.rev_eid[j] = nullptr; t.m_src[j] = 0;
        }
    }
    t.njobs = njobs;
    hipStream_t st = as_stream(stream);
    char* ws = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const int64_t n = rows + 1;                        // one extra counter: ptrs[rows] = all edges that have a source
    const size_t seg = align_up(sizeof(int32_t) * (size_t)n);
    int32_t* cnt = reinterpret_cast<int32_t*>(ws);
    int32_t* cursor = reinterpret_cast<int32_t*>(ws + seg);
    int32_t* ptrs = reinterpret_cast<int32_t*>(ws + 2 * seg);
    int32_t* sums = reinterpret_cast<int32_t*>(ws + 3 * seg);
    int32_t* tmp = reinterpret_cast<int32_t*>(ws + 3 * seg + align_up(sizeof(int32_t) * scan_block_sums(n)));
    int64_t zg = cdiv((int64_t)(2 * seg / 4), 256);
    if (zg > 1024) zg = 1024;
    hipLaunchKernelGGL(zero_i32_kernel, dim3((unsigned)zg), dim3(256), 0, st, cnt, (int64_t)(2 * seg / 4));
    CRF_LAUNCH_CHECK();
    hipLaunchKernelGGL(revb_count_kernel, dim3((unsigned)chunks), dim3(256), 0, st, t, cnt);
    CRF_LAUNCH_CHECK();
    exclusive_scan_i32(cnt, ptrs, n, sums, st);
    CRF_LAUNCH_CHECK();
    hipLaunchKernelGGL(revb_fill_kernel, dim3((unsigned)chunks), dim3(256), 0, st, t, ptrs, cursor, tmp);
    CRF_LAUNCH_CHECK();
    hipLaunchKernelGGL(revb_rows_kernel, dim3((unsigned)cdiv(rowchunks, 4)), dim3(256), 0, st, t, ptrs, tmp);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}
