// Exact batched kNN on gfx950 -- replaces the reference's nanoflann KD-tree extension
// (utils/nearest_neighbors/knn_.cxx:22-135, knn_.h:2-19) with a uniform-grid search:
//   1. per-cloud bounding box            (one block per cloud)
//   2. cell id + histogram               (flat, int atomics)
//   3. exclusive scan of cell counts     (scan.hpp: this library's own scratch-free scan)
//   4. counting-sort scatter             (points packed as float4 {x, y, z, id})
//   5. one thread per query: expanding Chebyshev rings of cells, register-resident top-K,
//      stops when the K-th distance is strictly inside the searched cube.
// Result contract (bit-exact against the oracle / the reference on tie-free input):
//   distance = fl(fl(fl(dx*dx) + fl(dy*dy)) + fl(dz*dz)) in float32, no FMA contraction
//   (nanoflann.hpp:342-347); neighbours ascending by (distance, point id).
#include "common.hpp"

#include <algorithm>
#include <cstdlib>
#include "scan.hpp"

namespace crf {

struct GridInfo {  // one per cloud, device resident
    float ox, oy, oz;   // origin (bbox min)
    float cell, inv;    // cell edge, 1 / cell
    int nx, ny, nz;
    float margin;       // absolute safety margin for the stop test
};

constexpr int QBLOCK = 128;

__device__ __forceinline__ float sqdist_exact(float qx, float qy, float qz, float px, float py, float pz) {
    const float dx = qx - px, dy = qy - py, dz = qz - pz;
    return add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz));
}

// ------------------------------------------------------------------ 1. bounding box -> grid
__global__ __launch_bounds__(1024) void bbox_kernel(const float* __restrict__ pts, int64_t npts, int G0,
                                                    GridInfo* __restrict__ info) {
    const float* p = pts + (int64_t)blockIdx.x * npts * 3;
    float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (int64_t i = threadIdx.x; i < npts; i += 1024) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = p[3 * i + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
        }
    }
    __shared__ float smn[16][3], smx[16][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float lo = mn[a], hi = mx[a];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo = fminf(lo, __shfl_xor(lo, o, WAVE));
            hi = fmaxf(hi, __shfl_xor(hi, o, WAVE));
        }
        if (lane == 0) { smn[wave][a] = lo; smx[wave][a] = hi; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float lo[3], hi[3];
        for (int a = 0; a < 3; ++a) {
            lo[a] = smn[0][a]; hi[a] = smx[0][a];
            for (int w = 1; w < 16; ++w) { lo[a] = fminf(lo[a], smn[w][a]); hi[a] = fmaxf(hi[a], smx[w][a]); }
        }
        const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
        float emax = fmaxf(ex, fmaxf(ey, ez));
        if (!(emax > 0.f)) emax = 1.f;  // all points coincide (or a single point)
        GridInfo g;
        g.ox = lo[0]; g.oy = lo[1]; g.oz = lo[2];
        g.cell = emax / (float)G0;
        g.inv = 1.0f / g.cell;
        g.nx = min(G0, (int)(ex * g.inv)) + 1;
        g.ny = min(G0, (int)(ey * g.inv)) + 1;
        g.nz = min(G0, (int)(ez * g.inv)) + 1;
        const float amax = fmaxf(fmaxf(fabsf(lo[0]), fabsf(hi[0])),
                                 fmaxf(fmaxf(fabsf(lo[1]), fabsf(hi[1])), fmaxf(fabsf(lo[2]), fabsf(hi[2]))));
        g.margin = 1e-5f * (amax + emax) + 1e-30f;
        info[blockIdx.x] = g;
    }
}

// ------------------------------------------------------------------ Morton (Z-order) codes of a batch of clouds
// code = 30 bits, ten per axis, of q = clamp((p - lo) / ext * 1023, 0, 1023) with lo = the cloud's per-axis minimum and ext =
// its largest extent (>= 1e-20): what data.morton_order computed with ~45 framework launches (min / max reductions, the bit
// spreading one elementwise op at a time), in two.  Every float operation is rounded on its own, as the framework's were.
__global__ __launch_bounds__(1024) void morton_bbox_kernel(const float* __restrict__ pts, int64_t npts, float* __restrict__ box) {
    const float* p = pts + (int64_t)blockIdx.x * npts * 3;
    float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (int64_t i = threadIdx.x; i < npts; i += 1024) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = p[3 * i + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
        }
    }
    __shared__ float smn[16][3], smx[16][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float lo = mn[a], hi = mx[a];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo = fminf(lo, __shfl_xor(lo, o, WAVE));
            hi = fmaxf(hi, __shfl_xor(hi, o, WAVE));
        }
        if (lane == 0) { smn[wave][a] = lo; smx[wave][a] = hi; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float ext = 0.f;
        for (int a = 0; a < 3; ++a) {
            float lo = smn[0][a], hi = smx[0][a];
            for (int w = 1; w < 16; ++w) { lo = fminf(lo, smn[w][a]); hi = fmaxf(hi, smx[w][a]); }
            box[4 * blockIdx.x + a] = lo;
            ext = fmaxf(ext, sub_rn(hi, lo));
        }
        box[4 * blockIdx.x + 3] = fmaxf(ext, 1e-20f);
    }
}

__device__ __forceinline__ long long morton_spread3(long long v) {
    v &= 0x3FF;
    v = (v | (v << 16)) & 0x030000FF;
    v = (v | (v << 8)) & 0x0300F00F;
    v = (v | (v << 4)) & 0x030C30C3;
    v = (v | (v << 2)) & 0x09249249;
    return v;
}

__global__ __launch_bounds__(256) void morton_code_kernel(const float* __restrict__ pts, int64_t npts,
                                                          const float* __restrict__ box, long long* __restrict__ code) {
    const int b = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npts) return;
    const float* p = pts + ((int64_t)b * npts + i) * 3;
    const float ext = box[4 * b + 3];
    long long q[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float t = mul_rn(sub_rn(p[a], box[4 * b + a]) / ext, 1023.0f);
        t = fminf(fmaxf(t, 0.f), 1023.f);
        q[a] = (long long)t;                               // truncation, as Tensor.to(int64)
    }
    code[(int64_t)b * npts + i] = morton_spread3(q[0]) | (morton_spread3(q[1]) << 1) | (morton_spread3(q[2]) << 2);
}

__device__ __forceinline__ int3 cell_of(const GridInfo& g, float x, float y, float z) {
    int cx = (int)floorf((x - g.ox) * g.inv), cy = (int)floorf((y - g.oy) * g.inv),
        cz = (int)floorf((z - g.oz) * g.inv);
    cx = min(max(cx, 0), g.nx - 1);
    cy = min(max(cy, 0), g.ny - 1);
    cz = min(max(cz, 0), g.nz - 1);
    return make_int3(cx, cy, cz);
}

__global__ __launch_bounds__(256) void zero_kernel(uint4* __restrict__ p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = make_uint4(0, 0, 0, 0);
}

// ------------------------------------------------------------------ 2. histogram
__global__ __launch_bounds__(256) void cell_count_kernel(const float* __restrict__ pts, int64_t npts,
                                                         const GridInfo* __restrict__ info, int ncell_alloc,
                                                         int32_t* __restrict__ cell_id,
                                                         int32_t* __restrict__ counts) {
    const int b = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npts) return;
    const GridInfo g = info[b];
    const float* p = pts + ((int64_t)b * npts + i) * 3;
    const int3 c = cell_of(g, p[0], p[1], p[2]);
    const int cid = (c.z * g.ny + c.y) * g.nx + c.x;
    cell_id[(int64_t)b * npts + i] = cid;
    atomicAdd(&counts[(int64_t)b * ncell_alloc + cid], 1);
}

// ------------------------------------------------------------------ 4. scatter
__global__ __launch_bounds__(256) void cell_scatter_kernel(const float* __restrict__ pts, int64_t npts,
                                                           int ncell_alloc,
                                                           const int32_t* __restrict__ cell_id,
                                                           const int32_t* __restrict__ starts,
                                                           int32_t* __restrict__ fill,
                                                           float4* __restrict__ sorted) {
    const int b = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npts) return;
    const int64_t g = (int64_t)b * npts + i;
    const int64_t c = (int64_t)b * ncell_alloc + cell_id[g];
    const int pos = starts[c] + atomicAdd(&fill[c], 1);
    sorted[pos] = make_float4(pts[3 * g], pts[3 * g + 1], pts[3 * g + 2], __int_as_float((int)i));
}

// ------------------------------------------------------------------ 5. query
// Candidates are ranked by ONE 64-bit key = (float32 distance bits << 32) | point id: squared distances are
// non-negative, so their bit patterns order like the values, and the key order is exactly (distance, id).
__device__ __forceinline__ unsigned long long knn_key(float d, int id) {
    return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned int)id;
}
constexpr unsigned long long KNN_KEY_INF = ((unsigned long long)0x7f800000u << 32) | 0x7fffffffu;

template <int KM>
struct TopK {
    unsigned long long key[KM];     // ascending
    // Only K <= KM slots are live: the first KM - K are pinned at key 0 (nothing is < 0, so nothing displaces them)
    // and the live ones end at the static position KM - 1 (a runtime "key[K - 1]" would push the array to scratch).
    __device__ __forceinline__ void init(int K) {
#pragma unroll
        for (int k = 0; k < KM; ++k) key[k] = (k < KM - K) ? 0ull : KNN_KEY_INF;
    }
    __device__ __forceinline__ unsigned long long kth_key() const { return key[KM - 1]; }
    __device__ __forceinline__ float kth() const { return __uint_as_float((unsigned int)(key[KM - 1] >> 32)); }
    __device__ __forceinline__ void offer(unsigned long long c) {
        if (!(c < key[KM - 1])) return;
        // sorted insert with selects only (static register indices, no conditional stores)
        bool below = c < key[KM - 1];                       // c < key[k], carried down the slots
#pragma unroll
        for (int k = KM - 1; k > 0; --k) {
            const bool below_prev = c < key[k - 1];         // slot k takes its upper neighbour
            key[k] = below_prev ? key[k - 1] : (below ? c : key[k]);
            below = below_prev;
        }
        key[0] = below ? c : key[0];
    }
};

constexpr int QCAP = 8;     // accepted candidates a lane may hold before the wavefront inserts them

template <int KM>
__global__ __launch_bounds__(QBLOCK) void knn_query_kernel(const float* __restrict__ queries, int64_t nq,
                                                           int64_t npts, int K,
                                                           const GridInfo* __restrict__ info,
                                                           int ncell_alloc,
                                                           const int32_t* __restrict__ starts,
                                                           const float4* __restrict__ sorted,
                                                           int64_t* __restrict__ out64,
                                                           int32_t* __restrict__ out32) {
    // The sorted insert is a ~100-instruction select chain and a wavefront runs it whenever ANY of its 64 queries
    // accepts a candidate -- with independent queries that is nearly every candidate.  Accepted keys are therefore
    // parked in a per-lane LDS queue (a cheap predicated store) and inserted in wave-wide batches: when some lane's
    // queue is full, and at the end of every ring (the stop test needs the true K-th distance).
    constexpr bool QUEUED = KM > 1;
    __shared__ unsigned long long s_queue[QUEUED ? QCAP * QBLOCK : 1];
    const int b = blockIdx.y;
    const int64_t qi = (int64_t)blockIdx.x * QBLOCK + threadIdx.x;
    if (qi >= nq) return;
    const GridInfo g = info[b];
    const float* qp = queries + ((int64_t)b * nq + qi) * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    const int3 c = cell_of(g, qx, qy, qz);
    const int32_t* st = starts + (int64_t)b * ncell_alloc;

    TopK<KM> best;
    best.init(K);
    int cnt = 0;
    unsigned long long bound = KNN_KEY_INF;                 // K-th key as of the last flush (only ever too large)
    auto flush = [&]() {
        if constexpr (QUEUED) {
            for (int i = 0; i < QCAP; ++i) {
                if (!__any(i < cnt)) break;
                if (i < cnt) best.offer(s_queue[i * QBLOCK + threadIdx.x]);
            }
            cnt = 0;
            bound = best.kth_key();
        }
    };
    auto consider = [&](const float4 s) {
        const unsigned long long key = knn_key(sqdist_exact(qx, qy, qz, s.x, s.y, s.z), __float_as_int(s.w));
        if constexpr (QUEUED) {
            if (key < bound) {
                s_queue[cnt * QBLOCK + threadIdx.x] = key;
                ++cnt;
            }
            if (__any(cnt == QCAP)) flush();
        } else {
            best.offer(key);
        }
    };
    const int rmax = max(g.nx, max(g.ny, g.nz));
    for (int r = 0; r <= rmax; ++r) {
        const int z0 = max(c.z - r, 0), z1 = min(c.z + r, g.nz - 1);
        const int y0 = max(c.y - r, 0), y1 = min(c.y + r, g.ny - 1);
        for (int z = z0; z <= z1; ++z) {
            const bool zface = (z == c.z - r) || (z == c.z + r);
            for (int y = y0; y <= y1; ++y) {
                const bool full = zface || (y == c.y - r) || (y == c.y + r);
                // full: whole x-span of the shell row; otherwise only its two end cells
                const int nseg = full ? 1 : (r == 0 ? 1 : 2);
                for (int sgi = 0; sgi < nseg; ++sgi) {
                    int xa, xb;
                    if (full) { xa = max(c.x - r, 0); xb = min(c.x + r, g.nx - 1); }
                    else if (sgi == 0) { xa = xb = c.x - r; }
                    else { xa = xb = c.x + r; }
                    if (xa < 0 || xb >= g.nx || xa > xb) continue;
                    const int rowbase = (z * g.ny + y) * g.nx;
                    const int pbeg = st[rowbase + xa], pend = st[rowbase + xb + 1];
                    int p = pbeg;
                    for (; p + 4 <= pend; p += 4) {          // four candidate loads in flight
                        const float4 s0 = sorted[p], s1 = sorted[p + 1], s2 = sorted[p + 2], s3 = sorted[p + 3];
                        consider(s0); consider(s1); consider(s2); consider(s3);
                    }
                    for (; p < pend; ++p) consider(sorted[p]);
                }
            }
        }
        flush();
        // every point outside the cube of rings <= r is at least `gap` away along some axis
        float gap = 3.4e38f;
        if (c.x - r > 0) gap = fminf(gap, qx - (g.ox + (float)(c.x - r) * g.cell));
        if (c.x + r + 1 < g.nx) gap = fminf(gap, (g.ox + (float)(c.x + r + 1) * g.cell) - qx);
        if (c.y - r > 0) gap = fminf(gap, qy - (g.oy + (float)(c.y - r) * g.cell));
        if (c.y + r + 1 < g.ny) gap = fminf(gap, (g.oy + (float)(c.y + r + 1) * g.cell) - qy);
        if (c.z - r > 0) gap = fminf(gap, qz - (g.oz + (float)(c.z - r) * g.cell));
        if (c.z + r + 1 < g.nz) gap = fminf(gap, (g.oz + (float)(c.z + r + 1) * g.cell) - qz);
        if (gap >= 3.0e38f) break;  // the cube covers the whole grid
        const float safe = gap - g.margin;
        if (safe > 0.f && best.kth() < safe * safe) break;
    }
    const int64_t o = ((int64_t)b * nq + qi) * K;
#pragma unroll
    for (int k = 0; k < KM; ++k) {
        if (k >= KM - K) {
            const int id = (int)(unsigned int)(best.key[k] & 0xffffffffull);
            if (out64) out64[o + k - (KM - K)] = id;
            if (out32) out32[o + k - (KM - K)] = id;
        }
    }
}

// ------------------------------------------------------------------ 5b. query, sixteen lanes per query (K <= 16)
// The one-lane search above is a chain of ~100 dependent round trips per query (cell start -> candidates, segment by
// segment).  Here the row segments of a ring are dealt over the SIXTEEN lanes of a query group (four queries per
// wavefront): ring 1 = the 3 x 3 x 3 cube as nine x-rows, ring r >= 2 = the shell's 8r face rows (full x-span, one
// contiguous candidate range each) plus two end cells for each of its (2r - 1)^2 inner rows.  (A first round of the
// nine centre rows FIVE cells wide, so that cube 2 needs only its sixteen face rows afterwards, measured slower -- 371
// against 233 us: indoor clouds are surfaces, an occupied cell holds several points, and the rank selection is quadratic
// in the candidates that arrive before a bound exists.)  Accepted keys
// (key < bound) are appended to a per-query LDS pool (LDS atomic slot counter); after every ring -- and whenever the
// pool could overflow -- the group ranks the pool entries against each other (entry i's rank = number of smaller keys)
// and the sixteen smallest land, sorted, in slots 0..15: that is the running top-K, its last key the new bound, and
// at the end the output row.  Same keys, same stop test -> same tables as the one-lane kernel.
#ifndef CO_LPQ_
#define CO_LPQ_ 16          // lanes per query; swept on 4 x 40960 self-queries: 8 lanes 262 us, 16 lanes 232 us, 32 lanes 289 us per call
#endif
#ifndef CO_CAP_
#define CO_CAP_ 128
#endif
constexpr int CO_LPQ = CO_LPQ_, CO_BLOCK = 256, CO_QPB = CO_BLOCK / CO_LPQ, CO_CAP = CO_CAP_, CO_CHUNK = 4;
constexpr int CO_ROUND = CO_LPQ * CO_CHUNK;          // appends of one chunk round at most

template <int KM>      // K <= KM <= 16
__global__ __launch_bounds__(CO_BLOCK) void knn_coop_kernel(const float* __restrict__ queries, int64_t nq, int K,
                                                            const GridInfo* __restrict__ info, int ncell_alloc,
                                                            const int32_t* __restrict__ starts,
                                                            const float4* __restrict__ sorted,
                                                            int64_t* __restrict__ out64, int32_t* __restrict__ out32) {
    static_assert(KM <= 16, "the running top-K lives in pool slots 0..15");
    __shared__ unsigned long long s_pool[2][CO_QPB][CO_CAP];       // double-buffered: a compaction writes the other buffer
    __shared__ int s_cnt[CO_QPB];
    const int b = blockIdx.y;
    const int l = threadIdx.x & (CO_LPQ - 1), qs = threadIdx.x / CO_LPQ;
    const int gshift = (threadIdx.x & 63) & ~(CO_LPQ - 1);         // first lane of this group inside its wavefront
    int64_t qi = (int64_t)blockIdx.x * CO_QPB + qs;
    const bool qvalid = qi < nq;
    if (!qvalid) qi = nq - 1;
    const GridInfo g = info[b];
    const float* qp = queries + ((int64_t)b * nq + qi) * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    const int3 c = cell_of(g, qx, qy, qz);
    const int32_t* st = starts + (int64_t)b * ncell_alloc;
    unsigned long long* pool = s_pool[0][qs];
    int which = 0;
    if (l == 0) s_cnt[qs] = 0;
    __builtin_amdgcn_wave_barrier();
    unsigned long long bound = KNN_KEY_INF;

    // rank selection: slots 0..min(n, 16)-1 of the OTHER buffer <- the smallest keys in ascending order
    auto compact = [&]() {
        __builtin_amdgcn_wave_barrier();
        const int n = s_cnt[qs];
        unsigned long long* dst = s_pool[which ^ 1][qs];
        for (int i = l; i < n; i += CO_LPQ) {                    // this lane's entries, one after the other
            const unsigned long long mine = pool[i];
            int rank = 0;
            int j = 0;
            for (; j + 4 <= n; j += 4) {                         // one address per group and read: LDS broadcasts
                const unsigned long long p0 = pool[j], p1 = pool[j + 1], p2 = pool[j + 2], p3 = pool[j + 3];
                rank += (p0 < mine ? 1 : 0) + (p1 < mine ? 1 : 0) + (p2 < mine ? 1 : 0) + (p3 < mine ? 1 : 0);
            }
            for (; j < n; ++j) rank += pool[j] < mine ? 1 : 0;
            if (rank < 16) dst[rank] = mine;
        }
        const int m = n < 16 ? n : 16;
        if (l == 0) s_cnt[qs] = m;
        which ^= 1;
        pool = dst;
        __builtin_amdgcn_wave_barrier();
        bound = m >= K ? pool[K - 1] : KNN_KEY_INF;
    };
    // this lane's candidate range [pb, pe) of the ring's slot `s`; empty when the slot lies outside the grid
    auto segment = [&](int r, int s, int& pb, int& pe) {
        pb = pe = 0;
        int dz, dy, xa, xb;
        const int side = 2 * r + 1;
        if (r == 1) {
            if (s >= 9) return;
            dz = s / 3 - 1; dy = s % 3 - 1; xa = c.x - 1; xb = c.x + 1;
        } else {
            const int nface = 8 * r, inner = side - 2;
            if (s < nface) {
                if (s < side) { dz = -r; dy = s - r; }
                else if (s < 2 * side) { dz = r; dy = s - side - r; }
                else { const int k = s - 2 * side; dz = (k >> 1) - r + 1; dy = (k & 1) ? r : -r; }
                xa = c.x - r; xb = c.x + r;
            } else {
                const int k = s - nface;
                if (k >= 2 * inner * inner) return;
                const int row = k >> 1;
                const int rq = (int)(((float)row + 0.5f) / (float)inner);      // row / inner (the margin 0.5 / inner >> float error)
                dz = rq - r + 1; dy = row - rq * inner - r + 1;
                xa = xb = (k & 1) ? c.x + r : c.x - r;
                if (xa < 0 || xa >= g.nx) return;
            }
        }
        const int z = c.z + dz, y = c.y + dy;
        if (z < 0 || z >= g.nz || y < 0 || y >= g.ny) return;
        xa = max(xa, 0); xb = min(xb, g.nx - 1);
        if (xa > xb) return;
        const int rowbase = (z * g.ny + y) * g.nx;
        pb = st[rowbase + xa];
        pe = st[rowbase + xb + 1];
    };
    const int rmax = max(g.nx, max(g.ny, g.nz));
    int room = CO_CAP;                                  // free pool slots, counted conservatively (group-uniform)
    for (int r = 1; r <= rmax; ++r) {
        const int side = 2 * r + 1;
        const int nslot = r == 1 ? 9 : 8 * r + 2 * (side - 2) * (side - 2);
        for (int s0 = 0; s0 < nslot; s0 += CO_LPQ) {
            int pb, pe;
            segment(r, s0 + l, pb, pe);
            for (;;) {
                const unsigned long long vote = __ballot(pb < pe);
                if (((vote >> gshift) & ((1ull << CO_LPQ) - 1ull)) == 0ull) break;
                if (room < CO_ROUND) {                              // the conservative count says the pool could overflow
                    __builtin_amdgcn_wave_barrier();
                    room = CO_CAP - s_cnt[qs];
                    if (room < CO_ROUND) {
                        compact();
                        room = CO_CAP - 16;
                    }
                }
                if (bound == KNN_KEY_INF && CO_CAP - room >= K) {
                    // no exact K-th key yet (fewer than K candidates at the last compaction), but the pool may hold K
                    // entries by now: the largest of ANY K candidates is an upper bound of the K-th smallest
                    __builtin_amdgcn_wave_barrier();
                    if (s_cnt[qs] >= K) {
                        unsigned long long v = 0ull;
                        for (int i = l; i < K; i += CO_LPQ) v = pool[i] > v ? pool[i] : v;
#pragma unroll
                        for (int o = 1; o < CO_LPQ; o <<= 1) {
                            const unsigned long long w = __shfl_xor(v, o, WAVE);
                            v = w > v ? w : v;
                        }
                        bound = v + 1ull;                           // the entry itself stays eligible (it is in the pool already)
                    }
                }
                float4 cand[CO_CHUNK];
#pragma unroll
                for (int u = 0; u < CO_CHUNK; ++u) cand[u] = sorted[pb + u < pe ? pb + u : (pb < pe ? pb : 0)];
                unsigned long long key[CO_CHUNK];
                bool ok[CO_CHUNK];
                int k = 0;
#pragma unroll
                for (int u = 0; u < CO_CHUNK; ++u) {
                    key[u] = knn_key(sqdist_exact(qx, qy, qz, cand[u].x, cand[u].y, cand[u].z), __float_as_int(cand[u].w));
                    ok[u] = pb + u < pe && key[u] < bound;
                    k += ok[u] ? 1 : 0;
                }
                if (k > 0) {                                        // one slot reservation per lane and round
                    int slot = atomicAdd(&s_cnt[qs], k);
#pragma unroll
                    for (int u = 0; u < CO_CHUNK; ++u)
                        if (ok[u]) pool[slot++] = key[u];
                }
                room -= CO_ROUND;
                pb += CO_CHUNK;
            }
        }
        compact();
        room = CO_CAP - 16;
        float gap = 3.4e38f;                // as the one-lane kernel: distance to the nearest face of the searched cube
        if (c.x - r > 0) gap = fminf(gap, qx - (g.ox + (float)(c.x - r) * g.cell));
        if (c.x + r + 1 < g.nx) gap = fminf(gap, (g.ox + (float)(c.x + r + 1) * g.cell) - qx);
        if (c.y - r > 0) gap = fminf(gap, qy - (g.oy + (float)(c.y - r) * g.cell));
        if (c.y + r + 1 < g.ny) gap = fminf(gap, (g.oy + (float)(c.y + r + 1) * g.cell) - qy);
        if (c.z - r > 0) gap = fminf(gap, qz - (g.oz + (float)(c.z - r) * g.cell));
        if (c.z + r + 1 < g.nz) gap = fminf(gap, (g.oz + (float)(c.z + r + 1) * g.cell) - qz);
        if (gap >= 3.0e38f) break;
        const float safe = gap - g.margin;
        const float kth = __uint_as_float((unsigned int)(bound >> 32));
        if (safe > 0.f && bound != KNN_KEY_INF && kth < safe * safe) break;
    }
    if (qvalid) {
        for (int i = l; i < K; i += CO_LPQ) {
            const int id = (int)(unsigned int)(pool[i] & 0xffffffffull);
            const int64_t o = ((int64_t)b * nq + qi) * K + i;
            if (out64) out64[o] = id;
            if (out32) out32[o] = id;
        }
    }
}

// ------------------------------------------------------------------ small clouds: no grid, sixteen lanes per query
// The grid search above is ONE lane per query walking ~100 dependent (cell start -> candidates) round trips: its time
// barely depends on the number of queries (213 us for 4 x 40960 self-queries, still 123 us for 4 x 320), so the coarse
// levels of the collate cost as much as the fine ones.  For clouds of <= BRUTE_MAX points every query instead looks at
// EVERY point of its cloud, sixteen lanes per query: lane l ranks candidates l, l + 16, ... (same 64-bit (distance, id)
// keys, same parked-insert queue), starting at the query's own neighbourhood in memory order -- the collate keeps every
// level Morton-sorted, so the first window already gives a tight bound -- then the sixteen sorted lists are merged by
// K rounds of group-wide minimum.  No grid construction (seven launches less per call), results identical: the K smallest
// keys of a set do not depend on the order they are offered in.
constexpr int BRUTE_MAX = 4096, BRUTE_LPQ = 16, BRUTE_BLOCK = 256, BRUTE_QPB = BRUTE_BLOCK / BRUTE_LPQ;

__device__ __forceinline__ unsigned long long group16_min(unsigned long long v) {
#pragma unroll
    for (int o = 1; o < BRUTE_LPQ; o <<= 1) {
        const unsigned long long w = __shfl_xor(v, o, WAVE);
        v = w < v ? w : v;
    }
    return v;
}

template <int KM>      // K == KM exactly
__global__ __launch_bounds__(BRUTE_BLOCK) void knn_brute_kernel(const float* __restrict__ pts, int64_t npts,
                                                                const float* __restrict__ queries, int64_t nq,
                                                                int64_t* __restrict__ out64, int32_t* __restrict__ out32) {
    constexpr bool QUEUED = KM > 1;
    __shared__ unsigned long long s_queue[QUEUED ? QCAP * BRUTE_BLOCK : 1];
    const int b = blockIdx.y;
    const int l = threadIdx.x & (BRUTE_LPQ - 1);
    int64_t qi = (int64_t)blockIdx.x * BRUTE_QPB + (threadIdx.x >> 4);
    const bool qvalid = qi < nq;
    if (!qvalid) qi = nq - 1;                              // keep every lane in the wave-wide votes and shuffles
    const float* qp = queries + ((int64_t)b * nq + qi) * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    const float* P = pts + (int64_t)b * npts * 3;
    const int n = (int)npts;
    // scan order: start 32 points before the query's proportional position in the candidate array, wrap around
    int start = ((int)((qi * npts) / nq) - 32) % n;
    if (start < 0) start += n;

    TopK<KM> best;
    best.init(KM);
    int cnt = 0;
    unsigned long long bound = KNN_KEY_INF;
    auto flush = [&]() {
        if constexpr (QUEUED) {
            for (int i = 0; i < QCAP; ++i) {
                if (!__any(i < cnt)) break;
                if (i < cnt) best.offer(s_queue[i * BRUTE_BLOCK + threadIdx.x]);
            }
            cnt = 0;
            // every lane of the group may use the smallest K-th key among the sixteen lists: that list alone already
            // holds K candidates at or below it
            bound = group16_min(best.kth_key());
        }
    };
    auto consider = [&](float px, float py, float pz, int id, bool live) {
        const unsigned long long key = knn_key(sqdist_exact(qx, qy, qz, px, py, pz), id);
        if constexpr (QUEUED) {
            if (live && key < bound) {
                s_queue[cnt * BRUTE_BLOCK + threadIdx.x] = key;
                ++cnt;
            }
            if (__any(cnt == QCAP)) flush();
        } else {
            if (live) best.offer(key);
        }
    };
    const int rounds = (n + BRUTE_LPQ - 1) / BRUTE_LPQ;
    int t = 0;
    for (; t + 4 <= rounds; t += 4) {                      // four candidates per lane in flight
        int id[4];
        float c[4][3];
        bool live[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = (t + u) * BRUTE_LPQ + l;
            live[u] = i < n;
            int j = start + (live[u] ? i : 0);
            if (j >= n) j -= n;
            id[u] = j;
            c[u][0] = P[3 * j]; c[u][1] = P[3 * j + 1]; c[u][2] = P[3 * j + 2];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) consider(c[u][0], c[u][1], c[u][2], id[u], live[u]);
        if (t == 0) flush();                               // the first window fixes a tight bound early
    }
    for (; t < rounds; ++t) {
        const int i = t * BRUTE_LPQ + l;
        const bool live = i < n;
        int j = start + (live ? i : 0);
        if (j >= n) j -= n;
        consider(P[3 * j], P[3 * j + 1], P[3 * j + 2], j, live);
    }
    flush();
    // merge: K rounds, each takes the smallest head of the sixteen lists; the owning lane drops it
    const int64_t o = ((int64_t)b * nq + qi) * KM;
#pragma unroll 1
    for (int k = 0; k < KM; ++k) {
        const unsigned long long mine = best.key[0];
        const unsigned long long mn = group16_min(mine);
        if (mine == mn) {                                  // keys are unique (ids are): exactly one lane
#pragma unroll
            for (int i = 0; i + 1 < KM; ++i) best.key[i] = best.key[i + 1];
            best.key[KM - 1] = KNN_KEY_INF;
        }
        if (l == 0 && qvalid) {
            const int id = (int)(unsigned int)(mn & 0xffffffffull);
            if (out64) out64[o + k] = id;
            if (out32) out32[o + k] = id;
        }
    }
}

static size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

static double knn_points_per_cell() { return 0.6; }      // sweep 0.35 .. 16 on the bench clouds: flat optimum around 0.6

static int grid_g0(size_t npts) {
    int g = (int)floor(cbrt((double)npts / knn_points_per_cell()));
    if (g < 1) g = 1;
    if (g > 160) g = 160;
    return g;
}

struct KnnLayout {
    int G0, ncell_alloc;
    size_t off_info, off_cellid, off_counts, off_starts, off_fill, off_sorted, off_temp, temp_bytes, total;
};

static KnnLayout knn_layout(size_t B, size_t npts) {
    KnnLayout L;
    L.G0 = grid_g0(npts);
    L.ncell_alloc = (L.G0 + 1) * (L.G0 + 1) * (L.G0 + 1);
    const size_t ncell_total = B * (size_t)L.ncell_alloc + 1;
    size_t o = 0;
    L.off_info = o;   o += al(sizeof(GridInfo) * B);
    L.off_cellid = o; o += al(sizeof(int32_t) * B * npts);
    L.off_counts = o; o += al(sizeof(int32_t) * ncell_total);
    L.off_fill = o;   o += al(sizeof(int32_t) * ncell_total);   // counts and fill are zeroed together
    L.off_starts = o; o += al(sizeof(int32_t) * ncell_total);
    L.off_sorted = o; o += al(sizeof(float4) * B * npts);
    const size_t tb = sizeof(int32_t) * scan_block_sums((int64_t)ncell_total);
    L.temp_bytes = tb;
    L.off_temp = o;   o += al(tb);
    L.total = o + 256;
    return L;
}

}  // namespace crf

using namespace crf;

extern "C" int crfconv_morton_codes(const float* pts, int64_t B, int64_t npts, float* box_ws, int64_t* codes,
                                    crf_stream_t stream) {
    CRF_REQUIRE(pts && box_ws && codes, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(B > 0 && npts > 0 && B < 65536, CRF_ERR_ARG, "bad shape B=%lld npts=%lld", (long long)B, (long long)npts);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(morton_bbox_kernel, dim3((unsigned)B), dim3(1024), 0, st, pts, npts, box_ws);
    hipLaunchKernelGGL(morton_code_kernel, dim3((unsigned)cdiv(npts, 256), (unsigned)B), dim3(256), 0, st, pts, npts, box_ws,
                       reinterpret_cast<long long*>(codes));
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" size_t crfconv_knn_batch_dev_workspace(size_t batch_size, size_t npts, size_t nqueries, size_t K) {
    (void)nqueries; (void)K;
    if (batch_size == 0 || npts == 0) return 0;
    return knn_layout(batch_size, npts).total;
}

extern "C" int crfconv_knn_batch_dev(const float* pts, size_t batch_size, size_t npts, size_t dim,
                                     const float* queries, size_t nqueries, size_t K, int64_t* out_i64,
                                     int32_t* out_i32, void* workspace, size_t workspace_bytes,
                                     crf_stream_t stream) {
    CRF_REQUIRE(pts && queries && workspace && (out_i64 || out_i32), CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(dim == 3, CRF_ERR_UNSUPPORTED, "kNN supports dim == 3 only (got %zu)", dim);
    CRF_REQUIRE(batch_size >= 1 && batch_size <= 65535 && npts >= 1 && nqueries >= 1, CRF_ERR_ARG, "empty input");
    CRF_REQUIRE(K >= 1 && K <= 64 && K <= npts, CRF_ERR_ARG, "K=%zu must satisfy 1 <= K <= min(64, npts=%zu)", K, npts);
    CRF_REQUIRE(batch_size * npts < ((size_t)1 << 31), CRF_ERR_UNSUPPORTED, "B*npts too large");
    const KnnLayout L = knn_layout(batch_size, npts);
    CRF_REQUIRE(workspace_bytes >= L.total, CRF_ERR_WORKSPACE, "knn workspace %zu < %zu", workspace_bytes, L.total);
    hipStream_t st = as_stream(stream);
    if (npts <= (size_t)BRUTE_MAX && (K == 1 || K == 8 || K == 16 || K == 32)) {
        const dim3 bgrid((unsigned)cdiv((int64_t)nqueries, BRUTE_QPB), (unsigned)batch_size);
#define KNN_BRUTE(KM)                                                                                       \
    hipLaunchKernelGGL(knn_brute_kernel<KM>, bgrid, dim3(BRUTE_BLOCK), 0, st, pts, (int64_t)npts, queries, \
                       (int64_t)nqueries, out_i64, out_i32)
        if (K == 1) KNN_BRUTE(1);
        else if (K == 8) KNN_BRUTE(8);
        else if (K == 16) KNN_BRUTE(16);
        else KNN_BRUTE(32);
#undef KNN_BRUTE
        CRF_LAUNCH_CHECK();
        return CRF_OK;
    }
    char* ws = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    GridInfo* info = reinterpret_cast<GridInfo*>(ws + L.off_info);
    int32_t* cell_id = reinterpret_cast<int32_t*>(ws + L.off_cellid);
    int32_t* counts = reinterpret_cast<int32_t*>(ws + L.off_counts);
    int32_t* fill = reinterpret_cast<int32_t*>(ws + L.off_fill);
    int32_t* starts = reinterpret_cast<int32_t*>(ws + L.off_starts);
    float4* sorted = reinterpret_cast<float4*>(ws + L.off_sorted);
    void* temp = ws + L.off_temp;
    const size_t ncell_total = batch_size * (size_t)L.ncell_alloc + 1;

    // counts + fill.  A kernel, not hipMemsetAsync: captured into a hipGraph, the memset NODE of this call faulted ("Memory
    // access fault ... write access to a read-only page") as soon as any eager launch ran between two replays (ROCm 7.2,
    // scratch/cg_diag2.py), while kernel nodes replay fine.
    {
        const size_t n16 = (L.off_starts - L.off_counts) / 16;            // the regions are 256-byte aligned
        hipLaunchKernelGGL(zero_kernel, dim3((unsigned)std::min<size_t>(1024, (n16 + 255) / 256)), dim3(256), 0, st,
                           reinterpret_cast<uint4*>(counts), n16);
        CRF_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(bbox_kernel, dim3((unsigned)batch_size), dim3(1024), 0, st, pts, (int64_t)npts, L.G0, info);
    CRF_LAUNCH_CHECK();
    const dim3 pgrid((unsigned)cdiv((int64_t)npts, 256), (unsigned)batch_size);
    hipLaunchKernelGGL(cell_count_kernel, pgrid, dim3(256), 0, st, pts, (int64_t)npts, info, L.ncell_alloc,
                       cell_id, counts);
    CRF_LAUNCH_CHECK();
    exclusive_scan_i32(counts, starts, (int64_t)ncell_total, reinterpret_cast<int32_t*>(temp), st);
    hipLaunchKernelGGL(cell_scatter_kernel, pgrid, dim3(256), 0, st, pts, (int64_t)npts, L.ncell_alloc, cell_id,
                       starts, fill, sorted);
    CRF_LAUNCH_CHECK();
    const dim3 qgrid((unsigned)cdiv((int64_t)nqueries, QBLOCK), (unsigned)batch_size);
#define KNN_LAUNCH(KM)                                                                                     \
    hipLaunchKernelGGL(knn_query_kernel<KM>, qgrid, dim3(QBLOCK), 0, st, queries, (int64_t)nqueries,       \
                       (int64_t)npts, (int)K, info, L.ncell_alloc, starts, sorted, out_i64, out_i32)
    if (K > 1 && K <= 16) {
        const dim3 cgrid((unsigned)cdiv((int64_t)nqueries, CO_QPB), (unsigned)batch_size);
        hipLaunchKernelGGL(knn_coop_kernel<16>, cgrid, dim3(CO_BLOCK), 0, st, queries, (int64_t)nqueries, (int)K, info,
                           L.ncell_alloc, starts, sorted, out_i64, out_i32);
        CRF_LAUNCH_CHECK();
        return CRF_OK;
    }
    if (K == 1) KNN_LAUNCH(1);
    else if (K <= 8) KNN_LAUNCH(8);
    else if (K <= 16) KNN_LAUNCH(16);
    else if (K <= 32) KNN_LAUNCH(32);
    else KNN_LAUNCH(64);
#undef KNN_LAUNCH
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

// ------------------------------------------------------------------ host-buffer forms (knn_.h signatures)
static int knn_host(const float* pts, size_t B, size_t npts, size_t dim, const float* queries, size_t nq,
                    size_t K, long* out) {
    CRF_REQUIRE(pts && queries && out, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(dim == 3, CRF_ERR_UNSUPPORTED, "kNN supports dim == 3 only (got %zu)", dim);
    const size_t wsb = crfconv_knn_batch_dev_workspace(B, npts, nq, K);
    float *dp = nullptr, *dq = nullptr;
    int64_t* dout = nullptr;
    void* ws = nullptr;
    int rc = CRF_OK;
    hipError_t e;
#define TRY(x) if ((e = (x)) != hipSuccess) { set_error("%s: %s", #x, hipGetErrorString(e)); rc = CRF_ERR_HIP; goto done; }
    TRY(hipMalloc(&dp, sizeof(float) * 3 * B * npts));
    TRY(hipMalloc(&dq, sizeof(float) * 3 * B * nq));
    TRY(hipMalloc(&dout, sizeof(int64_t) * B * nq * K));
    TRY(hipMalloc(&ws, wsb));
    TRY(hipMemcpy(dp, pts, sizeof(float) * 3 * B * npts, hipMemcpyHostToDevice));
    TRY(hipMemcpy(dq, queries, sizeof(float) * 3 * B * nq, hipMemcpyHostToDevice));
    rc = crfconv_knn_batch_dev(dp, B, npts, dim, dq, nq, K, dout, nullptr, ws, wsb, nullptr);
    if (rc != CRF_OK) goto done;
    TRY(hipStreamSynchronize(nullptr));
    static_assert(sizeof(long) == sizeof(int64_t), "LP64 expected");
    TRY(hipMemcpy(out, dout, sizeof(int64_t) * B * nq * K, hipMemcpyDeviceToHost));
#undef TRY
done:
    if (dp) (void)hipFree(dp);
    if (dq) (void)hipFree(dq);
    if (dout) (void)hipFree(dout);
    if (ws) (void)hipFree(ws);
    return rc;
}

extern "C" int crfconv_knn(const float* points, size_t npts, size_t dim, const float* queries,
                           size_t nqueries, size_t K, long* indices) {
    return knn_host(points, 1, npts, dim, queries, nqueries, K, indices);
}
extern "C" int crfconv_knn_omp(const float* points, size_t npts, size_t dim, const float* queries,
                               size_t nqueries, size_t K, long* indices) {
    return knn_host(points, 1, npts, dim, queries, nqueries, K, indices);
}
extern "C" int crfconv_knn_batch(const float* batch_data, size_t batch_size, size_t npts, size_t dim,
                                 const float* queries, size_t nqueries, size_t K, long* batch_indices) {
    return knn_host(batch_data, batch_size, npts, dim, queries, nqueries, K, batch_indices);
}
extern "C" int crfconv_knn_batch_omp(const float* batch_data, size_t batch_size, size_t npts, size_t dim,
                                     const float* queries, size_t nqueries, size_t K, long* batch_indices) {
    return knn_host(batch_data, batch_size, npts, dim, queries, nqueries, K, batch_indices);
}

// ------------------------------------------------------------------ farthest point sampling
// torch_cluster.fps as used by the reference's sparse graph builder (models/point_conv.py:381).  Inherently
// sequential in the number of samples; one 1024-thread workgroup per cloud keeps the running min-distance
// array in global memory (L2-resident) and does min-update + argmax (ties -> lower index) per iteration.
namespace crf {
__global__ __launch_bounds__(1024) void fps_kernel(const float* __restrict__ pos, const int64_t* __restrict__ seg_start,
                                                   const int64_t* __restrict__ seg_count,
                                                   const int64_t* __restrict__ out_start,
                                                   const int64_t* __restrict__ n_sample,
                                                   const int64_t* __restrict__ first, float* __restrict__ dist,
                                                   int64_t* __restrict__ out) {
    const int c = blockIdx.x;
    const int64_t s0 = seg_start[c], n = seg_count[c], m = n_sample[c];
    const float* p = pos + 3 * s0;
    float* d = dist + s0;
    int64_t* o = out + out_start[c];
    __shared__ float sval[16];
    __shared__ int sidx[16];
    __shared__ int scur;
    for (int64_t i = threadIdx.x; i < n; i += 1024) d[i] = 3.4e38f;
    if (threadIdx.x == 0) scur = (int)first[c];
    __syncthreads();
    for (int64_t t = 0; t < m; ++t) {
        const int cur = scur;
        if (threadIdx.x == 0) o[t] = s0 + cur;
        const float cx = p[3 * cur], cy = p[3 * cur + 1], cz = p[3 * cur + 2];
        float best = -1.f;
        int bi = 0x7fffffff;
        for (int64_t i = threadIdx.x; i < n; i += 1024) {
            const float dx = p[3 * i] - cx, dy = p[3 * i + 1] - cy, dz = p[3 * i + 2] - cz;
            const float dd = fminf(d[i], dx * dx + dy * dy + dz * dz);
            d[i] = dd;
            if (dd > best) { best = dd; bi = (int)i; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(best, off, WAVE);
            const int oi = __shfl_xor(bi, off, WAVE);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        __syncthreads();                                   // scur consumed by everyone
        if (lane == 0) { sval[wave] = best; sidx[wave] = bi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float bv = sval[0];
            int bj = sidx[0];
            for (int w = 1; w < 16; ++w)
                if (sval[w] > bv || (sval[w] == bv && sidx[w] < bj)) { bv = sval[w]; bj = sidx[w]; }
            scur = bj;
        }
        __syncthreads();
    }
}
}  // namespace crf

extern "C" int crfconv_fps(const float* pos, int n_clouds, const int64_t* seg_start, const int64_t* seg_count,
                           const int64_t* out_start, const int64_t* n_sample, const int64_t* first, float* dist_ws,
                           int64_t* out, crf_stream_t stream) {
    CRF_REQUIRE(pos && seg_start && seg_count && out_start && n_sample && first && dist_ws && out, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(n_clouds >= 1 && n_clouds <= 65535, CRF_ERR_ARG, "n_clouds=%d out of range", n_clouds);
    hipLaunchKernelGGL(crf::fps_kernel, dim3(n_clouds), dim3(1024), 0, crf::as_stream(stream), pos, seg_start, seg_count,
                       out_start, n_sample, first, dist_ws, out);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}
