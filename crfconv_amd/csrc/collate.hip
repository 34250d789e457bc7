// Device-side pieces of the collate (datasets/semantic3d_dataset.py:512-528) that used to run on the host or in framework
// kernels with scratch memory, so that data.CollateGraph is ONE hipGraph replay with no host work per batch:
//
//   crfconv_random_subsets   the per-level random subsets `randperm(n)[: n // ratio]` (:517), kept in ascending order:
//                            every point gets a 64-bit counter-based key (splitmix64 of seed, batch counter, level, point),
//                            the s smallest keys are selected by a radix select and a block-wide scan writes their indices
//                            in ascending order.  One workgroup per level, all levels in one launch.  The DRAWS differ from
//                            torch.randperm's (any uniform subset is the reference's semantics); reproducible from the seed.
//   crfconv_argsort_codes    stable argsort of the 30-bit Morton codes per cloud (torch.argsort(stable=True) needs a private
//                            segment: not replayable from a captured graph on ROCm 7.2): bucket by the top 16 bits
//                            (histogram -> exclusive scan -> fill), then every element ranks itself inside its bucket by
//                            (code, index).  Deterministic, scratch-free, six small launches.
#include "common.hpp"
#include "scan.hpp"

namespace crf {

__device__ __forceinline__ unsigned long long subset_key(unsigned long long seed, unsigned long long ctr, unsigned level,
                                                         unsigned i) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (ctr + 1ull) + (unsigned long long)level * 0xC2B2AE3D27D4EB4Full
                           + (unsigned long long)i * 0xD1B54A32D192ED03ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (z & ~0xFFFFFull) | (unsigned long long)i;        // low 20 bits = the point: keys are distinct (n <= 2^20)
}

constexpr int SUB_MAX_LEVELS = 8, SUB_NT = 1024;
struct SubsetJobs {
    int n[SUB_MAX_LEVELS];
    int s[SUB_MAX_LEVELS];
    long long* out[SUB_MAX_LEVELS];
    int* rank[SUB_MAX_LEVELS];        // may be null: rank[i] = position of point i in the subset, -1 for a point outside it
    int nlevels;
};

// One workgroup per level.  Radix select (8-bit digits from the top) of the s-th smallest key; it stops as soon as the selected bin is
// taken whole (two or three passes for uniform 64-bit keys).  Then flags -> block scan -> ascending indices.  A thread owns a
// contiguous chunk of `per` points and goes over it five times (the passes, the count, the write): up to SUB_CACHE keys per thread are
// hashed ONCE and their HIGH words kept in registers (CACHED: levels of up to 40 960 points -- the level-0 workgroup spent 60 us
// hashing every key five times on one CU).  The high word decides every pass down to bit 32 and the final membership test whenever
// the selection stopped there (it does for uniform keys: two or three passes); a pass below bit 32, and larger levels, recompute.
constexpr int SUB_CACHE = 40;
template <bool CACHED>
__device__ __forceinline__ void random_subset_level(const int n, const int s, long long* __restrict__ out, int* __restrict__ rank,
                                                    const unsigned long long seed, const unsigned long long ctr, const int level,
                                                    int* s_hist, int* s_sel, int* s_cnt) {
    const int tid = threadIdx.x;
    const int per = (n + SUB_NT - 1) / SUB_NT, lo = tid * per, hi = lo + per < n ? lo + per : n;   // a contiguous chunk per thread
    [[maybe_unused]] unsigned kc[CACHED ? SUB_CACHE : 1];
    if constexpr (CACHED) {
#pragma unroll
        for (int j = 0; j < SUB_CACHE; ++j) kc[j] = (unsigned)(subset_key(seed, ctr, level, lo + j) >> 32);      // (entries past the chunk: never looked at)
    }
    // visit(low, f): f(i, key) for every point of the chunk in ascending order; low (uniform) = f looks at bits below 32
    auto visit = [&](const bool low, auto&& f) {
        if constexpr (CACHED) {
            if (!low) {
                int cnt = hi - lo, lo2 = lo;
                asm volatile("" : "+v"(cnt), "+v"(lo2));        // (re-derived per pass: forty hoisted lane masks / point indices from the hashing loop would be kept alive -- and spilled)
#pragma unroll
                for (int j = 0; j < SUB_CACHE; ++j) {
                    if (j < cnt) f(lo2 + j, (unsigned long long)kc[j] << 32);
                    __builtin_amdgcn_sched_barrier(0);          // one point at a time: forty interleaved bodies spill at 128 registers
                }
                return;
            }
        }
        for (int i = lo; i < hi; ++i) f(i, subset_key(seed, ctr, level, i));
    };
    unsigned long long prefix = 0ull, mask = 0ull;             // keys with (key & mask) == prefix are still candidates
    int need = s;                                              // how many of the candidates belong to the subset
    bool whole = (s >= n);                                     // every candidate is in: nothing left to select
    for (int shift = 56; shift >= 0 && !whole; shift -= 8) {
        if (tid < 256) s_hist[tid] = 0;
        __syncthreads();
        visit(shift < 32, [&](int, unsigned long long k) {
            if ((k & mask) == prefix) atomicAdd(&s_hist[(int)((k >> shift) & 0xFF)], 1);
        });
        __syncthreads();
        {
            // the bin that holds the need-th candidate: an inclusive scan over the 256 bins by the first four wavefronts (one thread
            // walking the bins was 256 dependent LDS reads per pass: 12 of the kernel's 16 us per pass)
            const int h = tid < 256 ? s_hist[tid] : 0;
            int incl = h;
#pragma unroll
            for (int o = 1; o < WAVE; o <<= 1) {
                const int v = __shfl_up(incl, o, WAVE);
                if ((tid & 63) >= o) incl += v;
            }
            if (tid < 256 && (tid & 63) == 63) s_sel[2 + (tid >> 6)] = incl;
            __syncthreads();
            if (tid < 256) {
                for (int w = 0; w < (tid >> 6); ++w) incl += s_sel[2 + w];
                if (incl >= need && incl - h < need) {         // exactly one bin (need >= 1, the bins sum to >= need)
                    s_sel[0] = tid;
                    s_sel[1] = incl - h;
                }
            }
        }
        __syncthreads();
        const int d = s_sel[0], below = s_sel[1], in_bin = s_hist[d];
        need -= below;                                         // bins below d are taken whole
        prefix |= (unsigned long long)d << shift;
        mask |= 0xFFull << shift;
        whole = (need == in_bin);                              // the selected bin is taken whole too: the threshold is its top
        __syncthreads();
    }
    // subset = {k : (k & mask) <= prefix}  (equal prefix = the last selected bin, taken whole)
    int c = 0;
    const bool low = (mask & 0xFFFFFFFFull) != 0ull;
    visit(low, [&](int, unsigned long long k) { c += ((k & mask) <= prefix) ? 1 : 0; });
    s_cnt[tid] = c;
    __syncthreads();
    for (int o = 1; o < SUB_NT; o <<= 1) {                     // inclusive scan of the per-thread counts
        const int v = tid >= o ? s_cnt[tid - o] : 0;
        __syncthreads();
        s_cnt[tid] += v;
        __syncthreads();
    }
    int pos = s_cnt[tid] - c;
    visit(low, [&](int i, unsigned long long k) {
        const bool member = (k & mask) <= prefix;
        if (member && pos < s) out[pos] = i;
        if (rank != nullptr) rank[i] = (member && pos < s) ? pos : -1;
        pos += member ? 1 : 0;
    });
}
__global__ __launch_bounds__(SUB_NT) void random_subsets_kernel(const SubsetJobs jobs, unsigned long long seed,
                                                                const long long* __restrict__ counter) {
    const int level = blockIdx.x;
    const int n = jobs.n[level], s = jobs.s[level];
    if (s <= 0) return;
    const unsigned long long ctr = (unsigned long long)counter[0];
    __shared__ int s_hist[256];
    __shared__ int s_sel[6];                                   // selected digit, keys strictly below the selected bin so far | the scan's four wavefront totals
    __shared__ int s_cnt[SUB_NT];
    if ((n + SUB_NT - 1) / SUB_NT <= SUB_CACHE) random_subset_level<true>(n, s, jobs.out[level], jobs.rank[level], seed, ctr, level, s_hist, s_sel, s_cnt);
    else random_subset_level<false>(n, s, jobs.out[level], jobs.rank[level], seed, ctr, level, s_hist, s_sel, s_cnt);
}

// ------------------------------------------------------------------ up-index from the fine level's own neighbour table
// up_idx[b][i] = the subset member nearest to point i (datasets/semantic3d_dataset.py:524: knn_batch(sub_pos, pos, 1), the order
// (distance, subset position) of utils/nearest_neighbors).  The level's K-nearest table (distance order, the point itself first) already
// holds the answer for almost every point: the nearest member overall is the nearest member IN the table whenever one is there strictly
// closer than the table's last entry -- at ratio 4 and K = 16 all but (3/4)^16 = 1 % of the points.  Those go onto a list, and a second
// launch gives each of them a wavefront that scans the subset's positions (same arithmetic as csrc/knn.hip: singly-rounded x, y, z
// accumulation, key = distance bits << 32 | position).  Two launches per level instead of a grid build + search (round 4: 120 us of
// the 0.93 ms collate).
__device__ __forceinline__ float up_sqdist(float qx, float qy, float qz, float px, float py, float pz) {
    const float dx = qx - px, dy = qy - py, dz = qz - pz;
    return add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz));
}
constexpr unsigned long long UP_KEY_INF = ((unsigned long long)0x7f800000u << 32) | 0x7fffffffu;
// state: {list length, wavefronts of the scan launch that are done} -- zero before the first use, left zero by the scan launch
template <int KT>      // KT > 0: the table width is known (K == KT): index row and membership entries are all requested at once
__global__ __launch_bounds__(256) void upindex_table_kernel(const float* __restrict__ pos, const long long* __restrict__ nbr, int K,
                                                            const int* __restrict__ rank, int N, long long* __restrict__ out,
                                                            int* __restrict__ state, int* __restrict__ list) {
    const int b = blockIdx.y;
    const int i0 = blockIdx.x * 256 + threadIdx.x;
    const bool valid = i0 < N;
    const int i = valid ? i0 : N - 1;
    const float* P = pos + (size_t)b * N * 3;
    const float qx = P[3 * i], qy = P[3 * i + 1], qz = P[3 * i + 2];
    const long long* row = nbr + ((size_t)b * N + i) * K;
    unsigned long long best = UP_KEY_INF;
    float dlast = 0.f;
    auto offer = [&](int k, int j, int r) {
        if (r >= 0 || k == K - 1) {
            const float d = up_sqdist(qx, qy, qz, P[3 * j], P[3 * j + 1], P[3 * j + 2]);
            if (r >= 0) {
                const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)r;
                best = key < best ? key : best;
            }
            if (k == K - 1) dlast = d;
        }
    };
    if constexpr (KT > 0) {
        int jj[KT], rr[KT];
#pragma unroll
        for (int k = 0; k < KT; k += 2) {
            const longlong2 v = *reinterpret_cast<const longlong2*>(row + k);       // rows are K * 8 bytes: 16-byte aligned for even K
            jj[k] = (int)v.x; jj[k + 1] = (int)v.y;
        }
#pragma unroll
        for (int k = 0; k < KT; ++k) rr[k] = rank[jj[k]];
#pragma unroll
        for (int k = 0; k < KT; ++k) offer(k, jj[k], rr[k]);
    } else {
        for (int k = 0; k < K; ++k) {
            const int j = (int)row[k];
            offer(k, j, rank[j]);
        }
    }
    // settled when a member of the table is strictly closer than the table's last entry (every point outside the table is at least that far)
    const bool need = valid && !(best != UP_KEY_INF && __uint_as_float((unsigned)(best >> 32)) < dlast);
    if (valid && !need) out[(size_t)b * N + i] = (long long)(best & 0xffffffffull);
    const unsigned long long pending = __ballot(need);
    if (pending != 0ull) {                                       // one counter update per wavefront, positions by lane order
        const int lane = threadIdx.x & 63;
        int base = 0;
        if (lane == __ffsll((long long)pending) - 1) base = atomicAdd(state, __popcll(pending));
        base = __shfl(base, __ffsll((long long)pending) - 1, 64);
        if (need) list[base + __popcll(pending & ((1ull << lane) - 1ull))] = b * N + i;
    }
}
// one wavefront per listed point, a fixed grid that loops when a batch lists more than it has wavefronts
constexpr int UP_SCAN_WGS = 512;
__global__ __launch_bounds__(256) void upindex_scan_kernel(const float* __restrict__ pos, const float* __restrict__ sub_pos, int N, int S,
                                                           long long* __restrict__ out, int* __restrict__ state, const int* __restrict__ list) {
    const int lane = threadIdx.x & 63;
    const int count = state[0];
    for (int w = blockIdx.x * 4 + (threadIdx.x >> 6); w < count; w += 4 * (int)gridDim.x) {
        const int gi = list[w], b = gi / N;
        const float qx = pos[3 * (size_t)gi], qy = pos[3 * (size_t)gi + 1], qz = pos[3 * (size_t)gi + 2];
        const float* SP = sub_pos + (size_t)b * S * 3;
        unsigned long long mine = UP_KEY_INF;
        constexpr int UN = 8;                                    // positions of eight candidates per lane in flight
        for (int t0 = lane; t0 < S; t0 += 64 * UN) {
            float px[UN], py[UN], pz[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int t = t0 + 64 * u, tc = t < S ? t : S - 1;
                px[u] = SP[3 * tc]; py[u] = SP[3 * tc + 1]; pz[u] = SP[3 * tc + 2];
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int t = t0 + 64 * u;
                const float d = up_sqdist(qx, qy, qz, px[u], py[u], pz[u]);
                const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)t;
                mine = (t < S && key < mine) ? key : mine;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned lo32 = (unsigned)__shfl_xor((int)(unsigned)mine, o, 64), hi32 = (unsigned)__shfl_xor((int)(unsigned)(mine >> 32), o, 64);
            const unsigned long long other = ((unsigned long long)hi32 << 32) | lo32;
            mine = other < mine ? other : mine;
        }
        if (lane == 0) out[gi] = (long long)(mine & 0xffffffffull);
    }
    __syncthreads();                                             // (every wavefront has read the list length)
    if (threadIdx.x == 0) {                                      // the last workgroup out leaves the state zero for the next use
        const int done = atomicAdd(state + 1, 1);
        if (done + 1 == (int)gridDim.x) { state[0] = 0; state[1] = 0; }
    }
}

// ------------------------------------------------------------------ stable argsort of 30-bit codes, per cloud
constexpr int AS_BITS = 16, AS_SHIFT = 30 - AS_BITS, AS_BUCKETS = 1 << AS_BITS, AS_HUB = 256;

__global__ __launch_bounds__(256) void as_zero_kernel(int32_t* __restrict__ p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = 0;
}
__global__ __launch_bounds__(256) void as_count_kernel(const long long* __restrict__ code, int64_t N, int32_t* __restrict__ cnt) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= N) return;
    const int bucket = (int)((code[b * N + i] >> AS_SHIFT) & (AS_BUCKETS - 1));
    atomicAdd(&cnt[(int64_t)b * AS_BUCKETS + bucket], 1);
}
__global__ __launch_bounds__(256) void as_fill_kernel(const long long* __restrict__ code, int64_t N, const int32_t* __restrict__ ptrs,
                                                      int32_t* __restrict__ cursor, int32_t* __restrict__ tmp) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= N) return;
    const int64_t g = (int64_t)b * AS_BUCKETS + (int)((code[b * N + i] >> AS_SHIFT) & (AS_BUCKETS - 1));
    tmp[ptrs[g] + atomicAdd(&cursor[g], 1)] = (int32_t)i;      // global slot (clouds are consecutive bucket ranges)
}
// every slot of tmp: rank of its element inside the bucket by (code, index) -> its place in the stable order
__global__ __launch_bounds__(256) void as_rank_kernel(const long long* __restrict__ code, int64_t N, int64_t total,
                                                      const int32_t* __restrict__ ptrs, const int32_t* __restrict__ cnt,
                                                      const int32_t* __restrict__ tmp, long long* __restrict__ order) {
    const int64_t slot_raw = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = slot_raw < total;                       // lanes past the end stay in the wavefront: they load hub tiles below
    const int64_t slot = valid ? slot_raw : total - 1;
    const int b = (int)(slot / N);
    const int i = tmp[slot];
    const long long ci = code[b * N + i];
    const int64_t g = (int64_t)b * AS_BUCKETS + (int)((ci >> AS_SHIFT) & (AS_BUCKETS - 1));
    const int beg = ptrs[g], len = cnt[g];
    const unsigned long long key = ((unsigned long long)ci << 32) | (unsigned)i;     // (code, index): the stable order
    int rank = 0;
    const bool hub = valid && len > AS_HUB;
    if (valid && !hub) {
        for (int k = 0; k < len; ++k) {                        // buckets hold ~N / 65536 elements
            const int j = tmp[beg + k];
            const long long cj = code[b * N + j];
            rank += (cj < ci || (cj == ci && j < i)) ? 1 : 0;
        }
    }
    // Hub buckets (a cloud of duplicated / near-degenerate points: up to N elements in one bucket): the per-element scan above
    // would be len dependent load pairs per element (1.7e9 for 40 960 coincident points).  The wavefront ranks them together
    // instead: the slots of a bucket are consecutive, so its lanes share (beg, len); 64 keys of the bucket per coalesced tile
    // load, compared through lane broadcasts -- len / 64 loads and len compare steps per element, no dependent load.
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(hub);
    while (todo != 0ull) {
        const int leader = __ffsll((long long)todo) - 1;
        const int lbeg = __shfl(beg, leader, 64), llen = __shfl(len, leader, 64);
        const int lb = __shfl(b, leader, 64);
        const bool mine = hub && beg == lbeg;
        for (int t0 = 0; t0 < llen; t0 += 64) {
            const bool have = t0 + lane < llen;
            const int j = have ? tmp[lbeg + t0 + lane] : 0;
            const unsigned long long kj = have ? (((unsigned long long)code[(int64_t)lb * N + j] << 32) | (unsigned)j) : ~0ull;
            const int nt = llen - t0 < 64 ? llen - t0 : 64;
            for (int k = 0; k < nt; ++k) {
                const unsigned lo = __shfl((unsigned)kj, k, 64), hi = __shfl((unsigned)(kj >> 32), k, 64);
                rank += (mine && (((unsigned long long)hi << 32) | lo) < key) ? 1 : 0;
            }
        }
        todo &= ~__ballot(mine);
    }
    if (valid) order[(int64_t)beg + rank] = i;                            // bucket offsets count from the start of ALL clouds: cloud b's slots are [b N, (b + 1) N)
}


// Rows of SEVERAL [B, N, *] tensors picked by ONE index list in one launch: the Morton permutation of (pos, x, y, point_idx)
// and, at every scale, the subset of (pos, neighbor_idx) (datasets/semantic3d_dataset.py:524-526: pos[:, choice],
// neighbor_idx[:, choice]).  dst[b][s] = src[b][index[s]] (shared index) or src[b][index[b][s]] (per-cloud index); a row is
// row_bytes / 4 dwords; consecutive threads write consecutive dwords.
constexpr int GR_MAX = 8;
struct GatherRowsJobs {
    const uint32_t* src[GR_MAX];
    uint32_t* dst[GR_MAX];
    int words[GR_MAX];
    int njobs;
};
__global__ __launch_bounds__(256) void gather_rows_batched_kernel(const GatherRowsJobs t, const int64_t* __restrict__ index,
                                                                  int per_cloud, int64_t B, int64_t N, int64_t S) {
    const int j = blockIdx.y;
    const int w = t.words[j];
    const uint32_t* __restrict__ src = t.src[j];
    uint32_t* __restrict__ dst = t.dst[j];
    const int64_t total = B * S * w;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        const int64_t row = q / w;
        const int c = (int)(q - row * w);
        const int64_t b = row / S, sl = row - b * S;
        int64_t pick = per_cloud ? index[row] : index[sl];
        pick = pick < 0 ? 0 : (pick >= N ? N - 1 : pick);     // clamped: a bad index cannot fault (the tables are validated elsewhere)
        dst[q] = src[(b * N + pick) * w + c];
    }
}
}  // namespace crf

using namespace crf;

extern "C" size_t crfconv_upindex_workspace(int64_t B, int64_t N) {
    if (B < 1 || N < 1) return 0;
    return 256 + sizeof(int32_t) * (size_t)(B * N);              // {list length, done} | the list (worst case: every point)
}

extern "C" int crfconv_upindex_from_table(const float* pos, const float* sub_pos, const int64_t* neighbor_idx, const int32_t* rank,
                                          int64_t B, int64_t N, int K, int64_t S, int64_t* up_idx, void* workspace, size_t workspace_bytes,
                                          crf_stream_t stream) {
    CRF_REQUIRE(pos && sub_pos && neighbor_idx && rank && up_idx && workspace, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(B >= 1 && B <= 65535 && N >= 1 && B * N < ((int64_t)1 << 30) && S >= 1 && S <= N && K >= 1 && K <= 64 && K <= N, CRF_ERR_ARG,
                "bad shape B=%lld N=%lld K=%d S=%lld", (long long)B, (long long)N, K, (long long)S);
    CRF_REQUIRE(workspace_bytes >= crfconv_upindex_workspace(B, N) && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0, CRF_ERR_WORKSPACE,
                "workspace too small or misaligned");
    hipStream_t st = as_stream(stream);
    int* state = reinterpret_cast<int*>(workspace);
    int* list = reinterpret_cast<int*>(static_cast<char*>(workspace) + 256);
    const dim3 tgrid((unsigned)((N + 255) / 256), (unsigned)B);
    if (K == 16)
        hipLaunchKernelGGL(upindex_table_kernel<16>, tgrid, dim3(256), 0, st, pos, reinterpret_cast<const long long*>(neighbor_idx), K, rank, (int)N,
                           reinterpret_cast<long long*>(up_idx), state, list);
    else
        hipLaunchKernelGGL(upindex_table_kernel<0>, tgrid, dim3(256), 0, st, pos, reinterpret_cast<const long long*>(neighbor_idx), K, rank, (int)N,
                           reinterpret_cast<long long*>(up_idx), state, list);
    CRF_LAUNCH_CHECK();
    // the list holds ~(1 - S / N)^K of the points (1 % at ratio 4, K = 16): a fixed grid of 2048 wavefronts, looping beyond that
    hipLaunchKernelGGL(upindex_scan_kernel, dim3(UP_SCAN_WGS), dim3(256), 0, st, pos, sub_pos, (int)N, (int)S,
                       reinterpret_cast<long long*>(up_idx), state, list);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_random_subsets(const int* n, const int* s, int64_t* const* out, int32_t* const* rank, int nlevels, uint64_t seed,
                                      const int64_t* counter, crf_stream_t stream) {
    CRF_REQUIRE(n && s && out && counter, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(nlevels >= 1 && nlevels <= SUB_MAX_LEVELS, CRF_ERR_ARG, "nlevels=%d outside [1, %d]", nlevels, SUB_MAX_LEVELS);
    SubsetJobs jobs;
    jobs.nlevels = nlevels;
    for (int l = 0; l < SUB_MAX_LEVELS; ++l) {
        jobs.n[l] = 0; jobs.s[l] = 0; jobs.out[l] = nullptr; jobs.rank[l] = nullptr;
        if (l < nlevels) {
            CRF_REQUIRE(n[l] >= 1 && n[l] <= (1 << 20) && s[l] >= 0 && s[l] <= n[l] && (s[l] == 0 || out[l]), CRF_ERR_ARG,
                        "level %d: n=%d s=%d (1 <= n <= 2^20, 0 <= s <= n)", l, n[l], s[l]);
            jobs.n[l] = n[l]; jobs.s[l] = s[l]; jobs.out[l] = reinterpret_cast<long long*>(out[l]);
            jobs.rank[l] = rank ? rank[l] : nullptr;
        }
    }
    hipLaunchKernelGGL(random_subsets_kernel, dim3((unsigned)nlevels), dim3(SUB_NT), 0, as_stream(stream), jobs,
                       (unsigned long long)seed, reinterpret_cast<const long long*>(counter));
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" size_t crfconv_argsort_codes_workspace(int64_t B, int64_t N) {
    if (B < 1 || N < 1) return 0;
    const int64_t nb = B * AS_BUCKETS;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    return up(sizeof(int32_t) * (size_t)nb) * 3 + up(sizeof(int32_t) * scan_block_sums(nb)) + up(sizeof(int32_t) * (size_t)(B * N));
}

// order [B, N] int64 = per-cloud stable argsort of code [B, N] int64 (values < 2^30).
extern "C" int crfconv_argsort_codes(const int64_t* code, int64_t B, int64_t N, int64_t* order, void* workspace,
                                     size_t workspace_bytes, crf_stream_t stream) {
    CRF_REQUIRE(code && order && workspace, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(B >= 1 && N >= 1 && B * N < ((int64_t)1 << 31) && B * AS_BUCKETS < ((int64_t)1 << 31), CRF_ERR_ARG, "bad shape");
    CRF_REQUIRE(workspace_bytes >= crfconv_argsort_codes_workspace(B, N) && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0,
                CRF_ERR_WORKSPACE, "workspace too small or misaligned");
    hipStream_t st = as_stream(stream);
    const int64_t nb = B * AS_BUCKETS;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    char* w = static_cast<char*>(workspace);
    int32_t* cnt = reinterpret_cast<int32_t*>(w); w += up(sizeof(int32_t) * (size_t)nb);
    int32_t* ptrs = reinterpret_cast<int32_t*>(w); w += up(sizeof(int32_t) * (size_t)nb);
    int32_t* cursor = reinterpret_cast<int32_t*>(w); w += up(sizeof(int32_t) * (size_t)nb);
    int32_t* bsum = reinterpret_cast<int32_t*>(w); w += up(sizeof(int32_t) * scan_block_sums(nb));
    int32_t* tmp = reinterpret_cast<int32_t*>(w);
    const long long* c = reinterpret_cast<const long long*>(code);
    // cnt and cursor are adjacent-by-layout but not contiguous (ptrs sits between): two zero launches merged into one by
    // zeroing [cnt, cursor + nb) would also wipe nothing else, ptrs being rewritten by the scan below
    hipLaunchKernelGGL(as_zero_kernel, dim3(512), dim3(256), 0, st, cnt, (int64_t)(reinterpret_cast<char*>(cursor + nb) - reinterpret_cast<char*>(cnt)) / 4);
    CRF_LAUNCH_CHECK();
    const dim3 grid((unsigned)cdiv(N, 256), (unsigned)B);
    hipLaunchKernelGGL(as_count_kernel, grid, dim3(256), 0, st, c, N, cnt);
    CRF_LAUNCH_CHECK();
    exclusive_scan_i32(cnt, ptrs, nb, bsum, st);
    CRF_LAUNCH_CHECK();
    hipLaunchKernelGGL(as_fill_kernel, grid, dim3(256), 0, st, c, N, ptrs, cursor, tmp);
    CRF_LAUNCH_CHECK();
    hipLaunchKernelGGL(as_rank_kernel, dim3((unsigned)cdiv(B * N, 256)), dim3(256), 0, st, c, N, B * N, ptrs, cnt, tmp,
                       reinterpret_cast<long long*>(order));
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

// dst[j] [B, S, row_bytes[j]] = rows of src[j] [B, N, row_bytes[j]] picked by index ([S] shared by all clouds, or [B, S] when
// per_cloud != 0), for up to 8 tensors in one launch.  row_bytes multiples of 4; src / dst / row_bytes are host arrays.
extern "C" int crfconv_gather_rows_batched(const void* const* src, void* const* dst, const int* row_bytes, int njobs,
                                           const int64_t* index, int per_cloud, int64_t B, int64_t N, int64_t S, void* stream) {
    CRF_REQUIRE(njobs >= 1 && njobs <= crf::GR_MAX, CRF_ERR_ARG, "njobs=%d outside [1, %d]", njobs, crf::GR_MAX);
    CRF_REQUIRE(src && dst && row_bytes && index, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(B > 0 && N > 0 && S >= 0, CRF_ERR_ARG, "bad shape B=%lld N=%lld S=%lld", (long long)B, (long long)N, (long long)S);
    if (S == 0) return CRF_OK;
    crf::GatherRowsJobs t;
    int wmax = 1;
    for (int j = 0; j < crf::GR_MAX; ++j) {
        if (j < njobs) {
            CRF_REQUIRE(src[j] && dst[j] && row_bytes[j] >= 4 && row_bytes[j] % 4 == 0, CRF_ERR_ARG, "job %d: bad row", j);
            t.src[j] = reinterpret_cast<const uint32_t*>(src[j]);
            t.dst[j] = reinterpret_cast<uint32_t*>(dst[j]);
            t.words[j] = row_bytes[j] / 4;
            if (t.words[j] > wmax) wmax = t.words[j];
        } else {
            t.src[j] = nullptr; t.dst[j] = nullptr; t.words[j] = 1;
        }
    }
    t.njobs = njobs;
    int64_t gx = (B * S * wmax + 255) / 256;
    if (gx > 2048) gx = 2048;
    hipLaunchKernelGGL(crf::gather_rows_batched_kernel, dim3((unsigned)gx, (unsigned)njobs), dim3(256), 0, crf::as_stream(stream), t,
                       index, per_cloud, B, N, S);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}
