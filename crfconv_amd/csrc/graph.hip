// Neighbour-table plumbing: int64 per-cloud tables -> int32 global rows, and the reverse
// (source-major) CSR every backward scatter walks.  Integer work, HBM-bound; the sort and the
// scan are rocPRIM device primitives, the rest are flat coalesced kernels.
#include "common.hpp"

#include <rocprim/rocprim.hpp>

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
__global__ __launch_bounds__(256) void narrow_sorted_kernel(const int64_t* __restrict__ idx64, int64_t rows,
                                                            int64_t rows_per_cloud, int K, int sort_from,
                                                            int64_t n_src, int32_t* __restrict__ idx32,
                                                            uint16_t* __restrict__ idx16, int32_t* __restrict__ bad) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    const int64_t b = row / rows_per_cloud;
    int v[KT];
    int nbad = 0;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        int64_t x = k < K ? idx64[row * K + k] : n_src;        // padding sorts behind every real id
        if (k < K && (x < 0 || x >= n_src)) { ++nbad; x = x < 0 ? 0 : n_src - 1; }
        v[k] = (int)x;
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
#pragma unroll
    for (int k = 0; k < KT; ++k)
        if (k < K) {
            idx32[row * K + k] = (int32_t)(b * n_src + v[k]);
            if (idx16 != nullptr) idx16[row * K + k] = (uint16_t)v[k];
        }
}

// edge ids 0..E-1 and sort keys: the source row, or m_src for "no neighbour" entries (< 0), which
// therefore sort behind every real row and fall outside rev_ptr[0 .. m_src].
__global__ __launch_bounds__(256) void iota_keys_kernel(const int32_t* __restrict__ idx, int64_t n, int64_t m_src,
                                                        uint32_t* __restrict__ ids, uint32_t* __restrict__ keys) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    ids[t] = (uint32_t)t;
    const int32_t v = idx[t];
    keys[t] = (v < 0 || v >= m_src) ? (uint32_t)m_src : (uint32_t)v;
}

// sorted keys -> rev_ptr: ptr[v] = first position p with key[p] >= v, for v in [0, m_src].
__global__ __launch_bounds__(256) void boundaries_kernel(const uint32_t* __restrict__ keys,
                                                         int64_t E, int64_t m_src,
                                                         int32_t* __restrict__ ptr) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p > E) return;
    // position p closes every source id in (key[p-1], key[p]]  (key[-1] = -1, key[E] = m_src)
    const int64_t lo = p == 0 ? -1 : (int64_t)keys[p - 1];
    int64_t hi = p == E ? m_src : (int64_t)keys[p];
    if (hi > m_src) hi = m_src;
    for (int64_t v = lo + 1; v <= hi; ++v) ptr[v] = (int32_t)p;
}

static size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

static int key_bits(int64_t m_src) {   // keys take values 0 .. m_src (m_src = the "missing" bucket)
    int bits = 1;
    while (((int64_t)1 << bits) <= m_src) ++bits;
    return bits;
}

static size_t sort_temp_bytes(int64_t E, int64_t m_src) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                    (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)E, 0,
                                    key_bits(m_src), (hipStream_t)0);
    return bytes;
}

}  // namespace crf

using namespace crf;

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

extern "C" size_t crfconv_reverse_csr_workspace(int64_t E, int64_t m_src) {
    if (E <= 0 || m_src <= 0) return 0;
    // [keys_in E][keys_out E][vals_in E][sort temp]
    return 3 * align_up(sizeof(uint32_t) * (size_t)E) + align_up(sort_temp_bytes(E, m_src)) + 256;
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
    const size_t seg = align_up(sizeof(uint32_t) * (size_t)E);
    uint32_t* keys_in = reinterpret_cast<uint32_t*>(ws);
    uint32_t* keys_out = reinterpret_cast<uint32_t*>(ws + seg);
    uint32_t* vals_in = reinterpret_cast<uint32_t*>(ws + 2 * seg);
    void* temp = ws + 3 * seg;
    size_t temp_bytes = sort_temp_bytes(E, m_src);

    hipLaunchKernelGGL(iota_keys_kernel, dim3((unsigned)cdiv(E, 256)), dim3(256), 0, st, idx32, E, m_src, vals_in, keys_in);
    CRF_LAUNCH_CHECK();
    // stable LSD radix sort by source row: edge ids stay ascending inside each group
    CRF_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in,
                                      reinterpret_cast<uint32_t*>(rev_eid), (size_t)E, 0, key_bits(m_src), st));
    hipLaunchKernelGGL(boundaries_kernel, dim3((unsigned)cdiv(E + 1, 256)), dim3(256), 0, st, keys_out, E,
                       m_src, rev_ptr);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}
