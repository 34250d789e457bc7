// Voxel-grid subsampling on gfx950 -- replaces the reference's unordered_map pass
// (utils/cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:5-106):
//   1. bounding box (ordered-int atomics)            2. voxel key per point (reference float ops,
//   3. stable radix sort of (key, point id)             every op separately rounded)
//   4. segment heads + scan -> voxel ordinal          5. one thread per voxel: sums in ARRIVAL order
// Because the sort is stable, each voxel's points are visited in ascending original index, i.e.
// the order the reference's single pass adds them, so barycentres and feature means are bit-exact.
// Output rows are in ascending voxel key (the reference: hash-map iteration order).
#include "common.hpp"

#include "radix_sort.hpp"

namespace crf {

struct GridDims {
    float org[3];
    float dl;
    unsigned long long NX, NY, NZ;
};

__device__ __forceinline__ unsigned ford(float f) {  // order-preserving float -> uint
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float funord(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__global__ __launch_bounds__(256) void gs_bbox_kernel(const float* __restrict__ pts, int64_t N,
                                                      unsigned* __restrict__ mnmx /*[6]: min xyz, max xyz*/) {
    float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = pts[3 * i + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float lo = mn[a], hi = mx[a];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo = fminf(lo, __shfl_xor(lo, o, WAVE));
            hi = fmaxf(hi, __shfl_xor(hi, o, WAVE));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&mnmx[a], ford(lo));
            atomicMax(&mnmx[3 + a], ford(hi));
        }
    }
}

__global__ void gs_init_kernel(unsigned* __restrict__ mnmx) {
    if (threadIdx.x < 6) mnmx[threadIdx.x] = threadIdx.x < 3 ? 0xffffffffu : 0u;
}

// origin / grid dims with the reference's float arithmetic (grid_subsampling.cpp:27-31)
__global__ void gs_dims_kernel(const unsigned* __restrict__ mnmx, float dl, GridDims* __restrict__ gd) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float inv = __fdiv_rn(1.0f, dl);
    GridDims g;
    g.dl = dl;
    unsigned long long n[3];
    for (int a = 0; a < 3; ++a) {
        const float mn = funord(mnmx[a]), mx = funord(mnmx[3 + a]);
        g.org[a] = mul_rn(floorf(mul_rn(mn, inv)), dl);
        n[a] = (unsigned long long)floorf(__fdiv_rn(sub_rn(mx, g.org[a]), dl)) + 1ull;
    }
    g.NX = n[0]; g.NY = n[1]; g.NZ = n[2];
    *gd = g;
}

__global__ __launch_bounds__(256) void gs_keys_kernel(const float* __restrict__ pts, int64_t N,
                                                      const GridDims* __restrict__ gd,
                                                      unsigned long long* __restrict__ keys,
                                                      uint32_t* __restrict__ ids) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const GridDims g = *gd;
    const unsigned long long ix = (unsigned long long)floorf(__fdiv_rn(sub_rn(pts[3 * i + 0], g.org[0]), g.dl));
    const unsigned long long iy = (unsigned long long)floorf(__fdiv_rn(sub_rn(pts[3 * i + 1], g.org[1]), g.dl));
    const unsigned long long iz = (unsigned long long)floorf(__fdiv_rn(sub_rn(pts[3 * i + 2], g.org[2]), g.dl));
    keys[i] = ix + g.NX * iy + g.NX * g.NY * iz;   // grid_subsampling.cpp:56
    ids[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void gs_heads_kernel(const unsigned long long* __restrict__ skeys, int64_t N,
                                                       int32_t* __restrict__ head) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= N) return;
    head[p] = (p == 0 || skeys[p] != skeys[p - 1]) ? 1 : 0;
}

__global__ __launch_bounds__(256) void gs_starts_kernel(const int32_t* __restrict__ head,
                                                        const int32_t* __restrict__ ordinal, int64_t N,
                                                        int32_t* __restrict__ seg_start,
                                                        int32_t* __restrict__ count_out) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= N) return;
    if (head[p]) seg_start[ordinal[p]] = (int32_t)p;
    if (p == N - 1) {
        const int32_t M = ordinal[p] + head[p];
        seg_start[M] = (int32_t)N;
        *count_out = M;
    }
}

// One thread per voxel.
__global__ __launch_bounds__(128) void gs_reduce_kernel(const float* __restrict__ pts,
                                                        const float* __restrict__ feats, int fdim,
                                                        const int32_t* __restrict__ classes, int ldim,
                                                        const uint32_t* __restrict__ sids,
                                                        const int32_t* __restrict__ seg_start,
                                                        const int32_t* __restrict__ count, int64_t cap,
                                                        float* __restrict__ out_pts,
                                                        float* __restrict__ out_feats,
                                                        int32_t* __restrict__ out_classes) {
    const int64_t v = (int64_t)blockIdx.x * 128 + threadIdx.x;
    const int64_t M = *count;
    if (v >= M || v >= cap) return;
    const int beg = seg_start[v], end = seg_start[v + 1];
    const int n = end - beg;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int p = beg; p < end; ++p) {
        const int64_t i = sids[p];
        sx = add_rn(sx, pts[3 * i]);
        sy = add_rn(sy, pts[3 * i + 1]);
        sz = add_rn(sz, pts[3 * i + 2]);
    }
    const float r = (float)(1.0 / (double)n);   // grid_subsampling.cpp:87: double reciprocal narrowed
    out_pts[3 * v + 0] = mul_rn(sx, r);
    out_pts[3 * v + 1] = mul_rn(sy, r);
    out_pts[3 * v + 2] = mul_rn(sz, r);
    if (feats) {
        const float fc = (float)n;
        for (int f = 0; f < fdim; ++f) {
            float acc = 0.f;
            for (int p = beg; p < end; ++p) acc = add_rn(acc, feats[(int64_t)sids[p] * fdim + f]);
            out_feats[v * fdim + f] = __fdiv_rn(acc, fc);
        }
    }
    if (classes) {
        for (int l = 0; l < ldim; ++l) {
            // mode of the column; ties -> smallest label value
            int best = 0, bestc = 0;
            for (int a = beg; a < end; ++a) {
                const int la = classes[(int64_t)sids[a] * ldim + l];
                bool seen = false;
                for (int b = beg; b < a; ++b)
                    if (classes[(int64_t)sids[b] * ldim + l] == la) { seen = true; break; }
                if (seen) continue;
                int c = 1;
                for (int b = a + 1; b < end; ++b) c += (classes[(int64_t)sids[b] * ldim + l] == la);
                if (c > bestc || (c == bestc && la < best)) { bestc = c; best = la; }
            }
            out_classes[v * ldim + l] = best;
        }
    }
}

static size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

struct GsLayout {
    size_t off_mnmx, off_gd, off_keys, off_skeys, off_ids, off_sids, off_head, off_ord, off_start, off_count,
        off_temp, temp_bytes, total;
};

static GsLayout gs_layout(int64_t N) {
    GsLayout L;
    size_t o = 0;
    L.off_mnmx = o;  o += al(sizeof(unsigned) * 6);
    L.off_gd = o;    o += al(sizeof(GridDims));
    L.off_keys = o;  o += al(sizeof(unsigned long long) * N);
    L.off_skeys = o; o += al(sizeof(unsigned long long) * N);
    L.off_ids = o;   o += al(sizeof(uint32_t) * N);
    L.off_sids = o;  o += al(sizeof(uint32_t) * N);
    L.off_head = o;  o += al(sizeof(int32_t) * N);
    L.off_ord = o;   o += al(sizeof(int32_t) * N);
    L.off_start = o; o += al(sizeof(int32_t) * (N + 1));
    L.off_count = o; o += al(sizeof(int32_t));
    // scratch of this library's radix sort (radix_sort.hpp) and of the scan over the voxel heads (scan.hpp)
    const size_t t1 = rsort_workspace(N), t2 = sizeof(int32_t) * scan_block_sums(N) + 256;
    L.temp_bytes = t1 > t2 ? t1 : t2;
    L.off_temp = o;  o += al(L.temp_bytes);
    L.total = o + 256;
    return L;
}

}  // namespace crf

using namespace crf;

extern "C" size_t crfconv_grid_subsample_dev_workspace(int64_t N, int fdim, int ldim) {
    (void)fdim; (void)ldim;
    if (N <= 0) return 0;
    return gs_layout(N).total;
}

extern "C" int64_t crfconv_grid_subsample_dev(const float* points, int64_t N, const float* feats, int fdim,
                                              const int32_t* classes, int ldim, float sampleDl,
                                              float* out_points, float* out_feats, int32_t* out_classes,
                                              int64_t cap, void* workspace, size_t workspace_bytes,
                                              crf_stream_t stream) {
    CRF_REQUIRE(points && out_points && workspace, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(N > 0 && N < ((int64_t)1 << 31), CRF_ERR_ARG, "N=%lld out of range", (long long)N);
    CRF_REQUIRE(sampleDl > 0.f, CRF_ERR_ARG, "sampleDl must be positive");
    CRF_REQUIRE((!feats || (fdim > 0 && out_feats)) && (!classes || (ldim > 0 && out_classes)), CRF_ERR_ARG,
                "feature / class buffers inconsistent");
    const GsLayout L = gs_layout(N);
    CRF_REQUIRE(workspace_bytes >= L.total, CRF_ERR_WORKSPACE, "grid_subsample workspace %zu < %zu",
                workspace_bytes, L.total);
    hipStream_t st = as_stream(stream);
    char* ws = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    unsigned* mnmx = reinterpret_cast<unsigned*>(ws + L.off_mnmx);
    GridDims* gd = reinterpret_cast<GridDims*>(ws + L.off_gd);
    auto* keys = reinterpret_cast<unsigned long long*>(ws + L.off_keys);
    auto* skeys = reinterpret_cast<unsigned long long*>(ws + L.off_skeys);
    uint32_t* ids = reinterpret_cast<uint32_t*>(ws + L.off_ids);
    uint32_t* sids = reinterpret_cast<uint32_t*>(ws + L.off_sids);
    int32_t* head = reinterpret_cast<int32_t*>(ws + L.off_head);
    int32_t* ord = reinterpret_cast<int32_t*>(ws + L.off_ord);
    int32_t* seg_start = reinterpret_cast<int32_t*>(ws + L.off_start);
    int32_t* count = reinterpret_cast<int32_t*>(ws + L.off_count);
    void* temp = ws + L.off_temp;

    hipLaunchKernelGGL(gs_init_kernel, dim3(1), dim3(64), 0, st, mnmx);
    CRF_LAUNCH_CHECK();
    const unsigned nb = (unsigned)(cdiv(N, 256) < 2048 ? cdiv(N, 256) : 2048);
    hipLaunchKernelGGL(gs_bbox_kernel, dim3(nb), dim3(256), 0, st, points, N, mnmx);
    CRF_LAUNCH_CHECK();
    hipLaunchKernelGGL(gs_dims_kernel, dim3(1), dim3(64), 0, st, mnmx, sampleDl, gd);
    CRF_LAUNCH_CHECK();
    const dim3 flat((unsigned)cdiv(N, 256));
    hipLaunchKernelGGL(gs_keys_kernel, flat, dim3(256), 0, st, points, N, gd, keys, ids);
    CRF_LAUNCH_CHECK();
    // stable sort of (voxel key, point id): points of a voxel stay in arrival order (grid_subsampling.cpp:58-63 sums in that order).
    // All eight digits are sorted: how many key bits are in use is device data (GridDims), and a pass whose digit is the same for
    // every point costs 32 B per point -- less than reading the answer back would
    if (rsort_pairs_u64(keys, ids, skeys, sids, N, 0, 64, temp, st) == 0) {
        auto* tk = keys; keys = skeys; skeys = tk;           // (eight passes end in the first pair of buffers)
        auto* ti = ids; ids = sids; sids = ti;
    }
    CRF_LAUNCH_CHECK();
    hipLaunchKernelGGL(gs_heads_kernel, flat, dim3(256), 0, st, skeys, N, head);
    CRF_LAUNCH_CHECK();
    exclusive_scan_i32(head, ord, N, reinterpret_cast<int32_t*>(temp), st);
    CRF_LAUNCH_CHECK();
    hipLaunchKernelGGL(gs_starts_kernel, flat, dim3(256), 0, st, head, ord, N, seg_start, count);
    CRF_LAUNCH_CHECK();
    // the reduce kernel reads M from device memory; launch for the worst case (M <= min(N, cap) rows written)
    const int64_t rows = N < cap ? N : cap;
    if (rows > 0) {
        hipLaunchKernelGGL(gs_reduce_kernel, dim3((unsigned)cdiv(rows, 128)), dim3(128), 0, st, points, feats,
                           fdim, classes, ldim, sids, seg_start, count, cap, out_points, out_feats, out_classes);
        CRF_LAUNCH_CHECK();
    }
    int32_t M = 0;
    CRF_HIP(hipMemcpyAsync(&M, count, sizeof(M), hipMemcpyDeviceToHost, st));
    CRF_HIP(hipStreamSynchronize(st));
    if ((int64_t)M > cap) {
        set_error("grid_subsample: %d voxels but capacity %lld", M, (long long)cap);
        return CRF_ERR_WORKSPACE;
    }
    return (int64_t)M;
}

extern "C" int64_t crfconv_grid_subsample(const float* points, int64_t N, const float* feats, int fdim,
                                          const int32_t* classes, int ldim, float sampleDl,
                                          float* out_points, float* out_feats, int32_t* out_classes,
                                          int64_t cap) {
    CRF_REQUIRE(points && out_points, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(N > 0, CRF_ERR_ARG, "empty cloud");
    if (cap > N) cap = N;
    const size_t wsb = crfconv_grid_subsample_dev_workspace(N, fdim, ldim);
    float *dp = nullptr, *df = nullptr, *dop = nullptr, *dof = nullptr;
    int32_t *dc = nullptr, *doc = nullptr;
    void* ws = nullptr;
    int64_t rc = CRF_OK;
    hipError_t e;
#define TRY(x) if ((e = (x)) != hipSuccess) { set_error("%s: %s", #x, hipGetErrorString(e)); rc = CRF_ERR_HIP; goto done; }
    TRY(hipMalloc(&dp, sizeof(float) * 3 * N));
    TRY(hipMalloc(&dop, sizeof(float) * 3 * cap));
    TRY(hipMemcpy(dp, points, sizeof(float) * 3 * N, hipMemcpyHostToDevice));
    if (feats) {
        TRY(hipMalloc(&df, sizeof(float) * fdim * N));
        TRY(hipMalloc(&dof, sizeof(float) * fdim * cap));
        TRY(hipMemcpy(df, feats, sizeof(float) * fdim * N, hipMemcpyHostToDevice));
    }
    if (classes) {
        TRY(hipMalloc(&dc, sizeof(int32_t) * ldim * N));
        TRY(hipMalloc(&doc, sizeof(int32_t) * ldim * cap));
        TRY(hipMemcpy(dc, classes, sizeof(int32_t) * ldim * N, hipMemcpyHostToDevice));
    }
    TRY(hipMalloc(&ws, wsb));
    rc = crfconv_grid_subsample_dev(dp, N, df, fdim, dc, ldim, sampleDl, dop, dof, doc, cap, ws, wsb, nullptr);
    if (rc < 0) goto done;
    TRY(hipMemcpy(out_points, dop, sizeof(float) * 3 * rc, hipMemcpyDeviceToHost));
    if (feats) TRY(hipMemcpy(out_feats, dof, sizeof(float) * fdim * rc, hipMemcpyDeviceToHost));
    if (classes) TRY(hipMemcpy(out_classes, doc, sizeof(int32_t) * ldim * rc, hipMemcpyDeviceToHost));
#undef TRY
done:
    if (dp) (void)hipFree(dp);
    if (df) (void)hipFree(df);
    if (dc) (void)hipFree(dc);
    if (dop) (void)hipFree(dop);
    if (dof) (void)hipFree(dof);
    if (doc) (void)hipFree(doc);
    if (ws) (void)hipFree(ws);
    return rc;
}
