// Micro-benchmark: what does a neighbour-row gather cost per CU on gfx950, by access shape?
// Every workgroup owns a tile of `rows` 32-byte rows (table resident in L2 / L1) and each lane gathers NG rows whose ids
// were drawn beforehand inside a window of +-W rows around its own.  Variants differ in lanes per row, bytes per lane,
// cache policy (plain / sc1), exec masking, and LDS instead of the vector-memory path.
//   hipcc --offload-arch=gfx950 -O3 -o gather_bench gather_bench.hip && ./gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int NG = 16;        // gathers per point per pass
constexpr int NT = 640;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
template <int AUX>
__device__ __forceinline__ float4 ldb(__amdgpu_buffer_rsrc_t r, int off, int soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, soff, AUX);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// MODE 0: one lane per row, two 16-B loads (LPP = 1)        MODE 1: two lanes per row, one 16-B load each (LPP = 2)
// MODE 2: four lanes per 64-B row (rows of 64 B)             MODE 3: LPP = 2, only rows flagged "far" are loaded (exec mask)
// MODE 4: LPP = 2 from LDS (tile staged once)                MODE 5: LPP = 1 from LDS
// MODE 6: LPP = 2, near rows from LDS, far rows from memory  MODE 7: as 3 but the far rows first (compacted per lane)
template <int MODE, int AUX>
__global__ __launch_bounds__(NT) void gather_kernel(const float* __restrict__ tab, const int* __restrict__ nbr, int rows_per_wg,
                                                    int total_rows, int passes, float* __restrict__ out, int farpct) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int LPP = (MODE == 0 || MODE == 5) ? 1 : (MODE == 2 ? 4 : 2);
    constexpr int ROWB = MODE == 2 ? 64 : 32;
    const int lane = threadIdx.x & 63, q = lane % LPP;
    const int pts_per_wg = NT / LPP;
    const int p = blockIdx.x * pts_per_wg + threadIdx.x / LPP;         // point (may exceed the tile: modulo below)
    const int row_base = blockIdx.x * rows_per_wg;
    const int me = row_base + (threadIdx.x / LPP) % rows_per_wg;
    const __amdgpu_buffer_rsrc_t tr = rsrc(tab, total_rows * ROWB);
    int j[NG];
    bool far[NG];
    int nfar = 0;
#pragma unroll
    for (int k = 0; k < NG; ++k) {
        const int v = nbr[(size_t)(blockIdx.x * NT / LPP + threadIdx.x / LPP) * NG + k];   // signed offset from own row
        int row = me + v;
        far[k] = row < row_base || row >= row_base + rows_per_wg;
        if (row < 0) row += total_rows;
        if (row >= total_rows) row -= total_rows;
        j[k] = row;
        nfar += far[k];
    }
    if (MODE == 7) {   // stable partition: far rows first
        int jj[NG]; int a = 0;
#pragma unroll
        for (int k = 0; k < NG; ++k) if (far[k]) jj[a++] = j[k];
#pragma unroll
        for (int k = 0; k < NG; ++k) if (k < nfar) j[k] = jj[k];
    }
    if (MODE == 4 || MODE == 5 || MODE == 6) {
        for (int t = threadIdx.x; t < rows_per_wg * 8; t += NT) lds[t] = tab[(size_t)row_base * 8 + t];
        __syncthreads();
    }
    int jl[NG];                                   // LDS slot of every row (far rows wrapped into the tile for MODE 4 / 5)
#pragma unroll
    for (int k = 0; k < NG; ++k) jl[k] = ((j[k] - row_base) % rows_per_wg + rows_per_wg) % rows_per_wg;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int it = 0; it < passes; ++it) {
#pragma unroll
        for (int k = 0; k < NG; ++k) asm volatile("" : "+v"(j[k]), "+v"(jl[k]));     // loads are not loop-invariant
        if constexpr (MODE == 0) {
            float4 g[NG][2];
#pragma unroll
            for (int k = 0; k < NG; ++k) { g[k][0] = ldb<AUX>(tr, j[k] * 32, 0); g[k][1] = ldb<AUX>(tr, j[k] * 32, 16); }
#pragma unroll
            for (int k = 0; k < NG; ++k) { acc.x += g[k][0].x + g[k][1].x; acc.y += g[k][0].y + g[k][1].w; }
        } else if constexpr (MODE == 1) {
            float4 g[NG];
#pragma unroll
            for (int k = 0; k < NG; ++k) g[k] = ldb<AUX>(tr, j[k] * 32 + q * 16, 0);
#pragma unroll
            for (int k = 0; k < NG; ++k) { acc.x += g[k].x; acc.y += g[k].w; }
        } else if constexpr (MODE == 2) {
            float4 g[NG];
#pragma unroll
            for (int k = 0; k < NG; ++k) g[k] = ldb<AUX>(tr, j[k] * 64 + q * 16, 0);
#pragma unroll
            for (int k = 0; k < NG; ++k) { acc.x += g[k].x; acc.y += g[k].w; }
        } else if constexpr (MODE == 3) {
            float4 g[NG];
#pragma unroll
            for (int k = 0; k < NG; ++k) g[k] = far[k] ? ldb<AUX>(tr, j[k] * 32 + q * 16, 0) : make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
            for (int k = 0; k < NG; ++k) { acc.x += g[k].x; acc.y += g[k].w; }
        } else if constexpr (MODE == 7) {
            float4 g[NG];
#pragma unroll
            for (int k = 0; k < NG; ++k) g[k] = k < nfar ? ldb<AUX>(tr, j[k] * 32 + q * 16, 0) : make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
            for (int k = 0; k < NG; ++k) { acc.x += g[k].x; acc.y += g[k].w; }
        } else if constexpr (MODE == 4) {
            const float4* l4 = reinterpret_cast<const float4*>(lds);
#pragma unroll
            for (int k = 0; k < NG; ++k) { const float4 g = l4[jl[k] * 2 + q]; acc.x += g.x; acc.y += g.w; }
        } else if constexpr (MODE == 5) {
            const float4* l4 = reinterpret_cast<const float4*>(lds);
#pragma unroll
            for (int k = 0; k < NG; ++k) { const float4 g0 = l4[jl[k] * 2], g1 = l4[jl[k] * 2 + 1]; acc.x += g0.x + g1.x; acc.y += g0.w + g1.w; }
        } else if constexpr (MODE == 6) {
            const float4* l4 = reinterpret_cast<const float4*>(lds);
            float4 g[NG];
#pragma unroll
            for (int k = 0; k < NG; ++k) g[k] = far[k] ? ldb<AUX>(tr, j[k] * 32 + q * 16, 0) : l4[jl[k] * 2 + q];
#pragma unroll
            for (int k = 0; k < NG; ++k) { acc.x += g[k].x; acc.y += g[k].w; }
        }
        // keep the loop from collapsing
        asm volatile("" : "+v"(acc.x), "+v"(acc.y));
    }
    if (p >= 0) out[(size_t)blockIdx.x * NT + threadIdx.x] = acc.x + acc.y;
}

template <int MODE, int AUX>
static float run(const float* tab, const int* nbr, int rows_per_wg, int total_rows, int passes, float* out, int grid, size_t lds_bytes) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((gather_kernel<MODE, AUX>), dim3(grid), dim3(NT), lds_bytes, 0, tab, nbr, rows_per_wg, total_rows, passes, out, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best * 1e3f;   // us
}

int main(int argc, char** argv) {
    const int grid = 256, passes = 200;
    const int W = argc > 1 ? atoi(argv[1]) : 300;              // neighbour window, +-W rows
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    printf("device %s, %d CUs, clock %d kHz; window +-%d rows, %d gathers/point/pass, %d passes\n", pr.name, pr.multiProcessorCount, pr.clockRate, W, NG, passes);
    for (int lpp : {1, 2, 4}) {
        const int pts = NT / lpp, rows_per_wg = pts, total_rows = rows_per_wg * grid;
        const int rowb = lpp == 4 ? 64 : 32;
        std::vector<float> h((size_t)total_rows * rowb / 4, 1.0f);
        std::vector<int> hn((size_t)grid * pts * NG);
        srand(7);
        for (auto& v : hn) { v = rand() % (2 * W + 1) - W; }
        float* tab; int* nbr; float* out;
        CK(hipMalloc(&tab, h.size() * 4)); CK(hipMalloc(&nbr, hn.size() * 4)); CK(hipMalloc(&out, (size_t)grid * NT * 4));
        CK(hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(nbr, hn.data(), hn.size() * 4, hipMemcpyHostToDevice));
        // fraction of far rows
        double farc = 0;
        for (int p = 0; p < pts; ++p) for (int k = 0; k < NG; ++k) { int r = p + hn[(size_t)p * NG + k]; farc += (r < 0 || r >= pts); }
        const double rows_moved = (double)grid * pts * NG * passes;
        auto report = [&](const char* name, float us, double frac = 1.0) {
            const double per_cu_clk = us * 1e-6 * 2.4e9 / passes;          // cycles per pass per CU (1 WG per CU)
            printf("  %-58s %8.1f us  %7.0f clk/pass/CU  %6.1f B/clk/CU  (%.2f row-requests/clk/CU)\n", name, us, per_cu_clk,
                   rows_moved * frac / grid / passes * rowb / per_cu_clk, (double)pts * NG * frac / per_cu_clk);
        };
        printf("lanes per row %d (%d-byte rows), %d points per workgroup, far fraction %.2f\n", lpp, rowb, pts, farc / (pts * NG));
        const size_t ldsb = (size_t)rows_per_wg * 32;
        if (lpp == 1) {
            report("global, 1 lane/row, 2 x 16 B, plain", run<0, 0>(tab, nbr, rows_per_wg, total_rows, passes, out, grid, 0));
            report("global, 1 lane/row, 2 x 16 B, sc1", run<0, 16>(tab, nbr, rows_per_wg, total_rows, passes, out, grid, 0));
            report("LDS, 1 lane/row, 2 x ds_read_b128", run<5, 0>(tab, nbr, rows_per_wg, total_rows, passes, out, grid, ldsb));
        } else if (lpp == 2) {
            report("global, 2 lanes/row, 16 B each, plain", run<1, 0>(tab, nbr, rows_per_wg, total_rows, passes, out, grid, 0));
            report("global, 2 lanes/row, 16 B each, sc1", run<1, 16>(tab, nbr, rows_per_wg, total_rows, passes, out, grid, 0));
            report("global, 2 lanes/row, far rows only (exec-masked), plain", run<3, 0>(tab, nbr, rows_per_wg, total_rows, passes, out, grid, 0), farc / (pts * NG));
            report("global, 2 lanes/row, far rows only, compacted first, plain", run<7, 0>(tab, nbr, rows_per_wg, total_rows, passes, out, grid, 0), farc / (pts * NG));
            report("global, 2 lanes/row, far rows only (exec-masked), sc1", run<3, 16>(tab, nbr, rows_per_wg, total_rows, passes, out, grid, 0), farc / (pts * NG));
            report("LDS, 2 lanes/row, ds_read_b128 (all rows, wrapped)", run<4, 0>(tab, nbr, rows_per_wg, total_rows, passes, out, grid, ldsb));
            report("near rows LDS + far rows global plain", run<6, 0>(tab, nbr, rows_per_wg, total_rows, passes, out, grid, ldsb));
            report("near rows LDS + far rows global sc1", run<6, 16>(tab, nbr, rows_per_wg, total_rows, passes, out, grid, ldsb));
        } else {
            report("global, 4 lanes/64-B row, plain", run<2, 0>(tab, nbr, rows_per_wg, total_rows, passes, out, grid, 0));
            report("global, 4 lanes/64-B row, sc1", run<2, 16>(tab, nbr, rows_per_wg, total_rows, passes, out, grid, 0));
        }
        hipFree(tab); hipFree(nbr); hipFree(out);
    }
    return 0;
}
