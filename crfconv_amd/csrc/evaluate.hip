// The callers either side of the network, on the device so that a training / voting loop
// never leaves it:
//   * confusion matrix of utils/metrics.py:13-27 (runningScore._fast_hist / update), optionally with the arg-max of
//     the logits fused in (trainval.py:108: y_pred.max(dim=1)[1]);
//   * vote accumulator of trainval.py:170-190 (running mean of soft-max probabilities per cloud point) and the
//     re-projection arg-max of trainval.py:198-203;
//   * possibility sampler of datasets/semantic3d_dataset.py:423-460 (_get_random): seed = arg-min possibility,
//     crop = the num_points nearest points of the (jittered) seed, possibility += (1 - d / d_max)^2 * weight.
// Integer / byte work and streaming reductions: HBM-bound, no LDS tiling beyond block-local histograms.
#include "common.hpp"

#include "radix_sort.hpp"

namespace crf {

constexpr int EV_BLOCK = 256;
constexpr int HIST_LDS_CLASSES = 64;        // n_class^2 uint32 counters in LDS up to this many classes (16 KiB)

__device__ __forceinline__ int argmax_first(const float* __restrict__ row, int C) {
    float best = row[0];
    int arg = 0;
    for (int c = 1; c < C; ++c) {
        const float v = row[c];
        if (v > best) { best = v; arg = c; }        // strict: first maximum wins, like np.argmax
    }
    return arg;
}

// hist[t * n + p] += 1 for every row whose true label t passes the reference's mask
// (0 <= t < n and t != ignore, metrics.py:14).  pred comes from y_pred or, if logits != NULL, arg-max of the row.
template <bool LDS_HIST>
__global__ __launch_bounds__(EV_BLOCK) void confusion_kernel(const int64_t* __restrict__ y_true,
                                                             const int64_t* __restrict__ y_pred,
                                                             const float* __restrict__ logits, int64_t n_rows, int n,
                                                             int64_t ignore_index, int64_t label_shift,
                                                             unsigned long long* __restrict__ hist,
                                                             int32_t* __restrict__ bad) {
    __shared__ unsigned int s_hist[LDS_HIST ? HIST_LDS_CLASSES * HIST_LDS_CLASSES : 1];
    if constexpr (LDS_HIST) {
        for (int i = threadIdx.x; i < n * n; i += EV_BLOCK) s_hist[i] = 0u;
        __syncthreads();
    }
    int nbad = 0;
    for (int64_t r = (int64_t)blockIdx.x * EV_BLOCK + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * EV_BLOCK) {
        const int64_t t = y_true[r] - label_shift;
        if (t < 0 || t >= n || t == ignore_index) continue;
        const int64_t p = logits ? (int64_t)argmax_first(logits + r * n, n) : y_pred[r];
        if (p < 0 || p >= n) { ++nbad; continue; }       // np.bincount would widen the histogram and reshape() fail
        if constexpr (LDS_HIST) atomicAdd(&s_hist[(int)t * n + (int)p], 1u);
        else atomicAdd(&hist[t * n + p], 1ull);
    }
    if (nbad) atomicAdd(bad, nbad);
    if constexpr (LDS_HIST) {
        __syncthreads();
        for (int i = threadIdx.x; i < n * n; i += EV_BLOCK)
            if (s_hist[i]) atomicAdd(&hist[i], (unsigned long long)s_hist[i]);
    }
}

// test_probs[p_idx[r]] = smooth * test_probs[p_idx[r]] + (1 - smooth) * prob[r]   (float32, products and sum each
// rounded once -- numpy evaluates the expression of trainval.py:188 array-op by array-op, so no fused multiply-add).
// prob = probs[r] or soft-max of logits[r].  Rows of one call must be distinct points (a crop is a kNN result).
__global__ __launch_bounds__(EV_BLOCK) void vote_kernel(const float* __restrict__ probs,
                                                        const float* __restrict__ logits,
                                                        const int64_t* __restrict__ point_idx, int64_t n_rows, int C,
                                                        float smooth, float one_minus, float* __restrict__ test_probs,
                                                        int64_t n_cloud, int32_t* __restrict__ bad, int32_t* __restrict__ visits) {
    const int64_t r = (int64_t)blockIdx.x * EV_BLOCK + threadIdx.x;
    if (r >= n_rows) return;
    const int64_t p = point_idx[r];
    if (p < 0 || p >= n_cloud) { atomicAdd(bad, 1); return; }
    if (visits != nullptr) visits[p] += 1;               // (rows of one call are distinct points: no two threads share p)
    float* dst = test_probs + p * C;
    if (logits) {
        const float* row = logits + r * C;
        float mx = row[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, row[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(row[c] - mx);
        const float inv = 1.0f / se;
        for (int c = 0; c < C; ++c)
            dst[c] = add_rn(mul_rn(smooth, dst[c]), mul_rn(one_minus, expf(row[c] - mx) * inv));
    } else {
        const float* row = probs + r * C;
        for (int c = 0; c < C; ++c) dst[c] = add_rn(mul_rn(smooth, dst[c]), mul_rn(one_minus, row[c]));
    }
}

// preds[i] = argmax_c test_probs[proj_idx[i], c] + label_offset   (trainval.py:200-203)
__global__ __launch_bounds__(EV_BLOCK) void project_kernel(const float* __restrict__ test_probs,
                                                           const int64_t* __restrict__ proj_idx, int64_t n_proj, int C,
                                                           int64_t n_cloud, int label_offset,
                                                           uint8_t* __restrict__ preds, int32_t* __restrict__ bad) {
    const int64_t i = (int64_t)blockIdx.x * EV_BLOCK + threadIdx.x;
    if (i >= n_proj) return;
    const int64_t p = proj_idx[i];
    if (p < 0 || p >= n_cloud) { atomicAdd(bad, 1); preds[i] = 0; return; }
    preds[i] = (uint8_t)(argmax_first(test_probs + p * C, C) + label_offset);
}

// ------------------------------------------------------------------------------------------ possibility sampler
// arg-min with the first index on ties (np.argmin), two passes: per-block candidates, then one block.
struct MinIdx {
    double v;
    int64_t i;
};
__device__ __forceinline__ MinIdx min_first(MinIdx a, MinIdx b) {
    return (b.v < a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
__device__ __forceinline__ MinIdx block_min_first(MinIdx m) {
    __shared__ double s_v[EV_BLOCK / WAVE];
    __shared__ int64_t s_i[EV_BLOCK / WAVE];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        MinIdx other;
        other.v = __shfl_xor(m.v, o, WAVE);
        other.i = __shfl_xor(m.i, o, WAVE);
        m = min_first(m, other);
    }
    if ((threadIdx.x & 63) == 0) { s_v[threadIdx.x >> 6] = m.v; s_i[threadIdx.x >> 6] = m.i; }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int w = 1; w < EV_BLOCK / WAVE; ++w) m = min_first(m, MinIdx{s_v[w], s_i[w]});
    return m;       // valid on thread 0
}

__global__ __launch_bounds__(EV_BLOCK) void argmin_partial_kernel(const double* __restrict__ v, int64_t n,
                                                                  double* __restrict__ pv, int64_t* __restrict__ pi) {
    MinIdx m{1.0 / 0.0, INT64_MAX};
    for (int64_t i = (int64_t)blockIdx.x * EV_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * EV_BLOCK)
        m = min_first(m, MinIdx{v[i], i});
    m = block_min_first(m);
    if (threadIdx.x == 0) { pv[blockIdx.x] = m.v; pi[blockIdx.x] = m.i; }
}

__global__ __launch_bounds__(EV_BLOCK) void argmin_final_kernel(const double* __restrict__ pv,
                                                                const int64_t* __restrict__ pi, int nblk,
                                                                double* __restrict__ out_v, int64_t* __restrict__ out_i) {
    MinIdx m{1.0 / 0.0, INT64_MAX};
    for (int b = threadIdx.x; b < nblk; b += EV_BLOCK) m = min_first(m, MinIdx{pv[b], pi[b]});
    m = block_min_first(m);
    if (threadIdx.x == 0) { *out_v = m.v; *out_i = m.i; }
}

// pick point = float64(points[pick]) + noise (semantic3d_dataset.py:426-430)
__global__ void pick_point_kernel(const float* __restrict__ points, const int64_t* __restrict__ pick,
                                  const double* __restrict__ noise, double* __restrict__ center) {
    if (threadIdx.x < 3) center[threadIdx.x] = (double)points[*pick * 3 + threadIdx.x] + (noise ? noise[threadIdx.x] : 0.0);
}

// sort key of point i = bit pattern of the float64 squared distance to the seed (sklearn's KDTree holds the points
// as float64 and ranks by the reduced distance sum (x - c)^2); non-negative doubles order like their bit patterns.
__global__ __launch_bounds__(EV_BLOCK) void crop_keys_kernel(const float* __restrict__ points, int64_t n,
                                                             const double* __restrict__ center,
                                                             unsigned long long* __restrict__ keys,
                                                             unsigned int* __restrict__ ids) {
    const int64_t i = (int64_t)blockIdx.x * EV_BLOCK + threadIdx.x;
    if (i >= n) return;
    const double dx = (double)points[3 * i] - center[0], dy = (double)points[3 * i + 1] - center[1],
                 dz = (double)points[3 * i + 2] - center[2];
    const double d = dadd_rn(dadd_rn(dmul_rn(dx, dx), dmul_rn(dy, dy)), dmul_rn(dz, dz));
    keys[i] = (unsigned long long)__double_as_longlong(d);
    ids[i] = (unsigned int)i;
}

// float32 distances of the crop as the reference forms them for the possibility update
// (np.sum(np.square(points[q] - pick).astype(np.float32), axis=1), :448): squares in float64, rounded to float32,
// added left to right in float32; plus the block-wise maximum.
__global__ __launch_bounds__(EV_BLOCK) void crop_dist_kernel(const float* __restrict__ points,
                                                             const unsigned int* __restrict__ sel, int64_t k,
                                                             const double* __restrict__ center,
                                                             float* __restrict__ dist, float* __restrict__ pmax) {
    __shared__ float s_red[EV_BLOCK / WAVE];
    const int64_t t = (int64_t)blockIdx.x * EV_BLOCK + threadIdx.x;
    float d = 0.f;
    if (t < k) {
        const int64_t i = sel[t];
        const double dx = (double)points[3 * i] - center[0], dy = (double)points[3 * i + 1] - center[1],
                     dz = (double)points[3 * i + 2] - center[2];
        d = add_rn(add_rn((float)dmul_rn(dx, dx), (float)dmul_rn(dy, dy)), (float)dmul_rn(dz, dz));
        dist[t] = d;
    }
    float mx = d;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, WAVE));
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < EV_BLOCK / WAVE; ++w) mx = fmaxf(mx, s_red[w]);
        pmax[blockIdx.x] = mx;
    }
}

// possibility[q] += (1 - d / d_max)^2 * weight[label_to_idx(labels[q])]     (:449-450), and the crop's outputs:
// point_idx (int64), xyz centred on the seed in x and y (:436-437, float64 subtraction rounded to float32).
__global__ __launch_bounds__(EV_BLOCK) void crop_update_kernel(const float* __restrict__ points,
                                                               const unsigned int* __restrict__ sel,
                                                               const int64_t* __restrict__ perm, int64_t k,
                                                               const double* __restrict__ center,
                                                               const float* __restrict__ dist,
                                                               const float* __restrict__ pmax, int nblk,
                                                               const double* __restrict__ point_weight,
                                                               double* __restrict__ possibility,
                                                               int64_t* __restrict__ out_idx,
                                                               float* __restrict__ out_xyz) {
    const int64_t t = (int64_t)blockIdx.x * EV_BLOCK + threadIdx.x;
    if (t >= k) return;
    float dmax = pmax[0];
    for (int b = 1; b < nblk; ++b) dmax = fmaxf(dmax, pmax[b]);
    const int64_t src = perm ? perm[t] : t;            // output row t shows selected element perm[t] (the shuffle)
    const int64_t i = sel[src];
    const float u = sub_rn(1.0f, __fdiv_rn(dist[src], dmax));
    const float sq = mul_rn(u, u);
    // the weights are float64 for train / val (class_weight array) and the python int 1 for test (float32 result)
    const double delta = point_weight ? dmul_rn((double)sq, point_weight[i]) : (double)sq;
    possibility[i] += delta;                            // rows of a crop are distinct points
    out_idx[t] = i;
    out_xyz[3 * t + 0] = (float)((double)points[3 * i + 0] - center[0]);
    out_xyz[3 * t + 1] = (float)((double)points[3 * i + 1] - center[1]);
    out_xyz[3 * t + 2] = points[3 * i + 2];
}

static size_t ev_align(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t crop_sort_temp(int64_t n) { return rsort_workspace(n); }      // this library's radix sort (radix_sort.hpp)

constexpr int ARGMIN_BLOCKS = 1024;

}  // namespace crf

using namespace crf;

extern "C" int crfconv_confusion_accumulate(const int64_t* y_true, const int64_t* y_pred, const float* logits,
                                            int64_t n_rows, int n_class, int64_t ignore_index, int64_t label_shift,
                                            int64_t* hist, int32_t* bad_count, crf_stream_t stream) {
    CRF_REQUIRE(y_true && hist && bad_count && (y_pred || logits), CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(n_class >= 1 && n_class <= 4096, CRF_ERR_ARG, "n_class=%d out of range", n_class);
    if (n_rows <= 0) return CRF_OK;
    int64_t blocks = cdiv(n_rows, EV_BLOCK);
    if (blocks > 2048) blocks = 2048;
    auto* h = reinterpret_cast<unsigned long long*>(hist);
    if (n_class <= HIST_LDS_CLASSES)
        hipLaunchKernelGGL(confusion_kernel<true>, dim3((unsigned)blocks), dim3(EV_BLOCK), 0, as_stream(stream), y_true,
                           y_pred, logits, n_rows, n_class, ignore_index, label_shift, h, bad_count);
    else
        hipLaunchKernelGGL(confusion_kernel<false>, dim3((unsigned)blocks), dim3(EV_BLOCK), 0, as_stream(stream), y_true,
                           y_pred, logits, n_rows, n_class, ignore_index, label_shift, h, bad_count);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

static int vote_accumulate_impl(const float* probs, const float* logits, const int64_t* point_idx, int64_t n_rows, int C, double smooth,
                                float* test_probs, int64_t n_cloud, int32_t* bad_count, int32_t* visits, crf_stream_t stream);

extern "C" int crfconv_vote_accumulate(const float* probs, const float* logits, const int64_t* point_idx,
                                       int64_t n_rows, int C, double smooth, float* test_probs, int64_t n_cloud,
                                       int32_t* bad_count, crf_stream_t stream) {
    return vote_accumulate_impl(probs, logits, point_idx, n_rows, C, smooth, test_probs, n_cloud, bad_count, nullptr, stream);
}

// The same, counting the updates of every point in visits [n_cloud] (int32): what crfconv_vote_fold needs to merge tables that
// were accumulated apart (crops of one scene sharded over ranks).
extern "C" int crfconv_vote_accumulate_counted(const float* probs, const float* logits, const int64_t* point_idx,
                                               int64_t n_rows, int C, double smooth, float* test_probs, int64_t n_cloud,
                                               int32_t* bad_count, int32_t* visits, crf_stream_t stream) {
    CRF_REQUIRE(visits, CRF_ERR_ARG, "null pointer");
    return vote_accumulate_impl(probs, logits, point_idx, n_rows, C, smooth, test_probs, n_cloud, bad_count, visits, stream);
}

// acc <- the table that results from applying `later`'s updates AFTER acc's:  a running mean v <- s v + (1 - s) p applied n times
// scales what was there by s^n, so  acc[p] = acc[p] s^later_visits[p] + later[p]  (s^n by n rounded multiplications, as the
// sequential updates round), acc_visits += later_visits.  Merging per-rank tables in rank order gives the table of ONE accumulator
// that saw rank 0's crops first, then rank 1's, ... -- the reference's order-dependent update (trainval.py:188-189) in that order.
__global__ __launch_bounds__(EV_BLOCK) void vote_fold_kernel(float* __restrict__ acc, int32_t* __restrict__ acc_visits,
                                                             const float* __restrict__ later, const int32_t* __restrict__ later_visits,
                                                             int64_t n, int C, float smooth) {
    const int64_t p = (int64_t)blockIdx.x * EV_BLOCK + threadIdx.x;
    if (p >= n) return;
    const int nv = later_visits[p];
    float d = 1.0f;
    for (int i = 0; i < nv; ++i) d = mul_rn(d, smooth);
    for (int c = 0; c < C; ++c) acc[p * C + c] = add_rn(mul_rn(acc[p * C + c], d), later[p * C + c]);
    acc_visits[p] += nv;
}

extern "C" int crfconv_vote_fold(float* acc, int32_t* acc_visits, const float* later, const int32_t* later_visits, int64_t n, int C,
                                 double smooth, crf_stream_t stream) {
    CRF_REQUIRE(acc && acc_visits && later && later_visits, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(C >= 1 && n >= 0, CRF_ERR_ARG, "C=%d n=%lld invalid", C, (long long)n);
    if (n == 0) return CRF_OK;
    hipLaunchKernelGGL(vote_fold_kernel, dim3((unsigned)cdiv(n, EV_BLOCK)), dim3(EV_BLOCK), 0, as_stream(stream), acc, acc_visits, later,
                       later_visits, n, C, (float)smooth);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

static int vote_accumulate_impl(const float* probs, const float* logits, const int64_t* point_idx, int64_t n_rows, int C, double smooth,
                                float* test_probs, int64_t n_cloud, int32_t* bad_count, int32_t* visits, crf_stream_t stream) {
    CRF_REQUIRE((probs || logits) && point_idx && test_probs && bad_count, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(C >= 1 && n_cloud > 0, CRF_ERR_ARG, "C=%d n_cloud=%lld invalid", C, (long long)n_cloud);
    if (n_rows <= 0) return CRF_OK;
    // (1 - test_smooth) is formed in float64 by the interpreter and rounded when it meets the float32 array
    // both coefficients are python floats (float64) that numpy rounds to float32 when they meet the float32 table
    const float one_minus = (float)(1.0 - smooth);
    hipLaunchKernelGGL(vote_kernel, dim3((unsigned)cdiv(n_rows, EV_BLOCK)), dim3(EV_BLOCK), 0, as_stream(stream), probs,
                       logits, point_idx, n_rows, C, (float)smooth, one_minus, test_probs, n_cloud, bad_count, visits);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_vote_project(const float* test_probs, const int64_t* proj_idx, int64_t n_proj, int C,
                                    int64_t n_cloud, int label_offset, uint8_t* preds, int32_t* bad_count,
                                    crf_stream_t stream) {
    CRF_REQUIRE(test_probs && proj_idx && preds && bad_count, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(C >= 1 && C + label_offset <= 256 && n_cloud > 0, CRF_ERR_ARG, "C=%d offset=%d invalid", C, label_offset);
    if (n_proj <= 0) return CRF_OK;
    hipLaunchKernelGGL(project_kernel, dim3((unsigned)cdiv(n_proj, EV_BLOCK)), dim3(EV_BLOCK), 0, as_stream(stream),
                       test_probs, proj_idx, n_proj, C, n_cloud, label_offset, preds, bad_count);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" size_t crfconv_argmin_workspace(void) { return ev_align(ARGMIN_BLOCKS * 8) * 2; }

extern "C" int crfconv_argmin_f64(const double* values, int64_t n, double* out_value, int64_t* out_index,
                                  void* workspace, size_t workspace_bytes, crf_stream_t stream) {
    CRF_REQUIRE(values && out_value && out_index && workspace, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(n > 0, CRF_ERR_ARG, "empty array");
    CRF_REQUIRE(workspace_bytes >= crfconv_argmin_workspace(), CRF_ERR_WORKSPACE, "argmin workspace too small");
    double* pv = reinterpret_cast<double*>(workspace);
    int64_t* pi = reinterpret_cast<int64_t*>(reinterpret_cast<char*>(workspace) + ev_align(ARGMIN_BLOCKS * 8));
    int64_t blocks = cdiv(n, EV_BLOCK);
    if (blocks > ARGMIN_BLOCKS) blocks = ARGMIN_BLOCKS;
    hipLaunchKernelGGL(argmin_partial_kernel, dim3((unsigned)blocks), dim3(EV_BLOCK), 0, as_stream(stream), values, n, pv, pi);
    CRF_LAUNCH_CHECK();
    hipLaunchKernelGGL(argmin_final_kernel, dim3(1), dim3(EV_BLOCK), 0, as_stream(stream), pv, pi, (int)blocks, out_value,
                       out_index);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" size_t crfconv_possibility_crop_workspace(int64_t n, int64_t k) {
    if (n <= 0 || k <= 0) return 0;
    // [keys_in n u64][keys_out n u64][ids_in n u32][ids_out n u32][dist k f32][pmax blocks f32][center 3 f64][sort temp]
    return 2 * ev_align(8 * (size_t)n) + 2 * ev_align(4 * (size_t)n) + ev_align(4 * (size_t)k) +
           ev_align(4 * (size_t)cdiv(k, EV_BLOCK)) + 256 + ev_align(crop_sort_temp(n)) + 256;
}

extern "C" int crfconv_possibility_crop(const float* points, int64_t n, int64_t k, const int64_t* pick_index,
                                        const double* noise, const int64_t* perm, const double* point_weight,
                                        double* possibility, int64_t* out_idx, float* out_xyz, double* out_center,
                                        void* workspace, size_t workspace_bytes, crf_stream_t stream) {
    CRF_REQUIRE(points && pick_index && possibility && out_idx && out_xyz && workspace, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(n > 0 && k > 0 && k <= n && n < ((int64_t)1 << 32), CRF_ERR_ARG, "n=%lld k=%lld invalid", (long long)n,
                (long long)k);
    CRF_REQUIRE(workspace_bytes >= crfconv_possibility_crop_workspace(n, k), CRF_ERR_WORKSPACE,
                "possibility_crop workspace %zu < %zu", workspace_bytes, crfconv_possibility_crop_workspace(n, k));
    hipStream_t st = as_stream(stream);
    char* ws = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    auto* keys_in = reinterpret_cast<unsigned long long*>(ws);   ws += ev_align(8 * (size_t)n);
    auto* keys_out = reinterpret_cast<unsigned long long*>(ws);  ws += ev_align(8 * (size_t)n);
    auto* ids_in = reinterpret_cast<unsigned int*>(ws);          ws += ev_align(4 * (size_t)n);
    auto* ids_out = reinterpret_cast<unsigned int*>(ws);         ws += ev_align(4 * (size_t)n);
    auto* dist = reinterpret_cast<float*>(ws);                   ws += ev_align(4 * (size_t)k);
    const int nblk = (int)cdiv(k, EV_BLOCK);
    auto* pmax = reinterpret_cast<float*>(ws);                   ws += ev_align(4 * (size_t)nblk);
    auto* center = reinterpret_cast<double*>(ws);                ws += 256;
    size_t temp_bytes = crop_sort_temp(n);

    hipLaunchKernelGGL(pick_point_kernel, dim3(1), dim3(64), 0, st, points, pick_index, noise, center);
    CRF_LAUNCH_CHECK();
    hipLaunchKernelGGL(crop_keys_kernel, dim3((unsigned)cdiv(n, EV_BLOCK)), dim3(EV_BLOCK), 0, st, points, n, center,
                       keys_in, ids_in);
    CRF_LAUNCH_CHECK();
    // stable LSD radix sort: equal distances keep ascending point order (the KD-tree's order on ties is unspecified)
    (void)temp_bytes;
    if (rsort_pairs_u64(keys_in, ids_in, keys_out, ids_out, n, 0, 64, ws, st) == 0) ids_out = ids_in;      // float64 keys: all eight digits
    CRF_LAUNCH_CHECK();
    hipLaunchKernelGGL(crop_dist_kernel, dim3((unsigned)nblk), dim3(EV_BLOCK), 0, st, points, ids_out, k, center, dist, pmax);
    CRF_LAUNCH_CHECK();
    hipLaunchKernelGGL(crop_update_kernel, dim3((unsigned)nblk), dim3(EV_BLOCK), 0, st, points, ids_out, perm, k, center,
                       dist, pmax, nblk, point_weight, possibility, out_idx, out_xyz);
    CRF_LAUNCH_CHECK();
    if (out_center) CRF_HIP(hipMemcpyAsync(out_center, center, 3 * sizeof(double), hipMemcpyDeviceToDevice, st));
    return CRF_OK;
}
