// Weighted softmax cross-entropy with an ignored label (trainval.py:101-104: F.cross_entropy(y_pred, y - 1,
// weight=class_weights, ignore_index=...), mean over the weights of the counted rows).
//   loss = sum_r w[t_r] (lse_r - z[r, t_r]) / sum_r w[t_r]           over rows with t_r != ignore
//   dz[r, c] = g * w[t_r] (exp(z[r,c] - lse_r) - [c == t_r]) / sum w
// Forward: one thread per row (rows of adjacent lanes are adjacent in memory, so a wavefront still reads whole
// cache lines), per-block float64 partials, a last single-block kernel folds them in fixed order.
// The framework's nll_loss kernels reduce on ONE workgroup (126 us forward + 126 us backward at 164 k rows).
#include "common.hpp"
#include "gridsync.hpp"

namespace crf {

constexpr int CE_BLOCK = 256;

__global__ __launch_bounds__(CE_BLOCK) void ce_fwd_kernel(const float* __restrict__ z,
                                                          const int64_t* __restrict__ target,
                                                          const float* __restrict__ weight, int64_t m, int C,
                                                          int64_t ignore_index, int64_t label_shift,
                                                          float* __restrict__ lse,
                                                          double* __restrict__ partial, unsigned* __restrict__ ticket,
                                                          double* __restrict__ sums, float* __restrict__ loss) {
    __shared__ double s_red[3][CE_BLOCK / WAVE];
    __shared__ int s_flag;
    const int64_t r = (int64_t)blockIdx.x * CE_BLOCK + threadIdx.x;
    double num = 0.0, den = 0.0, bad = 0.0;
    if (r < m) {
        const float* row = z + r * C;
        float mx = row[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, row[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(row[c] - mx);
        const float l = mx + logf(se);
        lse[r] = l;
        const int64_t t = target[r] - label_shift;
        if (t != ignore_index) {
            if (t >= 0 && t < C) {
                const float w = weight ? weight[t] : 1.f;
                num = (double)w * (double)(l - row[t]);
                den = (double)w;
            } else {
                bad = 1.0;                                        // label outside [0, C): counted, not used
            }
        }
    }
    double v[3] = {num, den, bad};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[i] += __shfl_xor(v[i], o, WAVE);
        if ((threadIdx.x & 63) == 0) s_red[i][threadIdx.x >> 6] = v[i];
    }
    __syncthreads();
    typedef unsigned int ce_u32x2 __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t pr = make_rsrc(partial, (int)gridDim.x * 3 * 8);
    if (threadIdx.x < 3) {
        double a = 0.0;
        for (int w = 0; w < CE_BLOCK / WAVE; ++w) a += s_red[threadIdx.x][w];
        // write-through: with a ticket the launch's last workgroup (another CU, maybe another XCD) reads the rows below
        const unsigned long long bits = (unsigned long long)__double_as_longlong(a);
        const ce_u32x2 u = {(unsigned)bits, (unsigned)(bits >> 32)};
        __builtin_amdgcn_raw_buffer_store_b64(u, pr, ((int)blockIdx.x * 3 + (int)threadIdx.x) * 8, 0, 16);
    }
    // "the last workgroup finishes" (gridsync.hpp): ce_finalize_kernel's arithmetic, in its order, without its launch
    if (ticket == nullptr || !last_workgroup(ticket, gridDim.x, &s_flag)) return;
    double t[3] = {0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < (int)gridDim.x; b += CE_BLOCK)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const ce_u32x2 w2 = __builtin_amdgcn_raw_buffer_load_b64(pr, (b * 3 + i) * 8, 0, 16);
            t[i] += __longlong_as_double((long long)(((unsigned long long)w2.y << 32) | w2.x));
        }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t[i] += __shfl_xor(t[i], o, WAVE);
        if ((threadIdx.x & 63) == 0) s_red[i][threadIdx.x >> 6] = t[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a[3];
        for (int i = 0; i < 3; ++i) a[i] = s_red[i][0] + s_red[i][1] + s_red[i][2] + s_red[i][3];
        sums[0] = a[0]; sums[1] = a[1]; sums[2] = a[2];
        *loss = (float)(a[0] / a[1]);
    }
}

// sums = {sum w nll, sum w, #bad labels}; loss = sums[0] / sums[1] (nan when nothing is counted, like the framework)
__global__ __launch_bounds__(256) void ce_finalize_kernel(const double* __restrict__ partial, int64_t nblk,
                                                          double* __restrict__ sums, float* __restrict__ loss) {
    __shared__ double s_red[3][256 / WAVE];
    double v[3] = {0.0, 0.0, 0.0};
    for (int64_t b = threadIdx.x; b < nblk; b += 256)
#pragma unroll
        for (int i = 0; i < 3; ++i) v[i] += partial[b * 3 + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[i] += __shfl_xor(v[i], o, WAVE);
        if ((threadIdx.x & 63) == 0) s_red[i][threadIdx.x >> 6] = v[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a[3];
        for (int i = 0; i < 3; ++i) a[i] = s_red[i][0] + s_red[i][1] + s_red[i][2] + s_red[i][3];
        sums[0] = a[0]; sums[1] = a[1]; sums[2] = a[2];
        *loss = (float)(a[0] / a[1]);
    }
}

// one thread per logit
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ z, const int64_t* __restrict__ target,
                                                     const float* __restrict__ weight, const float* __restrict__ lse,
                                                     const double* __restrict__ sums, const float* __restrict__ gloss,
                                                     int64_t m, int C, int64_t ignore_index, int64_t label_shift,
                                                     float* __restrict__ dz) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m * C) return;
    const int64_t r = i / C;
    const int c = (int)(i - r * C);
    const int64_t t = target[r] - label_shift;
    float out = 0.f;
    if (t != ignore_index && t >= 0 && t < C) {
        const float w = weight ? weight[t] : 1.f;
        const float scale = (float)((double)gloss[0] * (double)w / sums[1]);
        out = scale * (expf(z[i] - lse[r]) - (c == t ? 1.f : 0.f));
    }
    dz[i] = out;
}

}  // namespace crf

using namespace crf;

extern "C" size_t crfconv_softmax_ce_workspace(int64_t m) {
    return m > 0 ? sizeof(double) * 3 * (size_t)cdiv(m, CE_BLOCK) + 256 : 0;
}

extern "C" int crfconv_softmax_ce_forward(const float* logits, const int64_t* target, const float* weight,
                                          int64_t m, int C, int64_t ignore_index, int64_t label_shift, float* lse,
                                          double* sums, float* loss, void* workspace, size_t workspace_bytes,
                                          unsigned* ticket, crf_stream_t stream) {
    CRF_REQUIRE(logits && target && lse && sums && loss && workspace, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(m > 0 && C >= 1 && C <= 4096 && m * C < ((int64_t)1 << 40), CRF_ERR_ARG, "m=%lld C=%d out of range",
                (long long)m, C);
    CRF_REQUIRE(workspace_bytes >= crfconv_softmax_ce_workspace(m), CRF_ERR_WORKSPACE, "workspace too small");
    double* partial = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const int64_t nblk = cdiv(m, CE_BLOCK);
    hipStream_t st = as_stream(stream);
    if (nblk * 3 * 8 >= ((int64_t)1 << 31)) ticket = nullptr;             // (the partial rows are addressed through a 32-bit buffer descriptor)
    hipLaunchKernelGGL(ce_fwd_kernel, dim3((unsigned)nblk), dim3(CE_BLOCK), 0, st, logits, target, weight, m, C,
                       ignore_index, label_shift, lse, partial, ticket, sums, loss);
    CRF_LAUNCH_CHECK();
    if (ticket == nullptr) {                                               // no ticket words: the fold as a launch of its own
        hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, st, partial, nblk, sums, loss);
        CRF_LAUNCH_CHECK();
    }
    return CRF_OK;
}

extern "C" int crfconv_softmax_ce_backward(const float* logits, const int64_t* target, const float* weight,
                                           const float* lse, const double* sums, const float* grad_loss, int64_t m,
                                           int C, int64_t ignore_index, int64_t label_shift, float* dlogits,
                                           crf_stream_t stream) {
    CRF_REQUIRE(logits && target && lse && sums && grad_loss && dlogits, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(m > 0 && C >= 1 && C <= 4096 && m * C < ((int64_t)1 << 40), CRF_ERR_ARG, "m=%lld C=%d out of range",
                (long long)m, C);
    hipLaunchKernelGGL(ce_bwd_kernel, dim3((unsigned)cdiv(m * C, 256)), dim3(256), 0, as_stream(stream), logits,
                       target, weight, lse, sums, grad_loss, m, C, ignore_index, label_shift, dlogits);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

// ====================================================================== optimizer step (trainval.py:69-72, 105)
// torch.optim.SGD(lr, momentum, dampening, weight_decay, nesterov) over ONE flat parameter vector:
//   g = grad + wd * p;  buf = first ? g : mu * buf + (1 - dampening) * g;  g = nesterov ? g + mu * buf : buf;  p -= lr * g
// (momentum == 0: p -= lr * g, buf untouched).  One launch for the whole model instead of ~15 multi-tensor ones.
namespace crf {
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                  int64_t n, float lr, float mu, float damp, float wd, int nesterov,
                                                  int first) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float w = p[i];
        float d = fmaf(wd, w, g[i]);
        if (mu != 0.f) {
            const float b = first ? d : fmaf(mu, buf[i], (1.f - damp) * d);
            buf[i] = b;
            d = nesterov ? fmaf(mu, b, d) : b;
        }
        p[i] = fmaf(-lr, d, w);
    }
}
// the same update with {lr, momentum, dampening, weight_decay} read from device memory: a captured hipGraph of the step
// then follows a learning-rate schedule (trainval.py:73 ExponentialLR) -- launch scalars are frozen at capture time
// the sticky failure words a guarded update looks at: the grid-barrier words of every barrier workspace of the device (a step may
// have run eagerly on one stream and as a captured graph on another: each has its own words) and, under data parallelism, the
// REDUCED flag -- a float slot behind the flat gradient bucket that every rank fills with its own state before the all-reduce, so
// that all replicas skip the update of a step in which ANY rank failed (its NaN gradient was summed into everybody's bucket)
constexpr int SGD_MAX_FAIL = 8;
struct SgdGuard {
    const unsigned* words[SGD_MAX_FAIL];
    int nwords;
    const float* reduced;      // may be null
};
__device__ __forceinline__ bool sgd_guard_set(const SgdGuard& gd) {
    bool bad = false;
    for (int i = 0; i < gd.nwords; ++i) bad |= *gd.words[i] != 0u;
    if (gd.reduced != nullptr) bad |= !(*gd.reduced == 0.f);          // (a NaN slot counts as set)
    return bad;
}
// slot = 1 when any of the local words is set, else 0 (one thread: the words are a handful)
__global__ void sgd_guard_publish_kernel(const SgdGuard gd, float* __restrict__ slot) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        bool bad = false;
        for (int i = 0; i < gd.nwords; ++i) bad |= *gd.words[i] != 0u;
        *slot = bad ? 1.f : 0.f;
    }
}
__global__ __launch_bounds__(256) void sgd_hyper_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ buf, int64_t n,
                                                        const float* __restrict__ hyper, int nesterov, int first,
                                                        const SgdGuard guard) {
    // guard: sticky failure words (gridsync.hpp FW_FAIL) and the rank-reduced flag.  A one-launch kernel whose barrier
    // gave up has NaN-poisoned its outputs, hence the gradient: the update is SKIPPED while any of them is set (uniform scalar
    // loads), so parameters and momentum survive until ops.check_gridsync reports the failure -- also in captured replays
    if (sgd_guard_set(guard)) return;
    // hyper[4] = gradient scale: 1 / world size when the bucket holds the all-reduce SUM of the ranks' gradients (the mean
    // then never exists as a pass of its own over the bucket); 1 otherwise (x * 1.0f is exact)
    const float lr = hyper[0], mu = hyper[1], damp = hyper[2], wd = hyper[3], gs = hyper[4];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float w = p[i];
        float d = fmaf(wd, w, g[i] * gs);
        if (mu != 0.f) {
            const float b = first ? d : fmaf(mu, buf[i], (1.f - damp) * d);
            buf[i] = b;
            d = nesterov ? fmaf(mu, b, d) : b;
        }
        p[i] = fmaf(-lr, d, w);
    }
}
}  // namespace crf

static int sgd_guard_of(const unsigned* const* fail_words, int nwords, const float* reduced, crf::SgdGuard& gd) {
    CRF_REQUIRE(nwords >= 0 && nwords <= crf::SGD_MAX_FAIL, CRF_ERR_ARG, "%d failure words (at most %d)", nwords, crf::SGD_MAX_FAIL);
    CRF_REQUIRE(nwords == 0 || fail_words != nullptr, CRF_ERR_ARG, "null pointer");
    for (int i = 0; i < crf::SGD_MAX_FAIL; ++i) gd.words[i] = nullptr;
    for (int i = 0; i < nwords; ++i) {
        CRF_REQUIRE(fail_words[i] != nullptr, CRF_ERR_ARG, "failure word %d: null pointer", i);
        gd.words[i] = fail_words[i];
    }
    gd.nwords = nwords;
    gd.reduced = reduced;
    return CRF_OK;
}

static int sgd_launch(float* param, const float* grad, float* momentum_buf, int64_t n, const float* hyper, int nesterov,
                      int first_step, const crf::SgdGuard& gd, crf_stream_t stream) {
    CRF_REQUIRE(param && grad && momentum_buf && hyper, CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(n > 0, CRF_ERR_ARG, "n=%lld <= 0", (long long)n);
    int64_t nb = cdiv(n, 256 * 4);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(crf::sgd_hyper_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), param, grad, momentum_buf,
                       n, hyper, nesterov, first_step, gd);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_sgd_step_hyper(float* param, const float* grad, float* momentum_buf, int64_t n,
                                      const float* hyper, int nesterov, int first_step, crf_stream_t stream) {
    crf::SgdGuard gd;
    if (int rc = sgd_guard_of(nullptr, 0, nullptr, gd)) return rc;
    return sgd_launch(param, grad, momentum_buf, n, hyper, nesterov, first_step, gd, stream);
}

extern "C" int crfconv_sgd_step_guarded(float* param, const float* grad, float* momentum_buf, int64_t n,
                                        const float* hyper, int nesterov, int first_step, const unsigned* fail_word,
                                        crf_stream_t stream) {
    crf::SgdGuard gd;
    if (int rc = sgd_guard_of(&fail_word, fail_word ? 1 : 0, nullptr, gd)) return rc;
    return sgd_launch(param, grad, momentum_buf, n, hyper, nesterov, first_step, gd, stream);
}

extern "C" int crfconv_sgd_step_guarded_all(float* param, const float* grad, float* momentum_buf, int64_t n,
                                            const float* hyper, int nesterov, int first_step,
                                            const unsigned* const* fail_words, int n_fail_words, const float* reduced_flag,
                                            crf_stream_t stream) {
    crf::SgdGuard gd;
    if (int rc = sgd_guard_of(fail_words, n_fail_words, reduced_flag, gd)) return rc;
    return sgd_launch(param, grad, momentum_buf, n, hyper, nesterov, first_step, gd, stream);
}

extern "C" int crfconv_sgd_guard_publish(const unsigned* const* fail_words, int n_fail_words, float* slot, crf_stream_t stream) {
    CRF_REQUIRE(slot, CRF_ERR_ARG, "null pointer");
    crf::SgdGuard gd;
    if (int rc = sgd_guard_of(fail_words, n_fail_words, nullptr, gd)) return rc;
    hipLaunchKernelGGL(crf::sgd_guard_publish_kernel, dim3(1), dim3(64), 0, as_stream(stream), gd, slot);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}

extern "C" int crfconv_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n, float lr,
                                float momentum, float dampening, float weight_decay, int nesterov, int first_step,
                                crf_stream_t stream) {
    CRF_REQUIRE(param && grad && (momentum_buf || momentum == 0.f), CRF_ERR_ARG, "null pointer");
    CRF_REQUIRE(n > 0, CRF_ERR_ARG, "n=%lld <= 0", (long long)n);
    int64_t nb = cdiv(n, 256 * 4);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(crf::sgd_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), param, grad, momentum_buf, n, lr,
                       momentum, dampening, weight_decay, nesterov, first_step);
    CRF_LAUNCH_CHECK();
    return CRF_OK;
}
