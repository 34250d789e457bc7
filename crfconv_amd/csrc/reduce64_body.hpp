// Sums of per-workgroup partial slabs, float64 results: out[slot] = sum_b partial[b][slot] -- one wavefront per (job, slot).  Shared by
// pointconv.hip (crfconv_reduce_jobs_f64) and linear.hip (crfconv_reduce_jobs_both: these sums and the float weight-gradient sums of a
// backward pass in ONE launch).
#pragma once
#include "common.hpp"

namespace crf {

// The same for SEVERAL reductions in one launch (crfconv_reduce_jobs_f64): the parameter-gradient partials of all PointConv layers
// (dW2 float slabs and dA1 | db1 float64 slabs) are summed once, at the end of the backward pass, in front of the batched fold --
// ten to fourteen single-purpose launches otherwise.  One wavefront per (job, slot); lane order and shuffle tree as above.
constexpr int R64_MAX = 32;
struct Reduce64Table {
    const void* partial[R64_MAX];
    double* out[R64_MAX];
    int is_float[R64_MAX], nblk[R64_MAX], nslots[R64_MAX];
    int wave_base[R64_MAX + 1];                        // prefix of nslots
    int njobs;
};
__device__ __forceinline__ void reduce_jobs_f64_body(const Reduce64Table& t, const unsigned blk) {
    const int gw = (int)blk * (256 / WAVE) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (gw >= t.wave_base[t.njobs]) return;
    int lo = 0, hi = t.njobs;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (t.wave_base[mid] <= gw) lo = mid; else hi = mid;
    }
    const int slot = gw - t.wave_base[lo], nslots = t.nslots[lo], nblk = t.nblk[lo];
    double a = 0.0;
    if (t.is_float[lo]) {
        const float* __restrict__ p = reinterpret_cast<const float*>(t.partial[lo]);
        for (int64_t b = lane; b < nblk; b += WAVE) a += (double)p[b * nslots + slot];
    } else {
        const double* __restrict__ p = reinterpret_cast<const double*>(t.partial[lo]);
        for (int64_t b = lane; b < nblk; b += WAVE) a += p[b * nslots + slot];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, WAVE);
    if (lane == 0) t.out[lo][slot] = a;
}

}  // namespace crf
