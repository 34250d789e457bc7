// Shared device/host helpers for the gfx950 kernels of libcrfconv_amd.so.
#pragma once
#include <cstring>
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <type_traits>

#include "../../include/crfconv_amd.h"

namespace crf {

// A value the whole wavefront agrees on (e.g. an entry of a by-value job table picked by blockIdx), pinned into scalar registers:
// without the hint the entries of a dynamically indexed kernel-argument array can end up replicated in vector registers.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uni(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
template <typename T>
__device__ __forceinline__ T* uni(T* p) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ long long uni(long long v) {
    const unsigned long long a = (unsigned long long)v;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}


constexpr int WAVE = 64;  // gfx950 wavefront

// ----------------------------------------------------------------------------- errors
void set_error(const char* fmt, ...);

#define CRF_REQUIRE(cond, code, ...)        \
    do {                                    \
        if (!(cond)) {                      \
            ::crf::set_error(__VA_ARGS__);  \
            return (code);                  \
        }                                   \
    } while (0)

#define CRF_HIP(expr)                                                                  \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) {                                                        \
            ::crf::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),    \
                             __FILE__, __LINE__);                                      \
            return CRF_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)

#define CRF_LAUNCH_CHECK() CRF_HIP(hipGetLastError())

inline hipStream_t as_stream(crf_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ----------------------------------------------------------------------------- device helpers
// XCD-aware block order.  MI355X deals workgroups round-robin over its 8 XCDs (block b -> XCD b % 8),
// each with a private 4 MiB L2.  Kernels here walk spatially sorted points and gather neighbour rows,
// so handing every XCD one CONTIGUOUS eighth of the blocks keeps an XCD's gathers inside its own
// eighth of the table (one L2 fill per row instead of up to eight).  Bijective for any grid size;
// the placement assumption only affects speed, never results.
__device__ __forceinline__ unsigned xcd_block_id() {
    const unsigned nb = gridDim.x, b = blockIdx.x;
    const unsigned xcd = b & 7u, within = b >> 3;
    const unsigned base = nb >> 3, rem = nb & 7u;
    return xcd * base + (xcd < rem ? xcd : rem) + within;
}

// Sum over aligned groups of L consecutive lanes (L a power of two <= 64); every lane of the
// group receives the total.
// lane ^ 1 / lane ^ 2 inside a DPP quad: one v_mov_b32_dpp instead of a ds_bpermute_b32
__device__ __forceinline__ float quad_xor1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));      // quad_perm [1, 0, 3, 2]
}
__device__ __forceinline__ float quad_xor2(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));      // quad_perm [2, 3, 0, 1]
}
template <int L>
__device__ __forceinline__ float group_sum(float v) {
    if constexpr (L == 2) {
        return v + quad_xor1(v);
    } else if constexpr (L == 4) {
        v += quad_xor2(v);
        return v + quad_xor1(v);
    } else {
#pragma unroll
        for (int o = L / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
        return v;
    }
}

template <int L>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WAVE));
    return v;
}

__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); }

// static_for<N>(f): f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N - 1>) -- a loop whose index is a
// compile-time constant inside the body (template arguments, DPP controls).
template <int I, int N, typename F>
__device__ __forceinline__ void static_for_impl(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_impl<I + 1, N>(f);
    }
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl<0, N>(f); }

// Value of lane (group base + HQ) for every lane of an aligned group of L adjacent lanes, HQ a compile-time constant.  For
// L <= 4 the group lies inside a DPP quad: one full-rate v_mov_b32_dpp with a quad_perm control instead of a
// ds_bpermute_b32 through the LDS pipeline (what __shfl compiles to) -- the narrow PointConv kernels issue 48 such
// broadcasts per edge.  Wider groups fall back to the shuffle.
template <int L, int HQ>
__device__ __forceinline__ float group_bcast(float v, int base_lane) {
    if constexpr (L == 1) {
        return v;
    } else if constexpr (L == 2) {
        constexpr int ctrl = HQ | (HQ << 2) | ((2 + HQ) << 4) | ((2 + HQ) << 6);           // quad_perm [HQ, HQ, 2 + HQ, 2 + HQ]
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, true));
    } else if constexpr (L == 4) {
        constexpr int ctrl = HQ * 0x55;                                                  // quad_perm [HQ, HQ, HQ, HQ]
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, true));
    } else {
        return __shfl(v, base_lane + HQ, WAVE);
    }
}


// Singly rounded float operations that the compiler may NOT fuse into an FMA.  (The __fmul_rn / __fadd_rn
// intrinsics of this toolchain are plain `*` / `+` and inherit hipcc's default -ffp-contract=fast: a product feeding
// a sum becomes v_fma_f32.)  Bit-exact parity paths -- kNN distances, voxel keys and barycentres, the vote and
// possibility updates -- must round every operation like the reference's scalar C++ / numpy code does.
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}
__device__ __forceinline__ double dmul_rn(double a, double b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ double dadd_rn(double a, double b) {
#pragma clang fp contract(off)
    return a + b;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

__device__ __forceinline__ float4 fma4(float a, float4 b, float4 c) {
    return make_float4(fmaf(a, b.x, c.x), fmaf(a, b.y, c.y), fmaf(a, b.z, c.z), fmaf(a, b.w, c.w));
}
__device__ __forceinline__ float dot4(float4 a, float4 b) {
    return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}
__device__ __forceinline__ float comp(const float4& v, int i) {
    return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
}

// Counter-based dropout mask shared by the forward and the backward pass: element e of the call with (seed, *counter) is kept
// iff a keyed 32-bit integer hash of e reaches `threshold` = p 2^32.  The two 32-bit keys are the halves of one splitmix64 round
// of (seed, counter) -- uniform per call, so the compiler hoists them out of every loop --; per element: the two-multiply
// "lowbias32" finalizer (x ^= x >> 16; x *= 0x7feb352d; x ^= x >> 15; x *= 0x846ca68b; x ^= x >> 16) with the second key added
// in front of the second multiply.  (Round 4: the 64-bit splitmix round per ELEMENT this replaces is ~190 cycles per wavefront
// on gfx950 -- eight quarter-rate 32-bit multiplies -- and bound every kernel that draws a mask: bn_apply_dropout 26 us for
// 21 M elements; this one is ~60.)  *counter is a DEVICE word that the caller advances between training steps (the classifier
// BatchNorm's num_batches_tracked), so a captured hipGraph draws a new mask at every replay while forward and backward of one
// step agree without a stored mask.  Host twin: ops.dropout_keep_mask.
__device__ __forceinline__ unsigned long long dropout_keys(unsigned long long seed, unsigned long long ctr) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (ctr + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ bool dropout_keep_keyed(unsigned long long keys, unsigned long long e, unsigned threshold) {
    const unsigned hi = (unsigned)(e >> 32);
    unsigned x = (unsigned)e + (unsigned)keys + ((hi << 13) | (hi >> 19));
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x += (unsigned)(keys >> 32);
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x >= threshold;
}
__device__ __forceinline__ bool dropout_keep(unsigned long long seed, unsigned long long ctr, unsigned long long e, unsigned threshold) {
    return dropout_keep_keyed(dropout_keys(seed, ctr), e, threshold);
}

inline unsigned dropout_threshold(float p) {
    const double t = (double)p * 4294967296.0;
    return t >= 4294967295.0 ? 0xffffffffu : (t <= 0.0 ? 0u : (unsigned)t);
}

}  // namespace crf
