// Variants of the level-0 mean-field step kernel (H = 8, K = 16, uint16 index rows) for A/B timing.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC scratch/mfv.hip -o scratch/libmfv.so
#include <hip/hip_runtime.h>
#include <cstdint>

namespace {
constexpr int H = 8, K = 16, L = 2, WAVE = 64;

__device__ __forceinline__ unsigned xcd_block_id() {
    const unsigned nb = gridDim.x, b = blockIdx.x;
    const unsigned xcd = b & 7u, within = b >> 3;
    const unsigned base = nb >> 3, rem = nb & 7u;
    return xcd * base + (xcd < rem ? xcd : rem) + within;
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
typedef float f4v __attribute__((ext_vector_type(4)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4nt(const float* p) {
    const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4nt(float* p, float4 v) {
    f4v t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<f4v*>(p));
}
__device__ __forceinline__ float4 fma4(float a, float4 b, float4 c) {
    return make_float4(fmaf(a, b.x, c.x), fmaf(a, b.y, c.y), fmaf(a, b.z, c.z), fmaf(a, b.w, c.w));
}
__device__ __forceinline__ float4 matvec_acc(float4 v, const float4* sM, int lane, int q, float4 acc) {
    const int base = lane - q;
#pragma unroll
    for (int hq = 0; hq < L; ++hq) {
        const float v0 = __shfl(v.x, base + hq, WAVE), v1 = __shfl(v.y, base + hq, WAVE);
        const float v2 = __shfl(v.z, base + hq, WAVE), v3 = __shfl(v.w, base + hq, WAVE);
        acc = fma4(v0, sM[(4 * hq + 0) * L + q], acc);
        acc = fma4(v1, sM[(4 * hq + 1) * L + q], acc);
        acc = fma4(v2, sM[(4 * hq + 2) * L + q], acc);
        acc = fma4(v3, sM[(4 * hq + 3) * L + q], acc);
    }
    return acc;
}
template <int NT>
__device__ __forceinline__ void load_matrix(float4* sM, const float* __restrict__ Mat) {
    float* s = reinterpret_cast<float*>(sM);
    for (int t = threadIdx.x; t < H * H; t += NT) s[t] = Mat[t];
}
template <bool NT_>
__device__ __forceinline__ void load_idx(const uint16_t* __restrict__ idx16, int64_t r, int base, int (&j)[K]) {
    const uint4* p = reinterpret_cast<const uint4*>(idx16 + r * K);
#pragma unroll
    for (int c = 0; c < K / 8; ++c) {
        unsigned w[4];
        if constexpr (NT_) {
            const u4v v = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(p + c));
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        } else {
            const uint4 v = p[c];
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            j[8 * c + 2 * e] = base + (int)(w[e] & 0xffffu);
            j[8 * c + 2 * e + 1] = base + (int)(w[e] >> 16);
        }
    }
}
template <bool NT_>
__device__ __forceinline__ void load_w(const float* __restrict__ p, float (&out)[K]) {
#pragma unroll
    for (int c = 0; c < K / 4; ++c) {
        const float4 v = NT_ ? ld4nt(p + 4 * c) : ld4(p + 4 * c);
        out[4 * c + 0] = v.x; out[4 * c + 1] = v.y; out[4 * c + 2] = v.z; out[4 * c + 3] = v.w;
    }
}

// MODE 0: baseline; 1: neighbours replaced by the own row (stream-only lower bound); 2: nontemporal streams;
// 3: gathers in two batches (fewer live registers)
template <int MODE>
__global__ __launch_bounds__(256) void step_kernel(const float* __restrict__ xin, const float* __restrict__ z,
                                                   const float* __restrict__ s, const uint16_t* __restrict__ idx16,
                                                   int n_tgt, int n_src, const float* __restrict__ Q,
                                                   const float* __restrict__ P, float* __restrict__ xout, int64_t m) {
    __shared__ float4 sP[H * L];
    __shared__ float4 sQ[H * L];
    load_matrix<256>(sP, P);
    load_matrix<256>(sQ, Q);
    const int lane = threadIdx.x & 63, q = lane % L;
    int64_t r = (int64_t)xcd_block_id() * 128 + (threadIdx.x >> 6) * 32 + lane / L;
    const bool valid = r < m;
    if (!valid) r = m - 1;
    int j[K];
    float w[K];
    constexpr bool NTL = MODE == 2 || MODE == 7;
    constexpr bool NTS = MODE == 2 || MODE == 6 || MODE == 7;
    if constexpr (MODE == 4 || MODE == 5) {
        // index / weight rows of the wave's 32 points as fully coalesced 1 KB loads, redistributed through LDS
        __shared__ float4 sS[4][128];
        __shared__ uint4 sI[4][64];
        const int wv = threadIdx.x >> 6;
        const int64_t row0 = (int64_t)xcd_block_id() * 128 + wv * 32;
        auto rowc = [&](int64_t rr) { return rr < m ? rr : m - 1; };
        const float4 sa = ld4(s + rowc(row0 + (lane >> 2)) * K + 4 * (lane & 3));
        const float4 sb = ld4(s + rowc(row0 + 16 + (lane >> 2)) * K + 4 * (lane & 3));
        const uint4 ia = reinterpret_cast<const uint4*>(idx16 + rowc(row0 + (lane >> 1)) * K)[lane & 1];
        sS[wv][lane] = sa; sS[wv][lane + 64] = sb; sI[wv][lane] = ia;
        __builtin_amdgcn_wave_barrier();
        const int p = lane >> 1, base = (int)(r / n_tgt) * n_src;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 v = sS[wv][p * 4 + c];
            w[4 * c] = v.x; w[4 * c + 1] = v.y; w[4 * c + 2] = v.z; w[4 * c + 3] = v.w;
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const uint4 v = sI[wv][p * 2 + c];
            const unsigned ww[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                j[8 * c + 2 * e] = base + (int)(ww[e] & 0xffffu);
                j[8 * c + 2 * e + 1] = base + (int)(ww[e] >> 16);
            }
        }
    } else {
        load_idx<NTL>(idx16, r, (int)(r / n_tgt) * n_src, j);
        load_w<NTL>(s + r * K, w);
    }
    const float4 zi = NTL ? ld4nt(z + r * H + 4 * q) : ld4(z + r * H + 4 * q);
    float4 msg = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (MODE == 3 || MODE == 5 || MODE == 6 || MODE == 7) {
        float4 nb[8];
#pragma unroll
        for (int k = 1; k < 8; ++k) nb[k] = ld4(xin + (int64_t)j[k] * H + 4 * q);
#pragma unroll
        for (int k = 1; k < 8; ++k) msg = fma4(w[k], nb[k], msg);
#pragma unroll
        for (int k = 0; k < 8; ++k) nb[k] = ld4(xin + (int64_t)j[8 + k] * H + 4 * q);
#pragma unroll
        for (int k = 0; k < 8; ++k) msg = fma4(w[8 + k], nb[k], msg);
    } else {
        float4 nb[K];
#pragma unroll
        for (int k = 1; k < K; ++k) nb[k] = ld4(xin + (int64_t)(MODE == 1 ? (int)r + (j[k] & 0) : j[k]) * H + 4 * q);
#pragma unroll
        for (int k = 1; k < K; ++k) msg = fma4(w[k], nb[k], msg);
    }
    __syncthreads();
    const float4 zqi = matvec_acc(zi, sQ, lane, q, make_float4(0.f, 0.f, 0.f, 0.f));
    const float4 o = matvec_acc(msg, sP, lane, q, zqi);
    if (valid) {
        if (NTS) st4nt(xout + r * H + 4 * q, o);
        else st4(xout + r * H + 4 * q, o);
    }
}

// LDS window: block of NT threads = NT/2 points stages rows [first - HALO, first + NT/2 + HALO) of xin.
template <int NT, int HALO>
__global__ __launch_bounds__(NT) void step_win_kernel(const float* __restrict__ xin, const float* __restrict__ z,
                                                      const float* __restrict__ s, const uint16_t* __restrict__ idx16,
                                                      int n_tgt, int n_src, const float* __restrict__ Q,
                                                      const float* __restrict__ P, float* __restrict__ xout, int64_t m) {
    constexpr int PPB = NT / L, ROWS = PPB + 2 * HALO;
    __shared__ float4 sP[H * L];
    __shared__ float4 sQ[H * L];
    __shared__ float4 sX[ROWS * L];
    int64_t w0 = (int64_t)xcd_block_id() * PPB - HALO;
    if (w0 > m - ROWS) w0 = m - ROWS;
    if (w0 < 0) w0 = 0;
    load_matrix<NT>(sP, P);
    load_matrix<NT>(sQ, Q);
    {
        const float4* s4 = reinterpret_cast<const float4*>(xin) + w0 * L;
        const int64_t lim = (m - w0) * L;
        for (int t = threadIdx.x; t < ROWS * L; t += NT) sX[t] = t < lim ? s4[t] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int lane = threadIdx.x & 63, q = lane % L;
    int64_t r = (int64_t)xcd_block_id() * PPB + (threadIdx.x >> 6) * 32 + lane / L;
    const bool valid = r < m;
    if (!valid) r = m - 1;
    int j[K];
    float w[K];
    load_idx<false>(idx16, r, (int)(r / n_tgt) * n_src, j);
    load_w<false>(s + r * K, w);
    const float4 zi = ld4(z + r * H + 4 * q);
    // out-of-window rows first (vector memory), then the barrier, then LDS
    float4 nb[K];
    const int iw0 = (int)w0;
#pragma unroll
    for (int k = 1; k < K; ++k) {
        const int jl = j[k] - iw0;
        if ((unsigned)jl >= (unsigned)ROWS) nb[k] = ld4(xin + (int64_t)j[k] * H + 4 * q);
    }
    __syncthreads();
#pragma unroll
    for (int k = 1; k < K; ++k) {
        const int jl = j[k] - iw0;
        if ((unsigned)jl < (unsigned)ROWS) nb[k] = sX[jl * L + q];
    }
    float4 msg = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 1; k < K; ++k) msg = fma4(w[k], nb[k], msg);
    const float4 zqi = matvec_acc(zi, sQ, lane, q, make_float4(0.f, 0.f, 0.f, 0.f));
    const float4 o = matvec_acc(msg, sP, lane, q, zqi);
    if (valid) st4(xout + r * H + 4 * q, o);
}

// Software-pipelined: each wave walks CH consecutive 32-point chunks; the index / weight / z rows of chunk c+1 are
// requested before the gathers of chunk c are consumed.
template <int CH>
__global__ __launch_bounds__(256) void step_pipe_kernel(const float* __restrict__ xin, const float* __restrict__ z,
                                                        const float* __restrict__ s, const uint16_t* __restrict__ idx16,
                                                        int n_tgt, int n_src, const float* __restrict__ Q,
                                                        const float* __restrict__ P, float* __restrict__ xout, int64_t m) {
    __shared__ float4 sP[H * L];
    __shared__ float4 sQ[H * L];
    load_matrix<256>(sP, P);
    load_matrix<256>(sQ, Q);
    __syncthreads();
    const int lane = threadIdx.x & 63, q = lane % L;
    const int64_t r0 = ((int64_t)xcd_block_id() * 4 + (threadIdx.x >> 6)) * (32 * CH) + lane / L;
    auto clampr = [&](int64_t r) { return r < m ? r : m - 1; };
    int j[K];
    float w[K];
    float4 zi;
    {
        const int64_t r = clampr(r0);
        load_idx<false>(idx16, r, (int)(r / n_tgt) * n_src, j);
        load_w<false>(s + r * K, w);
        zi = ld4(z + r * H + 4 * q);
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int64_t rr = r0 + 32 * c;
        const int64_t r = clampr(rr);
        float4 nb[K];
#pragma unroll
        for (int k = 1; k < K; ++k) nb[k] = ld4(xin + (int64_t)j[k] * H + 4 * q);
        float wc[K];
#pragma unroll
        for (int k = 0; k < K; ++k) wc[k] = w[k];
        const float4 zc = zi;
        if (c + 1 < CH) {
            const int64_t rn = clampr(rr + 32);
            load_idx<false>(idx16, rn, (int)(rn / n_tgt) * n_src, j);
            load_w<false>(s + rn * K, w);
            zi = ld4(z + rn * H + 4 * q);
        }
        float4 msg = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 1; k < K; ++k) msg = fma4(wc[k], nb[k], msg);
        const float4 zqi = matvec_acc(zc, sQ, lane, q, make_float4(0.f, 0.f, 0.f, 0.f));
        const float4 o = matvec_acc(msg, sP, lane, q, zqi);
        if (rr < m) st4(xout + r * H + 4 * q, o);
    }
}

__device__ __forceinline__ float4 sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float dot4(float4 a, float4 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w))); }
__device__ __forceinline__ float pair_sum(float v) { return v + __shfl_xor(v, 1, WAVE); }

// First kernel: s = softmax_k(-|y_i - y_j|^2), x1 = z Q + (sum_k s z_j) P.
// MODE 0: baseline; 1: own-row instead of gathers; 2: packed indices + batches of 8; 3: y/z interleaved batches of 8
template <int MODE, int MINW, int FL = 0>
__global__ __launch_bounds__(256, MINW) void sim_kernel(const float* __restrict__ y, const float* __restrict__ z,
                                                        const uint16_t* __restrict__ idx16, int n_tgt, int n_src,
                                                        const float* __restrict__ Q, const float* __restrict__ P,
                                                        float* __restrict__ s, float* __restrict__ x1, int64_t m) {
    __shared__ float4 sP[H * L];
    __shared__ float4 sQ[H * L];
    load_matrix<256>(sP, P);
    load_matrix<256>(sQ, Q);
    const int lane = threadIdx.x & 63, q = lane % L;
    int64_t r = (int64_t)xcd_block_id() * 128 + (threadIdx.x >> 6) * 32 + lane / L;
    const bool valid = r < m;
    if (!valid) r = m - 1;
    const int base = (int)(r / n_tgt) * n_src;
    float d[K];
    float4 msg = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 yi = ld4(y + r * H + 4 * q);
    if constexpr (MODE <= 1) {
        int j[K];
        load_idx<false>(idx16, r, base, j);
        if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < K; ++k) j[k] = (int)r + (j[k] & 0);
        }
        float4 nb[K];
#pragma unroll
        for (int k = 1; k < K; ++k) nb[k] = ld4(y + (int64_t)j[k] * H + 4 * q);
        float dmin = 3.4e38f;
#pragma unroll
        for (int k = 1; k < K; ++k) {
            const float4 df = sub4(yi, nb[k]);
            d[k] = pair_sum(dot4(df, df));
            dmin = fminf(dmin, d[k]);
        }
#pragma unroll
        for (int k = 1; k < K; ++k) nb[k] = ld4(z + (int64_t)j[k] * H + 4 * q);
        float den = 0.f;
#pragma unroll
        for (int k = 1; k < K; ++k) { d[k] = __expf(dmin - d[k]); den += d[k]; }
        const float inv = 1.0f / den;
        d[0] = 0.f;
#pragma unroll
        for (int k = 1; k < K; ++k) d[k] *= inv;
#pragma unroll
        for (int k = 1; k < K; ++k) msg = fma4(d[k], nb[k], msg);
    } else {
        const uint4* pi = reinterpret_cast<const uint4*>(idx16 + r * K);
        const uint4 p0 = pi[0], p1 = pi[1];
        const unsigned pk[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
        auto jk = [&](int k) { return (int64_t)(base + (int)((k & 1) ? (pk[k >> 1] >> 16) : (pk[k >> 1] & 0xffffu))); };
        if constexpr (MODE == 2) {
            float4 nb[8];
            float dmin = 3.4e38f;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
#pragma unroll
                for (int k = 0; k < 8; ++k) if (8 * b + k) nb[k] = ld4(y + jk(8 * b + k) * H + 4 * q);
#pragma unroll
                for (int k = 0; k < 8; ++k) if (8 * b + k) {
                    const float4 df = sub4(yi, nb[k]);
                    d[8 * b + k] = pair_sum(dot4(df, df));
                    dmin = fminf(dmin, d[8 * b + k]);
                }
            }
#pragma unroll
            for (int k = 1; k < 8; ++k) nb[k] = ld4(z + jk(k) * H + 4 * q);
            float den = 0.f;
#pragma unroll
            for (int k = 1; k < K; ++k) { d[k] = __expf(dmin - d[k]); den += d[k]; }
            const float inv = 1.0f / den;
            d[0] = 0.f;
#pragma unroll
            for (int k = 1; k < K; ++k) d[k] *= inv;
#pragma unroll
            for (int k = 1; k < 8; ++k) msg = fma4(d[k], nb[k], msg);
#pragma unroll
            for (int k = 0; k < 8; ++k) nb[k] = ld4(z + jk(8 + k) * H + 4 * q);
#pragma unroll
            for (int k = 0; k < 8; ++k) msg = fma4(d[8 + k], nb[k], msg);
        } else {
            // y and z rows of a neighbour requested together; unnormalised accumulation with a running minimum
            float4 ny[8], nz[8];
            float dmin = 3.4e38f, den = 0.f;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
#pragma unroll
                for (int k = 0; k < 8; ++k) if (8 * b + k) {
                    ny[k] = ld4(y + jk(8 * b + k) * H + 4 * q);
                    nz[k] = ld4(z + jk(8 * b + k) * H + 4 * q);
                }
                float bmin = dmin;
#pragma unroll
                for (int k = 0; k < 8; ++k) if (8 * b + k) {
                    const float4 df = sub4(yi, ny[k]);
                    d[8 * b + k] = pair_sum(dot4(df, df));
                    bmin = fminf(bmin, d[8 * b + k]);
                }
                const float resc = __expf(bmin - dmin);      // 0 on the first batch (dmin = 3.4e38)
                den *= resc;
                msg = make_float4(msg.x * resc, msg.y * resc, msg.z * resc, msg.w * resc);
                dmin = bmin;
#pragma unroll
                for (int k = 0; k < 8; ++k) if (8 * b + k) {
                    const float e = __expf(dmin - d[8 * b + k]);
                    den += e;
                    msg = fma4(e, nz[k], msg);
                }
            }
            const float inv = 1.0f / den;
            msg = make_float4(msg.x * inv, msg.y * inv, msg.z * inv, msg.w * inv);
            d[0] = 0.f;
#pragma unroll
            for (int k = 1; k < K; ++k) d[k] = __expf(dmin - d[k]) * inv;
        }
    }
#pragma unroll
    for (int c = 0; c < K / 4; ++c)
        if (valid && (c % L) == q && !(FL & 2) && !(FL & 16)) {
            const float4 v = make_float4(d[4 * c], d[4 * c + 1], d[4 * c + 2], d[4 * c + 3]);
            if (FL & 4) st4nt(s + r * K + 4 * c, v); else st4(s + r * K + 4 * c, v);
        }
    if (FL & 2) msg.x += d[3] + d[7] + d[9] + d[14];
    if constexpr ((FL & 16) != 0) {      // s rows of the wave (32 points x 64 B, contiguous) transposed through LDS
        __shared__ float4 sS[4][128];
        float4* mine = sS[threadIdx.x >> 6];
        const int p = lane >> 1;
        mine[p * 4 + q] = q ? make_float4(d[4], d[5], d[6], d[7]) : make_float4(d[0], d[1], d[2], d[3]);
        mine[p * 4 + q + 2] = q ? make_float4(d[12], d[13], d[14], d[15]) : make_float4(d[8], d[9], d[10], d[11]);
        __builtin_amdgcn_wave_barrier();
        const int64_t row0 = (int64_t)xcd_block_id() * 128 + (threadIdx.x >> 6) * 32;
        const float4 a = mine[lane], b = mine[lane + 64];
        if (FL & 32) {
            if (row0 + (lane >> 2) < m) st4nt(s + row0 * K + 4 * lane, a);
            if (row0 + 16 + (lane >> 2) < m) st4nt(s + row0 * K + 256 + 4 * lane, b);
        } else {
            if (row0 + (lane >> 2) < m) st4(s + row0 * K + 4 * lane, a);
            if (row0 + 16 + (lane >> 2) < m) st4(s + row0 * K + 256 + 4 * lane, b);
        }
    }
    const float4 zi = ld4(z + r * H + 4 * q);
    __syncthreads();
    const float4 zqi = matvec_acc(zi, sQ, lane, q, make_float4(0.f, 0.f, 0.f, 0.f));
    const float4 o = matvec_acc(msg, sP, lane, q, zqi);
    if (valid) { if (FL & 8) st4nt(x1 + r * H + 4 * q, o); else st4(x1 + r * H + 4 * q, o); }
}
// First kernel on an INTERLEAVED table yz [m, 16] = [y_i (8) | z_i (8)]: four lanes per point fetch one contiguous
// 64-byte row per neighbour (lanes 0-1 the y half, lanes 2-3 the z half), 15 gathers instead of 30.
__global__ __launch_bounds__(256, 4) void sim_yz_kernel(const float* __restrict__ yz, const uint16_t* __restrict__ idx16,
                                                        int n_tgt, int n_src, const float* __restrict__ Q,
                                                        const float* __restrict__ P, float* __restrict__ s,
                                                        float* __restrict__ x1, int64_t m) {
    __shared__ float4 sP[H * L];
    __shared__ float4 sQ[H * L];
    load_matrix<256>(sP, P);
    load_matrix<256>(sQ, Q);
    const int lane = threadIdx.x & 63, q4 = lane & 3;
    int64_t r = (int64_t)xcd_block_id() * 64 + (threadIdx.x >> 6) * 16 + (lane >> 2);
    const bool valid = r < m;
    if (!valid) r = m - 1;
    const int base = (int)(r / n_tgt) * n_src;
    int j[K];
    load_idx<false>(idx16, r, base, j);
    const float4 own = ld4(yz + r * 16 + 4 * q4);
    float4 nb[K];
#pragma unroll
    for (int k = 1; k < K; ++k) nb[k] = ld4(yz + (int64_t)j[k] * 16 + 4 * q4);
    float d[K];
    float dmin = 3.4e38f;
    const bool ylane = q4 < 2;
#pragma unroll
    for (int k = 1; k < K; ++k) {
        const float4 df = sub4(own, nb[k]);
        float part = ylane ? dot4(df, df) : 0.f;
        part += __shfl_xor(part, 1, WAVE);
        part += __shfl_xor(part, 2, WAVE);
        d[k] = part;
        dmin = fminf(dmin, part);
    }
    float den = 0.f;
#pragma unroll
    for (int k = 1; k < K; ++k) { d[k] = __expf(dmin - d[k]); den += d[k]; }
    const float inv = 1.0f / den;
    d[0] = 0.f;
#pragma unroll
    for (int k = 1; k < K; ++k) d[k] *= inv;
    float4 msg = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 1; k < K; ++k) msg = fma4(d[k], nb[k], msg);      // meaningful on the z lanes
    // s row: lane q4 of the quad owns chunk q4 -> the wave's 64 lanes write 1 KiB contiguous
    float4 mine;
    mine.x = q4 == 0 ? d[0] : q4 == 1 ? d[4] : q4 == 2 ? d[8] : d[12];
    mine.y = q4 == 0 ? d[1] : q4 == 1 ? d[5] : q4 == 2 ? d[9] : d[13];
    mine.z = q4 == 0 ? d[2] : q4 == 1 ? d[6] : q4 == 2 ? d[10] : d[14];
    mine.w = q4 == 0 ? d[3] : q4 == 1 ? d[7] : q4 == 2 ? d[11] : d[15];
    if (valid) st4(s + r * K + 4 * q4, mine);
    __syncthreads();
    // matvecs on the z lanes: vector broadcast from lanes (quad base + 2 + hq)
    const int zb = (lane & ~3) + 2, q = q4 & 1;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int hq = 0; hq < L; ++hq) {
        const float v0 = __shfl(own.x, zb + hq, WAVE), v1 = __shfl(own.y, zb + hq, WAVE);
        const float v2 = __shfl(own.z, zb + hq, WAVE), v3 = __shfl(own.w, zb + hq, WAVE);
        acc = fma4(v0, sQ[(4 * hq + 0) * L + q], acc); acc = fma4(v1, sQ[(4 * hq + 1) * L + q], acc);
        acc = fma4(v2, sQ[(4 * hq + 2) * L + q], acc); acc = fma4(v3, sQ[(4 * hq + 3) * L + q], acc);
        const float m0 = __shfl(msg.x, zb + hq, WAVE), m1 = __shfl(msg.y, zb + hq, WAVE);
        const float m2 = __shfl(msg.z, zb + hq, WAVE), m3 = __shfl(msg.w, zb + hq, WAVE);
        acc = fma4(m0, sP[(4 * hq + 0) * L + q], acc); acc = fma4(m1, sP[(4 * hq + 1) * L + q], acc);
        acc = fma4(m2, sP[(4 * hq + 2) * L + q], acc); acc = fma4(m3, sP[(4 * hq + 3) * L + q], acc);
    }
    if (valid && !ylane) st4(x1 + r * H + 4 * q, acc);
}
}  // namespace

extern "C" int mfv_sim_yz(const float* yz, const uint16_t* idx16, int n_tgt, int n_src, const float* Q, const float* P,
                          float* s, float* x1, int64_t m, void* stream) {
    hipLaunchKernelGGL(sim_yz_kernel, dim3((unsigned)((m + 63) / 64)), dim3(256), 0, (hipStream_t)stream, yz, idx16, n_tgt,
                       n_src, Q, P, s, x1, m);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int mfv_sim(int variant, const float* y, const float* z, const uint16_t* idx16, int n_tgt, int n_src,
                       const float* Q, const float* P, float* s, float* x1, int64_t m, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const unsigned g128 = (unsigned)((m + 127) / 128);
#define SARGS y, z, idx16, n_tgt, n_src, Q, P, s, x1, m
    switch (variant) {
        case 0: hipLaunchKernelGGL((sim_kernel<0, 4>), dim3(g128), dim3(256), 0, st, SARGS); break;
        case 1: hipLaunchKernelGGL((sim_kernel<1, 4>), dim3(g128), dim3(256), 0, st, SARGS); break;
        case 2: hipLaunchKernelGGL((sim_kernel<2, 5>), dim3(g128), dim3(256), 0, st, SARGS); break;
        case 3: hipLaunchKernelGGL((sim_kernel<3, 4>), dim3(g128), dim3(256), 0, st, SARGS); break;
        case 4: hipLaunchKernelGGL((sim_kernel<2, 6>), dim3(g128), dim3(256), 0, st, SARGS); break;
        case 5: hipLaunchKernelGGL((sim_kernel<3, 5>), dim3(g128), dim3(256), 0, st, SARGS); break;
        case 6: hipLaunchKernelGGL((sim_kernel<0, 5>), dim3(g128), dim3(256), 0, st, SARGS); break;
        case 7: hipLaunchKernelGGL((sim_kernel<1, 4, 2>), dim3(g128), dim3(256), 0, st, SARGS); break;
        case 8: hipLaunchKernelGGL((sim_kernel<1, 4, 4>), dim3(g128), dim3(256), 0, st, SARGS); break;
        case 9: hipLaunchKernelGGL((sim_kernel<3, 4, 4>), dim3(g128), dim3(256), 0, st, SARGS); break;
        case 10: hipLaunchKernelGGL((sim_kernel<3, 4, 12>), dim3(g128), dim3(256), 0, st, SARGS); break;
        case 11: hipLaunchKernelGGL((sim_kernel<3, 4, 2>), dim3(g128), dim3(256), 0, st, SARGS); break;
        case 12: hipLaunchKernelGGL((sim_kernel<3, 4, 16>), dim3(g128), dim3(256), 0, st, SARGS); break;
        case 13: hipLaunchKernelGGL((sim_kernel<0, 4, 16>), dim3(g128), dim3(256), 0, st, SARGS); break;
        case 14: hipLaunchKernelGGL((sim_kernel<1, 4, 16>), dim3(g128), dim3(256), 0, st, SARGS); break;
        case 15: hipLaunchKernelGGL((sim_kernel<0, 4, 16 | 32>), dim3(g128), dim3(256), 0, st, SARGS); break;
        case 16: hipLaunchKernelGGL((sim_kernel<0, 4, 16 | 32 | 8>), dim3(g128), dim3(256), 0, st, SARGS); break;
        default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int mfv_step(int variant, const float* xin, const float* z, const float* s, const uint16_t* idx16, int n_tgt,
                        int n_src, const float* Q, const float* P, float* xout, int64_t m, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const unsigned g128 = (unsigned)((m + 127) / 128);
#define ARGS xin, z, s, idx16, n_tgt, n_src, Q, P, xout, m
    switch (variant) {
        case 0: hipLaunchKernelGGL(step_kernel<0>, dim3(g128), dim3(256), 0, st, ARGS); break;
        case 1: hipLaunchKernelGGL(step_kernel<1>, dim3(g128), dim3(256), 0, st, ARGS); break;
        case 2: hipLaunchKernelGGL(step_kernel<2>, dim3(g128), dim3(256), 0, st, ARGS); break;
        case 3: hipLaunchKernelGGL(step_kernel<3>, dim3(g128), dim3(256), 0, st, ARGS); break;
        case 10: hipLaunchKernelGGL(step_kernel<4>, dim3(g128), dim3(256), 0, st, ARGS); break;
        case 11: hipLaunchKernelGGL(step_kernel<5>, dim3(g128), dim3(256), 0, st, ARGS); break;
        case 12: hipLaunchKernelGGL(step_kernel<6>, dim3(g128), dim3(256), 0, st, ARGS); break;
        case 13: hipLaunchKernelGGL(step_kernel<7>, dim3(g128), dim3(256), 0, st, ARGS); break;
        case 4: hipLaunchKernelGGL((step_win_kernel<512, 128>), dim3((unsigned)((m + 255) / 256)), dim3(512), 0, st, ARGS); break;
        case 5: hipLaunchKernelGGL((step_win_kernel<1024, 256>), dim3((unsigned)((m + 511) / 512)), dim3(1024), 0, st, ARGS); break;
        case 6: hipLaunchKernelGGL((step_win_kernel<256, 64>), dim3(g128), dim3(256), 0, st, ARGS); break;
        case 7: hipLaunchKernelGGL(step_pipe_kernel<2>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, ARGS); break;
        case 8: hipLaunchKernelGGL(step_pipe_kernel<4>, dim3((unsigned)((m + 511) / 512)), dim3(256), 0, st, ARGS); break;
        case 9: hipLaunchKernelGGL((step_win_kernel<1024, 512>), dim3((unsigned)((m + 511) / 512)), dim3(1024), 0, st, ARGS); break;
        default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
