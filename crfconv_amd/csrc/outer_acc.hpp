// Sum over a wavefront's points of outer products on the matrix pipe (shared by the CRF backward and PointConv's parameter pass).
#pragma once
#include "common.hpp"

namespace crf {

// Sum over a wavefront's points of the outer products a_p^T b_p (H x H, H in {8, 16}) on the matrix pipe, which is idle in
// these gather-bound kernels: the wave's a / b rows go through a per-wave LDS tile [64 / L points][H] whose row-major
// order IS the 16x16x4 fragment layout (lane l reads float 64 n + l of the tile), 64 / H points per MFMA, four MFMAs per
// call.  At H = 8 a tile row pair fills the 16 fragment rows, so D holds two valid 8 x 8 diagonal blocks that are added
// at the end.  Replaces the separate  dP = m^T G  /  dQ = z^T sum G  streaming reductions (and the m_t arrays they read).
using f32x4_t = __attribute__((ext_vector_type(4))) float;

template <int H, int NT = 256>
struct OuterAcc {
    static_assert(H == 8 || H == 16, "in-kernel outer products: H in {8, 16}");
    f32x4_t d = {0.f, 0.f, 0.f, 0.f};
    // every lane holds a float4 of its point's row (lane order = tile order)
    __device__ __forceinline__ void add_rows(float4 a, float4 b, float* tile_a, float* tile_b, int lane) {
        *reinterpret_cast<float4*>(tile_a + 4 * lane) = a;
        *reinterpret_cast<float4*>(tile_b + 4 * lane) = b;
        __builtin_amdgcn_wave_barrier();           // LDS operations of one wave complete in order
#pragma unroll
        for (int n = 0; n < 4; ++n) d = __builtin_amdgcn_mfma_f32_16x16x4f32(tile_a[64 * n + lane], tile_b[64 * n + lane], d, 0, 0, 0);
        __builtin_amdgcn_wave_barrier();
    }
    // 64 floats of each operand already in the tiles (one MFMA)
    __device__ __forceinline__ void add_tile64(const float* tile_a, const float* tile_b, int lane) {
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(tile_a[lane], tile_b[lane], d, 0, 0, 0);
    }
    // block sum -> partial[blockIdx.x][H * H]; s_red: [NT / WAVE][H * H] floats
    __device__ __forceinline__ void store_partial(float* s_red, float* __restrict__ partial, int lane) {
        const int wave = threadIdx.x >> 6;
        float* mine = s_red + wave * H * H;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * (lane >> 4) + r, col = lane & 15;       // D layout of 16x16x4
            if constexpr (H == 16) {
                mine[row * 16 + col] = d[r];
            } else {
                const float other = __shfl(d[r], lane + 40, WAVE);      // D[row + 8][col + 8]
                if (lane < 32 && col < 8) mine[row * 8 + col] = d[r] + other;
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < H * H; t += NT) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NT / WAVE; ++w) v += s_red[w * H * H + t];
            partial[(size_t)blockIdx.x * H * H + t] = v;
        }
    }
};


}  // namespace crf
