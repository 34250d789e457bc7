// Grid-wide arrival barrier for kernels whose workgroups are ALL resident at once (the host checks the grid against the
// occupancy query), plus the write-through / L1-bypassing buffer accesses that carry data between workgroups inside one
// launch.  Used by the one-launch mean-field forward (crf.hip) and the small-level Linear + BatchNorm kernels
// (mlp_small.hip).
//
// Protocol (MI355X_MICROARCH.md, inter-workgroup visibility): a producer stores write-through (`sc1`), drains its stores
// (`s_waitcnt vmcnt(0)`), the workgroup arrives; consumers read what other workgroups produced with `sc1` loads (L1 is
// never refreshed by another CU's stores) after leaving the barrier.  Two levels: groups of blockIdx % 8 (one XCD under
// the dispatcher's round-robin placement -- for speed only), then one top counter; 1.0-1.7 us after the last arrival.
#pragma once
#include "common.hpp"

namespace crf {

constexpr int FW_LINE = 32;                   // words per 128-byte line
constexpr int FW_CNT = 0, FW_TOP = 8, FW_GEN = 9, FW_FAIL = 17, FW_EXIT = 18, FW_WORDS = 19 * FW_LINE;
// In-launch waits of one producer / consumer pair per job slot (gemm.hip mlp_small_bwd_jobs_kernel): behind the barrier words, per job slot
// (4) a block of WL_LINES lines -- WL_REPL replicas of the producers' count (a consumer polls replica index % WL_REPL: ~30 pollers per
// line instead of 1 000), WL_GROUPS exit-ticket lines and their top word (the last consumer out zeroes the block).
constexpr int WL_REPL = 32, WL_GROUPS = 16, WL_LINES = 64, WL_JOBS = 4, FW_WAIT = 19;
constexpr int FW_TOTAL_WORDS = (FW_WAIT + WL_JOBS * WL_LINES) * FW_LINE;
constexpr unsigned FW_SPIN_LIMIT = 1u << 21;  // ~1 s: a stranded workgroup gives up with a code instead of hanging

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 ld4_sc1(__amdgpu_buffer_rsrc_t r, int byte_off, int sbase) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, sbase, 16);    // aux 16 = sc1
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float4 ld4_buf(__amdgpu_buffer_rsrc_t r, int byte_off) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void st4_sc1(__amdgpu_buffer_rsrc_t r, int byte_off, float4 v) {
    const u32x4_t u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(u, r, byte_off, 0, 16);
}

// phase = 1, 2, ...: counters only grow within a launch (zero at entry: the previous launch's last workgroup out resets
// them, fused_exit_reset).  Returns false (for the whole workgroup) when the spin gave up.
// (Polling the generation word with a returning atomic, or a fresh word per phase, measured the same: the barrier costs
// 1.0-1.3 us after the last arrival either way -- profiles/r2a_fused_meanfield_stamps.txt.)
template <bool STAMP>
__device__ __forceinline__ bool fused_grid_sync(unsigned* ws, unsigned phase, unsigned n_in_group, unsigned n_groups,
                                                int* s_ok, unsigned long long* dbg, unsigned bid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its write-through stores
    __syncthreads();
    if (threadIdx.x == 0) {
        if constexpr (STAMP) dbg[(size_t)bid * 64 + 8 * phase + 4] = __builtin_amdgcn_s_memrealtime();
        const unsigned g = bid & 7u;
        unsigned* gen = ws + (FW_GEN + g) * FW_LINE;
        const unsigned old = __hip_atomic_fetch_add(ws + (FW_CNT + g) * FW_LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == n_in_group * phase) {
            const unsigned o2 = __hip_atomic_fetch_add(ws + FW_TOP * FW_LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (o2 + 1 == n_groups * phase) {
                if constexpr (STAMP) dbg[(size_t)bid * 64 + 8 * phase + 6] = __builtin_amdgcn_s_memrealtime();
                for (unsigned g2 = 0; g2 < n_groups; ++g2)
                    __hip_atomic_store(ws + (FW_GEN + g2) * FW_LINE, phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if constexpr (STAMP) dbg[(size_t)bid * 64 + 8 * phase + 5] = __builtin_amdgcn_s_memrealtime();
        int ok = 1;
        unsigned spins = 0;
        for (;;) {
            if (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= phase) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > FW_SPIN_LIMIT) {
                __hip_atomic_store(ws + FW_FAIL * FW_LINE, 0x100u | phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        *s_ok = ok;
    }
    __syncthreads();
    return *s_ok != 0;
}

// Last workgroup out zeroes the barrier words for the next launch (every workgroup has passed every barrier by then):
// no memset node in front of the kernel (4 us of stream time per forward).  ws must be zero before the FIRST launch.
// Group sizes for a launch of nblk workgroups numbered bid = 0..nblk-1 (group = bid % 8).
__device__ __forceinline__ void grid_sync_groups(unsigned nblk, unsigned bid, unsigned& n_in_group, unsigned& n_groups) {
    const unsigned grp = bid & 7u;
    n_in_group = nblk / 8u + (grp < (nblk & 7u) ? 1u : 0u);
    n_groups = nblk < 8u ? nblk : 8u;
}

__device__ __forceinline__ void fused_exit_reset(unsigned* ws, unsigned nblk, int T, unsigned bid) {
    if (threadIdx.x == 0 && T > 1) {
        // counted out in the barrier's two levels (word 1 of the group's line, then FW_EXIT): hundreds of workgroups that finish
        // together and count on ONE word are served one after the other (~7 ns each) at the end of the kernel
        unsigned n_in_group, n_groups;
        grid_sync_groups(nblk, bid, n_in_group, n_groups);
#ifdef FW_FLAT_EXIT_       // A/B: the round-3 form
        if (__hip_atomic_fetch_add(ws + FW_EXIT * FW_LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == nblk)
            for (int l = 0; l < FW_WORDS / FW_LINE; ++l)
                if (l != FW_FAIL) __hip_atomic_store(ws + l * FW_LINE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
#endif
        const unsigned old = __hip_atomic_fetch_add(ws + (FW_CNT + (bid & 7u)) * FW_LINE + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 != n_in_group) return;
        const unsigned old2 = __hip_atomic_fetch_add(ws + FW_EXIT * FW_LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old2 + 1 == n_groups) {
            for (int l = 0; l < FW_WORDS / FW_LINE; ++l)
                if (l != FW_FAIL) __hip_atomic_store(ws + l * FW_LINE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int g = 0; g < 8; ++g) __hip_atomic_store(ws + (FW_CNT + g) * FW_LINE + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

#ifndef LASTWG_GROUPS_
#define LASTWG_GROUPS_ 32
#endif
#ifndef LASTWG_FLAT_
#define LASTWG_FLAT_ 64
#endif
// ------------------------------------------------------------------ "the last workgroup finishes"
// A launch whose workgroups each leave a row of partial sums and whose LAST workgroup to finish adds the rows: the reduction
// launch behind it disappears (4-5 us of stream time each on the coarse levels) and no co-residency is needed -- nobody waits.
// Protocol as above: partial rows stored write-through (st1_sc1 / st4_sc1), drained, one ticket per workgroup; the workgroup that
// draws the last ticket reads the rows past L1 (ld4_sc1) in a fixed order (bitwise reproducible) and zeroes the ticket for the
// next launch.  ticket: LW_TICKET_WORDS zero device words per stream (ops._ticket).
__device__ __forceinline__ void st1_sc1(__amdgpu_buffer_rsrc_t r, int byte_off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, byte_off, 0, 16);
}
// A burst of tickets on ONE word is served one after the other (~7 ns each from eight XCDs: 1 280 workgroups that finish together
// cost 9 us, measured on uvstats_kernel<8>), so launches of more than LW_FLAT workgroups draw in two levels: workgroup b takes a
// ticket of group b % LW_GROUPS (one 128-byte line each), the last of a group one of the top word.
// ticket: LW_TICKET_WORDS zero words (all left zero).
constexpr unsigned LW_GROUPS = LASTWG_GROUPS_, LW_FLAT = LASTWG_FLAT_, LW_TICKET_WORDS = (1 + 32) * FW_LINE;
static_assert(LW_GROUPS <= 32, "ticket lines");
// bid: this workgroup's index among the nblk that draw on `ticket` (a launch that hosts a second job numbers each job's workgroups itself)
__device__ __forceinline__ bool last_workgroup_of(unsigned* ticket, unsigned nblk, int* s_flag, const unsigned bid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        int last = 0;
        if (nblk <= LW_FLAT) {
            const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = old + 1 == nblk;
        } else {
            const unsigned g = bid % LW_GROUPS, n_in_group = nblk / LW_GROUPS + (g < nblk % LW_GROUPS ? 1u : 0u);
            unsigned* gt = ticket + (1 + g) * FW_LINE;
            const unsigned old = __hip_atomic_fetch_add(gt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == n_in_group) {
                __hip_atomic_store(gt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned old2 = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = old2 + 1 == LW_GROUPS;
            }
        }
        if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_flag = last;
    }
    __syncthreads();
    return *s_flag != 0;
}
__device__ __forceinline__ bool last_workgroup(unsigned* ticket, unsigned nblk, int* s_flag) {
    return last_workgroup_of(ticket, nblk, s_flag, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
}
// In the last workgroup (every thread of it calls; the first NT take part): s_tot[slot] = sum_b partial[b][slot] in double for
// nslots (a multiple of 4, nslots / 4 dividing NT) slots of nblk rows.  Thread (group g, quad) adds rows g, g + groups, ... in
// that order with LASTWG_INFLIGHT_ 16-byte loads in flight, then the groups are added in group order.  s_buf: 4 NT doubles.
#ifndef LASTWG_INFLIGHT_
#define LASTWG_INFLIGHT_ 8
#endif
template <int NT>
__device__ __forceinline__ void sum_partial_rows_f64(__amdgpu_buffer_rsrc_t pr, int nblk, int nslots, double* s_buf, double* s_tot) {
    constexpr int UF = LASTWG_INFLIGHT_;
    const int nq = nslots >> 2, groups = NT / nq, t = threadIdx.x;
    if (t < NT) {
        const int grp = t / nq, quad = t % nq;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        for (int b = grp; b < nblk; b += UF * groups) {
            float4 v[UF];
#pragma unroll
            for (int u = 0; u < UF; ++u) {
                const int bb = b + u * groups;
                v[u] = bb < nblk ? ld4_sc1(pr, (bb * nslots + 4 * quad) * 4, 0) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < UF; ++u) { a0 += (double)v[u].x; a1 += (double)v[u].y; a2 += (double)v[u].z; a3 += (double)v[u].w; }
        }
        s_buf[t] = a0; s_buf[NT + t] = a1; s_buf[2 * NT + t] = a2; s_buf[3 * NT + t] = a3;
    }
    __syncthreads();
    if (t < nslots) {
        const int quad = t >> 2, e = t & 3;
        double v = s_buf[e * NT + quad];
        for (int g2 = 1; g2 < groups; ++g2) v += s_buf[e * NT + g2 * nq + quad];
        s_tot[t] = v;
    }
    __syncthreads();
}

}  // namespace crf
