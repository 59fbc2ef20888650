// wave.hpp -- wave64 primitives for gfx950 (CDNA4).  Device code only.
//
// Everything here assumes a 64-lane wavefront and full EXEC (call from wave-uniform control
// flow only).  Scans use DPP row shifts + row broadcasts (the GFX9 cross-lane path that needs no
// LDS traffic), because the accumulator kernels keep the LDS pipe busy with atomics.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bsp {

typedef unsigned long long u64;
typedef unsigned int u32;

__device__ __forceinline__ int lane_id()
{
    return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// bits below / up to and including this lane
__device__ __forceinline__ u64 mask_lt(int lane) { return (1ull << lane) - 1ull; }
__device__ __forceinline__ u64 mask_le(int lane) { return (2ull << lane) - 1ull; }

// DPP controls (GFX9): row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_or_zero(int x)
{
    // lanes without a valid source, or rows masked out, receive `old` = 0
    return __builtin_amdgcn_update_dpp(0, x, CTRL, ROW_MASK, 0xF, false);
}

// inclusive prefix sum over the 64 lanes of a wave
__device__ __forceinline__ int wave_incl_scan(int x)
{
    x += dpp_or_zero<0x111, 0xF>(x);   // row_shr:1
    x += dpp_or_zero<0x112, 0xF>(x);   // row_shr:2
    x += dpp_or_zero<0x114, 0xF>(x);   // row_shr:4
    x += dpp_or_zero<0x118, 0xF>(x);   // row_shr:8   -> scan inside each row of 16
    x += dpp_or_zero<0x142, 0xA>(x);   // row_bcast:15 into rows 1 and 3
    x += dpp_or_zero<0x143, 0xC>(x);   // row_bcast:31 into rows 2 and 3
    return x;
}

__device__ __forceinline__ int wave_bcast(int x, int lane)   // lane must be wave-uniform
{
    return __builtin_amdgcn_readlane(x, lane);
}

__device__ __forceinline__ u64 wave_bcast64(u64 x, int lane)
{
    u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)x, lane);
    u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(x >> 32), lane);
    return ((u64)hi << 32) | lo;
}

__device__ __forceinline__ int wave_first(int x) { return __builtin_amdgcn_readfirstlane(x); }

// 64-bit inclusive scan for the (cold) device-wide scans; LDS-crossbar shuffles are fine there
__device__ __forceinline__ long long wave_incl_scan64(long long x)
{
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        long long y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    return x;
}

// orders this wave's LDS traffic for the compiler between phases that hand data from lane to
// lane through LDS.  DS operations of one wave execute in issue order, so no hardware barrier is
// needed; the fence only stops the compiler from moving accesses across the phase boundary.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Blocked scan of a slot array of 64*W words: lane l owns the W consecutive words [l*W, l*W+W)
// (vector LDS reads), does its own running popcount in registers, ONE wave scan joins the lanes,
// and the W prefixes go back as one vector store.  pre[t] = number of set bits in P[0..t).
// Cost is fixed per row (the arrays are zero beyond the live slots) and replaces a loop of
// ceil(n/64) dependent scan steps.
template <int W>
__device__ __forceinline__ int scan_blocked(const u32 *P, unsigned short *pre, int lane)
{
    u32 x[W];
#pragma unroll
    for (int k = 0; k < W; k++) x[k] = P[lane * W + k];
    int e[W];
    int run = 0;
#pragma unroll
    for (int k = 0; k < W; k++) {
        e[k] = run;
        run += __popc(x[k]);
    }
    const int inc = wave_incl_scan(run);
    const int base = inc - run;
#pragma unroll
    for (int k = 0; k < W; k++) pre[lane * W + k] = (unsigned short)(base + e[k]);
    return wave_bcast(inc, 63);
}

// Staging index swizzle: lanes write runs of ~W consecutive outputs, i.e. at a stride of ~W words
// (8-way bank conflicts at stride 8); XOR-ing the low 5 bits with the next 5 spreads any such
// stride over all 32 banks and keeps the final contiguous read-out conflict-free.
__device__ __forceinline__ int stage_swz(int x) { return x ^ ((x >> 5) & 31); }

template <int W>
__device__ __forceinline__ void clear_blocked(u32 *P, int lane)
{
#pragma unroll
    for (int k = 0; k < W; k++) P[lane * W + k] = 0u;
}

}  // namespace bsp
