// common.h -- shared helpers for the gfx950 kernels (wave64 only; no other target is supported).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/p2c_hip.h"

#define P2C_WAVE 64

#define P2C_LAUNCH_CHECK()                         \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

static inline int p2c_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- wave-level helpers (DPP; every lane of each 16-lane row ends with the row result) -----------

template <int CTRL>
__device__ __forceinline__ int p2c_dpp(int v)
{
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}

// max over each row of 16 lanes, result in all 16 lanes (signed int compare)
__device__ __forceinline__ int p2c_row16_max_i32(int v)
{
    v = max(v, p2c_dpp<0xB1>(v));   // quad_perm [1,0,3,2]
    v = max(v, p2c_dpp<0x4E>(v));   // quad_perm [2,3,0,1]
    v = max(v, p2c_dpp<0x141>(v));  // row_half_mirror
    v = max(v, p2c_dpp<0x140>(v));  // row_mirror
    return v;
}

// max over the 64 lanes of the wave, uniform result
__device__ __forceinline__ int p2c_wave_max_i32(int v)
{
    v = p2c_row16_max_i32(v);
    int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return max(max(a, b), max(c, d));
}

__device__ __forceinline__ float p2c_wave_sum_f32(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double p2c_wave_sum_f64(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
