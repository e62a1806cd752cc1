// common.h -- shared helpers for the gfx950 kernels (wave64 only; no other target is supported).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/p2c_hip.h"

#define P2C_WAVE 64
#define P2C_STAT_SLOTS 64   // fp64 accumulator rows every per-channel reduction is spread over (workgroup b -> row b % 64)

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

// Sum over the lanes of a wave that share lane % STRIDE (STRIDE a power of two <= 16), every lane ending with the sum of ITS class:
// rotations inside the 16-lane row on the DPP path (row_ror: a VALU operand modifier), the two cross-row steps with gfx950's
// v_permlane16_swap / v_permlane32_swap (VALU) - no ds_bpermute round trip (~100 cycles each; __shfl_xor always takes that route).
// With both operands the same register a swap leaves {rows 0,0,2,2 | rows 1,1,3,3} (16) resp. {lanes 0-31 twice | lanes 32-63 twice} (32)
// in the two results: their sum is the xor-16 / xor-32 exchange sum.
template <int STRIDE>
__device__ __forceinline__ float p2c_wave_class_sum_f32(float v)
{
    static_assert(STRIDE == 1 || STRIDE == 2 || STRIDE == 4 || STRIDE == 8 || STRIDE == 16, "");
    if (STRIDE <= 1) v += __int_as_float(p2c_dpp<0x121>(__float_as_int(v)));      // row_ror:1
    if (STRIDE <= 2) v += __int_as_float(p2c_dpp<0x122>(__float_as_int(v)));      // row_ror:2
    if (STRIDE <= 4) v += __int_as_float(p2c_dpp<0x124>(__float_as_int(v)));      // row_ror:4
    if (STRIDE <= 8) v += __int_as_float(p2c_dpp<0x128>(__float_as_int(v)));      // row_ror:8
    {
        const unsigned u = (unsigned)__float_as_int(v);
        const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        v = __int_as_float((int)r[0]) + __int_as_float((int)r[1]);
    }
    {
        const unsigned u = (unsigned)__float_as_int(v);
        const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        v = __int_as_float((int)r[0]) + __int_as_float((int)r[1]);
    }
    return v;
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

// Pre-activation of a folded first layer (3 input channels + zero pad, see bn.hip p2c_input_moments_f32): the SAME
// expression wherever it is recomputed (forward staging, backward staging), so the ReLU decisions agree everywhere.
__device__ __forceinline__ float p2c_l0_preact(float wx, float wy, float wz, float b, float x, float y, float z)
{
    return __builtin_fmaf(wz, z, __builtin_fmaf(wy, y, wx * x)) + b;
}

// Counter-based dropout bits.  Stateless, so the forward and the backward kernels regenerate the same mask from (seed, element index)
// instead of storing M x C bytes.  Four consecutive elements share ONE hash (13 integer operations - as many as the bf16 split of an
// element; hashing every element made the heads' forward and backward VALU-bound): element e = row*C + col keeps when byte (e & 3) of
// hash(seed, e >> 2) >= the top byte of the threshold, i.e. the drop probability is taken in steps of 1/256 (exact for the reference's 0.5).
__device__ __forceinline__ uint32_t p2c_hash32(uint32_t seed_lo, uint32_t seed_hi, uint32_t idx)
{
    uint32_t x = idx * 0x9E3779B1u ^ seed_lo;
    x ^= x >> 16; x *= 0x85EBCA6Bu;
    x ^= x >> 13; x *= 0xC2B2AE35u;
    x ^= x >> 16; x += seed_hi * 0x27D4EB2Fu;
    x ^= x >> 15; x *= 0x2C1B3C6Du;
    x ^= x >> 12;
    return x;
}
__device__ __forceinline__ bool p2c_keep4(uint32_t h, int j, uint32_t thr) { return ((h >> (8 * j)) & 0xFFu) >= (thr >> 24); }
__device__ __forceinline__ bool p2c_keep(uint32_t seed_lo, uint32_t seed_hi, uint32_t e, uint32_t thr)
{
    return p2c_keep4(p2c_hash32(seed_lo, seed_hi, e >> 2), (int)(e & 3u), thr);
}
// The hashed mask compares ONE BYTE with the top byte of the threshold, so only drop probabilities that are multiples of 1/256 are realised
// exactly; with any other p the keep-scale 1/(1-p) would no longer match the realised keep rate and the expectation would be biased.  The
// entry points that take a hashed mask refuse such a scale (P2C_EINVAL) instead of training on a biased dropout.
static inline bool p2c_drop_scale_representable(float scale)
{
    if (!(scale >= 1.0f)) return false;
    const double p256 = (1.0 - 1.0 / (double)scale) * 256.0;
    const double r = p256 - (double)(long long)(p256 + 0.5);
    return (r < 0 ? -r : r) < 1e-4 && p256 < 255.5;
}
static inline uint32_t p2c_drop_threshold(float scale)     // scale = 1/(1-p)  ->  p * 2^32
{
    double p = 1.0 - 1.0 / (double)scale;
    if (p < 0) p = 0;
    if (p > 0.999999) p = 0.999999;
    return (uint32_t)(p * 4294967296.0);
}

// -DP2C_TRACE (tools/fused_trace.py builds one source file that way into a throw-away library; never the product):
// workgroup 0 stamps the shader clock at the phase boundaries of a few iterations of a ping-pong kernel (variables
// `tid`, `half`, `it` of the kernel), so the overlap of the two halves can be read off directly.
#ifdef P2C_TRACE
#define P2C_TR_IT 12
#define P2C_TR_PT 8
__device__ unsigned long long p2c_trace_buf[2][P2C_TR_IT][P2C_TR_PT];
#define P2C_TR(pt)                                                                                   \
    do {                                                                                             \
        if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && it < P2C_TR_IT) p2c_trace_buf[half][it][pt] = __builtin_readcyclecounter(); \
    } while (0)
// every workgroup: shader clock (s_memtime) and the constant 100 MHz counter (s_memrealtime) at its start (e = 0) and end (e = 1):
// per-workgroup durations (is there a tail?) and the clock the kernel actually ran at
__device__ unsigned long long p2c_trace_wg[1024][4];
__device__ unsigned long long p2c_trace_wg_mid[1024][8];      // shader clock when the main loop starts / ends (prologue and flush lengths), [2..7]: prologue stages
#define P2C_TR_WG_MID(e) do { if (threadIdx.x == 0 && blockIdx.x < 1024) p2c_trace_wg_mid[blockIdx.x][e] = __builtin_readcyclecounter(); } while (0)
#define P2C_TR_WG(e)                                                                                 \
    do {                                                                                             \
        if (threadIdx.x == 0 && blockIdx.x < 1024) {                                                 \
            p2c_trace_wg[blockIdx.x][2 * (e)] = __builtin_readcyclecounter();                        \
            p2c_trace_wg[blockIdx.x][2 * (e) + 1] = __builtin_amdgcn_s_memrealtime();                \
        }                                                                                            \
    } while (0)
extern "C" int p2c_trace_read(void *host_out)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(p2c_trace_buf), sizeof(unsigned long long) * 2 * P2C_TR_IT * P2C_TR_PT);
}
extern "C" int p2c_trace_read_wg(void *host_out)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(p2c_trace_wg), sizeof(unsigned long long) * 1024 * 4);
}
extern "C" int p2c_trace_read_wg_mid(void *host_out)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(p2c_trace_wg_mid), sizeof(unsigned long long) * 1024 * 8);
}
#else
#define P2C_TR(pt) do { } while (0)
#define P2C_TR_WG(e) do { } while (0)
#define P2C_TR_WG_MID(e) do { } while (0)
#endif
