// Does a wave streaming back-to-back MFMAs starve the other wave on its SIMD?  512-thread workgroups: waves 0-3 run an
// MFMA stream (dependent chain or 4 independent accumulators), waves 4-7 (same SIMDs) run a fixed amount of VALU / LDS /
// global-store work and time it; compare with the MFMA half idle.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WORK, int MF>   // WORK 0: VALU chain, 1: LDS write+read, 2: global stores ; MF 0: idle partner, 1: dependent MFMA chain, 2: 4 accumulators
__global__ void __launch_bounds__(512) k(float *out, unsigned long long *ticks, int iters)
{
    __shared__ float lds[8192];
    const int tid = threadIdx.x, half = tid >> 8;
    lds[tid] = tid; lds[tid + 512] = tid;
    __syncthreads();
    if (half == 0) {
        f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
        float x = tid * 1e-3f, y = 1.f;
        unsigned long long ta = __builtin_readcyclecounter();
        if (MF) {
            for (int it = 0; it < iters * 8; ++it) {
                if (MF >= 3) {                      // MFMA chain that yields the issue port: s_nop between the MFMAs
#define P2C_YIELD()                                                        \
    do {                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                 \
        if (MF == 3) { asm volatile("s_nop 15"); asm volatile("s_nop 15"); asm volatile("s_nop 15"); }                      \
        if (MF == 4) { asm volatile("s_nop 15"); asm volatile("s_nop 15"); asm volatile("s_nop 15"); asm volatile("s_nop 7"); } \
        if (MF == 5) { __builtin_amdgcn_s_sleep(1); }                      \
        if (MF == 6) { asm volatile("s_nop 15"); }                          \
        if (MF == 7) { asm volatile("s_nop 15"); asm volatile("s_nop 15"); } \
        if (MF == 8) { asm volatile("s_nop 7"); }                           \
        if (MF == 9) { asm volatile("s_nop 3"); }                           \
        if (MF == 10) { asm volatile("s_nop 0"); }                          \
        __builtin_amdgcn_sched_barrier(0);                                 \
    } while (0)
                    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0); P2C_YIELD();
                    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0); P2C_YIELD();
                    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0); P2C_YIELD();
                    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0); P2C_YIELD();
                } else if (MF == 1) {
                    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                } else {
                    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
                    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
                }
            }
        }
        unsigned long long tb = __builtin_readcyclecounter();
        float s = 0.f;
        for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
        out[blockIdx.x * 512 + tid] = s;
        if (tid == 0 && blockIdx.x == 0) ticks[1] = tb - ta;
    } else {
        const int t = tid & 255;
        float v = t * 0.5f, w = 1.0001f;
        unsigned long long t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
            if (WORK == 0) {
#pragma unroll
                for (int u = 0; u < 32; ++u) v = v * w + 0.25f;
            } else if (WORK == 1) {
#pragma unroll
                for (int u = 0; u < 8; ++u) { lds[1024 + t + 256 * (u & 3)] = v; v += lds[1024 + ((t + 64 * u) & 1023)]; }
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u) out[(size_t)(blockIdx.x * 256 + t) + (size_t)gridDim.x * 512 + (size_t)(u + 8 * (it & 63)) * gridDim.x * 256] = v + u;
            }
        }
        unsigned long long t1 = __builtin_readcyclecounter();
        out[blockIdx.x * 512 + tid] = v;
        if (t == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
    }
}

template <int WORK, int MF>
static void run(const char *name, int iters)
{
    float *out; unsigned long long *ticks, h[2] = {0, 0};
    (void)hipMalloc(&out, sizeof(float) * 256 * 512 * 600); (void)hipMalloc(&ticks, 16);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<WORK, MF>), dim3(256), dim3(512), 0, 0, out, ticks, iters);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
    printf("%-44s %10.1f ticks per iteration of the non-MFMA wave | MFMA wave: %6.1f ticks per MFMA\n", name, (double)h[0] / iters,
           MF ? (double)h[1] / (iters * 32.0) : 0.0);
    (void)hipFree(out); (void)hipFree(ticks);
}

int main()
{
    const int it = 2000;
    run<0, 0>("VALU | idle", it);
    run<0, 1>("VALU | MFMA chain", it);
    run<0, 10>("VALU | MFMA + s_nop 0", it);
    run<0, 9>("VALU | MFMA + s_nop 3", it);
    run<0, 8>("VALU | MFMA + s_nop 7", it);
    run<0, 6>("VALU | MFMA + s_nop 15", it);
    run<0, 7>("VALU | MFMA + 2 x s_nop 15", it);
    run<0, 3>("VALU | MFMA + 3 x s_nop 15", it);
    run<1, 8>("LDS  | MFMA + s_nop 7", it);
    run<1, 6>("LDS  | MFMA + s_nop 15", it);
    run<2, 6>("STORE| MFMA + s_nop 15", it);
    return 0;
}
