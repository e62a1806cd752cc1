// Calibration: what does v_mfma_f32_32x32x2_f32 sustain, per SIMD (shader-clock ticks per MFMA) and chip-wide (TFLOP/s),
// with 1 or 2 waves per SIMD, and with an LDS read + a few VALU ops between the MFMAs (the shape of the real loops)?
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_peak.hip -o tools/ubench/mfma_peak && tools/ubench/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void __launch_bounds__(512) k(float *out, unsigned long long *ticks, int iters)
{
    __shared__ float lds[8192];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += blockDim.x) lds[i] = (float)(i & 7) * 0.001f;
    __syncthreads();
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = tid * 1e-3f, y = 1.0f + tid * 1e-4f, sc = 1.0001f, sh = 0.1f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {                       // pure MFMA, 4 independent accumulators
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
        } else if (MODE == 1) {                // one dependent chain
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        } else if (MODE == 4) {                // k-packed: 4 ds_read_b128 + 8 VALU feed 16 MFMAs
            const float4 *l4 = reinterpret_cast<const float4 *>(lds);
            const int o = (it * 64 + tid) & 255;
            float4 A0 = l4[o], A1 = l4[o + 256], B0 = l4[o + 512], B1 = l4[o + 768];
            __builtin_amdgcn_sched_barrier(0);
            B0.x = fmaxf(sc * B0.x + sh, 0.f); B0.y = fmaxf(sc * B0.y + sh, 0.f); B0.z = fmaxf(sc * B0.z + sh, 0.f); B0.w = fmaxf(sc * B0.w + sh, 0.f);
            B1.x = fmaxf(sc * B1.x + sh, 0.f); B1.y = fmaxf(sc * B1.y + sh, 0.f); B1.z = fmaxf(sc * B1.z + sh, 0.f); B1.w = fmaxf(sc * B1.w + sh, 0.f);
#define P2C_Q(e)                                                            \
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.e, B0.e, a0, 0, 0, 0); \
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.e, B1.e, a1, 0, 0, 0); \
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.e, B0.e, a2, 0, 0, 0); \
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.e, B1.e, a3, 0, 0, 0);
            P2C_Q(x) P2C_Q(y) P2C_Q(z) P2C_Q(w)
#undef P2C_Q
            it += 3;                           // 16 MFMAs this trip: keep the MFMA count per launch comparable
        } else if (MODE == 3) {                // same work, non-MFMA instructions spread into the shadow of EVERY MFMA
            const int o = (it * 64 + tid) & 4095;
            float b0 = fmaxf(sc * x + sh, 0.f);
            __builtin_amdgcn_sched_barrier(0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, b0, a0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            float an0 = lds[o], an1 = lds[o + 1024];
            float b1 = fmaxf(sc * y + sh, 0.f);
            __builtin_amdgcn_sched_barrier(0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, b1, a1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            float bn0 = lds[o + 2048], bn1 = lds[o + 3072];
            __builtin_amdgcn_sched_barrier(0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, b0, a2, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, b1, a3, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            x = an0 + bn0; y = an1 + bn1;
        } else {                               // the dW step: 4 LDS reads for the next step, transform, 4 MFMAs
            const int o = (it * 64 + tid) & 4095;
            float an0 = lds[o], an1 = lds[o + 1024], bn0 = lds[o + 2048], bn1 = lds[o + 3072];
            __builtin_amdgcn_sched_barrier(0);
            float b0 = fmaxf(sc * x + sh, 0.f), b1 = fmaxf(sc * y + sh, 0.f);
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, b0, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, b1, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, b0, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, b1, a3, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            x = an0 + bn0; y = an1 + bn1;
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    out[blockIdx.x * blockDim.x + tid] = s;
    if (tid == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int MODE>
static void run(const char *name, int threads, int grid, int iters)
{
    float *out; unsigned long long *ticks, h = 0;
    hipMalloc(&out, sizeof(float) * grid * threads); hipMalloc(&ticks, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(threads), 0, 0, out, ticks, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(threads), 0, 0, out, ticks, iters); hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
    const double mfmas_per_wave = 4.0 * iters, waves = (double)grid * threads / 64;
    const double flops = mfmas_per_wave * waves * 2.0 * 32 * 32 * 2;
    printf("%-34s threads %3d grid %4d: %7.1f ticks/MFMA(wave) | %8.3f ms | %6.1f TFLOP/s | tick rate %.3f GHz\n", name, threads, grid,
           (double)h / mfmas_per_wave, ms, flops / (ms * 1e-3) / 1e12, (double)h / (ms * 1e-3) / 1e9);
    hipFree(out); hipFree(ticks);
}

int main()
{
    const int it = 20000;
    run<0>("4 independent acc", 256, 256, it);      // 1 wave / SIMD
    run<0>("4 independent acc", 512, 256, it);      // 2 waves / SIMD
    run<0>("4 independent acc", 256, 1024, it);
    run<1>("1 dependent chain", 256, 256, it);
    run<1>("1 dependent chain", 512, 256, it);
    run<2>("dW step (4 ds_read + VALU + 4 MFMA)", 256, 256, it);
    run<2>("dW step (4 ds_read + VALU + 4 MFMA)", 512, 256, it);
    run<4>("k-packed (4 b128 + VALU per 16 MFMA)", 256, 256, it);
    run<4>("k-packed (4 b128 + VALU per 16 MFMA)", 512, 256, it);
    run<3>("dW step, interleaved", 256, 256, it);
    run<3>("dW step, interleaved", 512, 256, it);
    return 0;
}
