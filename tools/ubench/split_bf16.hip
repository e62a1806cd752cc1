// Feasibility probe for the NEXT round: can the fp32 layers run on the bf16 matrix pipe (16x the fp32 MFMA rate on
// gfx950) by splitting every fp32 operand into bf16 pieces, without giving up fp32-level accuracy?
//   x = hi + mid + lo, each a bf16 (8 significant bits): hi+mid carries 16 bits, hi+mid+lo 24.
//   2 pieces, 3 MFMAs per k-step  (hh, hm, mh)                    -> error ~2^-16 per product
//   3 pieces, 6 MFMAs per k-step  (hh, hm, mh, hl, lh, mm)        -> error ~2^-22..2^-24 per product (fp32 class)
// Part 1 (accuracy): one wave computes a 32 x 32 block of C = A B^T for K = 128 with v_mfma_f32_32x32x16_bf16 in the four variants
//   (1 piece, 2 pieces, 3 pieces, and the fp32 MFMA the product uses today); the host compares with float64.
// Part 2 (rate): ticks per MFMA of v_mfma_f32_32x32x16_bf16 from registers, 1 and 2 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/split_bf16.hip -o tools/ubench/split_bf16 && tools/ubench/split_bf16
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ __bf16 to_bf16_rn(float x) { return (__bf16)x; }

// mode 0: fp32 MFMA (32x32x2)   1: bf16 x1   2: bf16 x2 (3 MFMAs)   3: bf16 x3 (6 MFMAs)
template <int MODE>
__global__ void __launch_bounds__(64) acc_kernel(const float *A, const float *B, float *C, int K)
{
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    const float *a = A + (size_t)(blockIdx.x * 32 + i) * K, *b = B + (size_t)(blockIdx.y * 32 + i) * K;
    f32x16 acc = {0};
    if (MODE == 0) {
        for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k + h], b[k + h], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 16) {
            bf16x8 ah, am, al, bh, bm, bl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = a[k + 8 * h + e], y = b[k + 8 * h + e];
                const __bf16 xh = to_bf16_rn(x), yh = to_bf16_rn(y);
                const float xr = x - (float)xh, yr = y - (float)yh;
                const __bf16 xm = to_bf16_rn(xr), ym = to_bf16_rn(yr);
                ah[e] = xh; am[e] = xm; al[e] = to_bf16_rn(xr - (float)xm);
                bh[e] = yh; bm[e] = ym; bl[e] = to_bf16_rn(yr - (float)ym);
            }
            // small terms first
            if (MODE >= 3) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
            }
            if (MODE >= 2) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        }
    }
    for (int r = 0; r < 16; ++r) {
        const int row = blockIdx.x * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, col = blockIdx.y * 32 + i;
        C[(size_t)row * (gridDim.y * 32) + col] = acc[r];
    }
}

__global__ void __launch_bounds__(512) rate_kernel(float *out, unsigned long long *ticks, int iters)
{
    bf16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(threadIdx.x * 1e-3f + e); y[e] = (__bf16)(1.f + e * 0.01f); }
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}

int main()
{
    const int M = 256, N = 128, K = 128;
    float *hA = (float *)malloc(M * K * 4), *hB = (float *)malloc(N * K * 4), *hC = (float *)malloc(M * N * 4);
    srand(1);
    auto rnd = []() { float u = 0.f; for (int i = 0; i < 12; ++i) u += rand() / (float)RAND_MAX; return u - 6.f; };   // ~N(0,1)
    for (int i = 0; i < M * K; ++i) { float v = rnd(); hA[i] = v > 0.f ? v : 0.f; }           // post-ReLU activations
    for (int i = 0; i < N * K; ++i) hB[i] = 0.1f * rnd();                                      // weights
    float *dA, *dB, *dC;
    hipMalloc(&dA, M * K * 4); hipMalloc(&dB, N * K * 4); hipMalloc(&dC, M * N * 4);
    hipMemcpy(dA, hA, M * K * 4, hipMemcpyHostToDevice); hipMemcpy(dB, hB, N * K * 4, hipMemcpyHostToDevice);
    const char *names[4] = {"fp32 MFMA 32x32x2          ", "bf16 x1 (1 MFMA / 16 k)     ", "bf16 x2 (3 MFMAs / 16 k)    ", "bf16 x3 (6 MFMAs / 16 k)    "};
    for (int mode = 0; mode < 4; ++mode) {
        dim3 g(M / 32, N / 32);
        if (mode == 0) hipLaunchKernelGGL(acc_kernel<0>, g, dim3(64), 0, 0, dA, dB, dC, K);
        if (mode == 1) hipLaunchKernelGGL(acc_kernel<1>, g, dim3(64), 0, 0, dA, dB, dC, K);
        if (mode == 2) hipLaunchKernelGGL(acc_kernel<2>, g, dim3(64), 0, 0, dA, dB, dC, K);
        if (mode == 3) hipLaunchKernelGGL(acc_kernel<3>, g, dim3(64), 0, 0, dA, dB, dC, K);
        hipMemcpy(hC, dC, M * N * 4, hipMemcpyDeviceToHost);
        double emax = 0, esum = 0, scale = 0;
        for (int m = 0; m < M; ++m)
            for (int n = 0; n < N; ++n) {
                double ref = 0, mag = 0;
                for (int k = 0; k < K; ++k) { ref += (double)hA[m * K + k] * hB[n * K + k]; mag += fabs((double)hA[m * K + k] * hB[n * K + k]); }
                const double e = fabs(hC[m * N + n] - ref) / mag;                               // error relative to sum |a_k b_k|
                emax = e > emax ? e : emax; esum += e; scale += mag;
            }
        printf("%s  max err / sum|ab| = %.3e   mean = %.3e\n", names[mode], emax, esum / (M * N));
    }
    unsigned long long *dt, ht; float *dout;
    hipMalloc(&dt, 8); hipMalloc(&dout, 256 * 512 * 4);
    for (int threads = 256; threads <= 512; threads += 256) {
        const int iters = 4096;
        hipLaunchKernelGGL(rate_kernel, dim3(256), dim3(threads), 0, 0, dout, dt, iters);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); hipLaunchKernelGGL(rate_kernel, dim3(256), dim3(threads), 0, 0, dout, dt, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&ht, dt, 8, hipMemcpyDeviceToHost);
        const double flops = 256.0 * (threads / 64) * iters * 4 * 2.0 * 32 * 32 * 16;
        printf("v_mfma_f32_32x32x16_bf16, %d waves/SIMD: %.1f ticks per MFMA per wave, %.0f TFLOP/s chip-wide\n", threads / 256, (double)ht / (iters * 4), flops / ms / 1e9);
    }
    return 0;
}
