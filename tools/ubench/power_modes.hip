// Steady-state loops for power measurements (tools/power_run.py polls rocm-smi while one of these runs for `seconds`):
//   mode 0: fp32 MFMA stream (v_mfma_f32_32x32x2_f32), 1 wave per SIMD      mode 1: the same, 2 waves per SIMD
//   mode 2: bf16 MFMA stream (v_mfma_f32_32x32x16_bf16), 1 wave per SIMD     mode 3: HBM copy (read 512 MB + write 512 MB per pass)
//   mode 4: VALU fma stream, 2 waves per SIMD                                  mode 5: HBM read only (512 MB per pass)
//   mode 6: every CU occupied by 8 waves that only sleep (s_sleep): the clocked-but-idle power      mode 7: LDS read stream (ds_read_b128)
//   mode 8: fp32 MFMA stream whose operands change every instruction (one VALU op per MFMA keeps them moving): data-dependent switching
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ void __launch_bounds__(512) k(float *buf, int iters, size_t n4)
{
    const int tid = threadIdx.x;
    if (MODE <= 2) {
        f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
        float x = tid * 1e-3f, y = 1.f;
        bf16x8 p, q;
        for (int i = 0; i < 8; ++i) { p[i] = (__bf16)(tid * 1e-3f); q[i] = (__bf16)1.f; }
        for (int it = 0; it < iters; ++it) {
            if (MODE == 2) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p, q, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p, q, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p, q, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p, q, a3, 0, 0, 0);
            } else {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
        if (s == 12345.f) buf[tid] = s;
    } else if (MODE == 8) {
        f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
        float x = tid * 1.37e-3f + 0.5f, y = 1.f - tid * 0.77e-3f;
        for (int it = 0; it < iters; ++it) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0); x = x * 1.000173f + 0.3711f;
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0); y = y * 0.999871f - 0.2913f;
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0); x = x * 0.999613f + 0.1177f;
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a3, 0, 0, 0); y = y * 1.000291f - 0.4421f;
        }
        float s = 0.f;
        for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
        if (s == 12345.f) buf[tid] = s;
    } else if (MODE == 6) {
        for (int it = 0; it < iters * 4; ++it) __builtin_amdgcn_s_sleep(127);
    } else if (MODE == 7) {
        __shared__ f32x4 lds[4096];
        lds[tid] = f32x4{1.f, 2.f, 3.f, 4.f}; lds[tid + 512] = lds[tid];
        __syncthreads();
        f32x4 acc = {0, 0, 0, 0};
        for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += lds[(tid + 64 * j + it) & 1023];
        }
        if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) buf[tid] = acc[0];
    } else if (MODE == 4) {
        float v[8];
        for (int j = 0; j < 8; ++j) v[j] = tid + j;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(v[j], 1.0001f, 0.25f);
        float s = 0.f;
        for (int j = 0; j < 8; ++j) s += v[j];
        if (s == 12345.f) buf[tid] = s;
    } else {
        const f32x4 *src = (const f32x4 *)buf;
        f32x4 *dst = (f32x4 *)buf + n4;
        f32x4 acc = {0, 0, 0, 0};
        for (size_t i = (size_t)blockIdx.x * blockDim.x + tid; i < n4; i += (size_t)gridDim.x * blockDim.x) {
            const f32x4 v = __builtin_nontemporal_load(src + i);
            if (MODE == 3) __builtin_nontemporal_store(v, dst + i); else acc += v;
        }
        if (MODE == 5 && acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) buf[tid] = acc[0];
    }
}

int main(int argc, char **argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const double seconds = argc > 2 ? atof(argv[2]) : 5.0;
    const size_t n4 = (size_t)32 << 20;                      // 512 MB of float4
    float *buf; (void)hipMalloc(&buf, n4 * 16 * 2);
    (void)hipMemset(buf, 0, n4 * 16 * 2);
    const int iters = 20000;
    auto launch = [&]() {
        switch (mode) {
        case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, buf, iters, n4); break;
        case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, buf, iters, n4); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, buf, iters, n4); break;
        case 3: hipLaunchKernelGGL(k<3>, dim3(2048), dim3(512), 0, 0, buf, iters, n4); break;
        case 4: hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, buf, iters * 8, n4); break;
        case 6: hipLaunchKernelGGL(k<6>, dim3(256), dim3(512), 0, 0, buf, iters, n4); break;
        case 8: hipLaunchKernelGGL(k<8>, dim3(256), dim3(256), 0, 0, buf, iters, n4); break;
        case 7: hipLaunchKernelGGL(k<7>, dim3(256), dim3(512), 0, 0, buf, iters, n4); break;
        default: hipLaunchKernelGGL(k<5>, dim3(2048), dim3(512), 0, 0, buf, iters, n4); break;
        }
    };
    launch(); (void)hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    long n = 0; double el = 0;
    while (el < seconds) {
        for (int i = 0; i < 4; ++i) launch();
        n += 4; (void)hipDeviceSynchronize();
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    const double per = el / n;
    if (mode == 8) {
        printf("mode 8: %.3f ms per launch, %.1f TFLOP/s (operands change every MFMA)\n", per * 1e3, 256.0 * 4 * iters * 4.0 * 4096.0 / per / 1e12);
    } else if (mode <= 2) {
        const double waves = 256.0 * (mode == 1 ? 8 : 4), flops = waves * iters * 4.0 * (mode == 2 ? 32768.0 : 4096.0);
        printf("mode %d: %.3f ms per launch, %.1f TFLOP/s\n", mode, per * 1e3, flops / per / 1e12);
    } else if (mode == 6) {
        printf("mode 6: %.3f ms per launch (sleeping waves on every CU)\n", per * 1e3);
    } else if (mode == 7) {
        printf("mode 7: %.3f ms per launch, %.1f TB/s of LDS reads\n", per * 1e3, 256.0 * 512 * 16.0 * 8 * iters * 4 / per / 1e12);
    } else if (mode == 4) {
        printf("mode 4: %.3f ms per launch, %.2f T VALU lane-fma/s\n", per * 1e3, 256.0 * 512 * 8.0 * iters * 8 / per / 1e12);
    } else {
        printf("mode %d: %.3f ms per launch, %.0f GB/s\n", mode, per * 1e3, n4 * 16.0 * (mode == 3 ? 2 : 1) / per / 1e9);
    }
    return 0;
}
