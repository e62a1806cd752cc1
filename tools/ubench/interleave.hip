// What do a wave's OWN non-MFMA instructions cost when they are interleaved with its fp32 MFMA stream (v_mfma_f32_32x32x2_f32, 64 cycles
// of pipe time each)?  Each wave runs   { MFMA ; K x op } x 8   per iteration with scheduling barriers between the groups; reported: cycles
// per MFMA.  64 = the ops are hidden in the MFMA's shadow.  WPS = waves per SIMD (256 or 512 threads per workgroup, one workgroup per CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int K, int WPS>   // KIND 0 VALU fma, 1 ds_write_b128, 2 ds_read_b128, 3 global_load x4, 4 global_store x4, 5 mixed (2 VALU + 1 of each of 1..4 per 6)
__global__ void __launch_bounds__(256 * WPS) k(float *buf, unsigned long long *ticks, int iters)
{
    __shared__ f32x4 lds[2048];
    const int tid = threadIdx.x;
    lds[tid] = f32x4{1.f, 2.f, 3.f, 4.f};
    lds[tid + 512] = lds[tid];
    __syncthreads();
    f32x16 a0 = {0}, a1 = {0};
    float x = tid * 1e-3f, y = 1.f;
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = tid * 0.25f + j;
    f32x4 r[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f32x4 *g = (f32x4 *)buf + (size_t)blockIdx.x * 4096 + tid;        // 64 KB per workgroup: stays in L2
    const unsigned la = (unsigned)(size_t)(lds) + tid * 16;    // LDS byte address of this lane's slot (generic -> local: low 32 bits)
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            // everything in the loop is `asm volatile`: strict program order, nothing hoisted or sunk by the IR passes
            if (m & 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(a1) : "v"(x), "v"(y));
            else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(a0) : "v"(x), "v"(y));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < K; ++u) {
                const int q = (m * K + u);
                int kind = KIND;
                if (KIND == 5) { const int s = q % 6; kind = s < 2 ? 0 : s - 1; }
                if (kind == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q & 7]) : "v"(y), "v"(x));
                // memory ops as inline asm: the compiler places no s_waitcnt for them, so what is measured is issue cost + queue
                // back-pressure, not the latency of an immediate use (one wait for everything after the loop)
                else if (kind == 1) asm volatile("ds_write_b128 %0, %1" ::"v"(la + 8192 * (q & 1)), "v"(r[0]) : "memory");
                else if (kind == 2) asm volatile("ds_read_b128 %0, %1" : "+v"(r[1 + (q & 1)]) : "v"(la + 1024 * (q & 7)) : "memory");
                else if (kind == 3) asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(r[3]) : "v"(g + 512 * (q & 3)) : "memory");
                else if (kind == 4) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(g + 2048 + 512 * (q & 3)), "v"(r[0]) : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_nop 15\n s_nop 15" ::: "memory");   // before the compiler reuses the landing registers
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += a0[j] + a1[j];
    for (int j = 0; j < 8; ++j) s += v[j];
    for (int j = 0; j < 4; ++j) s += r[j][0] + r[j][1] + r[j][2] + r[j][3];
    buf[(size_t)256 * 4096 * 4 + blockIdx.x * 512 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int KIND, int K, int WPS>
static double run(int iters)
{
    float *buf; unsigned long long *ticks, h = 0;
    (void)hipMalloc(&buf, sizeof(float) * (256 * 4096 * 4 + 256 * 512)); (void)hipMalloc(&ticks, 16);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<KIND, K, WPS>), dim3(256), dim3(256 * WPS), 0, 0, buf, ticks, iters);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("KIND %d K %d WPS %d: %s\n", KIND, K, WPS, hipGetErrorString(e)); fflush(stdout); }
    (void)hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
    (void)hipFree(buf); (void)hipFree(ticks);
    return (double)h / (iters * 8.0);
}

template <int KIND, int WPS>
static void row(const char *name)
{
    const int it = 1000;
    printf("%s ...\n", name); fflush(stdout);
    printf("%-22s %d wave/SIMD | K=0 %6.1f  K=1 %6.1f  K=2 %6.1f  K=3 %6.1f  K=4 %6.1f  K=6 %6.1f  K=8 %6.1f  K=12 %6.1f  K=16 %6.1f\n", name, WPS,
           run<KIND, 0, WPS>(it), run<KIND, 1, WPS>(it), run<KIND, 2, WPS>(it), run<KIND, 3, WPS>(it), run<KIND, 4, WPS>(it), run<KIND, 6, WPS>(it),
           run<KIND, 8, WPS>(it), run<KIND, 12, WPS>(it), run<KIND, 16, WPS>(it));
}

int main()
{
    printf("cycles per MFMA of one wave, {MFMA; K ops} interleaved in the SAME wave\n");
    row<0, 1>("VALU fma");
    row<1, 1>("ds_write_b128");
    row<2, 1>("ds_read_b128");
    row<3, 1>("global_load x4 (L2)");
    row<4, 1>("global_store x4");
    row<5, 1>("mixed 2V+W+R+L+S");
    row<0, 2>("VALU fma");
    row<1, 2>("ds_write_b128");
    row<2, 2>("ds_read_b128");
    row<3, 2>("global_load x4 (L2)");
    row<4, 2>("global_store x4");
    row<5, 2>("mixed 2V+W+R+L+S");
    return 0;
}
