// Issue rate of the vector instructions the one-pass fitting kernel's stream phase is made of (csrc/fit.hip): cycles per wave64 instruction
// and SIMD with 1, 2 and 4 waves per SIMD, each wave running a stream of INDEPENDENT instructions of one kind.
//   hipcc -O3 --offload-arch=gfx950 -o tools/ubench/valu_rate.bin tools/ubench/valu_rate.hip && tools/ubench/valu_rate.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(X) X X X X X X X X
template <int OP>
__global__ void __launch_bounds__(1024) k(float *out, unsigned long long *ticks, int iters)
{
    const int tid = threadIdx.x;
    v2f a0 = {1.f, 2.f}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
    v2f x = {tid * 1e-9f, 1e-9f}, y = {1.f, 1.f};
    int m = tid & 1;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) {          // v_fma_f32
            REP8(asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
                              "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7"
                              : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(x.x), "v"(y.x));)
        } else if (OP == 1) {   // v_pk_fma_f32
            REP8(asm volatile("v_pk_fma_f32 %0, %8, %9, %0\n v_pk_fma_f32 %1, %8, %9, %1\n v_pk_fma_f32 %2, %8, %9, %2\n v_pk_fma_f32 %3, %8, %9, %3\n"
                              "v_pk_fma_f32 %4, %8, %9, %4\n v_pk_fma_f32 %5, %8, %9, %5\n v_pk_fma_f32 %6, %8, %9, %6\n v_pk_fma_f32 %7, %8, %9, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));)
        } else if (OP == 2) {   // v_pk_fma_f32 with a broadcast half (op_sel_hi:[0,1,1])
            REP8(asm volatile("v_pk_fma_f32 %0, %8, %9, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %1, %8, %9, %1 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %2, %8, %9, %2 op_sel_hi:[0,1,1]\n"
                              "v_pk_fma_f32 %3, %8, %9, %3 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %4, %8, %9, %4 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %5, %8, %9, %5 op_sel_hi:[0,1,1]\n"
                              "v_pk_fma_f32 %6, %8, %9, %6 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %7, %8, %9, %7 op_sel_hi:[0,1,1]"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));)
        } else if (OP == 3) {   // v_mov_b32_dpp quad broadcast
            REP8(asm volatile("v_mov_b32_dpp %0, %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %2, %8 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %4, %9 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %9 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %6, %9 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %9 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf"
                              : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(x.x), "v"(y.x));)
        } else if (OP == 4) {   // v_cmp -> SGPR pair, v_cndmask from it (the pair the masked sums are built from)
            REP8(asm volatile("v_cmp_eq_u32_e64 s[20:21], %8, 1\n v_cndmask_b32_e64 %0, 0, 1.0, s[20:21]\n v_cmp_eq_u32_e64 s[22:23], %8, 0\n v_cndmask_b32_e64 %1, 0, 1.0, s[22:23]\n"
                              "v_cmp_eq_u32_e64 s[24:25], %8, 1\n v_cndmask_b32_e64 %2, 0, 1.0, s[24:25]\n v_cmp_eq_u32_e64 s[26:27], %8, 0\n v_cndmask_b32_e64 %3, 0, 1.0, s[26:27]"
                              : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(m), "v"(y.x)
                              : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
        } else if (OP == 5) {   // v_pk_mul_f32
            REP8(asm volatile("v_pk_mul_f32 %0, %8, %9\n v_pk_mul_f32 %1, %8, %9\n v_pk_mul_f32 %2, %8, %9\n v_pk_mul_f32 %3, %8, %9\n"
                              "v_pk_mul_f32 %4, %8, %9\n v_pk_mul_f32 %5, %8, %9\n v_pk_mul_f32 %6, %8, %9\n v_pk_mul_f32 %7, %8, %9"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));)
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 1024 + tid] = a0.x + a1.x + a2.x + a3.x + a4.x + a5.x + a6.x + a7.x + a0.y + a1.y + a2.y + a3.y + a4.y + a5.y + a6.y + a7.y;
    if ((tid & 63) == 0) ticks[tid >> 6] = t1 - t0;
}

template <int OP>
static void run(const char *name, float *out, unsigned long long *ticks)
{
    const int iters = 2000, per_iter = 64;
    for (int threads : {256, 512, 1024}) {
        hipLaunchKernelGGL(k<OP>, dim3(1), dim3(threads), 0, 0, out, ticks, iters);
        hipLaunchKernelGGL(k<OP>, dim3(1), dim3(threads), 0, 0, out, ticks, iters);
        hipDeviceSynchronize();
        unsigned long long h[16];
        hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
        unsigned long long mx = 0;
        for (int w = 0; w < threads / 64; ++w) mx = h[w] > mx ? h[w] : mx;
        const double waves_per_simd = threads / 256.0;
        printf("%-46s %d wave(s) per SIMD: %.2f shader-clock ticks per instruction and SIMD (slowest wave %llu ticks for %d instructions)\n", name,
               (int)waves_per_simd, (double)mx / (iters * per_iter * waves_per_simd), mx, iters * per_iter);
    }
}

int main()
{
    float *out; unsigned long long *ticks;
    hipMalloc(&out, 1024 * 4); hipMalloc(&ticks, 16 * 8);
    run<0>("v_fma_f32", out, ticks);
    run<1>("v_pk_fma_f32", out, ticks);
    run<2>("v_pk_fma_f32 op_sel_hi:[0,1,1]", out, ticks);
    run<5>("v_pk_mul_f32", out, ticks);
    run<3>("v_mov_b32_dpp quad_perm", out, ticks);
    run<4>("v_cmp_eq_u32_e64 -> s[..] + v_cndmask_b32_e64", out, ticks);
    printf("(__builtin_readcyclecounter: s_memtime, the counter the phase traces of tools/fit_trace.py read)\n");
    return 0;
}
