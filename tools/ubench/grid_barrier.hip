// What a layer boundary INSIDE one cooperative kernel costs on gfx950, against the ~5 us a dependent launch costs inside a HIP-graph replay
// ("one cooperative kernel per low-resolution level: grid barrier between BN-separated layers").  256 workgroups (one per CU)
// run L "layers"; in each, a workgroup writes its slice of an activation tensor (`bytes` per workgroup, the SA3 / FP3 / FP2 layers move
// 4 - 16 MB per layer over the chip) and adds to a small fp64 statistics array (device-scope atomics, as the BatchNorm sums do); then the
// grid barrier; then it reads a slice ANOTHER workgroup wrote (on another XCD: workgroup b runs on XCD b % 8) and the statistics - which
// is what the next layer's operand load does.
//   mode 0: barrier only (arrive counter + spin, agent scope, relaxed data: INCORRECT for the data, the floor of the barrier itself)
//   mode 1: correct: release fence (agent scope) before arriving, acquire after leaving - the producer's L2 has to give up its dirty lines
//           and the consumer's L2 / vector caches have to drop theirs (XCD L2s are not coherent with each other)
//   mode 2: correct without cache maintenance: the activation is written with nontemporal ("write-through, no allocate") stores and read
//           with nontemporal loads, statistics by atomics; only the barrier orders them
//   mode 3: the same L layers as L dependent launches of an ordinary kernel in a captured HIP graph (what the step does today)
// Reported: microseconds per layer boundary beyond the data movement itself (a run with 1 layer is subtracted), and whether the data read
// after the barrier was the data written before it.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/grid_barrier.hip -o tools/ubench/grid_barrier.bin && tools/ubench/grid_barrier.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned int *cnt, unsigned int want, bool fences)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        if (fences) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
        if (fences) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

typedef float vf4 __attribute__((ext_vector_type(4)));

template <int MODE>
__device__ __forceinline__ void layer_body(float4 *act, double *stats, int layer, int nwg, size_t f4_per_wg, int b, unsigned int *bad)
{
    // read what workgroup (b + 3) % nwg wrote in the previous layer (another XCD), check it, write this layer's slice
    float4 *dst = act + ((size_t)(layer & 1) * nwg + b) * f4_per_wg;
    const float4 *src = act + ((size_t)((layer + 1) & 1) * nwg + (b + 3) % nwg) * f4_per_wg;
    const float expect = (float)(layer - 1) * 1000.f + (float)((b + 3) % nwg);
    float s = 0.f;
    for (size_t i = threadIdx.x; i < f4_per_wg; i += blockDim.x) {
        float4 v;
        if (layer > 0) {
            if (MODE == 2) { const vf4 t = __builtin_nontemporal_load(reinterpret_cast<const vf4 *>(src + i)); v = make_float4(t.x, t.y, t.z, t.w); } else v = src[i];
            if (v.x != expect) atomicAdd(bad, 1u);
            s += v.y;
        }
        const float mine = (float)layer * 1000.f + (float)b;
        float4 o = make_float4(mine, s * 1e-9f + 1.f, 0.f, 0.f);
        if (MODE == 2) { vf4 t = {o.x, o.y, o.z, o.w}; __builtin_nontemporal_store(t, reinterpret_cast<vf4 *>(dst + i)); } else dst[i] = o;
    }
    if (threadIdx.x < 64) atomicAdd(stats + (layer & 3) * 64 + threadIdx.x, (double)s + 1.0);      // BatchNorm-style sums: 64 slots x fp64
}

template <int MODE>
__global__ void __launch_bounds__(512) coop_kernel(float4 *act, double *stats, unsigned int *cnt, unsigned int *bad, int layers, size_t f4_per_wg)
{
    const int b = blockIdx.x, nwg = gridDim.x;
    for (int l = 0; l < layers; ++l) {
        layer_body<MODE>(act, stats, l, nwg, f4_per_wg, b, bad);
        if (l + 1 < layers) grid_barrier(cnt, (unsigned int)nwg * (unsigned int)(l + 1), MODE == 1);
    }
}

__global__ void __launch_bounds__(512) one_layer_kernel(float4 *act, double *stats, unsigned int *bad, int layer, size_t f4_per_wg)
{
    layer_body<0>(act, stats, layer, gridDim.x, f4_per_wg, blockIdx.x, bad);
}

int main()
{
    const int nwg = 256;
    hipStream_t st; CK(hipStreamCreate(&st));
    for (size_t kb : {16, 64}) {            // per workgroup and layer: 256 x 16 KB = 4 MB (SA3 / FP3 size), 256 x 64 KB = 16 MB (FP2)
        const size_t f4 = kb * 1024 / 16;
        float4 *act; double *stats; unsigned int *cnt, *bad;
        CK(hipMalloc(&act, 2 * nwg * f4 * 16)); CK(hipMalloc(&stats, 4 * 64 * 8)); CK(hipMalloc(&cnt, 4)); CK(hipMalloc(&bad, 4));
        auto time_coop = [&](int mode, int layers) {
            float best = 1e9f; unsigned int nbad = 0;
            for (int rep = 0; rep < 7; ++rep) {
                CK(hipMemsetAsync(cnt, 0, 4, st)); CK(hipMemsetAsync(bad, 0, 4, st)); CK(hipMemsetAsync(stats, 0, 4 * 64 * 8, st));
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                CK(hipEventRecord(e0, st));
                if (mode == 0) hipLaunchKernelGGL(coop_kernel<0>, dim3(nwg), dim3(512), 0, st, act, stats, cnt, bad, layers, f4);
                if (mode == 1) hipLaunchKernelGGL(coop_kernel<1>, dim3(nwg), dim3(512), 0, st, act, stats, cnt, bad, layers, f4);
                if (mode == 2) hipLaunchKernelGGL(coop_kernel<2>, dim3(nwg), dim3(512), 0, st, act, stats, cnt, bad, layers, f4);
                CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
                CK(hipMemcpy(&nbad, bad, 4, hipMemcpyDeviceToHost));
                CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
            }
            return std::pair<float, unsigned int>(best * 1e3f, nbad);
        };
        auto time_graph = [&](int layers) {
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int l = 0; l < layers; ++l) hipLaunchKernelGGL(one_layer_kernel, dim3(nwg), dim3(512), 0, st, act, stats, bad, l, f4);
            CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            float best = 1e9f; unsigned int nbad = 0;
            for (int rep = 0; rep < 7; ++rep) {
                CK(hipMemsetAsync(bad, 0, 4, st));
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
                CK(hipMemcpy(&nbad, bad, 4, hipMemcpyDeviceToHost));
                CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
            }
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            return std::pair<float, unsigned int>(best * 1e3f, nbad);
        };
        const int L = 9;
        printf("%zu KB per workgroup and layer (%zu MB per layer over %d workgroups), %d layers:\n", kb, kb * nwg / 1024, nwg, L);
        const char *names[3] = {"barrier only, no fences (data race)", "release / acquire fences at agent scope", "nontemporal stores + loads, no fences"};
        for (int mode = 0; mode < 3; ++mode) {
            auto one = time_coop(mode, 1), many = time_coop(mode, L);
            printf("  cooperative, %-42s: %8.1f us for %d layers, 1 layer %6.1f us -> %6.2f us per layer boundary + layer; stale reads: %u\n", names[mode],
                   many.first, L, one.first, (many.first - one.first) / (L - 1), many.second);
        }
        auto one = time_graph(1), many = time_graph(L);
        printf("  %-55s: %8.1f us for %d layers, 1 layer %6.1f us -> %6.2f us per layer boundary + layer; stale reads: %u\n",
               "HIP graph of dependent launches (today)", many.first, L, one.first, (many.first - one.first) / (L - 1), many.second);
        CK(hipFree(act)); CK(hipFree(stats)); CK(hipFree(cnt)); CK(hipFree(bad));
    }
    return 0;
}
