// What one iteration of a several-workgroups-per-cloud farthest point sampling would pay for its exchange alone:
// G clouds x W workgroups, every iteration each workgroup publishes one 64-bit candidate (distance bits | ~index | tag) and needs the
// maximum over the W candidates of its cloud before it can go on.  Nothing else is done per iteration - no distance update, no in-workgroup
// arg-max - so the figure is the floor the exchange adds to the per-iteration time of the one-CU kernel (1.66 us at N = 8192).
//   variant 0: one 64-bit atomicMax per workgroup into slot[cloud][it % 3] + an arrival counter, spin on the counter (agent scope)
//   variant 1: no read-modify-write: every workgroup stores its candidate (with a 16-bit iteration tag) into its own slot, W lanes poll the
//              W slots until all carry this iteration's tag, shuffle-reduce
//   same_xcd = 1 places the W workgroups of a cloud on one XCD (workgroup b runs on XCD b % 8), 0 spreads them over the XCDs.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/fps_xchg.hip -o tools/ubench/fps_xchg.bin && tools/ubench/fps_xchg.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int VARIANT>
__global__ void __launch_bounds__(1024) xchg_kernel(unsigned long long *slots, unsigned int *counters, unsigned long long *out, int G, int W, int iters,
                                                   int same_xcd, int threads_used)
{
    const int b = blockIdx.x;
    int g, w;
    if (same_xcd) { g = (b & 7) + 8 * ((b >> 3) / W); w = (b >> 3) % W; }       // all W workgroups of cloud g have b % 8 == g % 8
    else { g = b / W; w = b % W; }
    if (g >= G) return;
    __shared__ unsigned long long best_s;
    unsigned long long acc = 0;
    unsigned long long mine = ((unsigned long long)(b * 2654435761u) & 0xffffffffffffull);    // 48-bit payload (31 distance bits + 17 index bits)
    for (int it = 0; it < iters; ++it) {
        mine = (mine * 6364136223846793005ull + 1442695040888963407ull) & 0xffffffffffffull;
        if (VARIANT == 0) {
            unsigned long long *slot = slots + ((size_t)g * 4 + (it % 3));
            unsigned int *cnt = counters + g;
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_max(slot, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned int want = (unsigned int)W * (unsigned int)(it + 1);
                while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) { }
                best_s = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            acc ^= best_s;
            // three slots in rotation: everybody has read slot it-1 (they arrived at it), nobody is past it+1 (which uses slot (it+1) % 3)
            if (threadIdx.x == 0 && w == 0 && it >= 1) __hip_atomic_store(slots + ((size_t)g * 4 + ((it - 1) % 3)), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
        } else {
            // two rows in rotation: a slot is rewritten two iterations later, which its owner reaches only after every workgroup of the
            // cloud has published it+1, i.e. finished reading iteration it
            unsigned long long *row = slots + (size_t)g * 32 + (it & 1) * 16;  // W <= 16 slots of a cloud in one 128-byte line
            const unsigned long long tag = (unsigned long long)((it + 1) & 0xffff) << 48;
            if (threadIdx.x == 0) __hip_atomic_store(row + w, mine | tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (threadIdx.x < 64) {
                unsigned long long v = tag;
                if ((int)threadIdx.x < W) {
                    do { v = __hip_atomic_load(row + threadIdx.x, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); } while ((v >> 48) != (tag >> 48));
                }
                v &= 0xffffffffffffull;
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) {
                    const unsigned long long u = __shfl_xor(v, o);
                    v = u > v ? u : v;
                }
                if (threadIdx.x == 0) best_s = v;
            }
            __syncthreads();
            acc ^= best_s;
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) out[b] = acc;
    (void)threads_used;
}

template <int VARIANT>
static float run(int G, int W, int iters, int same_xcd, int threads)
{
    unsigned long long *slots, *out;
    unsigned int *cnt;
    CK(hipMalloc(&slots, (size_t)G * 32 * 8 + 4096));
    CK(hipMalloc(&cnt, (size_t)G * 4 + 64));
    CK(hipMalloc(&out, (size_t)G * W * 8 + 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemset(slots, 0, (size_t)G * 32 * 8 + 4096));
        CK(hipMemset(cnt, 0, (size_t)G * 4 + 64));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(xchg_kernel<VARIANT>, dim3(G * W), dim3(threads), 0, 0, slots, cnt, out, G, W, iters, same_xcd, threads);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    CK(hipFree(slots)); CK(hipFree(cnt)); CK(hipFree(out));
    return best * 1e3f / iters;
}

int main()
{
    const int iters = 2048;
    printf("exchange alone, microseconds per iteration (best of 3 launches of %d iterations); 32 clouds unless said otherwise\n", iters);
    for (int W : {1, 2, 4, 8}) {
        for (int sx : {1, 0}) {
            const float a = run<0>(32, W, iters, sx, 256);
            printf("W=%d %-9s | atomicMax+counter %6.3f us", W, sx ? "same XCD" : "spread", a);
            const float b = run<1>(32, W, iters, sx, 256);
            printf(" | tagged slots %6.3f us\n", b);
            fflush(stdout);
        }
    }
    printf("8 clouds, one per XCD (latency floor): W=8 same XCD tagged slots %6.3f us, atomicMax %6.3f us\n", run<1>(8, 8, iters, 1, 256),
           run<0>(8, 8, iters, 1, 256));
    return 0;
}
