// geom.hip -- farthest point sampling, ball query, 3-NN for gfx950 (wave64).
//
// These three replace the reference's Python loop / full-row sorts (models/pointnet_util.py:63-107,
// :301-307).  Their outputs are integer indices, so the floating-point ORDER OF OPERATIONS of the
// reference is reproduced exactly; this file must be compiled with
// -ffp-contract=off and fused multiply-adds are spelled __builtin_fmaf() where the reference's CPU
// BLAS uses them.
#include "common.h"
#include <stdlib.h>

// squared distance exactly as square_distance() rounds it (pointnet_util.py:37-39):
//   ((-2*dot + |s|^2) + |d|^2),  dot = fma(sz,dz, fma(sy,dy, sx*dx)),  |v|^2 = (x*x + y*y) + z*z
__device__ __forceinline__ float p2c_norm2(float x, float y, float z) { return (x * x + y * y) + z * z; }
__device__ __forceinline__ float p2c_sqdist(float sx, float sy, float sz, float sn, float dx, float dy, float dz, float dn)
{
    const float dot = __builtin_fmaf(sz, dz, __builtin_fmaf(sy, dy, sx * dx));
    return __builtin_fmaf(-2.0f, dot, sn) + dn;      // = (-2 dot + sn) + dn bit for bit (the product by 2 is exact), one instruction instead of two
}

// =============================================================================================
// Farthest point sampling: one workgroup per cloud, the whole cloud resident on chip.
//   - coordinates: SoA copy in LDS (read once per iteration by every lane: the new centroid) and the
//     PPT points a thread owns in registers together with their running min distance;
//   - thread t owns the CONTIGUOUS points [t*PPT, (t+1)*PPT) so "lowest lane" == "lowest index": the
//     first-index tie rule of torch.max (:83) becomes ballot + find-first-set;
//   - per iteration: VALU update, DPP max inside each wave, one LDS slot per wave, ONE barrier
//     (slots are double buffered), then every wave reduces the <=16 slots with the same DPP tree.
// The loop is latency bound (npoint dependent steps); only B workgroups run.
// =============================================================================================
// LDSXYZ: keep the SoA copy of the cloud in LDS for the centroid broadcast (N*12 B must fit in 160 KB);
// otherwise the centroid is re-read from global memory each iteration (large-N fallback, slower).
// REGXYZ = false (clouds above 16384 points, PPT 32 / 64): only the running distances stay in registers, the coordinates are re-read
// from memory (L2) in every iteration - the same arithmetic and tie rule, for sizes the reference's own (B, N) loop handles slowly too.
template <int PPT, bool LDSXYZ, bool REGXYZ = true>
__global__ void __launch_bounds__(1024) fps_kernel(const float *__restrict__ xyz, int N, const int64_t *__restrict__ start,
                                                   int npoint, int32_t *__restrict__ idx_out, float *__restrict__ new_xyz_out)
{
    extern __shared__ float smem[];
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwaves = nthreads >> 6;
    const int b = blockIdx.x;
    const int Npad = (N + 3) & ~3;
    float *sx = smem, *sy = smem + Npad, *sz = smem + 2 * Npad;
    int *red = reinterpret_cast<int *>(LDSXYZ ? smem + 3 * Npad : smem);   // [2][16][2] (value bits, index)
    const float *p = xyz + (size_t)b * N * 3;
    if (LDSXYZ) {
        for (int i = tid; i < N * 3; i += nthreads) {
            const float v = p[i];
            const int n = i / 3, c = i - n * 3;
            smem[c * Npad + n] = v;
        }
    }
    if (tid < 64) { red[tid] = (tid & 1) ? 0 : (int)0xBF800000; }   // value slots = -1.0f, index slots = 0
    __syncthreads();

    float px[REGXYZ ? PPT : 1], py[REGXYZ ? PPT : 1], pz[REGXYZ ? PPT : 1], dist[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int n = tid * PPT + j;
        const bool ok = n < N;
        if (REGXYZ) {
            px[j] = ok ? (LDSXYZ ? sx[n] : p[n * 3 + 0]) : 0.f;
            py[j] = ok ? (LDSXYZ ? sy[n] : p[n * 3 + 1]) : 0.f;
            pz[j] = ok ? (LDSXYZ ? sz[n] : p[n * 3 + 2]) : 0.f;
        }
        dist[j] = ok ? 1e10f : -1.0f;     // :74; padded slots can never win the argmax
    }
    int far = (int)start[b];
    int32_t *out = idx_out + (size_t)b * npoint;
    float *oxyz = new_xyz_out ? new_xyz_out + (size_t)b * npoint * 3 : nullptr;
    int buf = 0;
    for (int it = 0; it < npoint; ++it) {
        const float cx = LDSXYZ ? sx[far] : p[far * 3 + 0], cy = LDSXYZ ? sy[far] : p[far * 3 + 1],
                    cz = LDSXYZ ? sz[far] : p[far * 3 + 2];
        if (tid == 0) {
            out[it] = far;
            if (oxyz) { oxyz[it * 3 + 0] = cx; oxyz[it * 3 + 1] = cy; oxyz[it * 3 + 2] = cz; }
        }
        float best = -1.0f;
        int bj = 0;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            float qx, qy, qz;
            if (REGXYZ) { qx = px[j]; qy = py[j]; qz = pz[j]; }
            else {
                const int n = min(tid * PPT + j, N - 1);          // clamped: a padded slot's distance stays -1 (min with -1 below)
                qx = p[n * 3 + 0]; qy = p[n * 3 + 1]; qz = p[n * 3 + 2];
            }
            const float dx = qx - cx, dy = qy - cy, dz = qz - cz;
            const float d = (dx * dx + dy * dy) + dz * dz;        // :80, no FMA
            const float nd = d < dist[j] ? d : dist[j];           // :81-82
            dist[j] = nd;
            if (nd > best) { best = nd; bj = j; }                 // strict: first index wins
        }
        // distances are >= 0 (or -1 for padding): their bit patterns order like signed ints
        const int bbits = __float_as_int(best);
        const int wmax = p2c_wave_max_i32(bbits);
        const unsigned long long vote = __ballot(bbits == wmax);
        const int src = __ffsll((long long)vote) - 1;
        const int widx = __builtin_amdgcn_readlane(tid * PPT + bj, src);
        if (nwaves == 1) {
            far = widx;
            continue;
        }
        if (lane == 0) {
            red[buf * 32 + wave * 2 + 0] = wmax;
            red[buf * 32 + wave * 2 + 1] = widx;
        }
        __syncthreads();
        const int2 e = *reinterpret_cast<const int2 *>(&red[buf * 32 + (lane & 15) * 2]);
        const int gmax = __builtin_amdgcn_readlane(p2c_row16_max_i32(e.x), 0);
        const unsigned long long v2 = __ballot(e.x == gmax) & 0xFFFFull;    // lowest wave == lowest index
        far = __builtin_amdgcn_readlane(e.y, __ffsll((long long)v2) - 1);
        buf ^= 1;
    }
}

extern "C" int p2c_fps_f32(const float *xyz, int B, int N, const int64_t *start, int npoint, int32_t *idx_out,
                           float *new_xyz_out, void *stream)
{
    // up to 65536 points per cloud (64 running distances per thread of a 1024-thread workgroup); the reference's largest is 8192
    if (!xyz || !start || !idx_out || B <= 0 || N <= 0 || npoint <= 0 || N > 65536) return P2C_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    int ppt = 1;
    while (ppt < 64 && (long long)ppt * 1024 < N) ppt *= 2;
    if (N <= 512) ppt = 8;                         // single wave per cloud: no barrier at all
    int threads = ((N + ppt - 1) / ppt + 63) & ~63;
    if (threads > 1024) return P2C_EINVAL;
    const int Npad = (N + 3) & ~3;
    // Clouds whose SoA copy is larger than 24 KB read the centroid from global memory (an L1/L2 hit, measured equally
    // fast) instead: FPS runs on a side stream under the persistent GEMM kernels, whose 135 KB workgroups cannot share a
    // CU with a 96 KB copy (N = 8192) and would queue behind the 32 FPS workgroups.
    constexpr size_t lds_cap = 24 * 1024;
    const bool in_lds = (size_t)(3 * Npad + 64) * sizeof(float) <= lds_cap;
    const size_t lds = in_lds ? (size_t)(3 * Npad + 64) * sizeof(float) : 64 * sizeof(float);
#define P2C_FPS_LAUNCH(P, L)                                                                                              \
    (void)hipFuncSetAttribute((const void *)fps_kernel<P, L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      \
    hipLaunchKernelGGL((fps_kernel<P, L>), dim3(B), dim3(threads), lds, s, xyz, N, start, npoint, idx_out, new_xyz_out)
#define P2C_FPS_CASE(P)                                    \
    case P:                                                \
        if (in_lds) { P2C_FPS_LAUNCH(P, true); }           \
        else { P2C_FPS_LAUNCH(P, false); }                 \
        break;
    switch (ppt) {
        P2C_FPS_CASE(1)
        P2C_FPS_CASE(2)
        P2C_FPS_CASE(4)
        P2C_FPS_CASE(8)
        P2C_FPS_CASE(16)
    case 32:
        (void)hipFuncSetAttribute((const void *)fps_kernel<32, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((fps_kernel<32, false, false>), dim3(B), dim3(threads), 64 * sizeof(float), s, xyz, N, start, npoint, idx_out, new_xyz_out);
        break;
    case 64:
        (void)hipFuncSetAttribute((const void *)fps_kernel<64, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((fps_kernel<64, false, false>), dim3(B), dim3(threads), 64 * sizeof(float), s, xyz, N, start, npoint, idx_out, new_xyz_out);
        break;
    default:
        return P2C_EINVAL;
    }
#undef P2C_FPS_CASE
#undef P2C_FPS_LAUNCH
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// =============================================================================================
// Ball query without the sort: a wave scans the cloud in ascending index order, 64 points per step;
// ballot + prefix popcount append the in-ball indices in order; stop at nsample; pad with the first.
// (The round-1 form staged the cloud through LDS for 16 queries per workgroup, which then moved in lockstep through two barriers per 1024
// points; it is in the history up to round 4.)
// =============================================================================================
// No shared staging: every wave walks the cloud on its own (the 96 KB of a cloud are L1 / L2 hits), next step's
// points requested before the current ones are tested; a wave retires as soon as ITS queries are complete (a centre in a sparse region
// scans the whole cloud while its neighbours are done after a few hundred points).
template <int QPW>
__global__ void __launch_bounds__(256) ball_query_direct_kernel(const float *__restrict__ xyz, const float *__restrict__ new_xyz, int N,
                                                                int S, float r2, int nsample, int32_t *__restrict__ idx_out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int q0 = (blockIdx.x * 4 + wave) * QPW;
    if (q0 >= S) return;
    const float *cloud = xyz + (size_t)b * N * 3;
    float cx[QPW], cy[QPW], cz[QPW], cn[QPW];
    int cnt[QPW], first[QPW];
#pragma unroll
    for (int q = 0; q < QPW; ++q) {
        const int s = q0 + q;
        const float *c = new_xyz + ((size_t)b * S + (s < S ? s : 0)) * 3;
        cx[q] = c[0]; cy[q] = c[1]; cz[q] = c[2];
        cn[q] = p2c_norm2(cx[q], cy[q], cz[q]);
        cnt[q] = s < S ? 0 : nsample;      // out-of-range queries are "done"
        first[q] = N;
    }
    // Four 64-point steps per group; the NEXT group's 12 loads are issued before the current group is tested.  With one step of look-ahead a
    // step cost a memory round trip (~0.5 us), and the kernel's length is set by its slowest wave - a centre in a sparse region that walks all
    // N / 64 steps.
    constexpr int G = 4;
    auto fetch = [&](int off, float (&x)[G], float (&y)[G], float (&z)[G]) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int i = min(off + 64 * g + lane, N - 1);
            x[g] = cloud[(size_t)i * 3]; y[g] = cloud[(size_t)i * 3 + 1]; z[g] = cloud[(size_t)i * 3 + 2];
        }
    };
    float nx[G], ny[G], nz[G];
    fetch(0, nx, ny, nz);
    for (int off0 = 0; off0 < N; off0 += 64 * G) {
        bool all_done = true;
#pragma unroll
        for (int q = 0; q < QPW; ++q) all_done = all_done && (cnt[q] >= nsample);
        if (all_done) break;                                     // wave-uniform
        float cxs[G], cys[G], czs[G];
#pragma unroll
        for (int g = 0; g < G; ++g) { cxs[g] = nx[g]; cys[g] = ny[g]; czs[g] = nz[g]; }
        fetch(min(off0 + 64 * G, N - 1), nx, ny, nz);            // unconditional (clamped): no loop-carried copies
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int off = off0 + 64 * g;
            if (off >= N) break;                                 // uniform
            const float px = cxs[g], py = cys[g], pz = czs[g];
            const int i = off + lane;
            const bool ok = i < N;
            const float pn = p2c_norm2(px, py, pz);
#pragma unroll
            for (int q = 0; q < QPW; ++q) {
                if (cnt[q] >= nsample) continue;                 // wave-uniform
                const float d = p2c_sqdist(cx[q], cy[q], cz[q], cn[q], px, py, pz, pn);
                const bool in = ok && !(d > r2);                 // :102 excludes only d > r^2
                const unsigned long long m = __ballot(in);
                if (m) {
                    const int pos = cnt[q] + __popcll(m & ((1ull << lane) - 1ull));
                    if (in && pos < nsample) idx_out[((size_t)b * S + q0 + q) * nsample + pos] = i;
                    if (first[q] == N) first[q] = off + (__ffsll((long long)m) - 1);
                    cnt[q] += __popcll(m);
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < QPW; ++q) {
        if (q0 + q >= S) continue;
        int32_t *o = idx_out + ((size_t)b * S + q0 + q) * nsample;
        const int c = min(cnt[q], nsample);
        for (int k = c + lane; k < nsample; k += 64) o[k] = first[q];      // :104-106
    }
}

extern "C" int p2c_ball_query_f32(const float *xyz, const float *new_xyz, int B, int N, int S, float radius2, int nsample,
                                  int32_t *idx_out, void *stream)
{
    if (!xyz || !new_xyz || !idx_out || B <= 0 || N <= 0 || S <= 0 || nsample <= 0) return P2C_EINVAL;
    // every wave walks the cloud on its own (L1 / L2 hits), two queries per wave: 86 -> 64 us for the two levels against the LDS-chunk
    // scan whose 16 queries per workgroup moved in lockstep (1 and 4 queries per wave measured slower)
    dim3 g(p2c_cdiv(S, 8), B);
    hipLaunchKernelGGL(ball_query_direct_kernel<2>, g, dim3(256), 0, (hipStream_t)stream, xyz, new_xyz, N, S, radius2, nsample, idx_out);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// =============================================================================================
// 3-NN: one thread per dense point, the sparse set staged in LDS as float4 (x,y,z,|p|^2) and read as a
// broadcast; running top-3 by strict '<' insertion in ascending index order == stable sort's first 3.
// =============================================================================================
#define NN_CHUNK 2048

__global__ void __launch_bounds__(256) three_nn_kernel(const float *__restrict__ xyz1, const float *__restrict__ xyz2, int N, int S,
                                                       int32_t *__restrict__ idx_out, float *__restrict__ w_out,
                                                       float *__restrict__ d_out)
{
    __shared__ float4 sc[NN_CHUNK];
    const int b = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    const bool ok = n < N;
    const float *q = xyz1 + ((size_t)b * N + (ok ? n : 0)) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    const float qn = p2c_norm2(qx, qy, qz);
    float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY;
    int i0 = 0, i1 = 0, i2 = 0;
    for (int base = 0; base < S; base += NN_CHUNK) {
        const int len = min(NN_CHUNK, S - base);
        __syncthreads();
        for (int i = threadIdx.x; i < len; i += 256) {
            const float *c = xyz2 + ((size_t)b * S + base + i) * 3;
            const float x = c[0], y = c[1], z = c[2];
            sc[i] = make_float4(x, y, z, p2c_norm2(x, y, z));
        }
        __syncthreads();
        // eight candidates per step: their distances first (eight independent LDS reads and expressions in flight), ONE test whether any of
        // them can enter the top 3, and only then - rarely - the insertions, in index order as before
        constexpr int NNB = 8;               // 16: no further gain
        int i = 0;
        for (; i + NNB <= len; i += NNB) {
            float d[NNB];
#pragma unroll
            for (int u = 0; u < NNB; ++u) {
                const float4 c = sc[i + u];
                d[u] = p2c_sqdist(qx, qy, qz, qn, c.x, c.y, c.z, c.w);       // src = xyz1, dst = xyz2 (:301)
            }
            float m = d[0];
#pragma unroll
            for (int u = 1; u < NNB; ++u) m = fminf(m, d[u]);
            if (!(m < d2)) continue;
#pragma unroll
            for (int u = 0; u < NNB; ++u) {
                const int s = base + i + u;
                if (d[u] < d2) {
                    if (d[u] < d1) {
                        d2 = d1; i2 = i1;
                        if (d[u] < d0) { d1 = d0; i1 = i0; d0 = d[u]; i0 = s; }
                        else { d1 = d[u]; i1 = s; }
                    } else { d2 = d[u]; i2 = s; }
                }
            }
        }
        for (; i < len; ++i) {
            const float4 c = sc[i];
            const float d = p2c_sqdist(qx, qy, qz, qn, c.x, c.y, c.z, c.w);
            const int s = base + i;
            if (d < d2) {
                if (d < d1) {
                    d2 = d1; i2 = i1;
                    if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = s; }
                    else { d1 = d; i1 = s; }
                } else { d2 = d; i2 = s; }
            }
        }
    }
    if (!ok) return;
    const size_t o = ((size_t)b * N + n) * 3;
    idx_out[o + 0] = i0; idx_out[o + 1] = i1; idx_out[o + 2] = i2;
    if (d_out) { d_out[o + 0] = d0; d_out[o + 1] = d1; d_out[o + 2] = d2; }
    const float r0 = 1.0f / (d0 + 1e-8f), r1 = 1.0f / (d1 + 1e-8f), r2 = 1.0f / (d2 + 1e-8f);   // :305
    const float norm = (r0 + r1) + r2;                                                        // :306
    w_out[o + 0] = r0 / norm; w_out[o + 1] = r1 / norm; w_out[o + 2] = r2 / norm;             // :307
}

extern "C" int p2c_three_nn_f32(const float *xyz1, const float *xyz2, int B, int N, int S, int32_t *idx_out, float *weight_out,
                                float *dist_out, void *stream)
{
    if (!xyz1 || !xyz2 || !idx_out || !weight_out || B <= 0 || N <= 0 || S < 3) return P2C_EINVAL;
    dim3 grid(p2c_cdiv(N, 256), B);
    hipLaunchKernelGGL(three_nn_kernel, grid, dim3(256), 0, (hipStream_t)stream, xyz1, xyz2, N, S, idx_out, weight_out, dist_out);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}
