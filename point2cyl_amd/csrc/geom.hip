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

// ---------------------------------------------------------------------------------------------------------------------------------
// The same selection for clouds whose points live in registers (PPT 2..16, coordinates NOT mirrored in LDS): the iteration is a chain of
// VALU work (4 waves per SIMD, each issuing its points' updates) -> wave reduction -> LDS exchange + barrier -> centroid fetch.
// Against fps_kernel this variant
//   * writes the update for PAIRS of points (float2 lanes: same IEEE operations, same order ((dx*dx + dy*dy) + dz*dz), nothing fused).  The
//     library is built WITHOUT packed fp32 instructions (point2cyl_amd/build.py), so these are scalar v_sub / v_mul / v_add: as
//     v_pk_add_f32 / v_pk_mul_f32 the low half of a result was read stale by the v_min_i32 two issue slots later whenever the kernel ran
//     under the MFMA kernels with <= 2 waves per SIMD - wrong picks in 7 % (8 waves per cloud) to 54 % (4 waves) of the graph replays, never
//     alone (tools/stress_prefetch.py, profiles/r06_fps_packed_hazard.log) - and the scalar form is FASTER here (a v_fma-class instruction
//     issues in 2.5 clocks, a packed one in 4.2: 443 -> 382 us).  The running minimum is an integer v_min, the lane's best a v_max3 tree - the index of the
//     best point is NOT tracked per point (two selects each): after the wave maximum is known, one compare per register slot yields the 64-bit
//     mask of lanes holding it, and the first (lane, slot) - lowest point index, torch.max's tie rule (:83) - is picked on the scalar unit;
//   * carries the winner's COORDINATES through the exchange (read from the winning lane's registers), so the next iteration starts from an
//     LDS slot instead of a dependent global load of xyz[far].
// Bit-identical indices by construction (tests/test_gpu_parity.py: G1, the oracle shapes, the duplicate-point tie test).
// ---------------------------------------------------------------------------------------------------------------------------------
typedef float p2c_f2 __attribute__((ext_vector_type(2)));

// max over a row of 16 lanes / the wave with the DPP operand folded into v_max_i32 (old = INT_MIN, the operation's identity: the DPP
// combiner then emits ONE instruction per step instead of mov + mov_dpp + max)
template <int CTRL>
__device__ __forceinline__ int p2c_dpp_maxid(int v)
{
    return max(v, __builtin_amdgcn_update_dpp((int)0x80000000, v, CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ int p2c_row16_max_i32_folded(int v)
{
    v = p2c_dpp_maxid<0xB1>(v);
    v = p2c_dpp_maxid<0x4E>(v);
    v = p2c_dpp_maxid<0x141>(v);
    v = p2c_dpp_maxid<0x140>(v);
    return v;
}

#ifdef P2C_FPS_TRACE       // tools/fps_trace.py: shader-clock stamps of workgroup 0 (first and last wave) inside a few iterations
__device__ unsigned long long p2c_fps_stamps[2][4][8];
extern "C" int p2c_fps_trace_read(void *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(p2c_fps_stamps), sizeof(p2c_fps_stamps)) == hipSuccess ? 0 : 1; }
#define FPS_TR(i) do { if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == nwaves - 1) && it >= 100 && it < 104) \
        p2c_fps_stamps[wave ? 1 : 0][it - 100][i] = __builtin_readcyclecounter(); } while (0)
#else
#define FPS_TR(i) do { } while (0)
#endif

template <int PPT>
__global__ void __launch_bounds__(PPT == 32 ? 256 : 1024) fps_pk_kernel(const float *__restrict__ xyz, int N, const int64_t *__restrict__ start,
                                                      int npoint, int32_t *__restrict__ idx_out, float *__restrict__ new_xyz_out)
{
    static_assert(PPT == 2 || PPT == 4 || PPT == 8 || PPT == 16 || PPT == 32, "");
    constexpr int H = PPT / 2;
    // The coordinates live in a 32-slot register vector [x | y] and a 16-slot one [z]: reading "slot bj of the winner" is then a register-relative
    // move with a wave-uniform index (s_set_gpr_idx) - for vectors of <= 16 elements the compiler expands a dynamic index into a
    // compare + select per element instead.
    typedef float vec32 __attribute__((ext_vector_type(32)));
    typedef float vec16 __attribute__((ext_vector_type(16)));
    // The exchange between the waves: ONE 64-bit LDS maximum per iteration.  key = (distance bits with the sign flipped: unsigned order,
    // padding's -1 below every real distance) << 32 | ~index - the largest key is the largest distance and, among equal distances, the
    // LOWEST index (torch.max's rule, :83).  Three keys rotate (the one for the next iteration is cleared before this iteration's
    // barrier); the candidates' coordinates travel in per-wave slots (double buffered) so that the next centroid is an LDS read, not a
    // global load that depends on the reduced index.
    __shared__ unsigned long long keys[3];
    __shared__ __attribute__((aligned(16))) float cslot[2][16][4];
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwaves = nthreads >> 6;
    const int b = blockIdx.x;
    const float *p = xyz + (size_t)b * N * 3;
    if (tid < 3) keys[tid] = 0ull;
    __syncthreads();
    // x_j, y_j, z_j: one 32-slot vector each at PPT = 32; [x | y] + [z] up to PPT = 16 (YV / YO, ZV / ZO: where y and z start)
    vec32 qa, qb, qc;
    constexpr int YO = PPT == 32 ? 0 : PPT, ZO = 0;
#define FPS_X(j) qa[(j)]
#define FPS_Y(j) (PPT == 32 ? qb[(j)] : qa[YO + (j)])
#define FPS_Z(j) (PPT == 32 ? qc[(j)] : qb[ZO + (j)])
    int dist[PPT];                                           // the running minimum distances, as their bit patterns
#pragma unroll
    for (int j = 0; j < 32; ++j) { qa[j] = 0.f; qb[j] = 0.f; qc[j] = 0.f; }
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int n = tid * PPT + j;
        const bool ok = n < N;
        const float vx = ok ? p[n * 3 + 0] : 0.f, vy = ok ? p[n * 3 + 1] : 0.f, vz = ok ? p[n * 3 + 2] : 0.f;
        qa[j] = vx;
        if (PPT == 32) { qb[j] = vy; qc[j] = vz; } else { qa[YO + j] = vy; qb[ZO + j] = vz; }
        dist[j] = __float_as_int(ok ? 1e10f : -1.0f);        // :74; padded slots can never win the argmax
    }
    int far = (int)start[b];
    float cx = p[far * 3 + 0], cy = p[far * 3 + 1], cz = p[far * 3 + 2];
    int32_t *out = idx_out + (size_t)b * npoint;
    float *oxyz = new_xyz_out ? new_xyz_out + (size_t)b * npoint * 3 : nullptr;
    int kb = 0;                                              // it % 3
    for (int it = 0; it < npoint; ++it) {
        if (tid == 0) {
            out[it] = far;
            if (oxyz) { oxyz[it * 3 + 0] = cx; oxyz[it * 3 + 1] = cy; oxyz[it * 3 + 2] = cz; }
        }
        FPS_TR(0);
        const p2c_f2 c2x = {cx, cx}, c2y = {cy, cy}, c2z = {cz, cz};
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const p2c_f2 x2 = {FPS_X(2 * h), FPS_X(2 * h + 1)}, y2 = {FPS_Y(2 * h), FPS_Y(2 * h + 1)}, z2 = {FPS_Z(2 * h), FPS_Z(2 * h + 1)};
            const p2c_f2 dx = x2 - c2x, dy = y2 - c2y, dz = z2 - c2z;
            const p2c_f2 d = (dx * dx + dy * dy) + dz * dz;                   // :80, no FMA (file is built with -ffp-contract=off)
            // :81-82.  Distances are >= +0 (padding: -1): their bit patterns order like signed ints, so the running minimum and every maximum
            // below are INTEGER min / max - the same selections, and no NaN-quieting instructions in front of them
            dist[2 * h] = min(__float_as_int(d[0]), dist[2 * h]);
            dist[2 * h + 1] = min(__float_as_int(d[1]), dist[2 * h + 1]);
        }
        int best = dist[0];
#pragma unroll
        for (int j = 1; j < PPT; ++j) best = max(best, dist[j]);
        FPS_TR(1);
        // the wave's maximum in every lane, on the vector unit throughout (DPP inside the rows, gfx950's permlane swaps across them): no
        // round trip through scalar registers
        int wmax = p2c_row16_max_i32_folded(best);
        {
            const auto r = __builtin_amdgcn_permlane16_swap((unsigned)wmax, (unsigned)wmax, false, false);
            wmax = max((int)r[0], (int)r[1]);
        }
        {
            const auto r = __builtin_amdgcn_permlane32_swap((unsigned)wmax, (unsigned)wmax, false, false);
            wmax = max((int)r[0], (int)r[1]);
        }
        FPS_TR(2);
        int bjl = PPT;                                       // this lane's first slot holding the wave maximum (PPT: none)
#pragma unroll
        for (int j = PPT - 1; j >= 0; --j) bjl = dist[j] == wmax ? j : bjl;
        const unsigned long long vote = __ballot(bjl < PPT);
        const int src = __ffsll((long long)vote) - 1;        // lowest lane holding it = lowest index block; its first slot = lowest index
        const int bj = __builtin_amdgcn_readlane(bjl, src);
        const int widx = (wave * 64 + src) * PPT + bj;
        if (nwaves == 1) {
            far = widx;
            cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(FPS_X(bj)), src));
            cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(FPS_Y(bj)), src));
            cz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(FPS_Z(bj)), src));
            continue;
        }
        FPS_TR(3);
        if (lane == src) {                                   // the winner lane publishes its point and the wave's key
            float4 c;
            c.x = FPS_X(bj); c.y = FPS_Y(bj); c.z = FPS_Z(bj); c.w = 0.f;       // register-relative moves (bj is wave-uniform)
            *reinterpret_cast<float4 *>(&cslot[it & 1][wave][0]) = c;
            const unsigned long long key = ((unsigned long long)((unsigned)wmax ^ 0x80000000u) << 32) | (unsigned)(~widx);
            // (one lane: the instruction itself, not atomicMax() - the compiler wraps that in a loop over the active lanes)
            const unsigned ka = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned long long *)&keys[kb];
            asm volatile("ds_max_u64 %0, %1" ::"v"(ka), "v"(key) : "memory");
            if (wave == 0) keys[kb == 2 ? 0 : kb + 1] = 0ull;               // next iteration's key (last read two barriers ago)
        }
        FPS_TR(4);
        __syncthreads();
        FPS_TR(5);
        const unsigned long long k = keys[kb];
        far = (int)~(unsigned)(k & 0xFFFFFFFFull);
        const float4 c = *reinterpret_cast<const float4 *>(&cslot[it & 1][(unsigned)far / (64u * PPT)][0]);
        cx = c.x; cy = c.y; cz = c.z;
        kb = kb == 2 ? 0 : kb + 1;
        FPS_TR(6);
    }
}
#undef FPS_X
#undef FPS_Y
#undef FPS_Z

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
    // P2C_FPS_V1=1: the round-1..5 kernel for every shape (A/B switch, tools/bench_sa1_forward.py)
    static const bool v1 = [] { const char *e = getenv("P2C_FPS_V1"); return e && atoi(e) != 0; }();
    // Points per lane of the register-resident kernel: 8 WAVES per cloud (N / 512 points per lane), not 16.  Same-box A/B at B = 32 x 8192
    // (tools/fps_step_ab.sh, tools/fps_legs_ab.sh): 16 waves x 8 points 472 us alone and the training step it runs under +0.12 ms (its waves
    // take issue slots from the MLP workgroups they share CUs with); 8 x 16: 442 us, step -0.04 ms against the round-5 kernel, best
    // pipelined forward and evaluation loop; 4 x 32: 476 us, step -0.05 ms.  P2C_FPS_PPT = 8 / 16 / 32 forces a shape (N <= 8192).
    static const int ppt_env = [] { const char *e = getenv("P2C_FPS_PPT"); return e ? atoi(e) : 0; }();
    if (!in_lds && !v1 && N <= 8192 && N > 512) {
        int want = ppt_env == 8 || ppt_env == 16 || ppt_env == 32 ? ppt_env : 2 * ppt;
        if (want > 16 && ppt_env != 32) want = 16;
        if (want >= ppt && (long long)want * (want == 32 ? 256 : 1024) >= N) { ppt = want; threads = ((N + ppt - 1) / ppt + 63) & ~63; }
    }
    const bool forced = !in_lds && !v1 && (ppt == 16 || ppt == 32) && threads <= (ppt == 32 ? 256 : 1024) && N <= 8192;
    if (!in_lds && !v1 && (ppt == 2 || ppt == 4 || ppt == 8 || forced)) {
        switch (ppt) {
        case 32: hipLaunchKernelGGL((fps_pk_kernel<32>), dim3(B), dim3(threads), 0, s, xyz, N, start, npoint, idx_out, new_xyz_out); break;
        case 16: hipLaunchKernelGGL((fps_pk_kernel<16>), dim3(B), dim3(threads), 0, s, xyz, N, start, npoint, idx_out, new_xyz_out); break;
        case 2: hipLaunchKernelGGL((fps_pk_kernel<2>), dim3(B), dim3(threads), 0, s, xyz, N, start, npoint, idx_out, new_xyz_out); break;
        case 4: hipLaunchKernelGGL((fps_pk_kernel<4>), dim3(B), dim3(threads), 0, s, xyz, N, start, npoint, idx_out, new_xyz_out); break;
        default: hipLaunchKernelGGL((fps_pk_kernel<8>), dim3(B), dim3(threads), 0, s, xyz, N, start, npoint, idx_out, new_xyz_out); break;
        }
        P2C_LAUNCH_CHECK();
        return P2C_OK;
    }
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
