// fwd_pp.hip -- forward of the narrow, long layers  Y = act_in(X) . W^T + b  (K <= 128 (+4), N <= 128 per workgroup,
// M = 262144 .. 1048576 rows) as a persistent, weight-stationary, ping-pong kernel.
//
// Why not the tiled kernel of gemm.hip for these: its workgroups run load -> MFMA -> epilogue in sequence and the
// two workgroups a CU holds start together, so the phases of the chip line up instead of overlapping (matrix pipes
// busy ~50 %); and an fp32 MFMA loop on gfx950 is bounded by the OTHER instructions the wave issues (they do not
// overlap its own MFMAs, tools/ubench/mfma_peak.hip), which its 4-byte operand reads make 1.5 per MFMA.
//
// Here (same skeleton as bwd_fused.hip):
//  * one 512-thread workgroup per CU, W [N,K] in LDS once, row tiles of 64 streamed through;
//  * two halves of 4 waves alternate: while one runs the MFMAs of its tile, the other - on the same SIMDs - does the
//    epilogue of its previous tile (bias, BatchNorm sums, stores), copies its next tile to LDS (act_in applied once
//    per element) and issues the global loads of the tile after that;
//  * LDS tiles keep k contiguous ([row][KP+4]): a global float4 is one ds_write_b128, and ONE ds_read_b128 per lane
//    feeds FOUR MFMAs of an operand block: lane (i = lane&31, h = lane>>5) holds k = 8q+4h .. +3 and MFMA e of group
//    q multiplies the e-th components (k pair {8q+e, 8q+4+e}: A and B use the same pairing, the sum is the same);
//  * everything outside the MFMA phase is WAVE-LOCAL (wave w owns rows 16w..16w+15 of the tile buffer: it parks its
//    32 x 32NT output fragment there, writes it out as whole 128/256-byte row pieces, then fills the same rows with
//    its share of the next tile), so the halves need no barrier of their own;
//  * BatchNorm sums stay in registers for the whole kernel: one set of fp64 atomics per workgroup.
#include "common.h"
#include "fwd_pp.h"
#include <stdlib.h>

#define P2C_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

// Register budget: at most 160 VGPRs.  Two waves per SIMD of this kernel then leave room for the four 48-register waves per SIMD of the
// farthest-point sampling kernel that the forked geometry stream runs on 32 of the CUs at the same time (2 x 160 + 4 x 48 = 512); a
// workgroup that cannot share its CU waits for another to finish and the whole launch takes twice as long (measured: +85 us per launch
// with a 196-register variant).  Check with -Rpass-analysis=kernel-resource-usage after touching the epilogue.
template <int KP, int NT, int MODE, int EX, bool POOL = false>
__global__ void __launch_bounds__(512, 2) fwd_pp_kernel(FwdPPArgs a)
{
    constexpr int LD = KP + 4, BM = 64, BN = 64 * NT;
    constexpr int FR = 32 * 32 * NT;                         // floats of a wave's output fragment
    constexpr int R = 16 * LD > FR ? 16 * LD : FR;           // floats of a wave's private region (16 tile rows or the fragment)
    constexpr int HB = 4 * R;
    constexpr int UPW = 16 * KP / 4 / 64;                    // float4 units a lane stages per tile
    constexpr int QW = KP / 4;                               // float4 per tile row
    constexpr int NQ = KP / 8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Ws = smem;                                        // [BN][LD]
    float *Hb = Ws + BN * LD;                                // [2 halves][HB]
    float *red = Hb + 2 * HB;                                // [2 halves][2 row blocks][2][BN]
    float *We = red + 8 * BN;                                // [BN][4]      (EX)
    float *Xe = We + (EX ? BN * 4 : 0);                      // [2 halves][2 tile parities][BM][4]   (EX)

    const int half = threadIdx.x >> 8, tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int j0 = blockIdx.y * BN;                          // column block of this workgroup
    const int ntiles = (a.M + BM - 1) / BM;
    const int nk = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int niter = (nk + 1) / 2;
    float *hb = Hb + half * HB;
    float *mine = hb + wave * R;                             // this wave's region: tile rows 16*wave .. +15
    float *xe2 = Xe + half * 2 * BM * 4;                     // parity-double-buffered: the partner wave restages while this one still reads
    auto tile_of = [&](int k) { return (int)blockIdx.x + k * (int)gridDim.x; };

    // ---- W (and the EX trailing columns) -> LDS, zero-padded to [BN][KP]
    for (int u = threadIdx.x; u < BN * QW; u += 512) {
        const int n = u / QW, kq = u % QW;
        v4f v = {0.f, 0.f, 0.f, 0.f};
        if (j0 + n < a.N && 4 * kq < a.K) v = *reinterpret_cast<const v4f *>(a.w + (size_t)(j0 + n) * a.ldw + 4 * kq);
        *reinterpret_cast<v4f *>(&Ws[n * LD + 4 * kq]) = v;
    }
    if (EX) {
        for (int n = threadIdx.x; n < BN; n += 512) {
            v4f v = {0.f, 0.f, 0.f, 0.f};
            if (j0 + n < a.N) v = *reinterpret_cast<const v4f *>(a.w + (size_t)(j0 + n) * a.ldw + a.K);
            *reinterpret_cast<v4f *>(&We[n * 4]) = v;
        }
    }

    // ---- per-lane constants of the staging map: unit u = lane + 64 i -> row 16*wave + u / QW, float4 column u % QW
    const int kq = lane % QW;                                // the same for every i (64 % QW == 0)
    const bool kok = 4 * kq < a.K;
    v4f isc = {1.f, 1.f, 1.f, 1.f}, ish = {0.f, 0.f, 0.f, 0.f};
    if (MODE >= 1 && kok) {
        isc = *reinterpret_cast<const v4f *>(a.in_scale + 4 * kq);
        ish = *reinterpret_cast<const v4f *>(a.in_shift + 4 * kq);
    }
    uint32_t slo = 0, shi = 0;
    if (MODE == 3) { slo = a.seed[0]; shi = a.seed[1]; }
    // MODE 4: the A operand is act(bn(W0 x + b0)) of a folded first layer, rebuilt from its 16-byte input row
    v4f w0r[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    v4f b0r = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 4 && kok) {
#pragma unroll
        for (int e = 0; e < 4; ++e) w0r[e] = *reinterpret_cast<const v4f *>(a.w0 + (size_t)(4 * kq + e) * 4);
        if (a.b0) b0r = *reinterpret_cast<const v4f *>(a.b0 + 4 * kq);
    }
    const int xcoloff = MODE == 4 ? 0 : (kok ? 4 * kq : 0);

    v4f rx[UPW];
    v4f rxe = {0.f, 0.f, 0.f, 0.f};
    // row pointer of the lane's first unit in tile 0; unit i is (64 / QW) rows further down
    const float *xlane = a.x + (size_t)(16 * wave + lane / QW) * a.ldx + xcoloff;
    const size_t xstep = (size_t)(64 / QW) * a.ldx;
    auto gload = [&](int t) {                                // raw rows of tile t (clamped; masked when staged)
        const int m0 = t * BM + 16 * wave;
        if (t * BM + BM <= a.M) {                            // uniform: only the last tile can be ragged
            const float *p = xlane + (size_t)t * BM * a.ldx;
#pragma unroll
            for (int i = 0; i < UPW; ++i) rx[i] = *reinterpret_cast<const v4f *>(p + i * xstep);
        } else {
#pragma unroll
            for (int i = 0; i < UPW; ++i) {
                const int rl = (lane + 64 * i) / QW;
                const int m = min(m0 + rl, a.M - 1);
                rx[i] = *reinterpret_cast<const v4f *>(a.x + (size_t)m * a.ldx + xcoloff);
            }
        }
        if (EX && lane < 16) rxe = *reinterpret_cast<const v4f *>(a.x + (size_t)min(m0 + lane, a.M - 1) * a.ldx + a.K);
    };
    auto stage = [&](int t, int par) {                       // act_in, zero outside [M, K], 16-byte LDS writes into this wave's rows
        const int m0 = t * BM + 16 * wave;
        const bool full = t * BM + BM <= a.M && a.K >= KP;   // uniform: nothing to mask
#pragma unroll
        for (int i = 0; i < UPW; ++i) {
            const int rl = (lane + 64 * i) / QW;
            const int m = m0 + rl;
            v4f v = rx[i];
            if (MODE == 4) {
                const v4f x = rx[i];
                v.x = p2c_l0_preact(w0r[0].x, w0r[0].y, w0r[0].z, b0r.x, x.x, x.y, x.z);
                v.y = p2c_l0_preact(w0r[1].x, w0r[1].y, w0r[1].z, b0r.y, x.x, x.y, x.z);
                v.z = p2c_l0_preact(w0r[2].x, w0r[2].y, w0r[2].z, b0r.z, x.x, x.y, x.z);
                v.w = p2c_l0_preact(w0r[3].x, w0r[3].y, w0r[3].z, b0r.w, x.x, x.y, x.z);
            }
            if (MODE >= 1) {         // mul + add, NOT an fma: the backward kernels rebuild act_in(x) and its ReLU mask with the same
                                     // two roundings; with an fma here the deep BatchNorm gradients drift 8x further from float64
                v.x = fmaxf(isc.x * v.x + ish.x, 0.f);
                v.y = fmaxf(isc.y * v.y + ish.y, 0.f);
                v.z = fmaxf(isc.z * v.z + ish.z, 0.f);
                v.w = fmaxf(isc.w * v.w + ish.w, 0.f);
            }
            if (MODE == 3) {
                const uint32_t e = (uint32_t)min(m, a.M - 1) * (uint32_t)a.Kfull + (uint32_t)(4 * kq);
                const uint32_t hq = p2c_hash32(slo, shi, e >> 2);            // e is a multiple of 4: one hash for the four elements
                v.x = p2c_keep4(hq, 0, a.thr) ? v.x * a.dscale : 0.f;
                v.y = p2c_keep4(hq, 1, a.thr) ? v.y * a.dscale : 0.f;
                v.z = p2c_keep4(hq, 2, a.thr) ? v.z * a.dscale : 0.f;
                v.w = p2c_keep4(hq, 3, a.thr) ? v.w * a.dscale : 0.f;
            }
            if (!full) v *= (m < a.M && kok) ? 1.f : 0.f;    // branch-free (the clamped loads only ever return finite data)
            *reinterpret_cast<v4f *>(&mine[rl * LD + 4 * kq]) = v;
        }
        if (EX && lane < 16) {
            const bool ok = m0 + lane < a.M;
            *reinterpret_cast<v4f *>(&xe2[par * BM * 4 + (16 * wave + lane) * 4]) = ok ? rxe : v4f{0.f, 0.f, 0.f, 0.f};
        }
    };

    // ---- per-lane constants of the MFMA / epilogue maps
    const int arow = wm * 32 + l31;
    const float *Ap = hb + (arow >> 4) * R + (arow & 15) * LD + 4 * lh;
    const float *Bp = Ws + (wn * (NT * 32) + l31) * LD + 4 * lh;
    float bias[NT];
    v4f we[NT];
#pragma unroll
    for (int y = 0; y < NT; ++y) {
        const int col = j0 + wn * (NT * 32) + y * 32 + l31;
        bias[y] = (a.bias && col < a.N) ? a.bias[col] : 0.f;
        we[y] = v4f{0.f, 0.f, 0.f, 0.f};
    }
    v2f s1v[NT], s2v[NT];
#pragma unroll
    for (int y = 0; y < NT; ++y) s1v[y] = s2v[y] = v2f{0.f, 0.f};
    f32x16 acc[NT];

    // ---- prologue: first tile of this half -> LDS, second -> registers
    gload(tile_of(half < nk ? half : 0));
    stage(half < nk ? tile_of(half) : ntiles, 0);            // tile index past the end: all rows masked to zero
    gload(tile_of(half + 2 < nk ? half + 2 : 0));
    __syncthreads();
    if (EX) {
#pragma unroll
        for (int y = 0; y < NT; ++y) we[y] = *reinterpret_cast<const v4f *>(&We[(wn * (NT * 32) + y * 32 + l31) * 4]);
    }
    if (half == 1) P2C_LDS_BARRIER();                        // run one phase behind half 0
    for (int it = 0; it < niter; ++it) {
        const int k = 2 * it + half;
        const bool valid = k < nk;                           // uniform within the half
        const int m0 = tile_of(valid ? k : 0) * BM;
        // ================= MFMA phase =================
        P2C_TR(0);
        if (valid) {
#pragma unroll
            for (int y = 0; y < NT; ++y)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[y][r] = 0.f;
            v4f a0 = *reinterpret_cast<const v4f *>(Ap), a1;
            v4f b0[NT], b1[NT];
#pragma unroll
            for (int y = 0; y < NT; ++y) b0[y] = *reinterpret_cast<const v4f *>(Bp + y * 32 * LD);
#pragma unroll
            for (int q = 0; q < NQ; q += 2) {
                // the reads of group q+1 are issued before the MFMAs of group q (pinned: the scheduler would sink them)
                a1 = *reinterpret_cast<const v4f *>(Ap + 8 * (q + 1));
#pragma unroll
                for (int y = 0; y < NT; ++y) b1[y] = *reinterpret_cast<const v4f *>(Bp + y * 32 * LD + 8 * (q + 1));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int y = 0; y < NT; ++y) acc[y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[y][e], acc[y], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (q + 2 < NQ) {
                    a0 = *reinterpret_cast<const v4f *>(Ap + 8 * (q + 2));
#pragma unroll
                    for (int y = 0; y < NT; ++y) b0[y] = *reinterpret_cast<const v4f *>(Bp + y * 32 * LD + 8 * (q + 2));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int y = 0; y < NT; ++y) acc[y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[y][e], acc[y], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        P2C_TR(1);
        P2C_LDS_BARRIER();
        P2C_TR(2);
        // ================= wave-local phase (the other half is in its MFMA phase) =================
        __builtin_amdgcn_s_setprio(1);
        // vmcnt retires in issue order and the compiler cannot count stores across this phase's branches: its wait for the prefetched
        // rows (consumed by stage() below) would land BEHIND the freshly issued global stores and drain them too - a write round trip
        // per tile.  Touching the last prefetched register here makes it wait while only loads are outstanding.
        asm volatile("" ::"v"(rx[UPW - 1]));
        if (EX) asm volatile("" ::"v"(rxe));
        if (valid) {
            // fragment (+ the EX trailing input columns, + bias) -> this wave's region; BatchNorm sums on the bias-free value.
            // Register pairs along r are adjacent, so the sums run as packed-fp32 ops on (r, r+1) pairs.
            float *out = mine;
            if (EX) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rf = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const v4f xv = *reinterpret_cast<const v4f *>(&xe2[(it & 1) * BM * 4 + (wm * 32 + rf) * 4]);
#pragma unroll
                    for (int y = 0; y < NT; ++y)
                        acc[y][r] += (xv.x * we[y].x + xv.y * we[y].y) + (xv.z * we[y].z + xv.w * we[y].w);
                }
            }
#pragma unroll
            for (int y = 0; y < NT; ++y) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const v2f v = {acc[y][r], acc[y][r + 1]};
                    s1v[y] += v;
                    s2v[y] = __builtin_elementwise_fma(v, v, s2v[y]);      // one packed fma instead of packed mul + add (every instruction of
                                                                          // this phase costs an issue slot of the other half's MFMA stream)
                    const v2f ov = v + v2f{bias[y], bias[y]};
                    const int rf = (r & 3) + 8 * (r >> 2) + 4 * lh;        // r even: rows rf and rf+1
                    out[rf * (32 * NT) + y * 32 + l31] = ov.x;
                    out[(rf + 1) * (32 * NT) + y * 32 + l31] = ov.y;
                }
            }
            P2C_TR(3);
            __builtin_amdgcn_wave_barrier();
            // whole 128/256-byte row pieces, 16 bytes per lane: all reads first, then the stores (no round trip per store)
            constexpr int V = 8 * NT, NS = 4 * NT;           // float4 per fragment row, stores per lane
            const int c4 = lane % V, rf0 = lane / V;         // the lane's column is the same for all its stores (64 % V == 0)
            const int col = j0 + wn * (NT * 32) + 4 * c4;
            v4f o[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) o[i] = *reinterpret_cast<const v4f *>(&out[(rf0 + i * (64 / V)) * (32 * NT) + 4 * c4]);
            P2C_TR(4);
            if (a.y != nullptr && col < a.N) {          // (POOL: Y may be absent - its backward needs no Y, csrc/bwd_pool.hip)
                float *yp = a.y + (size_t)(m0 + wm * 32 + rf0) * a.ldy + col;
                if (m0 + BM <= a.M) {
#pragma unroll
                    for (int i = 0; i < NS; ++i) *reinterpret_cast<v4f *>(yp + (size_t)(i * (64 / V)) * a.ldy) = o[i];
                } else {
#pragma unroll
                    for (int i = 0; i < NS; ++i)
                        if (m0 + wm * 32 + rf0 + i * (64 / V) < a.M) *reinterpret_cast<v4f *>(yp + (size_t)(i * (64 / V)) * a.ldy) = o[i];
                }
            }
            if (POOL) {
                // max over the 64 neighbours of relu(bn(y)) (pointnet_util.py:205) needs, per column, only the largest pre-BN value
                // (BatchNorm scale > 0) or the smallest (scale < 0) of the group and its row - but the scale is known only after the whole
                // layer's statistics.  So each wave emits both extremes of its 32 rows x 64 columns, read back from the fragment it has
                // just parked (one column per lane, rows in ascending order, strict compares keep the first; the accumulators are dead
                // here, so the kernel's register count - it must share CUs with the sampling kernel of the forked stream - does not
                // grow); p2c_pool_select_f32 picks when the affine exists.  The 268-537 MB pass that re-read Y for the pooling is gone.
                static_assert(!POOL || NT == 2, "one column per lane");
                float vmax = -INFINITY, vmin = INFINITY;
                int imax = 0, imin = 0;
#pragma unroll 8
                for (int r = 0; r < 32; ++r) {              // (a full unroll costs 6 more registers: 165 > the 160 budget above)
                    const float v = out[r * (32 * NT) + lane];
                    if (v > vmax) { vmax = v; imax = r; }
                    if (v < vmin) { vmin = v; imin = r; }
                }
                const int pcol = j0 + wn * (NT * 32) + lane;
                if (pcol < a.N) {
                    const size_t po = ((size_t)2 * tile_of(k) + wm) * a.N + pcol;
                    a.pool_max[po] = vmax;
                    a.pool_min[po] = vmin;
                    a.pool_idx[po] = (wm * 32 + imax) | ((wm * 32 + imin) << 16);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        P2C_TR(5);
        {
            const int k2 = k + 2, k4 = k + 4;
            if (k2 < nk) stage(tile_of(k2), (it + 1) & 1);
            P2C_TR(6);
            gload(tile_of(k4 < nk ? k4 : 0));                // unconditional: stays in registers
        }
        __builtin_amdgcn_s_setprio(0);
        P2C_LDS_BARRIER();                                   // the prefetch stays in flight across this barrier
        P2C_TR(7);
    }
    if (half == 0) P2C_LDS_BARRIER();
    // ---- BatchNorm sums: every (half, row block) parks its column sums in its own LDS slot, summed in a fixed order
    //      (bitwise reproducible), then one fp64 atomic per column into this workgroup's slot row
    if (a.partials) {
#pragma unroll
        for (int y = 0; y < NT; ++y) {
            const float u1 = s1v[y].x + s1v[y].y, u2 = s2v[y].x + s2v[y].y;
            const float t1 = u1 + __shfl_xor(u1, 32), t2 = u2 + __shfl_xor(u2, 32);
            if (lh == 0) {
                float *r = red + (half * 2 + wm) * 2 * BN;
                r[wn * (NT * 32) + y * 32 + l31] = t1;
                r[BN + wn * (NT * 32) + y * 32 + l31] = t2;
            }
        }
        __syncthreads();
        if (threadIdx.x < BN && j0 + threadIdx.x < a.N) {
            const int c = threadIdx.x;
            const float q1 = (red[c] + red[2 * BN + c]) + (red[4 * BN + c] + red[6 * BN + c]);
            const float q2 = (red[BN + c] + red[3 * BN + c]) + (red[5 * BN + c] + red[7 * BN + c]);
            double *o = a.partials + (size_t)(blockIdx.x % P2C_STAT_SLOTS) * 2 * a.N;
            atomicAdd(&o[j0 + c], (double)q1);
            atomicAdd(&o[a.N + j0 + c], (double)q2);
        }
    }
}

template <int KP, int NT, int MODE, int EX, bool POOL = false>
static int launch_pp(const FwdPPArgs &a, hipStream_t s)
{
    constexpr int LD = KP + 4, BN = 64 * NT, FR = 32 * 32 * NT, R = 16 * LD > FR ? 16 * LD : FR;
    const size_t lds = (size_t)(BN * LD + 2 * 4 * R + 8 * BN + (EX ? BN * 4 + 4 * 64 * 4 : 0)) * sizeof(float);
    const int gy = (a.N + BN - 1) / BN;
    const int ntiles = (a.M + 63) / 64;
    int gx = 256 / gy;
    if (gx > ntiles) gx = ntiles;
    (void)hipFuncSetAttribute((const void *)fwd_pp_kernel<KP, NT, MODE, EX, POOL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((fwd_pp_kernel<KP, NT, MODE, EX, POOL>), dim3(gx, gy), dim3(512), lds, s, a);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// Which (M, N, K, in_mode) the persistent kernel takes; K == 132 is the grouped layer [128 features | xyz | pad].
extern "C" int p2c_linear_fwd_pp_supported(int M, int N, int K, int in_mode)
{
    if (in_mode == 2 || M < 8192 || (K & 3) || (N & 3)) return 0;
    if (K > 128 && K != 132) return 0;
    if (N > 256) return 0;
    if (in_mode == 3 && K > 128) return 0;
    return 1;
}

// The pooled forward: BatchNorm'ed input (in_mode 1), 64 or 128 input channels, a multiple of 128 output channels, whole groups of
// exactly 64 rows (a row tile IS a neighbourhood).
extern "C" int p2c_linear_fwd_pool_supported(int M, int N, int K, int in_mode, int ns)
{
    return in_mode == 1 && ns == 64 && M >= 8192 && (M % 64) == 0 && (K == 64 || K == 128) && (N % 128) == 0 && N <= 256;
}

static int g_mfma_split = -1;
int p2c_mfma_split()
{
    if (g_mfma_split < 0) {
        const char *e = getenv("P2C_MFMA");
        g_mfma_split = (e && (e[0] == 'f' || e[0] == 'F' || e[0] == '0')) ? 0 : 1;      // "f32" / "0": the fp32-MFMA kernels
    }
    return g_mfma_split;
}
extern "C" int p2c_set_mfma_mode(int split)
{
    const int old = p2c_mfma_split();
    g_mfma_split = split ? 1 : 0;
    return old;
}
extern "C" int p2c_get_mfma_mode(void) { return p2c_mfma_split(); }

int p2c_fwd_pp_launch(const FwdPPArgs &a, int in_mode, hipStream_t s)
{
    if (p2c_mfma_split()) return p2c_fwd_pp3_launch(a, in_mode, s);
    const int K = a.K;          // EX columns already split off by the caller
    if (a.pool_max) {
        if (!a.pool_min || !a.pool_idx || !p2c_linear_fwd_pool_supported(a.M, a.N, K, in_mode, 64)) return P2C_EINVAL;
        if (K == 64) return launch_pp<64, 2, 1, 0, true>(a, s);
        return launch_pp<128, 2, 1, 0, true>(a, s);
    }
    const bool ex = a.Kfull == 132 && in_mode != 3;
    const int nt = a.N <= 64 ? 1 : 2;
#define P2C_PP(KP_, NT_, MODE_, EX_) return launch_pp<KP_, NT_, MODE_, EX_>(a, s)
#define P2C_PPK(NT_, MODE_)                          \
    do {                                             \
        if (ex) P2C_PP(128, NT_, MODE_, 4);          \
        if (K <= 32) P2C_PP(32, NT_, MODE_, 0);      \
        if (K <= 64) P2C_PP(64, NT_, MODE_, 0);      \
        P2C_PP(128, NT_, MODE_, 0);                  \
    } while (0)
    if (in_mode == 0) { if (nt == 1) P2C_PPK(1, 0); P2C_PPK(2, 0); }
    if (in_mode == 1) { if (nt == 1) P2C_PPK(1, 1); P2C_PPK(2, 1); }
    if (in_mode == 4) {                  // folded first layer of 64 channels
        if (K != 64) return P2C_EINVAL;
        if (nt == 1) P2C_PP(64, 1, 4, 0);
        P2C_PP(64, 2, 4, 0);
    }
    if (in_mode == 3) {
        if (nt == 1) { if (K <= 64) P2C_PP(64, 1, 3, 0); P2C_PP(128, 1, 3, 0); }
        if (K <= 64) P2C_PP(64, 2, 3, 0);
        P2C_PP(128, 2, 3, 0);
    }
#undef P2C_PPK
#undef P2C_PP
    return P2C_EINVAL;
}
