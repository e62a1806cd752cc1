// bwd_fused.hip -- one-pass backward of a 1x1-conv layer for the narrow layers (Co, Ci in {64,128}) that
// carry most of the traffic (M = 262144 .. 1048576 rows).
//
// The two backward GEMMs of a layer,  dX = dY . W  and  dW = dY^T . act_in(X),  read the same three row
// streams (dZ, this layer's pre-BN Y, the layer-below pre-BN X).  Run separately they stream ~7 M*C*4 bytes
// and are HBM-bound (fp32 MFMA at 64 cycles per 32x32x2 block gives ~16-32 FLOP/B at these widths); fused
// they stream each tensor ONCE (3 reads + 1 write) and the matrix pipe sees twice the work per byte, which puts
// HBM and MFMA in balance (~8 B/cycle/CU at full MFMA rate).
//
// Structure (persistent, weight-stationary): one workgroup per CU; W [Co,Ci] is loaded into LDS once; the
// workgroup walks row tiles of BM rows:  global->register prefetch of tile t+1 overlaps the MFMAs of tile t,
// one barrier per tile, tiles double-buffered in LDS.  dW lives in the accumulators for the whole kernel
// (one atomic flush at the end); the dX tile is produced, stored, and reduced on the fly into the
// ReLU+BatchNorm-backward sums of the layer below (s1 = sum g, s2 = sum g*xhat), which also stay in
// registers until the end: one partial row per workgroup.
//
//   LDS:  Ws [Co][Ci+4]            B operand of dX  (k = co, j = ci)
//         dYs[2][Co][BM+1]         A operand of dX  (k = co, i = m)  and A operand of dW (i = co, k = m)
//         Xr [2][BM][Ci+4]         raw X tile: B operand of dW after act_in on the fly (k = m, j = ci), and the
//                                  y values the fused reduction needs in the dX epilogue
#include "common.h"
#include "fwd_pp.h"          // p2c_mfma_split()
#include <stdlib.h>

// bf16x3-split twin (bwd_fused3.hip): one wave per SIMD, W resident in registers; same arguments, fp32-accurate results
int p2c_bwd_fused3_launch(const float *dZ, int lddz, const float *Yfwd, int ldy, int grad_mode, const float *coef, const int32_t *pool_arg,
                          int pool_ns, const float *X, int ldx, int in_mode, const float *in_scale, const float *in_shift, const float *W, int ldw,
                          float *dX, int lddx, float *dW, int lddw, long long dw_slot_stride, float *dbias, const float *prev_stat,
                          double *bwd_partials, int M, int Co, int Ci, int coef_ld, int arg_ld, int dx_atomic, hipStream_t s);


// Workgroup barrier that only drains the LDS queue.  __syncthreads() also waits for every outstanding global
// load/store of the wave (vmcnt(0)), which would turn the register prefetch that is meant to stay in flight across
// the barrier into an exposed HBM round trip per phase.  LDS hand-offs only need lgkmcnt(0).
#define P2C_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));   // native vector: stays in registers (HIP's float4 struct copy can pin an array in scratch)
typedef float v2f __attribute__((ext_vector_type(2)));

struct BwdFusedArgs {
    const float *dz; int lddz;            // upstream gradient (or pooled gradient [G,Co] when gmode == 2)
    const float *y; int ldy;              // this layer's pre-BN output [M,Co]
    const float *coef;                    // [5][Co] (gmode >= 1)
    const int32_t *arg; int ns;           // gmode == 2
    const float *x; int ldx;              // layer-below pre-BN output (or raw input) [M,Ci]
    const float *in_scale, *in_shift;     // imode == 1
    const float *w; int ldw;              // [Co,Ci]
    float *dx; int lddx;                  // [M,Ci] or NULL
    float *dw; int lddw;                  // [Co,Ci], accumulated atomically
    float *dbias;                         // [Co] or NULL
    const float *pstat;                   // [4][Ci] of the layer below, or NULL
    double *partials;                     // [P2C_STAT_SLOTS][2][Ci] fp64 accumulators (atomic)
    int M;
    long long dw_slot_stride;        // elements between the 8 per-XCD copies of dW (0: a single copy)
    const float *w0, *b0;            // IMODE 2 (folded first layer below, bn.hip): x = its input [M,4], w0 [Ci,4], b0 [Ci] or NULL
    // a layer wider than the kernel's Co (256 -> two launches over 128 output channels each): row strides of coef / arg in the FULL
    // layer (0: Co), and dX accumulated with fire-and-forget atomics on top of what the first launch stored (the ReLU + BatchNorm
    // backward sums are linear in dX, so each launch adds its part)
    int coef_ld, arg_ld, dx_atomic;
};

// ------------------------------------------------------------------------------------------------
// LDS layouts.  What bounds an fp32-MFMA loop on gfx950 is the number of OTHER instructions the SIMD has to issue: they
// do not overlap the MFMAs - neither the wave's own nor, beyond one per MFMA, those of the second wave on the SIMD
// (tools/ubench/mfma_peak.hip, coissue.hip).  So every operand tile keeps the REDUCTION index contiguous and one
// ds_read_b128 per lane feeds four MFMAs: lane (i = lane&31, h = lane>>5) reads k = 8q+4h .. +3 and MFMA e of group q
// multiplies the e-th components (k pair {8q+e, 8q+4+e}; A and B use the same pairing, so the sum is the same).
//   dW = dY^T . act(X)   reduces over the rows m:  dYs[co][m], Xt[ci][m]   (both transposed on the way in)
//   dX = dY . W          reduces over co:          Wt[ci][co] via 16-byte reads; dYs[co][m] (m across lanes) via ds_read2_b32
// A transposing store of a float4 (4 channels of one row) is 4 scalar writes to 4 channel rows.  Row stride BM+4 alone
// would put the 16 float4-columns a wave writes in one instruction on 2 banks; rotating channel row c by 4*((c>>3)&7)
// positions (mod BM, so groups of 4 rows stay aligned for the 16-byte reads) spreads them over 8, and mapping a wave's
// 64 lanes to 16 float4-columns x 4 rows covers the other 4: conflict-free.
// ------------------------------------------------------------------------------------------------
// Thread -> tile elements.  A thread owns, per unit u, R CONSECUTIVE rows of one float4 column c4 (4 channels): p = tid + 256 u,
// c4 = p % (C/4), row group g = p / (C/4), rows R g .. R g + R - 1, with R = 4 (R = 2 only for the 64-channel x 32-row tile, which has
// half a 4-row block per thread).  Global loads: the lanes of a wave walk c4 first, so load j of a unit covers whole 256/512-byte rows.
// LDS: the thread transposes its R x 4 block in registers and writes ONE 16-byte (8-byte) piece per channel - R consecutive m of
// channel row c - instead of R x 4 scalar stores (ds_write_b128: 8 lanes per cycle = c4 0..7 at one g; with the rotation of fused_pos
// those start at banks 0,16,4,20,8,24,12,28: conflict-free).
template <int C, int BM>
struct TileMap {
    static constexpr int C4 = C / 4;
    static constexpr int R = (BM * C4 / 256) >= 4 ? 4 : 2;            // rows per unit
    static constexpr int NU = BM * C4 / (256 * R);                    // units per thread
    static constexpr int NV = NU * R;                                 // float4 registers per thread (= BM * C / 4 / 256)
    static __device__ __forceinline__ void unit(int tid, int u, int &row0, int &c4)
    {
        const int p = tid + 256 * u;
        c4 = p % C4;
        row0 = (p / C4) * R;
    }
};
template <int BM>
__device__ __forceinline__ int fused_pos(int c, int m) { return c * (BM + 4) + ((m + 4 * ((c >> 3) & 7)) & (BM - 1)); }

template <int R>
__device__ __forceinline__ void lds_store_rows(float *d, const float (&v)[4])
{
    if (R == 4) *reinterpret_cast<v4f *>(d) = v4f{v[0], v[1], v[2], v[3]};
    else *reinterpret_cast<v2f *>(d) = v2f{v[0], v[1]};
}

// Raw tile fetch (no arithmetic: the values stay in flight during the MFMAs of the previous tile).
// GMODE 2 (the layer feeds the max-pool): the gradient and the winner index are per GROUP of ns rows, and a tile spans at
// most two groups (host-checked: ns % 16 == 0, 2 ns >= BM), so a thread fetches its float4 column of those two group rows
// once instead of once per row: UDZ = 2 register sets instead of UDY.
// LDZ / LDY / LDX > 0: the row strides are compile-time constants (dense tensors: stride = channel count), so the R rows of a unit are
// ONE 64-bit address plus immediate offsets; 0: the stride comes from the argument block (a 64-bit multiply-add per row).
template <int GMODE, int IMODE, int Co, int Ci, int BM, int UDY, int UDZ, int UX, int LDZ, int LDY, int LDX>
__device__ __forceinline__ void fused_load_tile(const BwdFusedArgs &a, int tid, int t, float4 (&rdz)[UDZ], float4 (&ry)[UDY],
                                                int4 (&rarg)[UDZ], v4f (&rx)[UX])
{
    using MY = TileMap<Co, BM>;
    using MX = TileMap<Ci, BM>;
    static_assert(MY::NV == UDY && MX::NV == UX, "");
    const int m0 = t * BM;
    const bool full = m0 + BM <= a.M;                 // uniform: only the last tile can be ragged
    const int lddz = LDZ ? LDZ : a.lddz, ldy = LDY ? LDY : a.ldy, ldx = LDX ? LDX : a.ldx;
    if (GMODE == 2) {
        int row, c4;
        MY::unit(tid, 0, row, c4);                    // c4 is the same for every unit of the thread (256 % C4 == 0)
        const int g0 = m0 / a.ns, glast = (a.M - 1) / a.ns;
#pragma unroll
        for (int sI = 0; sI < UDZ; ++sI) {
            const int grp = min(g0 + sI, glast);
            rdz[sI] = *reinterpret_cast<const float4 *>(a.dz + (size_t)grp * a.lddz + c4 * 4);
            rarg[sI] = *reinterpret_cast<const int4 *>(a.arg + (size_t)grp * (a.arg_ld ? a.arg_ld : Co) + c4 * 4);
        }
    }
#pragma unroll
    for (int u = 0; u < MY::NU; ++u) {
        int row0, c4;
        MY::unit(tid, u, row0, c4);
        if (full) {
            const float *pz = a.dz + (size_t)(m0 + row0) * lddz + c4 * 4, *py = a.y + (size_t)(m0 + row0) * ldy + c4 * 4;
#pragma unroll
            for (int j = 0; j < MY::R; ++j) {
                if (GMODE != 2) rdz[u * MY::R + j] = *reinterpret_cast<const float4 *>(pz + j * lddz);
                if (GMODE >= 1) ry[u * MY::R + j] = *reinterpret_cast<const float4 *>(py + j * ldy);
            }
        } else {
#pragma unroll
            for (int j = 0; j < MY::R; ++j) {
                const int r = min(m0 + row0 + j, a.M - 1);
                if (GMODE != 2) rdz[u * MY::R + j] = *reinterpret_cast<const float4 *>(a.dz + (size_t)r * lddz + c4 * 4);
                if (GMODE >= 1) ry[u * MY::R + j] = *reinterpret_cast<const float4 *>(a.y + (size_t)r * ldy + c4 * 4);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < MX::NU; ++u) {
        int row0, c4;
        MX::unit(tid, u, row0, c4);
        if (full) {
            const float *px = a.x + (size_t)(m0 + row0) * ldx + (IMODE == 2 ? 0 : c4 * 4);
#pragma unroll
            for (int j = 0; j < MX::R; ++j) rx[u * MX::R + j] = *reinterpret_cast<const v4f *>(px + j * ldx);
        } else {
#pragma unroll
            for (int j = 0; j < MX::R; ++j) {
                const int r = min(m0 + row0 + j, a.M - 1);
                rx[u * MX::R + j] = *reinterpret_cast<const v4f *>(a.x + (size_t)r * ldx + (IMODE == 2 ? 0 : c4 * 4));
            }
        }
    }
}

// dY = gs*(dZ*[scale*Y+shift>0]) + q*Y + p from the raw registers (cf[] = this thread's 4 channels of coef) -> dYs[co][m];
// raw X -> Xt[ci][m] (act_in is applied when the dW operand is read: the dX epilogue needs the raw values).
// IMODE 2: rx[] holds the 16-byte INPUT rows of a folded first layer; the pre-BN values y0 = W0 x + b0 of the thread's four
// channels are rebuilt here (w0r/b0r = its rows of W0, b0) and the input row itself is parked in x0s[row][4] for the
// weight-gradient sums of that layer (see the epilogue).
template <int GMODE, int IMODE, int Co, int Ci, int BM, int UDY, int UDZ, int UX>
__device__ __forceinline__ void fused_store_tile(const BwdFusedArgs &a, int tid, int t, float *dy, float *xt, const float4 (&rdz)[UDZ],
                                                 const float4 (&ry)[UDY], const int4 (&rarg)[UDZ], const v4f (&rx)[UX],
                                                 const float4 (&cf)[5], const v4f (&w0r)[4], const v4f &b0r, float *x0s)
{
    using MY = TileMap<Co, BM>;
    using MX = TileMap<Ci, BM>;
    const int m0 = t * BM;
    const bool full = m0 + BM <= a.M;                 // uniform
#pragma unroll
    for (int u = 0; u < MY::NU; ++u) {
        int row0, c4;
        MY::unit(tid, u, row0, c4);
        float ox[4], oy[4], oz[4], ow[4];             // [row j] of channels 4c4 .. 4c4+3
#pragma unroll
        for (int j = 0; j < MY::R; ++j) {
            const int m = m0 + row0 + j;
            float4 g;
            if (GMODE == 2) {
                const int gi = m / a.ns;              // rows R g .. R g + R - 1 lie in one group (ns % 16 == 0)
                const int jj = m - gi * a.ns;         // the row's index inside its group
                const bool second = gi != m0 / a.ns;
                const float4 d = second ? rdz[UDZ - 1] : rdz[0];
                const int4 w = second ? rarg[UDZ - 1] : rarg[0];
                g = make_float4(w.x == jj ? d.x : 0.f, w.y == jj ? d.y : 0.f, w.z == jj ? d.z : 0.f, w.w == jj ? d.w : 0.f);
            } else {
                g = rdz[u * MY::R + j];
            }
            float4 o = g;
            if (GMODE >= 1) {
                const float4 yy = ry[u * MY::R + j];
                // the ReLU mask with the forward's own two roundings (mul, add); the affine rest as fmas
                o.x = __builtin_fmaf(cf[2].x, (cf[0].x * yy.x + cf[1].x > 0.f) ? g.x : 0.f, __builtin_fmaf(cf[3].x, yy.x, cf[4].x));
                o.y = __builtin_fmaf(cf[2].y, (cf[0].y * yy.y + cf[1].y > 0.f) ? g.y : 0.f, __builtin_fmaf(cf[3].y, yy.y, cf[4].y));
                o.z = __builtin_fmaf(cf[2].z, (cf[0].z * yy.z + cf[1].z > 0.f) ? g.z : 0.f, __builtin_fmaf(cf[3].z, yy.z, cf[4].z));
                o.w = __builtin_fmaf(cf[2].w, (cf[0].w * yy.w + cf[1].w > 0.f) ? g.w : 0.f, __builtin_fmaf(cf[3].w, yy.w, cf[4].w));
            }
            if (!full) {                              // rows past M contribute nothing to dW / dbias
                const float ok = m < a.M ? 1.f : 0.f;
                o.x *= ok; o.y *= ok; o.z *= ok; o.w *= ok;
            }
            ox[j] = o.x; oy[j] = o.y; oz[j] = o.z; ow[j] = o.w;
        }
        float *d = dy + fused_pos<BM>(c4 * 4, row0);  // channels 4c4 .. 4c4+3 share (c >> 3): same rotation; row0 % R == 0
        lds_store_rows<MY::R>(d, ox);
        lds_store_rows<MY::R>(d + (BM + 4), oy);
        lds_store_rows<MY::R>(d + 2 * (BM + 4), oz);
        lds_store_rows<MY::R>(d + 3 * (BM + 4), ow);
    }
#pragma unroll
    for (int u = 0; u < MX::NU; ++u) {
        int row0, c4;
        MX::unit(tid, u, row0, c4);
        float vx[4], vy[4], vz[4], vw[4];
#pragma unroll
        for (int j = 0; j < MX::R; ++j) {
            v4f v = rx[u * MX::R + j];
            if (IMODE == 2) {
                const v4f x = v;
                v.x = p2c_l0_preact(w0r[0].x, w0r[0].y, w0r[0].z, b0r.x, x.x, x.y, x.z);
                v.y = p2c_l0_preact(w0r[1].x, w0r[1].y, w0r[1].z, b0r.y, x.x, x.y, x.z);
                v.z = p2c_l0_preact(w0r[2].x, w0r[2].y, w0r[2].z, b0r.z, x.x, x.y, x.z);
                v.w = p2c_l0_preact(w0r[3].x, w0r[3].y, w0r[3].z, b0r.w, x.x, x.y, x.z);
                if (c4 == 0) *reinterpret_cast<v4f *>(&x0s[(row0 + j) * 4]) = (m0 + row0 + j < a.M) ? x : v4f{0.f, 0.f, 0.f, 0.f};
            }
            vx[j] = v.x; vy[j] = v.y; vz[j] = v.z; vw[j] = v.w;
        }
        float *d = xt + fused_pos<BM>(c4 * 4, row0);
        lds_store_rows<MX::R>(d, vx);
        lds_store_rows<MX::R>(d + (BM + 4), vy);
        lds_store_rows<MX::R>(d + 2 * (BM + 4), vz);
        lds_store_rows<MX::R>(d + 3 * (BM + 4), vw);
    }
}

// COT = Co/64, CIT = Ci/64 (1 or 2).  Waves: for dW a 2x2 grid over Co x Ci (wave tile COT*32 x CIT*32);
// for dX a WR x WC grid with WC = Ci/32 columns of 32, WR = 4/WC, so BM = 32*WR rows per tile.
// ------------------------------------------------------------------------------------------------
// Ping-pong structure: 512 threads = two half-workgroups of 4 waves that share W in LDS and own one tile buffer
// each, one phase apart (an extra barrier at the start of half 1 / end of half 0): while half A streams the MFMAs of
// its tile, half B - on the same SIMDs - stores its dX tile, reduces the BN-backward sums, copies its next tile to LDS
// and issues the global loads of the one after.  The instruction streams of the two do not overlap (see above), but the
// memory latencies of one hide behind the MFMAs of the other.  Every barrier is workgroup-wide; both halves execute the
// same number of them.
// ------------------------------------------------------------------------------------------------
// EX: extra input columns [Ci, Ci+EX) of X (the 3 relative coordinates + pad of a grouped layer, laid out AFTER the
// feature block): they only contribute EX more columns of dW (no dX, no act_in), accumulated on the VALU.
// DENSE (compile-time row strides for dense tensors: immediates instead of 64-bit address arithmetic) is kept as a switch but not
// instantiated: measured on the 128x128 layer it changes nothing (232-247 us either way) and costs two spilled registers.
template <int COT, int CIT, int GMODE, int IMODE, bool NEED_DX, bool HAS_STATS, int EX, bool DENSE = false>
__global__ void __launch_bounds__(512, 2) bwd_fused_pp_kernel(BwdFusedArgs a)
{
    constexpr int Co = 64 * COT, Ci = 64 * CIT;
    constexpr int LDZ = (DENSE && GMODE != 2) ? Co : 0, LDY = DENSE ? Co : 0, LDX = DENSE ? (IMODE == 2 ? 4 : Ci + EX) : 0;
    constexpr int WC = Ci / 32, WR = 4 / WC, BM = 32 * WR;
    constexpr int LDT = Co + 4, LDM = BM + 4, NG = BM / 8, NQ = Co / 8;
    constexpr int UDY = BM * Co / 4 / 256, UX = BM * Ci / 4 / 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Wt = smem;                              // [Ci][LDT]   W transposed: co contiguous      shared by both halves
    float *dYs = Wt + Ci * LDT;                    // [2 halves][Co][LDM]
    float *Xt = dYs + 2 * Co * LDM;                // [2 halves][Ci][LDM]
    float *red = Xt + 2 * Ci * LDM;                // [2][Ci]
    float *Xe = red + 6 * Ci;                      // [2 halves][BM][4] extra input columns (EX > 0); IMODE 2: [2 halves][2 parities][BM][4] input rows
    static_assert(!(EX > 0 && IMODE == 2), "");

    const int half = threadIdx.x >> 8, tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    P2C_TR_WG(0);
    const int l31 = lane & 31, lh = lane >> 5;
    const int wi = wave >> 1, wj = wave & 1;       // dW wave grid (within the half)
    const int wr = wave / WC, wc = wave % WC;      // dX wave grid
    const int ntiles = (a.M + BM - 1) / BM;
    const int nk = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // tiles of this workgroup: blockIdx.x + k*gridDim.x
    const int niter = (nk + 1) / 2;                // per half
    float *dy = dYs + half * Co * LDM;
    float *xt = Xt + half * Ci * LDM;
    float *xe = Xe + half * BM * 4;
    constexpr int EPT = EX > 0 ? (EX * Co + 255) / 256 : 1;     // extra dW columns per thread (thread -> co = tid % Co)
    float acce[EPT];
#pragma unroll
    for (int i = 0; i < EPT; ++i) acce[i] = 0.f;
    v4f rxe = {0.f, 0.f, 0.f, 0.f};

    constexpr int UDZ = GMODE == 2 ? 2 : UDY;
    float4 rdz[UDZ], ry[UDY];
    v4f rx[UX];
    int4 rarg[UDZ];
    auto tile_of = [&](int k) { return (int)blockIdx.x + k * (int)gridDim.x; };
    // the first tile's rows are requested BEFORE W is staged: their HBM round trip then runs under the weight copy and the constant set-up
    // below instead of after them (the prologue was 13-29 k cycles of a workgroup's 110-750 k)
    fused_load_tile<GMODE, IMODE, Co, Ci, BM, UDY, UDZ, UX, LDZ, LDY, LDX>(a, tid, tile_of(half < nk ? half : 0), rdz, ry, rarg, rx);
    P2C_TR_WG_MID(2);
    if (NEED_DX) {
        // W [Co,Ci] -> Wt[ci][co]: 4 x 4 blocks transposed in registers, 16-byte LDS stores.  Lanes 0-7 of each group of 8 take 8
        // consecutive co-quads (one contiguous 128-byte piece of a Wt row: conflict-free), the 8 groups of a wave 8 consecutive ci-quads
        // (so the four row loads of a lane group read whole 128-byte pieces of 8 rows of W).  The scalar transposing stores this replaces
        // put a wave's 64 writes on two banks and made the copy 11 k of the kernel's 21 k prologue cycles.
        constexpr int RH = Co / 32, NBLK = (Co / 4) * (Ci / 4), NWU = (NBLK + 511) / 512;      // 4 x 4 blocks: 256 .. 1024, per thread 1 or 2
        v4f v[NWU][4];
#pragma unroll
        for (int u = 0; u < NWU; ++u) {                // every row load of the thread in flight before the first store
            const int p = min((int)threadIdx.x + 512 * u, NBLK - 1), rest = p >> 6;
            const int r4 = (rest % RH) * 8 + (p & 7), c4 = (rest / RH) * 8 + ((p >> 3) & 7);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[u][j] = *reinterpret_cast<const v4f *>(a.w + (size_t)(4 * r4 + j) * a.ldw + 4 * c4);
        }
#pragma unroll
        for (int u = 0; u < NWU; ++u) {
            const int p = threadIdx.x + 512 * u, rest = p >> 6;
            const int r4 = (rest % RH) * 8 + (p & 7), c4 = (rest / RH) * 8 + ((p >> 3) & 7);
            if (p < NBLK) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<v4f *>(&Wt[(4 * c4 + i) * LDT + 4 * r4]) = v4f{v[u][0][i], v[u][1][i], v[u][2][i], v[u][3][i]};
            }
        }
    }
    P2C_TR_WG_MID(3);
    float isc[CIT], ish[CIT];
#pragma unroll
    for (int t = 0; t < CIT; ++t) {
        const int ci = wj * (CIT * 32) + t * 32 + l31;
        isc[t] = IMODE >= 1 ? a.in_scale[ci] : 1.f;
        ish[t] = IMODE >= 1 ? a.in_shift[ci] : 0.f;
    }
    const int xcol = wc * 32 + l31;
    float psc = 0.f, psh = 0.f, pmu = 0.f, pis = 0.f;
    if (NEED_DX && HAS_STATS) { psc = a.pstat[xcol]; psh = a.pstat[Ci + xcol]; pmu = a.pstat[2 * Ci + xcol]; pis = a.pstat[3 * Ci + xcol]; }
    const float npm = -pmu * pis;
    // coef of the thread's float4 column (it does not depend on the unit index i)
    float4 cf[5];
    int row0_, c40;
    TileMap<Co, BM>::unit(tid, 0, row0_, c40);
    auto load_cf = [&]() {
#pragma unroll
        for (int i = 0; i < 5; ++i)
            cf[i] = GMODE >= 1 ? *reinterpret_cast<const float4 *>(a.coef + i * (a.coef_ld ? a.coef_ld : Co) + c40 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    load_cf();
    v4f w0r[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    v4f b0r = {0.f, 0.f, 0.f, 0.f};
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;            // IMODE 2: sum_m g[m, xcol] * x0[m, 0..2] (weight gradient of the folded layer)
    if (IMODE == 2) {
        int rowx_, c4x;
        TileMap<Ci, BM>::unit(tid, 0, rowx_, c4x);
#pragma unroll
        for (int e = 0; e < 4; ++e) w0r[e] = *reinterpret_cast<const v4f *>(a.w0 + (size_t)(4 * c4x + e) * 4);
        if (a.b0) b0r = *reinterpret_cast<const v4f *>(a.b0 + 4 * c4x);
    }
    float *x0h = Xe + half * 2 * BM * 4;
    // per-lane LDS offsets (floats) of the operand reads; they are the same for every tile
    int adw[COT][NG], bdw[CIT][NG];                // dW: group q of block i / j  (16-byte reads over m)
#pragma unroll
    for (int q = 0; q < NG; ++q) {
#pragma unroll
        for (int i = 0; i < COT; ++i) adw[i][q] = fused_pos<BM>(wi * (COT * 32) + i * 32 + l31, 8 * q + 4 * lh);
#pragma unroll
        for (int j = 0; j < CIT; ++j) bdw[j][q] = fused_pos<BM>(wj * (CIT * 32) + j * 32 + l31, 8 * q + 4 * lh);
    }
    int adx[8];                                    // dX: A element (co = 8q+4h+e, m = wr*32+l31): rotation class q & 7
#pragma unroll
    for (int r = 0; r < 8; ++r) adx[r] = 4 * lh * LDM + ((wr * 32 + l31 + 4 * r) & (BM - 1));
    const int bdx = xcol * LDT + 4 * lh;           // dX: B = Wt[ci][8q + 4h .. +3]
    int ayp[4];                                    // raw x of the lane's output column at its 16 rows (4 groups of 4)
#pragma unroll
    for (int g = 0; g < 4; ++g) ayp[g] = fused_pos<BM>(xcol, wr * 32 + 8 * g + 4 * lh);

    float s1 = 0.f, s2 = 0.f, dbacc = 0.f;
    f32x16 accW[COT][CIT];
#pragma unroll
    for (int i = 0; i < COT; ++i)
#pragma unroll
        for (int j = 0; j < CIT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) accW[i][j][r] = 0.f;
    f32x16 accX;
    float yp[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { accX[r] = 0.f; yp[r] = 0.f; }

    P2C_TR_WG_MID(4);
    // prologue: this half's first tile -> its LDS buffer (its rows were requested at the top of the kernel); its second tile -> registers (in flight)
    {
        const int k0 = half, k1 = half + 2;
        fused_store_tile<GMODE, IMODE, Co, Ci, BM, UDY, UDZ, UX>(a, tid, tile_of(k0 < nk ? k0 : 0), dy, xt, rdz, ry, rarg, rx, cf, w0r, b0r, x0h);
        if (EX > 0 && tid < BM) {
            const int m = min(tile_of(k0 < nk ? k0 : 0) * BM + tid, a.M - 1);
            *reinterpret_cast<v4f *>(&xe[tid * 4]) = *reinterpret_cast<const v4f *>(a.x + (size_t)m * a.ldx + Ci);
        }
        fused_load_tile<GMODE, IMODE, Co, Ci, BM, UDY, UDZ, UX, LDZ, LDY, LDX>(a, tid, tile_of(k1 < nk ? k1 : 0), rdz, ry, rarg, rx);
        if (EX > 0 && tid < BM) {
            const int m = min(tile_of(k1 < nk ? k1 : 0) * BM + tid, a.M - 1);
            rxe = *reinterpret_cast<const v4f *>(a.x + (size_t)m * a.ldx + Ci);
        }
    }
    P2C_TR_WG_MID(5);
    __syncthreads();
#ifndef P2C_LOCKSTEP                               // (tools/fused_trace.py --lockstep: both halves in the same phase - measured slower)
    if (half == 1) P2C_LDS_BARRIER();             // run one phase behind half 0
#endif
    P2C_TR_WG_MID(0);
    for (int it = 0; it < niter; ++it) {
        const int k = 2 * it + half;
        const bool valid = k < nk;                 // uniform within the half
        const int m0 = tile_of(valid ? k : 0) * BM;
        // ================= MFMA phase =================
        P2C_TR(0);
        if (valid) {
            // ---- dW += dY^T . act(X): per group of 8 rows, COT + CIT 16-byte reads feed 4*COT*CIT MFMAs; the reads of group
            //      q+1 are issued before the MFMAs of group q (pinned: the scheduler would sink them next to their use)
            v4f av[COT], bv[CIT], an[COT], bn[CIT];
#pragma unroll
            for (int i = 0; i < COT; ++i) av[i] = *reinterpret_cast<const v4f *>(dy + adw[i][0]);
#pragma unroll
            for (int j = 0; j < CIT; ++j) bv[j] = *reinterpret_cast<const v4f *>(xt + bdw[j][0]);
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                if (q + 1 < NG) {
#pragma unroll
                    for (int i = 0; i < COT; ++i) an[i] = *reinterpret_cast<const v4f *>(dy + adw[i][q + 1]);
#pragma unroll
                    for (int j = 0; j < CIT; ++j) bn[j] = *reinterpret_cast<const v4f *>(xt + bdw[j][q + 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (IMODE >= 1) {
#pragma unroll
                    for (int j = 0; j < CIT; ++j) {
                        bv[j].x = fmaxf(isc[j] * bv[j].x + ish[j], 0.f);
                        bv[j].y = fmaxf(isc[j] * bv[j].y + ish[j], 0.f);
                        bv[j].z = fmaxf(isc[j] * bv[j].z + ish[j], 0.f);
                        bv[j].w = fmaxf(isc[j] * bv[j].w + ish[j], 0.f);
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < COT; ++i)
#pragma unroll
                        for (int j = 0; j < CIT; ++j) accW[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][e], bv[j][e], accW[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < COT; ++i) av[i] = an[i];
#pragma unroll
                for (int j = 0; j < CIT; ++j) bv[j] = bn[j];
            }
            P2C_TR(1);
            if (GMODE == 0 && a.dbias && tid < Co) {
                float sb = 0.f;
#pragma unroll 8
                for (int s = 0; s < BM; ++s) sb += dy[tid * LDM + s];      // the BM row entries in their rotated order
                dbacc += sb;
            }
            if (EX > 0) {                               // dW[co, Ci + e] += sum_m dY[m,co] * x[m, Ci + e]   (rows past M have dY == 0)
                const int co = tid % Co, e0 = (tid / Co) * EPT;
#pragma unroll 8
                for (int s = 0; s < BM; ++s) {
                    const float d = dy[fused_pos<BM>(co, s)];
#pragma unroll
                    for (int i = 0; i < EPT; ++i) acce[i] += d * xe[s * 4 + e0 + i];
                }
            }
            if (NEED_DX) {
                // ---- dX = dY . W: per group of 8 output channels co, one 16-byte read of Wt and two ds_read2_b32 of dYs
                //      (rows co, co+1 / co+2, co+3 at this lane's m) feed 4 MFMAs; group q+1 is fetched under the MFMAs of q
#pragma unroll
                for (int r = 0; r < 16; ++r) accX[r] = 0.f;
                const float *ap = dy, *bp = Wt + bdx;
                float a0[4], a1[4];
                v4f b0, b1;
#define P2C_DXLOAD(A_, B_, Q_)                                            \
    do {                                                                  \
        const float *p_ = ap + adx[(Q_) & 7] + 8 * (Q_) * LDM;            \
        A_[0] = p_[0]; A_[1] = p_[LDM]; A_[2] = p_[2 * LDM]; A_[3] = p_[3 * LDM]; \
        B_ = *reinterpret_cast<const v4f *>(bp + 8 * (Q_));               \
    } while (0)
                P2C_DXLOAD(a0, b0, 0);
#pragma unroll
                for (int q = 0; q < NQ; q += 2) {
                    P2C_DXLOAD(a1, b1, q + 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) accX = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], accX, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (q + 2 < NQ) P2C_DXLOAD(a0, b0, q + 2);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) accX = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], accX, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#undef P2C_DXLOAD
                // ReLU + BatchNorm-backward sums of the layer below, at the tail of the MFMA phase (before the barrier) rather than in the phase
                // after it: there they ran in the shadow of the other half's MFMA stream, where a wave gets about one issue slot per MFMA
                // (tools/ubench/coissue.hip, interleave.hip); here they compete with the other half's memory phase only, and accX / yp
                // are dead before the next tile is staged (236 instead of 256 registers; the folded-input variant no longer spills).
                // Measured in the step: 4.91 -> 4.87 ms.  Going further - the dY transform of the next tile in front of the dX MFMAs, the
                // sums under the next tile's dW MFMAs, coef in LDS - shortens the period of a workgroup from 13.4 k to 11.4 k cycles and
                // makes the step SLOWER (4.93 ms): the kernel runs at the socket power limit, the shader clock gives way.
                if (HAS_STATS) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const v4f v = *reinterpret_cast<const v4f *>(xt + ayp[g]);
                        yp[4 * g] = v.x; yp[4 * g + 1] = v.y; yp[4 * g + 2] = v.z; yp[4 * g + 3] = v.w;
                    }
                    const float *x0t = x0h + (it & 1) * BM * 4;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = accX[r];
                        const float g = (psc * yp[r] + psh > 0.f) ? v : 0.f;          // mask: the forward's two roundings
                        s1 += g;
                        s2 = __builtin_fmaf(g, __builtin_fmaf(yp[r], pis, npm), s2);    // g * (y - mean) * invstd
                        if (IMODE == 2) {
                            const v4f x = *reinterpret_cast<const v4f *>(&x0t[(wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 4]);   // zero past M
                            g0 = __builtin_fmaf(g, x.x, g0);
                            g1 = __builtin_fmaf(g, x.y, g1);
                            g2 = __builtin_fmaf(g, x.z, g2);
                        }
                    }
                }
            }
        }
        P2C_TR(2);
        P2C_LDS_BARRIER();
        P2C_TR(3);
        // ================= non-MFMA phase (the other half is in its MFMA phase) =================
        __builtin_amdgcn_s_setprio(1);
#ifndef P2C_TRACE_NODATA
        // Order matters for the memory counters: the next tile is staged FIRST (its operands were fetched a whole phase ago), the
        // prefetch of the tile after it is issued SECOND, and the dX stores of the finished tile go out LAST.  vmcnt retires in issue
        // order and the compiler cannot count the stores across this block's branches, so a wait for the prefetched registers placed
        // after freshly issued stores drains those stores too (a full write round trip per tile in the previous order: staging
        // started with s_waitcnt vmcnt(8..0) right behind 16 global stores).
        auto stage_next = [&]() {
        if (IMODE != 2) asm volatile("" ::"v"(rx[UX - 1]));        // the newest prefetch register: wait here, while nothing younger is outstanding
            if (EX > 0) asm volatile("" ::"v"(rxe));
            {
                const int k2 = k + 2, k4 = k + 4;
                if (k2 < nk) fused_store_tile<GMODE, IMODE, Co, Ci, BM, UDY, UDZ, UX>(a, tid, tile_of(k2), dy, xt, rdz, ry, rarg, rx, cf, w0r, b0r, x0h + ((it + 1) & 1) * BM * 4);
                P2C_TR(4);
                if (EX > 0 && tid < BM) *reinterpret_cast<v4f *>(&xe[tid * 4]) = rxe;
                fused_load_tile<GMODE, IMODE, Co, Ci, BM, UDY, UDZ, UX, LDZ, LDY, LDX>(a, tid, tile_of(k4 < nk ? k4 : 0), rdz, ry, rarg, rx);   // unconditional: stays in registers
                if (EX > 0 && tid < BM) {
                    const int m = min(tile_of(k4 < nk ? k4 : 0) * BM + tid, a.M - 1);
                    rxe = *reinterpret_cast<const v4f *>(a.x + (size_t)m * a.ldx + Ci);
                }
            }
        };
        auto epilogue = [&]() {
        if (valid && NEED_DX) {
                if (IMODE == 2) {
                    // the layer below is folded: nobody reads dX; what its backward needs from it are the BatchNorm sums (below) and
                    // G[c, :] = sum_m g[m, c] x0[m, :], from which its weight gradient is assembled (p2c_fold0_bwd_finalize_f32)
                } else if (m0 + BM <= a.M) {
                    const int lddx = DENSE ? Ci : a.lddx;
                    float *dxp = a.dx + (size_t)(m0 + wr * 32 + 4 * lh) * lddx + xcol;      // one 64-bit address; the 16 rows are offsets of it
                    if (a.dx_atomic) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) atomicAdd(dxp + ((r & 3) + 8 * (r >> 2)) * lddx, accX[r]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) dxp[((r & 3) + 8 * (r >> 2)) * lddx] = accX[r];
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (m < a.M) { if (a.dx_atomic) atomicAdd(&a.dx[(size_t)m * a.lddx + xcol], accX[r]); else a.dx[(size_t)m * a.lddx + xcol] = accX[r]; }
                    }
                }
            }
        };
        if (IMODE == 2) {          // folded layer below: no dX stores at all - statistics first (its registers die before the staging)
            epilogue();
            P2C_TR(5);
            stage_next();
        } else {
            stage_next();
            P2C_TR(5);
            epilogue();
        }
#endif
        __builtin_amdgcn_s_setprio(0);
        P2C_TR(6);
        P2C_LDS_BARRIER();                         // the prefetch above stays in flight across this barrier
        P2C_TR(7);
    }
#ifndef P2C_LOCKSTEP
    if (half == 0) P2C_LDS_BARRIER();
#endif
    P2C_TR_WG_MID(1);
    // ---------------- flush: half 1 hands its dW accumulators to half 0 through LDS (the tile buffers and W are dead
    // now), half 0 adds them and issues ONE set of atomics per workgroup into the slot of its XCD (blockIdx % 8 is the
    // XCD the dispatcher places the workgroup on, so the read-modify-writes stay inside one L2; the host sums the 8
    // slots).  256 workgroups hammering one copy of dW cost ~50 us of serialized L2 atomics per launch.
    __syncthreads();
    {
        float *ex = smem;                          // [COT*CIT*16][256] floats <= Co*Ci: fits in the W region for every shape
        if (half == 1) {
#pragma unroll
            for (int i = 0; i < COT; ++i)
#pragma unroll
                for (int j = 0; j < CIT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ex[((i * CIT + j) * 16 + r) * 256 + tid] = accW[i][j][r];
        }
        __syncthreads();
        if (half == 0) {
            float *dws = a.dw + (size_t)(blockIdx.x & 7) * a.dw_slot_stride;
#pragma unroll
            for (int i = 0; i < COT; ++i)
#pragma unroll
                for (int j = 0; j < CIT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = wi * (COT * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const int ci = wj * (CIT * 32) + j * 32 + l31;
                        atomicAdd(&dws[(size_t)co * a.lddw + ci], accW[i][j][r] + ex[((i * CIT + j) * 16 + r) * 256 + tid]);
                    }
        }
    }
    if (EX > 0) {
        float *dws = a.dw + (size_t)(blockIdx.x & 7) * a.dw_slot_stride;
        const int co = tid % Co, e0 = (tid / Co) * EPT;
#pragma unroll
        for (int i = 0; i < EPT; ++i)
            if (e0 + i < EX) atomicAdd(&dws[(size_t)co * a.lddw + Ci + e0 + i], acce[i]);
    }
    if (GMODE == 0 && a.dbias && tid < Co) atomicAdd(&a.dbias[tid], dbacc);
    if (NEED_DX && HAS_STATS) {
        constexpr int NS_ = IMODE == 2 ? 5 : 2;         // sums per column: s1, s2 (+ G[.,0..2] for a folded layer below)
        for (int u = threadIdx.x; u < NS_ * Ci; u += 512) red[u] = 0.f;
        __syncthreads();
        float sv[5] = {s1, s2, g0, g1, g2};
#pragma unroll
        for (int q = 0; q < NS_; ++q) {
            const float t = sv[q] + __shfl_xor(sv[q], 32);
            if (lh == 0) atomicAdd(&red[q * Ci + xcol], t);
        }
        __syncthreads();
        for (int u = threadIdx.x; u < NS_ * Ci; u += 512) {
            double *o = a.partials + (size_t)(blockIdx.x % P2C_STAT_SLOTS) * NS_ * Ci;
            atomicAdd(&o[u], (double)red[u]);
        }
    }
    P2C_TR_WG(1);
}

static int fused_grid(int M, int Ci)
{
    const int BM = Ci >= 128 ? 32 : 64;
    const int ntiles = (M + BM - 1) / BM;
    return ntiles < 256 ? ntiles : 256;
}

extern "C" int p2c_linear_bwd_fused_parts(int M, int Ci) { return fused_grid(M, Ci); }
extern "C" int p2c_linear_bwd_fused_supported(int Co, int Ci, int in_mode)
{
    if (Co == 128 && Ci == 132 && in_mode == 0) return 2;      // 128 feature columns + 4 trailing (xyz | pad) columns
    if (Co == 256 && Ci == 128 && (in_mode == 0 || in_mode == 1)) return 3;      // two launches over 128 output channels each
    return (Co == 64 || Co == 128) && (Ci == 64 || Ci == 128) && (in_mode == 0 || in_mode == 1);
}

template <int COT, int CIT, int GMODE, int IMODE>
static int launch_fused(const BwdFusedArgs &a, int extra, hipStream_t s)
{
    constexpr int Co = 64 * COT, Ci = 64 * CIT;
    constexpr int WC = Ci / 32, WR = 4 / WC, BM = 32 * WR;
    const size_t lds = (size_t)(Ci * (Co + 4) + 2 * Co * (BM + 4) + 2 * Ci * (BM + 4) + 6 * Ci + 4 * BM * 4) * sizeof(float);
    const int grid = fused_grid(a.M, Ci);
    if (extra) {      // grouped first layer: [feats(128) | xyz(3) | pad] -> 4 extra dW columns (only this shape needs it)
        if constexpr (COT == 2 && CIT == 2 && GMODE == 1 && IMODE == 0) {
            if (!a.dx || a.pstat) return P2C_EINVAL;
            (void)hipFuncSetAttribute((const void *)bwd_fused_pp_kernel<2, 2, 1, 0, true, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds);
            hipLaunchKernelGGL((bwd_fused_pp_kernel<2, 2, 1, 0, true, false, 4>), dim3(grid), dim3(512), lds, s, a);
            P2C_LAUNCH_CHECK();
            return P2C_OK;
        } else {
            return P2C_EINVAL;
        }
    }
#define P2C_FL(DX_, ST_)                                                                                                             \
    do {                                                                                                                             \
        (void)hipFuncSetAttribute((const void *)bwd_fused_pp_kernel<COT, CIT, GMODE, IMODE, DX_, ST_, 0>,                               \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                             \
        hipLaunchKernelGGL((bwd_fused_pp_kernel<COT, CIT, GMODE, IMODE, DX_, ST_, 0>), dim3(grid), dim3(512), lds, s, a);                 \
    } while (0)
    if (a.dx && a.pstat) P2C_FL(true, true);
    else if (a.dx) P2C_FL(true, false);
    else P2C_FL(false, false);
#undef P2C_FL
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

template <int GMODE, int IMODE>
static int dispatch_shape(int Co, int Ci, int extra, const BwdFusedArgs &a, hipStream_t s)
{
    if (Co == 128 && Ci == 128) return launch_fused<2, 2, GMODE, IMODE>(a, extra, s);
    if (Co == 128 && Ci == 64) return launch_fused<2, 1, GMODE, IMODE>(a, extra, s);
    if (Co == 64 && Ci == 128) return launch_fused<1, 2, GMODE, IMODE>(a, extra, s);
    return launch_fused<1, 1, GMODE, IMODE>(a, extra, s);
}

extern "C" int p2c_linear_bwd_fused_f32(const float *dZ, int lddz, const float *Yfwd, int ldy, int grad_mode, const float *coef,
                                        const int32_t *pool_arg, int pool_ns, const float *X, int ldx, int in_mode, const float *in_scale,
                                        const float *in_shift, const float *W, int ldw, float *dX, int lddx, float *dW, int lddw,
                                        long long dw_slot_stride, float *dbias, const float *prev_stat, double *bwd_partials, int M, int Co,
                                        int Ci, void *stream)
{
    if (!dZ || !X || !W || !dW || M <= 0 || grad_mode < 0 || grad_mode > 2) return P2C_EINVAL;
    const int sup = p2c_linear_bwd_fused_supported(Co, Ci, in_mode);
    if (!sup) return P2C_EINVAL;
    const int extra = sup == 2 ? 4 : 0;
    if (extra) Ci -= extra;
    if (grad_mode >= 1 && (!Yfwd || !coef)) return P2C_EINVAL;
    if (grad_mode == 2 && (!pool_arg || pool_ns < 32 || (pool_ns & 15))) return P2C_EINVAL;    // a 64-row tile spans <= 2 groups
    if (in_mode == 1 && (!in_scale || !in_shift)) return P2C_EINVAL;
    if (prev_stat && (!bwd_partials || !dX)) return P2C_EINVAL;
    if ((lddz & 3) || (ldx & 3) || (ldw & 3) || (grad_mode >= 1 && (ldy & 3)) || ((uintptr_t)dZ & 15) || ((uintptr_t)X & 15) || ((uintptr_t)W & 15))
        return P2C_EALIGN;
    BwdFusedArgs a{dZ, lddz, Yfwd, ldy, coef, pool_arg, pool_ns, X, ldx, in_scale, in_shift, W, ldw, dX, lddx, dW, lddw, dbias, prev_stat,
                   bwd_partials, M, dw_slot_stride, nullptr, nullptr, 0, 0, 0};
    hipStream_t s = (hipStream_t)stream;
    if (sup == 3) {
        // Co = 256: W (133 KB) and the tiles do not fit in LDS together, so the layer runs as two passes over 128 output channels; the
        // second adds its dX on top of the first's (atomics without return value: no register, no wait), each accumulates its own rows
        // of dW and its share of the (linear) BatchNorm-backward sums
        if (!dX || (grad_mode == 0 && dbias)) return P2C_EINVAL;
        for (int h = 0; h < 2; ++h) {
            BwdFusedArgs b = a;
            const int c0 = 128 * h;
            if (p2c_mfma_split()) {
                const int rc3 = p2c_bwd_fused3_launch(dZ + c0, lddz, Yfwd ? Yfwd + c0 : nullptr, ldy, grad_mode, coef ? coef + c0 : nullptr,
                                                      pool_arg ? pool_arg + c0 : nullptr, pool_ns, X, ldx, in_mode, in_scale, in_shift,
                                                      W + (size_t)c0 * ldw, ldw, dX, lddx, dW + (size_t)c0 * lddw, lddw, dw_slot_stride, nullptr,
                                                      prev_stat, bwd_partials, M, 128, 128, 256, 256, h, s);
                if (rc3 != P2C_OK) return rc3;
                continue;
            }
            b.dz = dZ + c0; b.y = Yfwd ? Yfwd + c0 : nullptr; b.coef = coef ? coef + c0 : nullptr; b.arg = pool_arg ? pool_arg + c0 : nullptr;
            b.w = W + (size_t)c0 * ldw; b.dw = dW + (size_t)c0 * lddw;
            b.coef_ld = 256; b.arg_ld = 256; b.dx_atomic = h;
            int rc;
#define P2C_H(G_, I_) rc = launch_fused<2, 2, G_, I_>(b, 0, s)
            if (grad_mode == 0) { if (in_mode == 0) P2C_H(0, 0); else P2C_H(0, 1); }
            else if (grad_mode == 1) { if (in_mode == 0) P2C_H(1, 0); else P2C_H(1, 1); }
            else { if (in_mode == 0) P2C_H(2, 0); else P2C_H(2, 1); }
#undef P2C_H
            if (rc != P2C_OK) return rc;
        }
        return P2C_OK;
    }
    if (!extra && p2c_mfma_split())
        return p2c_bwd_fused3_launch(dZ, lddz, Yfwd, ldy, grad_mode, coef, pool_arg, pool_ns, X, ldx, in_mode, in_scale, in_shift, W, ldw, dX, lddx,
                                     dW, lddw, dw_slot_stride, dbias, prev_stat, bwd_partials, M, Co, Ci, 0, 0, 0, s);
#define P2C_F(G_, I_) return dispatch_shape<G_, I_>(Co, Ci, extra, a, s)
    if (grad_mode == 0) { if (in_mode == 0) P2C_F(0, 0); P2C_F(0, 1); }
    if (grad_mode == 1) { if (in_mode == 0) P2C_F(1, 0); P2C_F(1, 1); }
    if (in_mode == 0) P2C_F(2, 0);
    P2C_F(2, 1);
#undef P2C_F
}

// Backward of the layer that FOLLOWS a folded first layer (bn.hip): same kernel, IMODE 2.  The X operand (that layer's
// pre-BN output) is rebuilt from its 16-byte input rows, dX is not stored, and per column c of the folded layer the kernel
// accumulates 5 sums into partials5 [P2C_STAT_SLOTS][5][C0] (zeroed by the caller): s1 = sum g, s2 = sum g*xhat (its
// BatchNorm backward) and G[c, 0..2] = sum_m g[m,c] x0[m, 0..2] (its weight gradient, finished by p2c_fold0_bwd_finalize_f32).
extern "C" int p2c_linear_bwd_fused_fold0_f32(const float *dZ, int lddz, const float *Yfwd, int ldy, const float *coef, const float *X0, int ldx0,
                                              const float *W0, const float *b0, const float *stat0, const float *W, int ldw, float *dW, int lddw,
                                              long long dw_slot_stride, double *partials5, int M, int Co, int C0, void *stream)
{
    if (!dZ || !Yfwd || !coef || !X0 || !W0 || !stat0 || !W || !dW || !partials5 || M <= 0 || C0 != 64 || (Co != 64 && Co != 128) || ldx0 != 4)
        return P2C_EINVAL;
    if ((lddz & 3) || (ldy & 3) || (ldw & 3) || ((uintptr_t)dZ & 15) || ((uintptr_t)X0 & 15) || ((uintptr_t)W & 15) || ((uintptr_t)W0 & 15))
        return P2C_EALIGN;
    BwdFusedArgs a{dZ, lddz, Yfwd, ldy, coef, nullptr, 0, X0, ldx0, stat0, stat0 + C0, W, ldw, nullptr, 0, dW, lddw, nullptr, stat0,
                   partials5, M, dw_slot_stride, W0, b0, 0, 0, 0};
    hipStream_t s = (hipStream_t)stream;
    constexpr int Ci = 64, BM = 64;
    const int grid = fused_grid(M, Ci);
#define P2C_F0(COT_)                                                                                                                 \
    do {                                                                                                                             \
        const size_t lds = (size_t)(Ci * (64 * COT_ + 4) + 2 * 64 * COT_ * (BM + 4) + 2 * Ci * (BM + 4) + 6 * Ci + 4 * BM * 4) * sizeof(float); \
        (void)hipFuncSetAttribute((const void *)bwd_fused_pp_kernel<COT_, 1, 1, 2, true, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)lds);                                                                                         \
        hipLaunchKernelGGL((bwd_fused_pp_kernel<COT_, 1, 1, 2, true, true, 0>), dim3(grid), dim3(512), lds, s, a);                       \
    } while (0)
    if (Co == 128) P2C_F0(2); else P2C_F0(1);
#undef P2C_F0
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}
