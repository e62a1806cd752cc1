// bwd_fused3.hip -- the one-pass backward of bwd_fused.hip ( dW = dY^T . act(X),  dX = dY . W,  ReLU + BatchNorm-backward sums of the
// layer below, each of dZ / Y / X read once ) on the bf16 matrix pipe at fp32 accuracy: every operand is split into three bf16 pieces
// (fwd_pp3.hip explains the arithmetic) and a 32x32x16 block costs 6 x 32 matrix-pipe cycles instead of 8 x 64.
//
// What measuring the split FORWARD kernel taught (profiles/r03_*): with two waves per SIMD one phase apart, the wave that is NOT in its
// MFMA phase gets about one issue slot per MFMA of its partner - for the 32-cycle bf16 instruction just as for the 64-cycle fp32 one -
// so the non-MFMA half of the work runs at ~40 cycles per instruction while the partner streams, and the kernel's period is the SUM of
// both.  Here therefore: ONE wave per SIMD (4 waves, 256 threads, up to 512 registers), no partner to starve, every phase of a tile in
// program order in the same wave, memory latency hidden by a register prefetch issued a whole tile ahead:
//
//   per tile of BM rows:   stage (raw registers -> dY transform, act_in, 3-way split -> LDS)   | barrier
//                          issue the global loads of the next tile (stay in flight)
//                          dW MFMAs (A = dY^T pieces, B = act(X)^T pieces from LDS, k = rows)
//                          dX MFMAs (A = dY pieces from LDS, B = the wave's W slab, RESIDENT IN REGISTERS, k = output channels)
//                          epilogue: ReLU + BatchNorm-backward sums, dX stores                      | barrier
//
//  * W never touches LDS: wave w owns the 32 input channels [32 w', 32 w'+32) of dX and keeps its slab of W^T as 3 x Co/16 bf16x8
//    fragments (96 registers at Co = 128).  Three bf16 planes of W (104 KB) plus the tiles would not fit the 160 KB.
//  * dY is stored twice (row-major pieces for dX, transposed pieces for dW), act(X) once (transposed) plus the raw fp32 X the
//    statistics need: 126 KB at 128 x 128.  A thread owns 4 rows x 4 channels; 8 consecutive lanes run along a row (a quarter-wave
//    fetches 2 rows x 128 B = whole cache lines: with lanes running over the row groups first every 128-byte line was requested by four
//    different quarter-waves and the L1's tag rate, not HBM, set the pace - 1200 of 9000 cycles per tile).  All LDS traffic is 8/16-byte;
//    the layouts come from an exhaustive search over strides and chunk rotations (tools/lds_layout_search.py): row-major dY rows in
//    natural order (stride 2 Co + 16), transposed rows with their 16-byte chunks rotated by ((c >> 2) ^ (c >> 4)) & 3 (conflict-free
//    operand reads, 2-way on the staging stores at BM = 32; the reverse at BM = 64), raw X chunks rotated by the column group.
//  * dW stays in the accumulators for the whole kernel (one accumulator set per workgroup: 16 blocks over 4 waves), flushed into one
//    of 8 per-XCD copies as before.
#include "common.h"
#include <stdlib.h>

#define P2C_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

struct BwdFused3Args {
    const float *dz; int lddz;
    const float *y; int ldy;
    const float *coef;
    const int32_t *arg; int ns;
    const float *x; int ldx;
    const float *in_scale, *in_shift;
    const float *w; int ldw;
    float *dx; int lddx;
    float *dw; int lddw;
    float *dbias;
    const float *pstat;
    double *partials;
    int M;
    long long dw_slot_stride;
    int coef_ld, arg_ld, dx_atomic;
};

__device__ __forceinline__ uint32_t b3_pk(float a, float b)
{
    const v2f v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float b3_lo(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float b3_hi(uint32_t p) { return __builtin_bit_cast(float, p & 0xFFFF0000u); }

// Two elements -> their three bf16 pieces: the PACKED dwords of the pair (hi, mid, lo: what an LDS row of pieces holds when the two
// elements are neighbours along the contiguous direction) and the pieces as fp32 VALUES (hi, mid exactly representable in bf16, lo =
// the rest), so that the same split can also be packed along the other direction without splitting twice.
struct B3Pair { uint32_t ph, pm, pl; float h0, h1, m0, m1, l0, l1; };
__device__ __forceinline__ B3Pair b3_split2(float x0, float x1)
{
    B3Pair r;
    r.ph = b3_pk(x0, x1);
    r.h0 = b3_lo(r.ph); r.h1 = b3_hi(r.ph);
    const float r0 = x0 - r.h0, r1 = x1 - r.h1;
    r.pm = b3_pk(r0, r1);
    r.m0 = b3_lo(r.pm); r.m1 = b3_hi(r.pm);
    r.l0 = r0 - r.m0; r.l1 = r1 - r.m1;
    r.pl = b3_pk(r.l0, r.l1);
    return r;
}

// chunk rotation of the transposed piece rows (channel c): see the layout search
__device__ __forceinline__ int b3_rot(int c) { return ((c >> 2) ^ (c >> 4)) & 3; }

struct B3True { static constexpr bool value = true; };
struct B3False { static constexpr bool value = false; };

template <int C, int BM>
struct Unit3 {          // thread -> (4 rows x 4 channels) units of a [BM x C] tile; 8 consecutive lanes along a row, then the row groups
    static constexpr int NRG = BM / 4, NCG = C / 4, NUNITS = NRG * NCG;
    static constexpr int NU = (NUNITS + 255) / 256;                       // units per thread (1 or 2); NUNITS == 128: threads >= 128 idle
    static __device__ __forceinline__ bool map(int tid, int u, int &c4, int &rg)
    {
        const int p = tid + 256 * u;
        c4 = (p & 7) + 8 * (p / (8 * NRG));
        rg = (p >> 3) % NRG;
        return p < NUNITS;
    }
};

// ROLES (512 threads, two waves per SIMD): the four waves of half 0 stage dY, run the dX MFMAs and the epilogue; the four waves of half 1
// stage X and run the dW MFMAs - IN THE SAME PHASE.  Each SIMD then holds one wave of either kind: their vector work (the staging) runs
// side by side and hides each other's latencies (a lone wave issues a dependent instruction chain at ~6 cycles per instruction), their
// MFMA streams share the matrix pipe, and the register-resident state splits in two (W slab + dX accumulators | dW accumulators), so
// both fit 256 registers.  Used for Ci = 128 (one staging unit per thread); the other shapes keep the one-wave-per-SIMD form.
template <int Co, int Ci, int GMODE, int IMODE, bool NEED_DX, bool HAS_STATS, bool ROLES>
__device__ __forceinline__ void bwd_fused3_body(const BwdFused3Args &a)
{
    constexpr int WC = Ci / 32, WR = 4 / WC, BM = 32 * WR;
    constexpr int COT = Co / 64, CIT = Ci / 64;
    constexpr int LDR = 2 * Co + 16, PLR = BM * LDR;                      // row-major dY pieces   [3][BM][LDR]
    constexpr int LDT = 2 * BM + (BM == 32 ? 32 : 16), PLY = Co * LDT, PLX = Ci * LDT;      // transposed pieces [3][Co][LDT], [3][Ci][LDT]
    constexpr int NCH = BM / 8;                                           // 16-byte chunks (8 rows) of a transposed row
    constexpr int LDXR = 4 * (BM == 32 ? 48 : 68);                        // raw X, transposed     [Ci][..] fp32, 16-byte chunks of 4 rows
    constexpr int NRGX = BM / 4;
    constexpr int NQX = Co / 16, NKW = BM / 16;
    using UY = Unit3<Co, BM>;
    using UX = Unit3<Ci, BM>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3b[];
    unsigned char *DYR = smem3b;
    unsigned char *DYT = DYR + (NEED_DX ? 3 * PLR : 0);
    unsigned char *XT = DYT + 3 * PLY;
    unsigned char *XR = XT + 3 * PLX;
    float *red = reinterpret_cast<float *>(XR + ((NEED_DX && HAS_STATS) ? Ci * LDXR : 0));      // [2][Ci] stats, [Co] dbias

    const int role = ROLES ? (int)(threadIdx.x >> 8) : 0;                 // ROLES: 0 = dY / dX / epilogue waves, 1 = X / dW waves
    const bool doA = !ROLES || role == 0, doB = !ROLES || role == 1;
    const int tid = ROLES ? (int)(threadIdx.x & 255) : (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    [[maybe_unused]] const int half = role;                               // (P2C_TR stamps: one row per role)
    P2C_TR_WG(0);
    const int l31 = lane & 31, lh = lane >> 5;
    const int wi = wave >> 1, wj = wave & 1;                              // dW wave grid
    const int wr = wave / WC, wc = wave % WC;                             // dX wave grid
    const int ntiles = (a.M + BM - 1) / BM;
    const int nk = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    auto tile_of = [&](int k) { return (int)blockIdx.x + k * (int)gridDim.x; };

    // ---------------- raw prefetch registers: (dZ, Y) of the next tile are requested during the dX phase, X during the dW phase
    v4f rdz[UY::NU][GMODE == 2 ? 1 : 4], ry[UY::NU][4], rx[UX::NU][4];
    v4i rarg[UY::NU];
    auto group_of = [&](int m0, int rg, int &r) {                        // GMODE 2: group of the unit's 4 rows (they lie in one group: ns % 16 == 0)
        const int g0 = m0 / a.ns;                                        // uniform
        r = m0 - g0 * a.ns + 4 * rg;                                     // < ns + BM <= 3 ns
        return g0 + (r >= a.ns ? 1 : 0) + (r >= 2 * a.ns ? 1 : 0);
    };
    // Addresses: a whole tile's rows are  UNIFORM base (tile, row j: scalar arithmetic)  +  one 32-bit per-lane offset per tensor, so a load
    // costs no vector arithmetic at all.  (One wave per SIMD has no second wave to hide the latency of dependent 64-bit address chains:
    // with a multiply-add and a clamp per row the twelve loads of a tile took 1500 cycles to issue.)  Only the ragged last tile clamps.
    int voy[UY::NU], vox[UX::NU];                                         // byte offsets of the unit's first row within a tile (ld in floats)
#pragma unroll
    for (int u = 0; u < UY::NU; ++u) { int c4, rg; UY::map(tid, u, c4, rg); voy[u] = 4 * rg; (void)c4; }
#pragma unroll
    for (int u = 0; u < UX::NU; ++u) { int c4, rg; UX::map(tid, u, c4, rg); vox[u] = 4 * rg; (void)c4; }
    auto gload_y = [&](int t, int which) {                              // which: 1 = dZ (+ winners), 2 = Y, 3 = both
        const int m0 = t * BM;
        const bool whole = m0 + BM <= a.M;                                // uniform
#pragma unroll
        for (int u = 0; u < UY::NU; ++u) {
            int c4, rg;
            if (!UY::map(tid, u, c4, rg)) continue;
            if (GMODE == 2 && (which & 1)) {
                int r;
                const int grp = min(group_of(m0, rg, r), (a.M - 1) / a.ns);
                rdz[u][0] = *reinterpret_cast<const v4f *>(a.dz + (size_t)grp * a.lddz + 4 * c4);
                rarg[u] = *reinterpret_cast<const v4i *>(a.arg + (size_t)grp * (a.arg_ld ? a.arg_ld : Co) + 4 * c4);
            }
            if (whole) {
                const uint32_t oz = (uint32_t)(voy[u] * a.lddz + 4 * c4) * 4u, oy = (uint32_t)(voy[u] * a.ldy + 4 * c4) * 4u;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (GMODE != 2 && (which & 1)) rdz[u][j] = *reinterpret_cast<const v4f *>(reinterpret_cast<const char *>(a.dz + (size_t)(m0 + j) * a.lddz) + oz);
                    if (GMODE >= 1 && (which & 2)) ry[u][j] = *reinterpret_cast<const v4f *>(reinterpret_cast<const char *>(a.y + (size_t)(m0 + j) * a.ldy) + oy);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = min(m0 + 4 * rg + j, a.M - 1);
                    if (GMODE != 2 && (which & 1)) rdz[u][j] = *reinterpret_cast<const v4f *>(a.dz + (size_t)m * a.lddz + 4 * c4);
                    if (GMODE >= 1 && (which & 2)) ry[u][j] = *reinterpret_cast<const v4f *>(a.y + (size_t)m * a.ldy + 4 * c4);
                }
            }
        }
    };
    auto gload_x = [&](int t) {
        const int m0 = t * BM;
        const bool whole = m0 + BM <= a.M;
#pragma unroll
        for (int u = 0; u < UX::NU; ++u) {
            int c4, rg;
            UX::map(tid, u, c4, rg);                                      // BM * Ci == 4096: every thread has its unit
            if (whole) {
                const uint32_t ox = (uint32_t)(vox[u] * a.ldx + 4 * c4) * 4u;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    rx[u][j] = *reinterpret_cast<const v4f *>(reinterpret_cast<const char *>(a.x + (size_t)(m0 + j) * a.ldx) + ox);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = min(m0 + 4 * rg + j, a.M - 1);
                    rx[u][j] = *reinterpret_cast<const v4f *>(a.x + (size_t)m * a.ldx + 4 * c4);
                }
            }
        }
    };
    if (doA) gload_y(tile_of(0), 3);                                      // in flight under the W set-up below
    if (doB) gload_x(tile_of(0));

    // ---------------- this wave's slab of W^T as register-resident bf16 fragments: lane (i, h), k-step q holds W[16q + 8h + e][32 wc + i]
    bf16x8 wh[NEED_DX ? NQX : 1], wm[NEED_DX ? NQX : 1], wl[NEED_DX ? NQX : 1];
    if (NEED_DX && doA) {
        const float *wp = a.w + (size_t)(8 * lh) * a.ldw + wc * 32 + l31;
#pragma unroll
        for (int q = 0; q < NQX; ++q) {
            v4u ph, pm, pl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const B3Pair sp = b3_split2(wp[(size_t)(16 * q + 2 * e) * a.ldw], wp[(size_t)(16 * q + 2 * e + 1) * a.ldw]);
                ph[e] = sp.ph; pm[e] = sp.pm; pl[e] = sp.pl;
            }
            wh[q] = __builtin_bit_cast(bf16x8, ph);
            wm[q] = __builtin_bit_cast(bf16x8, pm);
            wl[q] = __builtin_bit_cast(bf16x8, pl);
        }
    }

    // ---------------- per-thread constants of the staging units
    v4f cf[UY::NU][5], isc[UX::NU], ish[UX::NU];
#pragma unroll
    for (int u = 0; u < UY::NU; ++u) {
        int c4, rg;
        UY::map(tid, u, c4, rg);
        c4 = min(c4, Co / 4 - 1);
#pragma unroll
        for (int i = 0; i < 5; ++i)
            cf[u][i] = GMODE >= 1 ? *reinterpret_cast<const v4f *>(a.coef + i * (a.coef_ld ? a.coef_ld : Co) + 4 * c4) : v4f{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < UX::NU; ++u) {
        int c4, rg;
        UX::map(tid, u, c4, rg);
        isc[u] = IMODE >= 1 ? *reinterpret_cast<const v4f *>(a.in_scale + 4 * c4) : v4f{1.f, 1.f, 1.f, 1.f};
        ish[u] = IMODE >= 1 ? *reinterpret_cast<const v4f *>(a.in_shift + 4 * c4) : v4f{0.f, 0.f, 0.f, 0.f};
    }
    const int xcol = wc * 32 + l31;
    float psc = 0.f, psh = 0.f, pmu = 0.f, pis = 0.f;
    if (NEED_DX && HAS_STATS) { psc = a.pstat[xcol]; psh = a.pstat[Ci + xcol]; pmu = a.pstat[2 * Ci + xcol]; pis = a.pstat[3 * Ci + xcol]; }
    const float npm = -pmu * pis;

    // ---------------- per-lane LDS offsets of the operand reads
    const int adx = (32 * wr + l31) * LDR + 16 * lh;                     // dX A: dY row, k = co
    int adw[COT][NKW], bdw[CIT][NKW];                                    // dW A / B: transposed row (co / ci), k-step s = chunk 2 s + h, rotated
#pragma unroll
    for (int ks = 0; ks < NKW; ++ks) {
#pragma unroll
        for (int i = 0; i < COT; ++i) {
            const int c = wi * (COT * 32) + i * 32 + l31;
            adw[i][ks] = c * LDT + 16 * ((2 * ks + lh + b3_rot(c)) % NCH);
        }
#pragma unroll
        for (int j = 0; j < CIT; ++j) {
            const int c = wj * (CIT * 32) + j * 32 + l31;
            bdw[j][ks] = c * LDT + 16 * ((2 * ks + lh + b3_rot(c)) % NCH);
        }
    }
    int ayp[4];                                                          // raw x of the lane's column: rows 8g + 4h .. +3 = chunk 8 wr + 2 g + h
#pragma unroll
    for (int g = 0; g < 4; ++g) ayp[g] = xcol * LDXR + 16 * ((8 * wr + 2 * g + lh + (BM == 32 ? (xcol >> 2) : (xcol >> 3))) % NRGX);

    double s1 = 0.0, s2 = 0.0;            // per-tile fp32 partial sums (16 rows) are added in fp64: 2 adds per tile
    v4f dbacc[UY::NU];
#pragma unroll
    for (int u = 0; u < UY::NU; ++u) dbacc[u] = v4f{0.f, 0.f, 0.f, 0.f};
    f32x16 accW[COT][CIT];
#pragma unroll
    for (int i = 0; i < COT; ++i)
#pragma unroll
        for (int j = 0; j < CIT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) accW[i][j][r] = 0.f;

    // ---------------- packed bf16 pieces of one tile's dY
    struct PY { v2u rh[4], rm[4], rl[4], th[4], tm[4], tl[4]; };          // row-major: [row j] 4 channels; transposed: [channel e] 4 rows
    PY py[UY::NU];
    auto write_py = [&](int u) {
        {
            int c4, rg;
            if (!UY::map(tid, u, c4, rg)) return;
            if (NEED_DX) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned char *d = DYR + (4 * rg + j) * LDR + 8 * c4;
                    *reinterpret_cast<v2u *>(d) = py[u].rh[j];
                    *reinterpret_cast<v2u *>(d + PLR) = py[u].rm[j];
                    *reinterpret_cast<v2u *>(d + 2 * PLR) = py[u].rl[j];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned char *d = DYT + (4 * c4 + e) * LDT + 16 * (((rg >> 1) + b3_rot(4 * c4 + e)) % NCH) + 8 * (rg & 1);
                *reinterpret_cast<v2u *>(d) = py[u].th[e];
                *reinterpret_cast<v2u *>(d + PLY) = py[u].tm[e];
                *reinterpret_cast<v2u *>(d + 2 * PLY) = py[u].tl[e];
            }
        }
    };
    // dY of the unit from the raw registers: gs * (dZ masked by the ReLU) + q * Y + p, three-way split, packed along both directions
    auto make_py = [&](int t, bool masked) {
        const int m0 = t * BM;
#pragma unroll
        for (int u = 0; u < UY::NU; ++u) {
            int c4, rg;
            if (!UY::map(tid, u, c4, rg)) continue;
            float H[4][4], Mi[4][4], L[4][4];                             // [row j][channel e] pieces as fp32 values
            int r0 = 0;
            if (GMODE == 2) (void)group_of(m0, rg, r0);
            const int rin = GMODE == 2 ? (r0 >= 2 * a.ns ? r0 - 2 * a.ns : (r0 >= a.ns ? r0 - a.ns : r0)) : 0;   // first row's index in its group
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v4f g;
                if (GMODE == 2) {
                    const int jj = rin + j;
                    const v4f d = rdz[u][0];
                    const v4i w = rarg[u];
                    g = v4f{w.x == jj ? d.x : 0.f, w.y == jj ? d.y : 0.f, w.z == jj ? d.z : 0.f, w.w == jj ? d.w : 0.f};
                } else {
                    g = rdz[u][j];
                }
                v4f o = g;
                if (GMODE >= 1) {
                    const v4f yy = ry[u][j];
                    // the ReLU mask with the forward's own two roundings (mul, add); the affine rest as fmas
                    o.x = __builtin_fmaf(cf[u][2].x, (cf[u][0].x * yy.x + cf[u][1].x > 0.f) ? g.x : 0.f, __builtin_fmaf(cf[u][3].x, yy.x, cf[u][4].x));
                    o.y = __builtin_fmaf(cf[u][2].y, (cf[u][0].y * yy.y + cf[u][1].y > 0.f) ? g.y : 0.f, __builtin_fmaf(cf[u][3].y, yy.y, cf[u][4].y));
                    o.z = __builtin_fmaf(cf[u][2].z, (cf[u][0].z * yy.z + cf[u][1].z > 0.f) ? g.z : 0.f, __builtin_fmaf(cf[u][3].z, yy.z, cf[u][4].z));
                    o.w = __builtin_fmaf(cf[u][2].w, (cf[u][0].w * yy.w + cf[u][1].w > 0.f) ? g.w : 0.f, __builtin_fmaf(cf[u][3].w, yy.w, cf[u][4].w));
                }
                if (masked) o *= (m0 + 4 * rg + j < a.M) ? 1.f : 0.f;    // rows past M contribute nothing to dW / dbias / dX
                if (GMODE == 0) dbacc[u] += o;
                // split along the channel pairs: the packed dwords ARE the row-major pieces of this row
                const B3Pair p0 = b3_split2(o.x, o.y), p1 = b3_split2(o.z, o.w);
                H[j][0] = p0.h0; H[j][1] = p0.h1; H[j][2] = p1.h0; H[j][3] = p1.h1;
                Mi[j][0] = p0.m0; Mi[j][1] = p0.m1; Mi[j][2] = p1.m0; Mi[j][3] = p1.m1;
                L[j][0] = p0.l0; L[j][1] = p0.l1; L[j][2] = p1.l0; L[j][3] = p1.l1;
                py[u].rh[j] = v2u{p0.ph, p1.ph};
                py[u].rm[j] = v2u{p0.pm, p1.pm};
                py[u].rl[j] = v2u{p0.pl, p1.pl};
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {                                 // transposed pieces: 4 rows of one channel
                py[u].th[e] = v2u{b3_pk(H[0][e], H[1][e]), b3_pk(H[2][e], H[3][e])};
                py[u].tm[e] = v2u{b3_pk(Mi[0][e], Mi[1][e]), b3_pk(Mi[2][e], Mi[3][e])};
                py[u].tl[e] = v2u{b3_pk(L[0][e], L[1][e]), b3_pk(L[2][e], L[3][e])};
            }
            write_py(u);                                                  // unit by unit: only one unit's packed pieces are alive at a time
        }
    };
    // act_in(X) of the unit from the raw registers, three-way split along the row pairs, transposed pieces + raw values -> LDS
    auto stage_x = [&]() {
#pragma unroll
        for (int u = 0; u < UX::NU; ++u) {
            int c4, rg;
            UX::map(tid, u, c4, rg);
            v4f av[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v4f v = rx[u][j];
                if (IMODE >= 1) {
                    v.x = fmaxf(isc[u].x * v.x + ish[u].x, 0.f);
                    v.y = fmaxf(isc[u].y * v.y + ish[u].y, 0.f);
                    v.z = fmaxf(isc[u].z * v.z + ish[u].z, 0.f);
                    v.w = fmaxf(isc[u].w * v.w + ish[u].w, 0.f);
                }
                av[j] = v;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const B3Pair p0 = b3_split2(av[0][e], av[1][e]), p1 = b3_split2(av[2][e], av[3][e]);
                unsigned char *d = XT + (4 * c4 + e) * LDT + 16 * (((rg >> 1) + b3_rot(4 * c4 + e)) % NCH) + 8 * (rg & 1);
                *reinterpret_cast<v2u *>(d) = v2u{p0.ph, p1.ph};
                *reinterpret_cast<v2u *>(d + PLX) = v2u{p0.pm, p1.pm};
                *reinterpret_cast<v2u *>(d + 2 * PLX) = v2u{p0.pl, p1.pl};
                if (NEED_DX && HAS_STATS)                                 // the raw values: 4 rows of one channel = 16 bytes
                    *reinterpret_cast<v4f *>(XR + (4 * c4 + e) * LDXR + 16 * ((rg + (BM == 32 ? c4 : (c4 >> 1))) % NRGX)) =
                        v4f{rx[u][0][e], rx[u][1][e], rx[u][2][e], rx[u][3][e]};
            }
        }
    };

    // ---------------- main loop.  Per tile t:  stage (dY, X) -> LDS | barrier | request tile t+1 | dW MFMAs | dX MFMAs | sums + stores | barrier.
    // Tried and measured slower (tools/fused_trace.py, 128 x 128, cycles per 32-row tile): a three-phase software pipeline - dX MFMAs
    // interleaved with the X staging, dW MFMAs interleaved with the sums and with the NEXT tile's dY transform + split into registers,
    // the interleave spelled out with __builtin_amdgcn_sched_group_barrier - 13.5 k against 9.0 k for the plain order below: an
    // instruction placed between two MFMAs that share an accumulator costs tens of cycles (MI355X_MICROARCH.md, "one extra issue slot
    // between two MFMAs"), and the live packed pieces pushed the 128-wide shapes past 512 registers (48-156 B of scratch per lane).
    P2C_TR_WG_MID(0);
    if constexpr (ROLES) {
        static_assert(!ROLES || (NEED_DX && UY::NU == 1 && UX::NU == 1), "");
        if (role == 0) {
            // ======================= role A: dY staging | dX MFMAs | sums + dX stores =======================
            auto touch_y = [&]() {                                       // the youngest load of gload_y (see the vmcnt note in the plain loop)
                if (GMODE >= 1) asm volatile("" ::"v"(ry[0][3])); else asm volatile("" ::"v"(rdz[0][3]));
            };
            touch_y();
            for (int k = 0; k < nk; ++k) {
                [[maybe_unused]] const int it = k;
                const int t = tile_of(k), m0 = t * BM;
                const int tn = tile_of(k + 1 < nk ? k + 1 : k);
                const bool full = m0 + BM <= a.M;
                P2C_TR(0);
                make_py(t, !full);
                P2C_TR(1);
                P2C_LDS_BARRIER();
                P2C_TR(2);
                gload_y(tn, 3);
                __builtin_amdgcn_sched_barrier(0);
                f32x16 accX0, accX1;
#pragma unroll
                for (int r = 0; r < 16; ++r) accX0[r] = accX1[r] = 0.f;
#pragma unroll
                for (int q = 0; q < NQX; q += 2) {
                    bf16x8 f0[3], f1[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        f0[p] = *reinterpret_cast<const bf16x8 *>(DYR + adx + p * PLR + 32 * q);
                        f1[p] = *reinterpret_cast<const bf16x8 *>(DYR + adx + p * PLR + 32 * (q + 1));
                    }
#define P2C_XR(ACC_, F_, Q_, PA_, WB_) ACC_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F_[PA_], WB_[Q_], ACC_, 0, 0, 0)
                    P2C_XR(accX0, f0, q, 1, wm); P2C_XR(accX1, f1, q + 1, 1, wm);
                    P2C_XR(accX0, f0, q, 0, wl); P2C_XR(accX1, f1, q + 1, 0, wl);
                    P2C_XR(accX0, f0, q, 2, wh); P2C_XR(accX1, f1, q + 1, 2, wh);
                    P2C_XR(accX0, f0, q, 0, wm); P2C_XR(accX1, f1, q + 1, 0, wm);
                    P2C_XR(accX0, f0, q, 1, wh); P2C_XR(accX1, f1, q + 1, 1, wh);
                    P2C_XR(accX0, f0, q, 0, wh); P2C_XR(accX1, f1, q + 1, 0, wh);
#undef P2C_XR
                }
                P2C_TR(3);
                float yp[16];
                if (HAS_STATS) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const v4f v = *reinterpret_cast<const v4f *>(XR + ayp[g]);
                        yp[4 * g] = v.x; yp[4 * g + 1] = v.y; yp[4 * g + 2] = v.z; yp[4 * g + 3] = v.w;
                    }
                }
                touch_y();                                               // wait for the prefetch while only loads are outstanding
                const uint32_t dxo = (uint32_t)((wr * 32 + 4 * lh) * a.lddx + xcol) * 4u;
                float *dxp = a.dx + (size_t)(m0 + wr * 32 + 4 * lh) * a.lddx + xcol;
                float vx[16];
                float t1[4] = {0.f, 0.f, 0.f, 0.f}, t2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = accX0[r] + accX1[r];
                    vx[r] = v;
                    if (HAS_STATS) {
                        const float g = (psc * yp[r] + psh > 0.f) ? v : 0.f;
                        t1[r & 3] += g;
                        t2[r & 3] = __builtin_fmaf(g, __builtin_fmaf(yp[r], pis, npm), t2[r & 3]);
                    }
                }
                if (HAS_STATS) { s1 += (double)((t1[0] + t1[1]) + (t1[2] + t1[3])); s2 += (double)((t2[0] + t2[1]) + (t2[2] + t2[3])); }
                if (full && !a.dx_atomic) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        *reinterpret_cast<float *>(reinterpret_cast<char *>(a.dx + (size_t)(m0 + (r & 3) + 8 * (r >> 2)) * a.lddx) + dxo) = vx[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ro = (r & 3) + 8 * (r >> 2);
                        if (m0 + wr * 32 + 4 * lh + ro < a.M) {
                            if (a.dx_atomic) atomicAdd(dxp + (size_t)ro * a.lddx, vx[r]);
                            else dxp[(size_t)ro * a.lddx] = vx[r];
                        }
                    }
                }
                P2C_TR(7);
                P2C_LDS_BARRIER();
            }
        } else {
            // ======================= role B: X staging | dW MFMAs =======================
            for (int k = 0; k < nk; ++k) {
                [[maybe_unused]] const int it = k;
                const int tn = tile_of(k + 1 < nk ? k + 1 : k);
                P2C_TR(0);
                stage_x();
                P2C_TR(1);
                P2C_LDS_BARRIER();
                P2C_TR(2);
                gload_x(tn);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int sk = 0; sk < NKW; ++sk) {
                    bf16x8 A[COT][3], B[CIT][3];
#pragma unroll
                    for (int i = 0; i < COT; ++i)
#pragma unroll
                        for (int p = 0; p < 3; ++p) A[i][p] = *reinterpret_cast<const bf16x8 *>(DYT + adw[i][sk] + p * PLY);
#pragma unroll
                    for (int j = 0; j < CIT; ++j)
#pragma unroll
                        for (int p = 0; p < 3; ++p) B[j][p] = *reinterpret_cast<const bf16x8 *>(XT + bdw[j][sk] + p * PLX);
#define P2C_WR(PA_, PB_)                                                                                             \
    _Pragma("unroll") for (int i = 0; i < COT; ++i) _Pragma("unroll") for (int j = 0; j < CIT; ++j)                  \
        accW[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][PA_], B[j][PB_], accW[i][j], 0, 0, 0)
                    P2C_WR(1, 1); P2C_WR(0, 2); P2C_WR(2, 0); P2C_WR(0, 1); P2C_WR(1, 0); P2C_WR(0, 0);
#undef P2C_WR
                }
                P2C_TR(3);
                P2C_TR(7);
                P2C_LDS_BARRIER();
            }
        }
    } else {
    asm volatile("" ::"v"(rx[UX::NU - 1][3]));            // both paths into the loop header see the prefetch registers settled (see below)
    for (int k = 0; k < nk; ++k) {
        [[maybe_unused]] const int it = k;
        const int t = tile_of(k), m0 = t * BM;
        const int tn = tile_of(k + 1 < nk ? k + 1 : k);
        const bool full = m0 + BM <= a.M;
        P2C_TR(0);
        // ================= stage tile t: raw registers -> LDS =================
        make_py(t, !full);
        P2C_TR(1);
        stage_x();
        P2C_TR(2);
        P2C_LDS_BARRIER();
        P2C_TR(3);
        // ================= next tile's rows: requested in three instalments between the MFMA groups, consumed one iteration later ==========
        // A CU tracks only so many outstanding misses: issued as one burst, the 48 KB of a tile stall the wave's (in-order) instruction
        // stream for ~1500 cycles before the first MFMA - and the chip-wide rate stayed at 3.3 TB/s.  Spread out, the later requests
        // queue up while the matrix pipe already works.
        gload_y(tn, 1);
        P2C_TR(4);
        // ================= dW += dY^T . act(X): k = the BM rows =================
        // the fragments of k-step s+1 are requested before the MFMAs of k-step s (pinned: the scheduler would sink them to their use)
        {
            bf16x8 A[2][COT][3], B[2][CIT][3];
#define P2C_LDW(SET_, S_)                                                                                            \
    do {                                                                                                             \
        _Pragma("unroll") for (int i = 0; i < COT; ++i) _Pragma("unroll") for (int p = 0; p < 3; ++p)                \
            A[SET_][i][p] = *reinterpret_cast<const bf16x8 *>(DYT + adw[i][S_] + p * PLY);                           \
        _Pragma("unroll") for (int j = 0; j < CIT; ++j) _Pragma("unroll") for (int p = 0; p < 3; ++p)                \
            B[SET_][j][p] = *reinterpret_cast<const bf16x8 *>(XT + bdw[j][S_] + p * PLX);                            \
    } while (0)
            // six products, smallest first; consecutive MFMAs go to different accumulators
#define P2C_W(SET_, PA_, PB_)                                                                                        \
    _Pragma("unroll") for (int i = 0; i < COT; ++i) _Pragma("unroll") for (int j = 0; j < CIT; ++j)                  \
        accW[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[SET_][i][PA_], B[SET_][j][PB_], accW[i][j], 0, 0, 0)
            P2C_LDW(0, 0);
#pragma unroll
            for (int s = 0; s < NKW; s += 2) {
                if (s + 1 < NKW) P2C_LDW(1, s + 1);
                __builtin_amdgcn_sched_barrier(0);
                P2C_W(0, 1, 1); P2C_W(0, 0, 2); P2C_W(0, 2, 0); P2C_W(0, 0, 1); P2C_W(0, 1, 0); P2C_W(0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (s == 0) { gload_y(tn, 2); __builtin_amdgcn_sched_barrier(0); }
                if (s + 1 < NKW) {
                    if (s + 2 < NKW) P2C_LDW(0, s + 2);
                    __builtin_amdgcn_sched_barrier(0);
                    P2C_W(1, 1, 1); P2C_W(1, 0, 2); P2C_W(1, 2, 0); P2C_W(1, 0, 1); P2C_W(1, 1, 0); P2C_W(1, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#undef P2C_W
#undef P2C_LDW
        }
        gload_x(tn);
        __builtin_amdgcn_sched_barrier(0);
        P2C_TR(5);
        if (NEED_DX) {
            // ================= dX = dY . W: k = the Co output channels, B operand from registers =================
            f32x16 accX0, accX1;
#pragma unroll
            for (int r = 0; r < 16; ++r) accX0[r] = accX1[r] = 0.f;
            {
                bf16x8 af[2][2][3];                                      // [set][q parity][piece]: two k-steps per set, two sets
#define P2C_LDX(SET_, Q_)                                                                                            \
    _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                                                  \
        af[SET_][0][p] = *reinterpret_cast<const bf16x8 *>(DYR + adx + p * PLR + 32 * (Q_));                         \
        af[SET_][1][p] = *reinterpret_cast<const bf16x8 *>(DYR + adx + p * PLR + 32 * ((Q_) + 1));                   \
    }
#define P2C_X(ACC_, SET_, PAR_, Q_, PA_, WB_) ACC_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[SET_][PAR_][PA_], WB_[Q_], ACC_, 0, 0, 0)
#define P2C_X12(SET_, Q_)                                                                \
    do {                                                                                 \
        P2C_X(accX0, SET_, 0, Q_, 1, wm); P2C_X(accX1, SET_, 1, (Q_) + 1, 1, wm);        \
        P2C_X(accX0, SET_, 0, Q_, 0, wl); P2C_X(accX1, SET_, 1, (Q_) + 1, 0, wl);        \
        P2C_X(accX0, SET_, 0, Q_, 2, wh); P2C_X(accX1, SET_, 1, (Q_) + 1, 2, wh);        \
        P2C_X(accX0, SET_, 0, Q_, 0, wm); P2C_X(accX1, SET_, 1, (Q_) + 1, 0, wm);        \
        P2C_X(accX0, SET_, 0, Q_, 1, wh); P2C_X(accX1, SET_, 1, (Q_) + 1, 1, wh);        \
        P2C_X(accX0, SET_, 0, Q_, 0, wh); P2C_X(accX1, SET_, 1, (Q_) + 1, 0, wh);        \
    } while (0)
                P2C_LDX(0, 0)
#pragma unroll
                for (int q = 0; q < NQX; q += 4) {
                    if (q + 2 < NQX) { P2C_LDX(1, q + 2) }
                    __builtin_amdgcn_sched_barrier(0);
                    P2C_X12(0, q);
                    __builtin_amdgcn_sched_barrier(0);
                    if (q + 2 < NQX) {
                        if (q + 4 < NQX) { P2C_LDX(0, q + 4) }
                        __builtin_amdgcn_sched_barrier(0);
                        P2C_X12(1, q + 2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#undef P2C_X12
#undef P2C_X
#undef P2C_LDX
            }
            P2C_TR(6);
            // ================= epilogue: sums of the layer below, dX stores =================
            float yp[16];
            if (HAS_STATS) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const v4f v = *reinterpret_cast<const v4f *>(XR + ayp[g]);
                    yp[4 * g] = v.x; yp[4 * g + 1] = v.y; yp[4 * g + 2] = v.z; yp[4 * g + 3] = v.w;
                }
            }
            // vmcnt retires in issue order and the compiler cannot count stores across the loop's back edge: without this, the wait for the
            // prefetched rows at the top of the next iteration drains the 16 stores issued below as well - a write round trip per tile.
            // Touching the youngest prefetch register HERE waits while only loads (requested a whole MFMA phase ago) are outstanding.
            asm volatile("" ::"v"(rx[UX::NU - 1][3]));
            float *dxp = a.dx + (size_t)(m0 + wr * 32 + 4 * lh) * a.lddx + xcol;
            const uint32_t dxo = (uint32_t)((wr * 32 + 4 * lh) * a.lddx + xcol) * 4u;     // per-lane byte offset inside the tile (tile rows: scalar)
            float vx[16];
            // four independent chains per sum: one wave per SIMD has no other wave to hide an add's latency
            float t1[4] = {0.f, 0.f, 0.f, 0.f}, t2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = accX0[r] + accX1[r];
                vx[r] = v;
                if (HAS_STATS) {
                    const float g = (psc * yp[r] + psh > 0.f) ? v : 0.f;          // mask: the forward's two roundings
                    t1[r & 3] += g;
                    t2[r & 3] = __builtin_fmaf(g, __builtin_fmaf(yp[r], pis, npm), t2[r & 3]);    // g * (y - mean) * invstd
                }
            }
            if (HAS_STATS) { s1 += (double)((t1[0] + t1[1]) + (t1[2] + t1[3])); s2 += (double)((t2[0] + t2[1]) + (t2[2] + t2[3])); }
            if (full && !a.dx_atomic) {                                   // uniform: straight-line stores, one base address
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    *reinterpret_cast<float *>(reinterpret_cast<char *>(a.dx + (size_t)(m0 + (r & 3) + 8 * (r >> 2)) * a.lddx) + dxo) = vx[r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = (r & 3) + 8 * (r >> 2);
                    if (m0 + wr * 32 + 4 * lh + ro < a.M) {
                        if (a.dx_atomic) atomicAdd(dxp + (size_t)ro * a.lddx, vx[r]);
                        else dxp[(size_t)ro * a.lddx] = vx[r];
                    }
                }
            }
        } else {
            asm volatile("" ::"v"(rx[UX::NU - 1][3]));
        }
        P2C_TR(7);
        P2C_LDS_BARRIER();                                               // every read of this tile's LDS image is done
    }
    }
    P2C_TR_WG_MID(1);

    // ---------------- flush: dW into the slot of this workgroup's XCD, the statistics / dbias through LDS into fp64 slot rows
    if (doB) {
        float *dws = a.dw + (size_t)(blockIdx.x & 7) * a.dw_slot_stride;
#pragma unroll
        for (int i = 0; i < COT; ++i)
#pragma unroll
            for (int j = 0; j < CIT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = wi * (COT * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const int ci = wj * (CIT * 32) + j * 32 + l31;
                    atomicAdd(&dws[(size_t)co * a.lddw + ci], accW[i][j][r]);
                }
    }
    const bool want_db = GMODE == 0 && a.dbias != nullptr;
    if ((NEED_DX && HAS_STATS) || want_db) {
        for (int u = tid; u < 2 * Ci + Co; u += 256) red[u] = 0.f;
        __syncthreads();
        if (NEED_DX && HAS_STATS && doA) {
            const double u1 = s1 + __shfl_xor(s1, 32), u2 = s2 + __shfl_xor(s2, 32);
            if (lh == 0) {                    // fp64 straight into this workgroup's slot row (one or two waves per column)
                double *o = a.partials + (size_t)(blockIdx.x % P2C_STAT_SLOTS) * 2 * Ci;
                atomicAdd(&o[xcol], u1);
                atomicAdd(&o[Ci + xcol], u2);
            }
        }
        if (want_db && doA) {
#pragma unroll
            for (int u = 0; u < UY::NU; ++u) {
                int c4, rg;
                if (!UY::map(tid, u, c4, rg)) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) atomicAdd(&red[2 * Ci + 4 * c4 + e], dbacc[u][e]);
            }
        }
        __syncthreads();
        if (want_db && doA && tid < Co) atomicAdd(&a.dbias[tid], red[2 * Ci + tid]);
    }
    P2C_TR_WG(1);
}

template <int Co, int Ci, int GMODE, int IMODE, bool NEED_DX, bool HAS_STATS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) bwd_fused3_kernel(BwdFused3Args a)
{
    bwd_fused3_body<Co, Ci, GMODE, IMODE, NEED_DX, HAS_STATS, false>(a);
}

template <int Co, int Ci, int GMODE, int IMODE, bool HAS_STATS>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) bwd_fused3r_kernel(BwdFused3Args a)
{
    bwd_fused3_body<Co, Ci, GMODE, IMODE, true, HAS_STATS, true>(a);
}

template <int Co, int Ci, int GMODE, int IMODE>
static int launch_fused3(const BwdFused3Args &a, hipStream_t s)
{
    constexpr int WC = Ci / 32, WR = 4 / WC, BM = 32 * WR;
    constexpr int LDR = 2 * Co + 16, LDT = 2 * BM + (BM == 32 ? 32 : 16), LDXR = 4 * (BM == 32 ? 48 : 68);
    static_assert(3 * BM * LDR + 3 * Co * LDT + 3 * Ci * LDT + Ci * LDXR + (2 * Ci + Co) * 4 <= 160 * 1024, "LDS");
    const int ntiles = (a.M + BM - 1) / BM;
    const int grid = ntiles < 256 ? ntiles : 256;
#define P2C_FL3(DX_, ST_)                                                                                                             \
    do {                                                                                                                              \
        const size_t lds = (size_t)(DX_ ? 3 * BM * LDR : 0) + 3 * Co * LDT + 3 * Ci * LDT + ((DX_ && ST_) ? Ci * LDXR : 0) + (2 * Ci + Co) * 4; \
        (void)hipFuncSetAttribute((const void *)bwd_fused3_kernel<Co, Ci, GMODE, IMODE, DX_, ST_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)lds);                                                                                          \
        hipLaunchKernelGGL((bwd_fused3_kernel<Co, Ci, GMODE, IMODE, DX_, ST_>), dim3(grid), dim3(256), lds, s, a);                    \
    } while (0)
    if constexpr (Ci == 128) {
        if (a.dx) {        // role split (two waves per SIMD in the same phase): 163 -> 142 us against one wave per SIMD, DESIGN.md section 3
            const size_t lds = (size_t)3 * BM * LDR + 3 * Co * LDT + 3 * Ci * LDT + Ci * LDXR + (2 * Ci + Co) * 4;
            if (a.pstat) {
                (void)hipFuncSetAttribute((const void *)bwd_fused3r_kernel<Co, Ci, GMODE, IMODE, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL((bwd_fused3r_kernel<Co, Ci, GMODE, IMODE, true>), dim3(grid), dim3(512), lds, s, a);
            } else {
                (void)hipFuncSetAttribute((const void *)bwd_fused3r_kernel<Co, Ci, GMODE, IMODE, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL((bwd_fused3r_kernel<Co, Ci, GMODE, IMODE, false>), dim3(grid), dim3(512), lds, s, a);
            }
            P2C_LAUNCH_CHECK();
            return P2C_OK;
        }
    }
    if (a.dx && a.pstat) P2C_FL3(true, true);
    else if (a.dx) P2C_FL3(true, false);
    else P2C_FL3(false, false);
#undef P2C_FL3
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

template <int GMODE, int IMODE>
static int dispatch3(int Co, int Ci, const BwdFused3Args &a, hipStream_t s)
{
    if (Co == 128 && Ci == 128) return launch_fused3<128, 128, GMODE, IMODE>(a, s);
    if (Co == 128 && Ci == 64) return launch_fused3<128, 64, GMODE, IMODE>(a, s);
    if (Co == 64 && Ci == 128) return launch_fused3<64, 128, GMODE, IMODE>(a, s);
    return launch_fused3<64, 64, GMODE, IMODE>(a, s);
}

// Called by p2c_linear_bwd_fused_f32 (bwd_fused.hip) for the shapes without trailing extra columns when the split path is on;
// arguments already validated there.  Co, Ci in {64, 128}.
int p2c_bwd_fused3_launch(const float *dZ, int lddz, const float *Yfwd, int ldy, int grad_mode, const float *coef, const int32_t *pool_arg,
                          int pool_ns, const float *X, int ldx, int in_mode, const float *in_scale, const float *in_shift, const float *W, int ldw,
                          float *dX, int lddx, float *dW, int lddw, long long dw_slot_stride, float *dbias, const float *prev_stat,
                          double *bwd_partials, int M, int Co, int Ci, int coef_ld, int arg_ld, int dx_atomic, hipStream_t s)
{
    BwdFused3Args a{dZ, lddz, Yfwd, ldy, coef, pool_arg, pool_ns, X, ldx, in_scale, in_shift, W, ldw, dX, lddx, dW, lddw, dbias, prev_stat,
                    bwd_partials, M, dw_slot_stride, coef_ld, arg_ld, dx_atomic};
#define P2C_F3(G_, I_) return dispatch3<G_, I_>(Co, Ci, a, s)
    if (grad_mode == 0) { if (in_mode == 0) P2C_F3(0, 0); P2C_F3(0, 1); }
    if (grad_mode == 1) { if (in_mode == 0) P2C_F3(1, 0); P2C_F3(1, 1); }
    if (in_mode == 0) P2C_F3(2, 0);
    P2C_F3(2, 1);
#undef P2C_F3
}
