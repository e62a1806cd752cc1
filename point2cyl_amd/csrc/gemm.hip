// gemm.hip -- the shared per-point MLP as fp32 MFMA GEMMs with fused prologues / epilogues (gfx950).
//
// One LDS-tiled kernel template, C[I,J] = sum_k A(i,k) * B(k,j), built on
// v_mfma_f32_32x32x2_f32 (exact fp32, bitwise an fmaf chain; 64 cycles per 32x32x2 block, so one wave
// per SIMD already saturates the matrix pipe and the LDS feed is 1 dword per operand per MFMA).
// The three GEMMs of a 1x1-conv layer are instances of it; what differs is how an operand ELEMENT is
// produced while its tile is loaded (the "operand" structs) and what the epilogue does:
//
//   forward       Y  = act_in(X) . W^T + b      A = act_in(X)[m,k]        B = W[n,k]         epi: +bias, store, BN partial sums
//   backward-data dX = dY . W                   A = dY(dZ,Y)[m,co]        B = W[co,ci]       epi: (*dropout mask), store,
//                                                                                                 ReLU+BN-backward partial sums of the layer below
//   backward-wgt  dW = dY^T . act_in(X)         A = dY(dZ,Y)[m,co] (k=m)  B = act_in(X)[m,ci] (k=m)   epi: atomicAdd (split over m)
//
// act_in folds the previous layer's BatchNorm+ReLU (+dropout) into the load, dY(dZ,Y) folds this layer's
// ReLU + train-mode BatchNorm backward into the load, so neither post-activation tensors nor dY tensors
// ever exist in HBM (reference: models/pointnet_util.py:201-205, :317-319, pointnet_extrusion.py:58-65).
//
// Tiling: 256 threads = 4 waves as 2x2; a wave owns (TM*32) x (TN*32) of C as TM*TN accumulators of 16
// VGPRs; workgroup tile (64*TM) x (64*TN); k-tile 32.  Operand tiles are staged k-major in LDS
// (As[k][i], Bs[k][j]) so the MFMA fragment read (lane -> i = lane&31, k = lane>>5) is a conflict-free
// ds_read_b32: tiles whose global layout is k-contiguous are transposed on the way in (row stride
// ROWS+1 keeps the 4-byte transposing stores conflict-free), tiles that are i/j-contiguous are stored
// with 16-byte writes (row stride ROWS+4).  Global loads for tile t+1 are issued before the MFMAs of
// tile t (register prefetch).  Every operand load is BRANCH-FREE: addresses are clamped into the matrix
// and out-of-range elements are zeroed by a select, all leading dimensions / channel counts are
// multiples of 4 (the host pads), so the compiler can keep the 16-byte loads in flight across the MFMAs.
#include "common.h"
#include "fwd_pp.h"
#include <type_traits>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GK 32   // k-tile

// ------------------------------------------------------------------------------------------------
// Operands: row-major [R, C] matrices, channel dim contiguous, C % 4 == 0, 16-byte aligned rows.
// load4(r, c) yields 4 consecutive channels of row r AFTER the element transform, zero outside [R, C].
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 sel4(bool ok, float4 v) { return ok ? v : zero4(); }

struct OpPlain {
    const float *p; int ld;
    __device__ __forceinline__ float4 load4(int r, int c, int R, int C) const
    {
        const int rr = min(r, R - 1), cc = min(c, C - 4);
        const float4 v = *reinterpret_cast<const float4 *>(p + (size_t)rr * ld + cc);
        return sel4(r < R && c < C, v);
    }
};

// act_in(X): MODE 0 identity, 1 relu(scale*x+shift), 2 same * dropout mask * dscale (mask bytes [M,C]),
//            3 same with the mask regenerated from a counter hash (mask -> const uint32_t seed[2], ldmask -> threshold)
template <int MODE>
struct OpActIn {
    const float *p; int ld;
    const float *scale; const float *shift;
    const uint8_t *mask; int ldmask; float dscale;
    __device__ __forceinline__ float4 load4(int r, int c, int R, int C) const
    {
        const int rr = min(r, R - 1), cc = min(c, C - 4);
        float4 v = *reinterpret_cast<const float4 *>(p + (size_t)rr * ld + cc);
        if (MODE >= 1) {
            const float4 s = *reinterpret_cast<const float4 *>(scale + cc), t = *reinterpret_cast<const float4 *>(shift + cc);
            v.x = fmaxf(s.x * v.x + t.x, 0.f);
            v.y = fmaxf(s.y * v.y + t.y, 0.f);
            v.z = fmaxf(s.z * v.z + t.z, 0.f);
            v.w = fmaxf(s.w * v.w + t.w, 0.f);
        }
        if (MODE == 2) {
            const uchar4 m = *reinterpret_cast<const uchar4 *>(mask + (size_t)rr * ldmask + cc);
            v.x = m.x ? v.x * dscale : 0.f;
            v.y = m.y ? v.y * dscale : 0.f;
            v.z = m.z ? v.z * dscale : 0.f;
            v.w = m.w ? v.w * dscale : 0.f;
        }
        if (MODE == 3) {
            const uint32_t *sd = reinterpret_cast<const uint32_t *>(mask);
            const uint32_t lo = sd[0], hi = sd[1], thr = (uint32_t)ldmask, e = (uint32_t)rr * (uint32_t)C + (uint32_t)cc;
            const uint32_t hq = p2c_hash32(lo, hi, e >> 2);            // e is a multiple of 4: one hash for the four elements
            v.x = p2c_keep4(hq, 0, thr) ? v.x * dscale : 0.f;
            v.y = p2c_keep4(hq, 1, thr) ? v.y * dscale : 0.f;
            v.z = p2c_keep4(hq, 2, thr) ? v.z * dscale : 0.f;
            v.w = p2c_keep4(hq, 3, thr) ? v.w * dscale : 0.f;
        }
        return sel4(r < R && c < C, v);
    }
};

// dY rebuilt from the upstream gradient dZ (w.r.t. relu(bn(Y))) and the saved pre-BN Y:
//   MODE 0: dY = dZ;   MODE 1: dY = gs*(dZ*[scale*Y+shift > 0]) + q*Y + p,  coef = [scale|shift|gs|q|p] x Cc
//   MODE 2: as 1, but dZ is never materialised: the layer is followed by the max-pool over `ns` neighbours, so
//           dZ[m,c] = dOut[m/ns, c] if arg[m/ns, c] == m%ns else 0  (dz = dOut [G,C], arg = winners [G,C])
template <int MODE>
struct OpGrad {
    const float *dz; int lddz;
    const float *y; int ldy;
    const float *coef; int Cc;
    const int32_t *arg; int ns;
    __device__ __forceinline__ float4 load4(int r, int c, int R, int C) const
    {
        const int rr = min(r, R - 1), cc = min(c, C - 4);
        float4 g;
        if (MODE == 2) {
            const int grp = rr / ns, j = rr - grp * ns;
            const float4 d = *reinterpret_cast<const float4 *>(dz + (size_t)grp * lddz + cc);
            const int4 a = *reinterpret_cast<const int4 *>(arg + (size_t)grp * Cc + cc);
            g = make_float4(a.x == j ? d.x : 0.f, a.y == j ? d.y : 0.f, a.z == j ? d.z : 0.f, a.w == j ? d.w : 0.f);
        } else {
            g = *reinterpret_cast<const float4 *>(dz + (size_t)rr * lddz + cc);
        }
        if (MODE == 0) return sel4(r < R && c < C, g);
        const float4 yy = *reinterpret_cast<const float4 *>(y + (size_t)rr * ldy + cc);
        const float4 s = *reinterpret_cast<const float4 *>(coef + cc), t = *reinterpret_cast<const float4 *>(coef + Cc + cc);
        const float4 gs = *reinterpret_cast<const float4 *>(coef + 2 * Cc + cc), q = *reinterpret_cast<const float4 *>(coef + 3 * Cc + cc),
                     pp = *reinterpret_cast<const float4 *>(coef + 4 * Cc + cc);
        float4 o;
        o.x = gs.x * ((s.x * yy.x + t.x > 0.f) ? g.x : 0.f) + q.x * yy.x + pp.x;
        o.y = gs.y * ((s.y * yy.y + t.y > 0.f) ? g.y : 0.f) + q.y * yy.y + pp.y;
        o.z = gs.z * ((s.z * yy.z + t.z > 0.f) ? g.z : 0.f) + q.z * yy.z + pp.z;
        o.w = gs.w * ((s.w * yy.w + t.w > 0.f) ? g.w : 0.f) + q.w * yy.w + pp.w;
        return sel4(r < R && c < C, o);
    }
};

// ------------------------------------------------------------------------------------------------
// Tile movers.  ROWS = extent of the non-k dim in the tile (64*TM or 64*TN).
// ------------------------------------------------------------------------------------------------
template <int ROWS, bool KCONTIG>
struct Tile {
    static constexpr int LD = KCONTIG ? ROWS + 1 : ROWS + 4;
    static constexpr int UNITS = ROWS * GK / 4 / 256;   // float4 units per thread
    template <class Op>
    static __device__ __forceinline__ void load(const Op &op, int o0, int Olim, int k0, int Klim, float4 (&regs)[UNITS])
    {
#pragma unroll
        for (int i = 0; i < UNITS; ++i) {
            const int u = threadIdx.x + 256 * i;
            if (KCONTIG) {
                const int nk = u >> 3, kq = u & 7;                     // 8 float4 per 32-wide k row
                regs[i] = op.load4(o0 + nk, k0 + kq * 4, Olim, Klim);
            } else {
                const int kr = u / (ROWS / 4), c4 = u % (ROWS / 4);
                regs[i] = op.load4(k0 + kr, o0 + c4 * 4, Klim, Olim);
            }
        }
    }
    static __device__ __forceinline__ void store(float *s, const float4 (&regs)[UNITS])
    {
#pragma unroll
        for (int i = 0; i < UNITS; ++i) {
            const int u = threadIdx.x + 256 * i;
            if (KCONTIG) {
                const int nk = u >> 3, kq = u & 7;
                s[(kq * 4 + 0) * LD + nk] = regs[i].x;
                s[(kq * 4 + 1) * LD + nk] = regs[i].y;
                s[(kq * 4 + 2) * LD + nk] = regs[i].z;
                s[(kq * 4 + 3) * LD + nk] = regs[i].w;
            } else {
                const int kr = u / (ROWS / 4), c4 = u % (ROWS / 4);
                *reinterpret_cast<float4 *>(&s[kr * LD + c4 * 4]) = regs[i];
            }
        }
    }
};

// ------------------------------------------------------------------------------------------------
// bf16x3-split operand tiles (SPLIT = true): the same ROWS x GK tile as three bf16 planes [3][ROWS][LDK], k contiguous, so that one
// ds_read_b128 per plane is the A / B fragment of a v_mfma_f32_32x32x16_bf16 (lane (i, h): k = 8h .. 8h+7 of row i).  Every fp32 element is
// split once, here, into hi + mid + lo (round to nearest each: x = hi + mid + lo up to 2^-26 |x|); the six products mm, hl, lh, hm, mh, hh
// into the fp32 accumulator give the fp32 product to 2-3e-7 of sum |a b| - what the fp32 MFMA gives (tools/ubench/split_bf16.hip) - at
// 6 x 32 cycles per 32 x 32 x 16 block instead of 8 x 64.  Row stride 80 bytes: 16-byte aligned; see Tile3 for the row permutation.
// ------------------------------------------------------------------------------------------------
typedef __bf16 g_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 g_bf16x2 __attribute__((ext_vector_type(2)));
typedef float g_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t g_pk_bf16(float a, float b)
{
    const g_v2f v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, g_bf16x2));
}
__device__ __forceinline__ float g_bf_lo(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float g_bf_hi(uint32_t p) { return __builtin_bit_cast(float, p & 0xFFFF0000u); }
// (a, b) -> the three packed pairs
__device__ __forceinline__ void g_split_pair(float a, float b, uint32_t &h, uint32_t &m, uint32_t &l)
{
    h = g_pk_bf16(a, b);
    const float ra = a - g_bf_lo(h), rb = b - g_bf_hi(h);
    m = g_pk_bf16(ra, rb);
    l = g_pk_bf16(ra - g_bf_lo(m), rb - g_bf_hi(m));
}

template <int ROWS, bool KCONTIG>
struct Tile3 {
    static constexpr int LDK = GK + 8;                 // bf16 elements per row
    // Rows are stored PERMUTED: row r lives at (r % 4) * RS + r / 4, RS = ROWS / 4 rounded up to 4 (mod 16).  A tile whose k runs down the
    // memory rows (KCONTIG = false) arrives as float4s of 4 consecutive ROWS at one k; a thread takes 4 (2 for ROWS = 64) consecutive k of
    // its row quad, splits, and writes for each of its 4 rows one 8-byte (4-byte) piece per plane - lanes with consecutive row quads then
    // hit consecutive physical rows (80 bytes apart: conflict-free), instead of rows 320 bytes apart (16 lanes on 4 banks).  20 * RS = 16
    // (mod 64 dwords) keeps the 16-lane groups of the fragment reads (rows r .. r + 15 -> 4 x 4 physical rows) on 16 distinct bank quads.
    static constexpr int RS = ROWS == 64 ? 20 : 36;
    static constexpr int PLANE = 4 * RS * LDK;         // elements per plane
    static constexpr int UNITS = ROWS * GK / 4 / 256;
    static_assert(ROWS == 64 || ROWS == 128, "");
    static __device__ __forceinline__ int prow(int r) { return (r & 3) * RS + (r >> 2); }
    // KCONTIG = false only: consecutive k per thread (Tile<>::load deals k = kr, kr + 8, ...)
    template <class Op>
    static __device__ __forceinline__ void load_t(const Op &op, int o0, int Olim, int k0, int Klim, float4 (&regs)[UNITS])
    {
        const int c4 = threadIdx.x % (ROWS / 4), kg = threadIdx.x / (ROWS / 4);
#pragma unroll
        for (int i = 0; i < UNITS; ++i) regs[i] = op.load4(k0 + kg * UNITS + i, o0 + c4 * 4, Klim, Olim);
    }
    static __device__ __forceinline__ void store(uint16_t *s, const float4 (&regs)[UNITS])
    {
        if (KCONTIG) {
#pragma unroll
            for (int i = 0; i < UNITS; ++i) {
                const int u = threadIdx.x + 256 * i;
                uint32_t h0, m0, l0, h1, m1, l1;
                g_split_pair(regs[i].x, regs[i].y, h0, m0, l0);
                g_split_pair(regs[i].z, regs[i].w, h1, m1, l1);
                const int nk = u >> 3, kq = u & 7;                      // 4 consecutive k of row nk: one 8-byte store per plane
                uint16_t *o = s + prow(nk) * LDK + kq * 4;
                *reinterpret_cast<uint2 *>(o) = make_uint2(h0, h1);
                *reinterpret_cast<uint2 *>(o + PLANE) = make_uint2(m0, m1);
                *reinterpret_cast<uint2 *>(o + 2 * PLANE) = make_uint2(l0, l1);
            }
        } else {
            const int c4 = threadIdx.x % (ROWS / 4), kg = threadIdx.x / (ROWS / 4);
            const float *f = reinterpret_cast<const float *>(&regs[0]);   // f[4 * i + e]: k = kg * UNITS + i, row = 4 * c4 + e
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint16_t *o = s + (e * RS + c4) * LDK + kg * UNITS;
                uint32_t h0, m0, l0;
                g_split_pair(f[e], f[4 + e], h0, m0, l0);
                if (UNITS == 4) {
                    uint32_t h1, m1, l1;
                    g_split_pair(f[8 + e], f[12 + e], h1, m1, l1);
                    *reinterpret_cast<uint2 *>(o) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2 *>(o + PLANE) = make_uint2(m0, m1);
                    *reinterpret_cast<uint2 *>(o + 2 * PLANE) = make_uint2(l0, l1);
                } else {
                    *reinterpret_cast<uint32_t *>(o) = h0;
                    *reinterpret_cast<uint32_t *>(o + PLANE) = m0;
                    *reinterpret_cast<uint32_t *>(o + 2 * PLANE) = l0;
                }
            }
        }
    }
};

// ------------------------------------------------------------------------------------------------
// Epilogues.  C/D fragment of mfma 32x32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
// ------------------------------------------------------------------------------------------------
struct EpiFwd {
    float *Y; int ldy; const float *bias; double *partials;  // [P2C_STAT_SLOTS][2][N] fp64 accumulators (atomic) or NULL
    // optional per-row-group term gbias[row / rpg, col] (rpg a multiple of the row tile): it is part of the value whose
    // BatchNorm sums are taken (a layer fed by [features | one vector repeated over the rows of a group], see FP3)
    const float *gbias; int ldgb; int rpg;
};
struct EpiBwdData {
    float *dX; int lddx; const uint8_t *mask; int ldmask; float mscale;   // ldmask < 0: hashed mask, mask -> seed[2], thr below
    uint32_t thr;
    // fused ReLU+BN-backward reduction of the layer BELOW (whose pre-BN output is Yp, same shape as dX):
    const float *Yp; int ldyp; const float *pstat;   // pstat [4][J]: scale, shift, mean, invstd
    double *partials;                                // [P2C_STAT_SLOTS][2][J] fp64 accumulators (atomic) or NULL
    // optional: dX *= sigmoid(beta * Z) (Z the pre-activation the product is the gradient of, same shape as dX): the derivative of the
    // softplus that precedes the layer, applied where the product leaves the registers (the implicit decoder, IGR/network.py:80-82)
    const float *spz; int ldspz; float sp_beta, sp_thr;
};
struct EpiAtomic {
    float *dW; int lddw; float *dbias;
    long long slot_stride;     // elements between the 8 copies of dW the split-k workgroups spread their atomics over (0: one copy)
};

// shared-memory floats one workgroup of the GEMM body needs
template <int TM, int TN, bool AK, bool BK_, bool SPLIT = false>
constexpr int gemm_smem_floats()
{
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int OPS = SPLIT ? 3 * (Tile3<BM, AK>::PLANE + Tile3<BN, BK_>::PLANE) / 2 : GK * Tile<BM, AK>::LD + GK * Tile<BN, BK_>::LD, OUT = BM * (BN + 4);
    return (OPS > OUT ? OPS : OUT) + 2 * BN + BM;
}

// The GEMM of one workgroup (256 threads) for tile (bx, by) of the output and k-split bz; smem: gemm_smem_floats<...>() floats.
// A __device__ body so that one launch can run the workgroups of two different GEMMs side by side (gemm_dual_kernel below).
template <int TM, int TN, bool AK, bool BK_, class OpA, class OpB, class Epi, bool SPLIT = false>
__device__ __forceinline__ void gemm_body(float *smem, const OpA &opA, const OpB &opB, const Epi &epi, int I, int J, int K, int k_per_split,
                                          int bx, int by, int bz)
{
    constexpr int BM = 64 * TM, BN = 64 * TN;
    using TA = Tile<BM, AK>;
    using TB = Tile<BN, BK_>;
    using TA3 = Tile3<BM, AK>;
    using TB3 = Tile3<BN, BK_>;
    // operand tiles + [2][BN] stat scratch (+BM), or the BM x (BN+4) output tile the epilogue stages for full-row stores
    constexpr int SMEM_OPS = SPLIT ? 3 * (TA3::PLANE + TB3::PLANE) / 2 : GK * TA::LD + GK * TB::LD, SMEM_OUT = BM * (BN + 4);
    float *As = smem, *Bs = smem + GK * TA::LD;
    uint16_t *Ap = reinterpret_cast<uint16_t *>(smem), *Bp = Ap + 3 * TA3::PLANE;       // SPLIT: [3][BM][LDK], [3][BN][LDK]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i0 = bx * BM, j0 = by * BN;
    const int kbeg = bz * k_per_split;
    const int kend = min(K, kbeg + k_per_split);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float4 ra[TA::UNITS], rb[TB::UNITS];
    float dbias_acc = 0.f;
    auto loadA = [&](int k0) { if constexpr (SPLIT && !AK) TA3::load_t(opA, i0, I, k0, kend, ra); else TA::load(opA, i0, I, k0, kend, ra); };
    auto loadB = [&](int k0) { if constexpr (SPLIT && !BK_) TB3::load_t(opB, j0, J, k0, kend, rb); else TB::load(opB, j0, J, k0, kend, rb); };
    loadA(kbeg);
    loadB(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += GK) {
        __syncthreads();              // previous tile fully consumed
        if constexpr (SPLIT) {
            TA3::store(Ap, ra);
            TB3::store(Bp, rb);
        } else {
            TA::store(As, ra);
            TB::store(Bs, rb);
        }
        __syncthreads();
        if (k0 + GK < kend) {         // prefetch the next tile while the matrix pipe works
            loadA(k0 + GK);
            loadB(k0 + GK);
        }
        if constexpr (std::is_same<Epi, EpiAtomic>::value) {
            if (epi.dbias && by == 0 && tid < BM) {
                float s = 0.f;
                if constexpr (SPLIT) {
#pragma unroll
                    for (int p = 0; p < 3; ++p)
#pragma unroll
                        for (int q = 0; q < GK / 8; ++q) {
                            const g_bf16x8 v = *reinterpret_cast<const g_bf16x8 *>(Ap + p * TA3::PLANE + TA3::prow(tid) * TA3::LDK + q * 8);
#pragma unroll
                            for (int e = 0; e < 8; ++e) s += (float)v[e];
                        }
                } else {
#pragma unroll 8
                    for (int kk = 0; kk < GK; ++kk) s += As[kk * TA::LD + tid];
                }
                dbias_acc += s;
            }
        }
        const int kl = lane >> 5, il = lane & 31;
        if constexpr (SPLIT) {
#pragma unroll
            for (int kc = 0; kc < GK / 16; ++kc) {
                g_bf16x8 a[TM][3], b[TN][3];
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        a[t][p] = *reinterpret_cast<const g_bf16x8 *>(Ap + p * TA3::PLANE + TA3::prow(wm * (TM * 32) + t * 32 + il) * TA3::LDK + kc * 16 + kl * 8);
#pragma unroll
                for (int t = 0; t < TN; ++t)
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        b[t][p] = *reinterpret_cast<const g_bf16x8 *>(Bp + p * TB3::PLANE + TB3::prow(wn * (TN * 32) + t * 32 + il) * TB3::LDK + kc * 16 + kl * 8);
                // smallest products first; consecutive MFMAs go to different accumulators where there are several
#define P2C_G3(PA_, PB_)                                                                                             \
    _Pragma("unroll") for (int ta = 0; ta < TM; ++ta) _Pragma("unroll") for (int tb = 0; tb < TN; ++tb)               \
        acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ta][PA_], b[tb][PB_], acc[ta][tb], 0, 0, 0)
                P2C_G3(1, 1); P2C_G3(0, 2); P2C_G3(2, 0); P2C_G3(0, 1); P2C_G3(1, 0); P2C_G3(0, 0);
#undef P2C_G3
            }
        } else
#pragma unroll
        for (int kk = 0; kk < GK; kk += 2) {
            float a[TM], b[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t) a[t] = As[(kk + kl) * TA::LD + wm * (TM * 32) + t * 32 + il];
#pragma unroll
            for (int t = 0; t < TN; ++t) b[t] = Bs[(kk + kl) * TB::LD + wn * (TN * 32) + t * 32 + il];
#pragma unroll
            for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                for (int tb = 0; tb < TN; ++tb)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
        }
    }

    // ---------------------------------------------------------------- epilogue
    const int col_l = lane & 31, rquad = lane >> 5;
    float *sstat = smem + (SMEM_OPS > SMEM_OUT ? SMEM_OPS : SMEM_OUT);      // [2][BN]: the two wm-waves of a column combine here
    if constexpr (std::is_same<Epi, EpiFwd>::value) {
        // The C/D fragment gives each lane one column of 16 scattered rows: stored directly, a wave instruction writes
        // two 128-byte pieces of two different rows.  Stage the tile in LDS instead (the operand tiles are dead) and
        // write whole rows, 16 bytes per lane: the 512-byte rows of Y then reach HBM as full bursts.
        constexpr int LDO = BN + 4;
        __syncthreads();
        if (epi.partials && tid < 2 * BN) sstat[tid] = 0.f;
#pragma unroll
        for (int tb = 0; tb < TN; ++tb) {
            const int cl = wn * (TN * 32) + tb * 32 + col_l;
            const int col = j0 + cl;
            const bool cok = col < J;
            const float bv = (epi.bias && cok) ? epi.bias[col] : 0.f;
            const float gbv = (epi.gbias && cok) ? epi.gbias[(size_t)(i0 / epi.rpg) * epi.ldgb + col] : 0.f;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = wm * (TM * 32) + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * rquad;
                    const float v = acc[ta][tb][r] + gbv;
                    smem[rl * LDO + cl] = v + bv;
                    if (i0 + rl < I && cok) {
                        s1 += v;
                        s2 += v * v;
                    }
                }
            if (epi.partials) {
                s1 += __shfl_xor(s1, 32);
                s2 += __shfl_xor(s2, 32);
                __syncthreads();                                              // sstat zeroed (first tb) / tile rows landed
                if (rquad == 0) {
                    atomicAdd(&sstat[cl], s1);                               // exactly two adds per slot: order-independent
                    atomicAdd(&sstat[BN + cl], s2);
                }
            }
        }
        __syncthreads();
        {
            constexpr int V = BN / 4;                                        // float4 per tile row
            const bool vec_ok = (epi.ldy & 3) == 0 && (((uintptr_t)epi.Y) & 15) == 0;
            for (int u = tid; u < BM * V; u += 256) {
                const int rl = u / V, c4 = (u % V) * 4;
                const int row = i0 + rl, col = j0 + c4;
                if (row >= I || col >= J) continue;
                const float4 v = *reinterpret_cast<const float4 *>(&smem[rl * LDO + c4]);
                float *o = epi.Y + (size_t)row * epi.ldy + col;
                if (vec_ok && col + 3 < J) {
                    *reinterpret_cast<float4 *>(o) = v;
                } else {
                    o[0] = v.x;
                    if (col + 1 < J) o[1] = v.y;
                    if (col + 2 < J) o[2] = v.z;
                    if (col + 3 < J) o[3] = v.w;
                }
            }
        }
        if (epi.partials) {
            if (tid < BN && j0 + tid < J) {        // one fp64 atomic per column and workgroup into its slot
                double *o = epi.partials + (size_t)(bx % P2C_STAT_SLOTS) * 2 * J;
                atomicAdd(&o[j0 + tid], (double)sstat[tid]);
                atomicAdd(&o[J + j0 + tid], (double)sstat[BN + tid]);
            }
        }
    } else if constexpr (std::is_same<Epi, EpiBwdData>::value) {
        if ((epi.lddx & 3) == 0 && (((uintptr_t)epi.dX) & 15) == 0 && (J & 3) == 0 &&
            (!epi.spz || ((epi.ldspz & 3) == 0 && (((uintptr_t)epi.spz) & 15) == 0)) &&
            (!epi.partials || ((epi.ldyp & 3) == 0 && (((uintptr_t)epi.Yp) & 15) == 0)) && !(epi.mask && epi.ldmask >= 0)) {
            // Like the forward: stage the tile in LDS (the operand tiles are dead) and move whole rows - dX out, and the pre-activations
            // the epilogue reads (Yp for the ReLU + BatchNorm-backward sums of the layer below, Z for the softplus derivative) in - as
            // 16-byte pieces per lane instead of two 128-byte pieces per wave instruction.  A thread keeps the same four columns over all
            // its rows, so the column sums stay in registers until one LDS add per thread.
            constexpr int LDO = BN + 4, V = BN / 4;
            __syncthreads();
            if (epi.partials && tid < 2 * BN) sstat[tid] = 0.f;
#pragma unroll
            for (int tb = 0; tb < TN; ++tb)
#pragma unroll
                for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        smem[(wm * (TM * 32) + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * rquad) * LDO + wn * (TN * 32) + tb * 32 + col_l] = acc[ta][tb][r];
            __syncthreads();
            static_assert(256 % V == 0, "");
            const int c4 = (tid % V) * 4, col = j0 + c4;
            float psc[4] = {0.f, 0.f, 0.f, 0.f}, psh[4] = {0.f, 0.f, 0.f, 0.f}, pmu[4] = {0.f, 0.f, 0.f, 0.f}, pis[4] = {0.f, 0.f, 0.f, 0.f};
            float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
            if (epi.partials && col < J) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { psc[c] = epi.pstat[col + c]; psh[c] = epi.pstat[J + col + c]; pmu[c] = epi.pstat[2 * J + col + c]; pis[c] = epi.pstat[3 * J + col + c]; }
            }
            for (int rl = tid / V; rl < BM; rl += 256 / V) {
                const int row = i0 + rl;
                if (row >= I || col >= J) continue;
                float4 v4 = *reinterpret_cast<const float4 *>(&smem[rl * LDO + c4]);
                float *vv = reinterpret_cast<float *>(&v4);
                if (epi.mask) {                                             // hashed keep-mask (ldmask < 0)
                    const uint32_t *sd = reinterpret_cast<const uint32_t *>(epi.mask);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        vv[c] = p2c_keep(sd[0], sd[1], (uint32_t)row * (uint32_t)J + (uint32_t)(col + c), epi.thr) ? vv[c] * epi.mscale : 0.f;
                }
                if (epi.spz) {
                    const float4 z = *reinterpret_cast<const float4 *>(epi.spz + (size_t)row * epi.ldspz + col);
                    const float zz[4] = {z.x, z.y, z.z, z.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float bz = zz[c] * epi.sp_beta;
                        if (!(bz > epi.sp_thr)) {                           // linear region of softplus: derivative 1
                            const float e = __expf(-fabsf(bz)), sg = 1.f / (1.f + e);
                            vv[c] *= bz >= 0.f ? sg : 1.f - sg;
                        }
                    }
                }
                *reinterpret_cast<float4 *>(epi.dX + (size_t)row * epi.lddx + col) = v4;
                if (epi.partials) {
                    const float4 y = *reinterpret_cast<const float4 *>(epi.Yp + (size_t)row * epi.ldyp + col);
                    const float yy[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float g = (psc[c] * yy[c] + psh[c] > 0.f) ? vv[c] : 0.f;
                        s1[c] += g;
                        s2[c] += g * ((yy[c] - pmu[c]) * pis[c]);
                    }
                }
            }
            if (epi.partials) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { atomicAdd(&sstat[c4 + c], s1[c]); atomicAdd(&sstat[BN + c4 + c], s2[c]); }
                __syncthreads();
                if (tid < BN && j0 + tid < J) {        // one fp64 atomic per column and workgroup into its slot
                    double *o = epi.partials + (size_t)(bx % P2C_STAT_SLOTS) * 2 * J;
                    atomicAdd(&o[j0 + tid], (double)sstat[tid]);
                    atomicAdd(&o[J + j0 + tid], (double)sstat[BN + tid]);
                }
            }
            return;
        }
        if (epi.partials) {
            __syncthreads();
            if (tid < 2 * BN) sstat[tid] = 0.f;
            __syncthreads();
        }
#pragma unroll
        for (int tb = 0; tb < TN; ++tb) {
            const int col = j0 + wn * (TN * 32) + tb * 32 + col_l;
            const bool cok = col < J;
            float psc = 0.f, psh = 0.f, pmu = 0.f, pis = 0.f;
            if (epi.partials && cok) {
                psc = epi.pstat[col]; psh = epi.pstat[J + col]; pmu = epi.pstat[2 * J + col]; pis = epi.pstat[3 * J + col];
            }
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i0 + wm * (TM * 32) + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * rquad;
                    if (row < I && cok) {
                        float v = acc[ta][tb][r];
                        if (epi.mask) {
                            bool keep;
                            if (epi.ldmask < 0) {
                                const uint32_t *sd = reinterpret_cast<const uint32_t *>(epi.mask);
                                keep = p2c_keep(sd[0], sd[1], (uint32_t)row * (uint32_t)J + (uint32_t)col, epi.thr);
                            } else {
                                keep = epi.mask[(size_t)row * epi.ldmask + col] != 0;
                            }
                            v = keep ? v * epi.mscale : 0.f;
                        }
                        if (epi.spz) {
                            const float bz = epi.spz[(size_t)row * epi.ldspz + col] * epi.sp_beta;
                            if (!(bz > epi.sp_thr)) {                       // linear region of softplus: derivative 1
                                const float e = __expf(-fabsf(bz)), sg = 1.f / (1.f + e);
                                v *= bz >= 0.f ? sg : 1.f - sg;
                            }
                        }
                        epi.dX[(size_t)row * epi.lddx + col] = v;
                        if (epi.partials) {
                            const float yp = epi.Yp[(size_t)row * epi.ldyp + col];
                            const float g = (psc * yp + psh > 0.f) ? v : 0.f;
                            s1 += g;
                            s2 += g * ((yp - pmu) * pis);
                        }
                    }
                }
            if (epi.partials) {
                s1 += __shfl_xor(s1, 32);
                s2 += __shfl_xor(s2, 32);
                if (rquad == 0) {
                    atomicAdd(&sstat[wn * (TN * 32) + tb * 32 + col_l], s1);
                    atomicAdd(&sstat[BN + wn * (TN * 32) + tb * 32 + col_l], s2);
                }
            }
        }
        if (epi.partials) {
            __syncthreads();
            if (tid < BN && j0 + tid < J) {        // one fp64 atomic per column and workgroup into its slot
                double *o = epi.partials + (size_t)(bx % P2C_STAT_SLOTS) * 2 * J;
                atomicAdd(&o[j0 + tid], (double)sstat[tid]);
                atomicAdd(&o[J + j0 + tid], (double)sstat[BN + tid]);
            }
        }
    } else {
#pragma unroll
        for (int tb = 0; tb < TN; ++tb) {
            const int col = j0 + wn * (TN * 32) + tb * 32 + col_l;
#pragma unroll
            for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i0 + wm * (TM * 32) + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * rquad;
                    if (row < I && col < J) atomicAdd(&epi.dW[(size_t)(bz & 7) * epi.slot_stride + (size_t)row * epi.lddw + col], acc[ta][tb][r]);
                }
        }
        if (epi.dbias && by == 0 && tid < BM && i0 + tid < I) atomicAdd(&epi.dbias[i0 + tid], dbias_acc);
    }
}

template <int TM, int TN, bool AK, bool BK_, class OpA, class OpB, class Epi, bool SPLIT = false>
__global__ void __launch_bounds__(256) gemm_kernel(OpA opA, OpB opB, Epi epi, int I, int J, int K, int k_per_split)
{
    __shared__ __attribute__((aligned(16))) float smem[gemm_smem_floats<TM, TN, AK, BK_, SPLIT>()];
    gemm_body<TM, TN, AK, BK_, OpA, OpB, Epi, SPLIT>(smem, opA, opB, epi, I, J, K, k_per_split, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Both backward GEMMs of one layer in ONE launch: workgroups [0, nA) run the data gradient (64 x 64*TNA tiles), the rest the weight
// gradient (TMB x TNB tiles, split over the rows).  For the 4 k - 16 k-row layers neither GEMM alone has enough workgroups to fill
// the chip and each is a chain of latency-bound steps; side by side they overlap (two streams would do the same at the price of
// fork / join events in the graph; one launch has no price).
template <int TNA, int TMB, int TNB, class OpG, class OpI, bool SPLIT = false>
__global__ void __launch_bounds__(256) gemm_dual_kernel(OpG gA, OpPlain wA, EpiBwdData eA, int MA, int KA, int NA, int kpsA, int nAx, int nA,
                                                        OpG gB, OpI xB, EpiAtomic eB, int NB, int KB, int MB, int kpsB, int nBx, int nBy)
{
    constexpr int FA = gemm_smem_floats<1, TNA, true, false, SPLIT>(), FB = gemm_smem_floats<TMB, TNB, false, false, SPLIT>();
    __shared__ __attribute__((aligned(16))) float smem[FA > FB ? FA : FB];
    const int b = blockIdx.x;
    if (b < nA) {
        gemm_body<1, TNA, true, false, OpG, OpPlain, EpiBwdData, SPLIT>(smem, gA, wA, eA, MA, KA, NA, kpsA, b % nAx, b / nAx, 0);
    } else {
        const int c = b - nA;
        gemm_body<TMB, TNB, false, false, OpG, OpI, EpiAtomic, SPLIT>(smem, gB, xB, eB, NB, KB, MB, kpsB, c % nBx, (c / nBx) % nBy, c / (nBx * nBy));
    }
}

// The tiled kernels follow the switch of the persistent ones (p2c_set_mfma_mode / P2C_MFMA=f32).
static bool gemm_split() { return p2c_mfma_split(); }

// ... and the products with an operand whose k runs down the rows in memory (both backward products: their tiles are transposed on the way
// into LDS, Tile3's permuted rows make that conflict-free): sa3.2 dX 83 -> 59 us, dW 70 -> 56 us; step -0.06 ms.
static bool gemm_split_t() { return gemm_split(); }

static inline bool al16(const void *p) { return ((uintptr_t)p & 15) == 0; }
#define P2C_REQ_ALIGNED(ptr_, ld_)                                   \
    do {                                                             \
        if (!al16(ptr_) || ((ld_) & 3)) return P2C_EALIGN;           \
    } while (0)

// Row-tile height of the forward / backward-data kernels (also the granularity of the per-tile partial sums).
static constexpr int tile_m() { return 64; }
extern "C" int p2c_linear_tile_m(void) { return tile_m(); }
// Small problems (the SA3 / FP3 / FP2 layers: 4 k - 16 k rows): with 64 x 128 tiles the grid has fewer workgroups than the chip has
// CUs, one workgroup per CU cannot hide the operand latency behind its own MFMAs, and the kernel runs at 15-40 TFLOP/s.  64 x 64
// tiles double the number of workgroups (twice the loads in flight per CU) at the price of one more LDS read per MFMA.
static bool narrow_tiles(int rows, int cols)
{
    return (long long)p2c_cdiv(rows, 64) * p2c_cdiv(cols, 128) < 512;
}
extern "C" int p2c_linear_stat_tiles(int M) { return (M + tile_m() - 1) / tile_m(); }

// ---- forward, very few rows ------------------------------------------------------------------------
// Y[M,N] = X[M,K] . W[N,K]^T + bias for M <= 32 (FP3's global feature: one 1024-vector per cloud against 256 weight rows).  The tiled
// kernel gives such a problem one row of tiles - N/64 workgroups walking K sequentially, 33 us for 17 MFLOP.  Here one WORKGROUP owns one
// output column: its 256 lanes split K in 16-byte pieces (the weight row is read once, the <= 32 input rows come from L2), 32
// accumulators per lane, a wave reduction per row and one LDS step across the four waves.
#define SKINNY_MAXM 32
__global__ void __launch_bounds__(256) skinny_fwd_kernel(const float *__restrict__ X, int ldx, const float *__restrict__ W, int ldw,
                                                         const float *__restrict__ bias, float *__restrict__ Y, int ldy, int M, int N, int K)
{
    __shared__ float red[4][SKINNY_MAXM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = blockIdx.x;      // one WORKGROUP per output column: K split over 256 lanes
    float acc[SKINNY_MAXM];
#pragma unroll
    for (int m = 0; m < SKINNY_MAXM; ++m) acc[m] = 0.f;
    for (int k = tid * 4; k < K; k += 1024) {
        const float4 w = *reinterpret_cast<const float4 *>(W + (size_t)n * ldw + k);
        // eight input rows in flight at a time (issued back to back, then consumed): one L2 round trip per batch instead of per row
#pragma unroll
        for (int m0 = 0; m0 < SKINNY_MAXM; m0 += 8) {
            float4 x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = *reinterpret_cast<const float4 *>(X + (size_t)min(m0 + j, M - 1) * ldx + k);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[m0 + j] += (x[j].x * w.x + x[j].y * w.y) + (x[j].z * w.z + x[j].w * w.w);
        }
    }
#pragma unroll
    for (int m = 0; m < SKINNY_MAXM; ++m) {
        float v = acc[m];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wave][m] = v;
    }
    __syncthreads();
    if (tid < M) Y[(size_t)tid * ldy + n] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]) + (bias ? bias[n] : 0.f);
}

// ---- forward --------------------------------------------------------------------------------------
template <int MODE>
static int launch_fwd(const float *X, int ldx, const float *W, int ldw, const float *bias, float *Y, int ldy, int M, int N, int K,
                      const float *in_scale, const float *in_shift, const uint8_t *drop_mask, int ldmask, float drop_scale,
                      double *stat_partials, hipStream_t s, const float *gbias = nullptr, int ldgb = 0, int rpg = 1)
{
    OpActIn<MODE> a{X, ldx, in_scale, in_shift, drop_mask, ldmask, drop_scale};
    OpPlain b{W, ldw};
    EpiFwd e{Y, ldy, bias, stat_partials, gbias, ldgb, rpg};
    const int kps = (K + GK - 1) / GK * GK;
#define P2C_FW_(TM_, TN_, SP_)                                                                                                    \
    hipLaunchKernelGGL((gemm_kernel<TM_, TN_, true, true, OpActIn<MODE>, OpPlain, EpiFwd, SP_>), dim3(p2c_cdiv(M, 64 * TM_), p2c_cdiv(N, 64 * TN_), 1), \
                       dim3(256), 0, s, a, b, e, M, N, K, kps)
    // split form from 65,536 rows on (the implicit decoder's 512-wide products: 101 -> 125 TFLOP/s); the 4 k - 16 k-row levels of the
    // backbone stay on the fp32 instructions - there the larger LDS / register footprint costs co-residency with the sampling kernel
    // of the forked stream and the step got 0.03 ms slower
    // ... except the one product of those levels that is large enough to pay (SA3's 512 -> 1024 layer: K * N >= 400,000; -0.02 ms)
    constexpr long long fw_kn = 400000;
#define P2C_FW(TM_, TN_) do { if (gemm_split() && (M >= 65536 || (long long)K * N >= fw_kn)) P2C_FW_(TM_, TN_, true); else P2C_FW_(TM_, TN_, false); } while (0)
    if (N > 64 && !narrow_tiles(M, N)) P2C_FW(1, 2); else P2C_FW(1, 1);
#undef P2C_FW
#undef P2C_FW_
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

extern "C" int p2c_linear_fwd_f32(const float *X, int ldx, const float *W, int ldw, const float *bias, float *Y, int ldy, int M, int N,
                                  int K, int in_mode, const float *in_scale, const float *in_shift, const uint8_t *drop_mask,
                                  int ldmask, float drop_scale, double *stat_partials, void *stream)
{
    if (!X || !W || !Y || M <= 0 || N <= 0 || K <= 0 || in_mode < 0 || in_mode > 3) return P2C_EINVAL;
    if (in_mode >= 1 && (!in_scale || !in_shift)) return P2C_EINVAL;
    if (in_mode == 2 && (!drop_mask || (ldmask & 3) || ((uintptr_t)drop_mask & 3))) return P2C_EINVAL;
    if (in_mode == 3) { if (!drop_mask || !p2c_drop_scale_representable(drop_scale)) return P2C_EINVAL; ldmask = (int)p2c_drop_threshold(drop_scale); }
    if (K & 3) return P2C_EALIGN;
    P2C_REQ_ALIGNED(X, ldx);
    P2C_REQ_ALIGNED(W, ldw);
    if (in_mode >= 1) { P2C_REQ_ALIGNED(in_scale, 0); P2C_REQ_ALIGNED(in_shift, 0); }
    hipStream_t s = (hipStream_t)stream;
    if (M <= SKINNY_MAXM && in_mode == 0 && !stat_partials && K >= 256) {
        hipLaunchKernelGGL(skinny_fwd_kernel, dim3(N), dim3(256), 0, s, X, ldx, W, ldw, bias, Y, ldy, M, N, K);
        P2C_LAUNCH_CHECK();
        return P2C_OK;
    }
    if (p2c_linear_fwd_pp_supported(M, N, K, in_mode)) {
        FwdPPArgs a{X, ldx, W, ldw, bias, Y, ldy, M, N, K == 132 ? 128 : K, in_scale, in_shift, (const uint32_t *)drop_mask,
                    (uint32_t)ldmask, drop_scale, K, stat_partials, nullptr, nullptr, nullptr, nullptr, nullptr};
        return p2c_fwd_pp_launch(a, in_mode, s);
    }
    switch (in_mode) {
    case 0: return launch_fwd<0>(X, ldx, W, ldw, bias, Y, ldy, M, N, K, in_scale, in_shift, drop_mask, ldmask, drop_scale, stat_partials, s);
    case 1: return launch_fwd<1>(X, ldx, W, ldw, bias, Y, ldy, M, N, K, in_scale, in_shift, drop_mask, ldmask, drop_scale, stat_partials, s);
    case 2: return launch_fwd<2>(X, ldx, W, ldw, bias, Y, ldy, M, N, K, in_scale, in_shift, drop_mask, ldmask, drop_scale, stat_partials, s);
    default: return launch_fwd<3>(X, ldx, W, ldw, bias, Y, ldy, M, N, K, in_scale, in_shift, drop_mask, ldmask, drop_scale, stat_partials, s);
    }
}

// ---- forward of the LAST layer of a set-abstraction stack (BatchNorm'ed input, neighbourhoods of 64 rows): Y as p2c_linear_fwd_f32
// writes it, plus the per-half-neighbourhood extremes the max-pool needs (fwd_pp.hip, POOL) - see p2c_pool_select_f32 in bn.hip.
extern "C" int p2c_linear_fwd_pool_f32(const float *X, int ldx, const float *W, int ldw, const float *bias, float *Y, int ldy, int M, int N,
                                       int K, const float *in_scale, const float *in_shift, double *stat_partials, float *pool_max,
                                       float *pool_min, int32_t *pool_idx, void *stream)
{
    // Y == NULL: the pre-BatchNorm output is not stored (the statistics and the extremes are all the pooled layer's forward AND its
    // backward through p2c_linear_bwd_pool_alg_f32 need)
    if (!X || !W || !in_scale || !in_shift || !pool_max || !pool_min || !pool_idx || M <= 0) return P2C_EINVAL;
    if (!p2c_linear_fwd_pool_supported(M, N, K, 1, 64)) return P2C_EINVAL;
    P2C_REQ_ALIGNED(X, ldx);
    P2C_REQ_ALIGNED(W, ldw);
    P2C_REQ_ALIGNED(in_scale, 0);
    P2C_REQ_ALIGNED(in_shift, 0);
    FwdPPArgs a{X, ldx, W, ldw, bias, Y, ldy, M, N, K, in_scale, in_shift, nullptr, 0u, 1.0f, K, stat_partials, nullptr, nullptr,
                pool_max, pool_min, pool_idx};
    return p2c_fwd_pp_launch(a, 1, (hipStream_t)stream);
}

// ---- forward with a per-row-group additive term: Y[m,:] = act_in(X)[m,:] . W^T + gbias[m / rows_per_group, :] + bias.
// The layer's input is [X | one vector per group repeated over its rows] (FP3: the global feature repeated over the 128 points,
// pointnet_util.py:298-299, :312); the repeated part's product is computed once per group by the caller.  rows_per_group must
// be a multiple of 64 and divide M.  in_mode 0 or 1.
extern "C" int p2c_linear_fwd_gbias_f32(const float *X, int ldx, const float *W, int ldw, const float *bias, const float *gbias, int ldgb,
                                        int rows_per_group, float *Y, int ldy, int M, int N, int K, int in_mode, const float *in_scale,
                                        const float *in_shift, double *stat_partials, void *stream)
{
    if (!X || !W || !Y || !gbias || M <= 0 || N <= 0 || K <= 0 || in_mode < 0 || in_mode > 1 || rows_per_group <= 0 ||
        (rows_per_group % 64) || (M % rows_per_group) || tile_m() != 64)
        return P2C_EINVAL;
    if (in_mode == 1 && (!in_scale || !in_shift)) return P2C_EINVAL;
    if (K & 3) return P2C_EALIGN;
    P2C_REQ_ALIGNED(X, ldx);
    P2C_REQ_ALIGNED(W, ldw);
    hipStream_t s = (hipStream_t)stream;
    if (in_mode == 0) return launch_fwd<0>(X, ldx, W, ldw, bias, Y, ldy, M, N, K, in_scale, in_shift, nullptr, 0, 1.f, stat_partials, s, gbias, ldgb, rows_per_group);
    return launch_fwd<1>(X, ldx, W, ldw, bias, Y, ldy, M, N, K, in_scale, in_shift, nullptr, 0, 1.f, stat_partials, s, gbias, ldgb, rows_per_group);
}

// ---- forward of the layer that FOLLOWS a folded first layer (see bn.hip): A = relu(bn0(X0 W0^T + b0)) rebuilt from X0 [M,4]
extern "C" int p2c_linear_fwd_fold0_f32(const float *X0, int ldx0, const float *W0, const float *b0, const float *scale0, const float *shift0,
                                        int C0, const float *W, int ldw, const float *bias, float *Y, int ldy, int M, int N,
                                        double *stat_partials, void *stream)
{
    if (!X0 || !W0 || !scale0 || !shift0 || !W || !Y || M < 8192 || N <= 0 || N > 256 || (N & 3) || C0 != 64 || ldx0 != 4) return P2C_EINVAL;
    P2C_REQ_ALIGNED(X0, ldx0);
    P2C_REQ_ALIGNED(W, ldw);
    P2C_REQ_ALIGNED(W0, 0);
    FwdPPArgs a{X0, ldx0, W, ldw, bias, Y, ldy, M, N, C0, scale0, shift0, nullptr, 0u, 1.f, C0, stat_partials, W0, b0};
    return p2c_fwd_pp_launch(a, 4, (hipStream_t)stream);
}

// ---- backward data ----------------------------------------------------------------------------------
template <int GMODE>
static int launch_bwd_data(const float *dZ, int lddz, const float *Yfwd, int ldy, const float *coef, const float *W, int ldw, float *dX,
                           int lddx, int M, int N, int K, const uint8_t *out_mask, int ldmask, float out_mask_scale, const float *Yprev,
                           int ldyp, const float *prev_stat, double *bwd_partials, const int32_t *pool_arg, int pool_ns, hipStream_t s,
                           const float *spz = nullptr, int ldspz = 0, float sp_beta = 0.f, float sp_thr = 0.f)
{
    // layer: Y[M,N] = in[M,K] . W[N,K]^T ; here the GEMM is dX[M,K] = dY[M,N] . W[N,K]
    if (out_mask && ldmask < 0 && !p2c_drop_scale_representable(out_mask_scale)) return P2C_EINVAL;      // hashed dropout (ldmask -1): p in steps of 1/256
    OpGrad<GMODE> a{dZ, lddz, Yfwd, ldy, coef, N, pool_arg, pool_ns};
    OpPlain b{W, ldw};
    EpiBwdData e{dX, lddx, out_mask, ldmask, out_mask_scale, p2c_drop_threshold(out_mask_scale), Yprev, ldyp, prev_stat, bwd_partials, spz, ldspz, sp_beta, sp_thr};
    const int kps = (N + GK - 1) / GK * GK;
#define P2C_BDL_(TM_, TN_, SP_)                                                                                                    \
    hipLaunchKernelGGL((gemm_kernel<TM_, TN_, true, false, OpGrad<GMODE>, OpPlain, EpiBwdData, SP_>),                               \
                       dim3(p2c_cdiv(M, 64 * TM_), p2c_cdiv(K, 64 * TN_), 1), dim3(256), 0, s, a, b, e, M, K, N, kps)
#define P2C_BDL(TM_, TN_) do { if (gemm_split_t()) P2C_BDL_(TM_, TN_, true); else P2C_BDL_(TM_, TN_, false); } while (0)
    if (K > 64 && !narrow_tiles(M, K)) P2C_BDL(1, 2); else P2C_BDL(1, 1);
#undef P2C_BDL
#undef P2C_BDL_
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

extern "C" int p2c_linear_bwd_data_f32(const float *dZ, int lddz, const float *Yfwd, int ldy, int grad_mode, const float *coef,
                                       const float *W, int ldw, float *dX, int lddx, int M, int N, int K, const uint8_t *out_mask,
                                       int ldmask, float out_mask_scale, const float *Yprev, int ldyp, const float *prev_stat,
                                       double *bwd_partials, const int32_t *pool_arg, int pool_ns, void *stream)
{
    if (!dZ || !W || !dX || M <= 0 || N <= 0 || K <= 0 || grad_mode < 0 || grad_mode > 2) return P2C_EINVAL;
    if (grad_mode >= 1 && (!Yfwd || !coef)) return P2C_EINVAL;
    if (grad_mode == 2 && (!pool_arg || pool_ns <= 0 || ((uintptr_t)pool_arg & 15))) return P2C_EINVAL;
    if (bwd_partials && (!Yprev || !prev_stat)) return P2C_EINVAL;
    if ((N & 3) || (K & 3)) return P2C_EALIGN;
    P2C_REQ_ALIGNED(dZ, lddz);
    P2C_REQ_ALIGNED(W, ldw);
    if (grad_mode >= 1) { P2C_REQ_ALIGNED(Yfwd, ldy); P2C_REQ_ALIGNED(coef, 0); }
    hipStream_t s = (hipStream_t)stream;
#define P2C_BD(G_)                                                                                                                        \
    return launch_bwd_data<G_>(dZ, lddz, Yfwd, ldy, coef, W, ldw, dX, lddx, M, N, K, out_mask, ldmask, out_mask_scale, Yprev, ldyp, prev_stat, \
                               bwd_partials, pool_arg, pool_ns, s)
    if (grad_mode == 0) P2C_BD(0);
    if (grad_mode == 1) P2C_BD(1);
    P2C_BD(2);
#undef P2C_BD
}

// dX[M,K] = (dZ[M,N] . W[N,K]) * sigmoid(beta * Z[M,K])  (1 where beta*Z > threshold): the data gradient of a linear layer whose
// input is softplus(Z) (nn.Softplus(beta), IGR/network.py:58-59, :80-82), with the activation's derivative in the epilogue.
extern "C" int p2c_linear_bwd_data_sig_f32(const float *dZ, int lddz, const float *W, int ldw, const float *Z, int ldz, float beta, float threshold,
                                           float *dX, int lddx, int M, int N, int K, void *stream)
{
    if (!dZ || !W || !Z || !dX || M <= 0 || N <= 0 || K <= 0 || beta <= 0.f) return P2C_EINVAL;
    if ((N & 3) || (K & 3)) return P2C_EALIGN;
    P2C_REQ_ALIGNED(dZ, lddz);
    P2C_REQ_ALIGNED(W, ldw);
    return launch_bwd_data<0>(dZ, lddz, nullptr, 0, nullptr, W, ldw, dX, lddx, M, N, K, nullptr, 0, 1.f, nullptr, 0, nullptr, nullptr, nullptr, 0,
                              (hipStream_t)stream, Z, ldz, beta, threshold);
}

// ---- backward weight --------------------------------------------------------------------------------
template <int GMODE, int IMODE>
static int launch_bwd_weight(const float *dZ, int lddz, const float *Yfwd, int ldy, const float *coef, const float *X, int ldx,
                             const float *in_scale, const float *in_shift, const uint8_t *drop_mask, int ldmask, float drop_scale, float *dW,
                             int lddw, long long slot_stride, float *dbias, int M, int N, int K, const int32_t *pool_arg, int pool_ns,
                             hipStream_t s)
{
    // dW[N,K] += sum_m dY[m,N]^T act_in(X)[m,K]: GEMM with I=N (co), J=K (ci), reduction over the M rows
    OpGrad<GMODE> a{dZ, lddz, Yfwd, ldy, coef, N, pool_arg, pool_ns};
    OpActIn<IMODE> b{X, ldx, in_scale, in_shift, drop_mask, ldmask, drop_scale};
    EpiAtomic e{dW, lddw, dbias, slot_stride};
    const int ti = N > 64 ? p2c_cdiv(N, 128) : 1, tj = K > 64 ? p2c_cdiv(K, 128) : 1;
    const int ktiles = (M + GK - 1) / GK;
    // workgroups = tiles x splits.  Long reductions (the 262 k / 1 M-row layers): ~2048 workgroups of >= 8 k-tiles.  Short ones
    // (4 k - 16 k rows): the atomic epilogue (one 128 x 128 tile per workgroup) is no longer small against the main loop, so aim at
    // ~512 workgroups, but of >= 4 k-tiles only (measured on the SA3 / FP3 / FP2 shapes, tools/gemm_bench.py).
    const bool longk = ktiles >= 2048;
    int splits = (longk ? 2048 : 512) / (ti * tj);
    if (splits < 1) splits = 1;
    const int mink = longk ? 8 : 4;
    if (splits > (ktiles + mink - 1) / mink) splits = (ktiles + mink - 1) / mink;
    const int kps = (ktiles + splits - 1) / splits * GK;
    splits = (M + kps - 1) / kps;
    dim3 grid(ti, tj, (unsigned)splits);
#define P2C_BW_(TM_, TN_, SP_)                                                                                                            \
    hipLaunchKernelGGL((gemm_kernel<TM_, TN_, false, false, OpGrad<GMODE>, OpActIn<IMODE>, EpiAtomic, SP_>), grid, dim3(256), 0, s, a, b, e, N, K, \
                       M, kps)
#define P2C_BW(TM_, TN_) do { if (gemm_split_t()) P2C_BW_(TM_, TN_, true); else P2C_BW_(TM_, TN_, false); } while (0)
    if (N > 64 && K > 64) P2C_BW(2, 2);
    else if (N > 64) P2C_BW(2, 1);
    else if (K > 64) P2C_BW(1, 2);
    else P2C_BW(1, 1);
#undef P2C_BW
#undef P2C_BW_
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

extern "C" int p2c_linear_bwd_weight_f32(const float *dZ, int lddz, const float *Yfwd, int ldy, int grad_mode, const float *coef,
                                         const float *X, int ldx, int in_mode, const float *in_scale, const float *in_shift,
                                         const uint8_t *drop_mask, int ldmask, float drop_scale, float *dW, int lddw,
                                         long long dw_slot_stride, float *dbias, int M, int N, int K, const int32_t *pool_arg, int pool_ns,
                                         void *stream)
{
    if (!dZ || !X || !dW || M <= 0 || N <= 0 || K <= 0 || grad_mode < 0 || grad_mode > 2 || in_mode < 0 || in_mode > 3) return P2C_EINVAL;
    if (in_mode == 3) { if (!drop_mask || !p2c_drop_scale_representable(drop_scale)) return P2C_EINVAL; ldmask = (int)p2c_drop_threshold(drop_scale); }
    if (grad_mode >= 1 && (!Yfwd || !coef)) return P2C_EINVAL;
    if (grad_mode == 2 && (!pool_arg || pool_ns <= 0 || ((uintptr_t)pool_arg & 15))) return P2C_EINVAL;
    if (in_mode >= 1 && (!in_scale || !in_shift)) return P2C_EINVAL;
    if (in_mode == 2 && (!drop_mask || (ldmask & 3) || ((uintptr_t)drop_mask & 3))) return P2C_EINVAL;
    if ((N & 3) || (K & 3)) return P2C_EALIGN;
    P2C_REQ_ALIGNED(dZ, lddz);
    P2C_REQ_ALIGNED(X, ldx);
    if (grad_mode >= 1) { P2C_REQ_ALIGNED(Yfwd, ldy); P2C_REQ_ALIGNED(coef, 0); }
    if (in_mode >= 1) { P2C_REQ_ALIGNED(in_scale, 0); P2C_REQ_ALIGNED(in_shift, 0); }
    hipStream_t s = (hipStream_t)stream;
#define P2C_DISPATCH(G_, I_)                                                                                                              \
    return launch_bwd_weight<G_, I_>(dZ, lddz, Yfwd, ldy, coef, X, ldx, in_scale, in_shift, drop_mask, ldmask, drop_scale, dW, lddw,        \
                                     dw_slot_stride, dbias, M, N, K, pool_arg, pool_ns, s)
    if (grad_mode == 0) {
        if (in_mode == 0) P2C_DISPATCH(0, 0);
        if (in_mode == 1) P2C_DISPATCH(0, 1);
        if (in_mode == 2) P2C_DISPATCH(0, 2);
        P2C_DISPATCH(0, 3);
    }
    if (in_mode == 3) return P2C_EINVAL;      // dropout only ever precedes the BN-less head layer (grad_mode 0)
    if (grad_mode == 1) {
        if (in_mode == 0) P2C_DISPATCH(1, 0);
        if (in_mode == 1) P2C_DISPATCH(1, 1);
        P2C_DISPATCH(1, 2);
    }
    if (in_mode == 0) P2C_DISPATCH(2, 0);
    P2C_DISPATCH(2, 1);
#undef P2C_DISPATCH
}

// ---- both backward GEMMs of a small layer in one launch ----------------------------------------------
template <int GMODE, int IMODE>
static int launch_bwd_both(const float *dZ, int lddz, const float *Yfwd, int ldy, const float *coef, const int32_t *pool_arg, int pool_ns,
                           const float *X, int ldx, const float *in_scale, const float *in_shift, const float *W, int ldw, float *dX, int lddx,
                           const float *Yprev, int ldyp, const float *prev_stat, double *bwd_partials, float *dW, int lddw, int M, int N, int K,
                           hipStream_t s)
{
    // A: dX[M,K] = dY[M,N] . W[N,K]     (64 x 64 tiles)           B: dW[N,K] += dY[M,N]^T act_in(X)[M,K]   (128 x 128 tiles, split over M)
    OpGrad<GMODE> g{dZ, lddz, Yfwd, ldy, coef, N, pool_arg, pool_ns};
    OpPlain w{W, ldw};
    EpiBwdData eA{dX, lddx, nullptr, 0, 1.f, 0u, Yprev, ldyp, prev_stat, bwd_partials, nullptr, 0, 0.f, 0.f};
    OpActIn<IMODE> x{X, ldx, in_scale, in_shift, nullptr, 0, 1.f};
    EpiAtomic eB{dW, lddw, nullptr, 0};
    // dX tiles: 64 x 64, or 64 x 128 when that still leaves >= 256 of them (SA3's 512 -> 1024 layer: half the re-reads of the rebuilt dY;
    // same-box A/B 3.839 -> 3.815 ms per step, the four dual launches 0.252 -> 0.232 ms).  Round 6 also swept the other launch knobs of
    // this file on the step (tile width of forward / backward-data, split thresholds, workgroup targets and minimum k-tiles of the split
    // reductions: tools history): every alternative within +-0.02 ms of the defaults or slower.
    const bool wideA = (long long)p2c_cdiv(M, 64) * p2c_cdiv(K, 128) >= 256;
    const int nAx = p2c_cdiv(M, 64), nAy = p2c_cdiv(K, wideA ? 128 : 64), nA = nAx * nAy;
    const int kpsA = (N + GK - 1) / GK * GK;
    const int ti = p2c_cdiv(N, 128), tj = p2c_cdiv(K, 128);
    const int ktiles = (M + GK - 1) / GK;
    int splits = 512 / (ti * tj);
    if (splits < 1) splits = 1;
    if (splits > (ktiles + 3) / 4) splits = (ktiles + 3) / 4;
    const int kpsB = (ktiles + splits - 1) / splits * GK;
    splits = (M + kpsB - 1) / kpsB;
    if (gemm_split_t() && wideA)
        hipLaunchKernelGGL((gemm_dual_kernel<2, 2, 2, OpGrad<GMODE>, OpActIn<IMODE>, true>), dim3(nA + ti * tj * splits), dim3(256), 0, s, g, w, eA, M, K, N,
                           kpsA, nAx, nA, g, x, eB, N, K, M, kpsB, ti, tj);
    else if (gemm_split_t())
        hipLaunchKernelGGL((gemm_dual_kernel<1, 2, 2, OpGrad<GMODE>, OpActIn<IMODE>, true>), dim3(nA + ti * tj * splits), dim3(256), 0, s, g, w, eA, M, K, N,
                           kpsA, nAx, nA, g, x, eB, N, K, M, kpsB, ti, tj);
    else
        hipLaunchKernelGGL((gemm_dual_kernel<1, 2, 2, OpGrad<GMODE>, OpActIn<IMODE>, false>), dim3(nA + ti * tj * splits), dim3(256), 0, s, g, w, eA, M, K, N,
                           kpsA, nAx, nA, g, x, eB, N, K, M, kpsB, ti, tj);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// dX and dW of one layer in ONE launch (see gemm_dual_kernel): for layers of a few thousand rows, where neither GEMM fills the chip.
// Same operands as p2c_linear_bwd_data_f32 + p2c_linear_bwd_weight_f32 with grad_mode 1 or 2, in_mode 0 or 1, no dropout masks;
// N, K > 64 and multiples of 4; dW (zeroed by the caller) is accumulated with atomics, one copy.
extern "C" int p2c_linear_bwd_both_f32(const float *dZ, int lddz, const float *Yfwd, int ldy, int grad_mode, const float *coef,
                                       const int32_t *pool_arg, int pool_ns, const float *X, int ldx, int in_mode, const float *in_scale,
                                       const float *in_shift, const float *W, int ldw, float *dX, int lddx, const float *Yprev, int ldyp,
                                       const float *prev_stat, double *bwd_partials, float *dW, int lddw, int M, int N, int K, void *stream)
{
    if (!dZ || !Yfwd || !coef || !X || !W || !dX || !dW || M <= 0 || N <= 64 || K <= 64 || grad_mode < 1 || grad_mode > 2 || in_mode < 0 || in_mode > 1)
        return P2C_EINVAL;
    if (grad_mode == 2 && (!pool_arg || pool_ns <= 0 || ((uintptr_t)pool_arg & 15))) return P2C_EINVAL;
    if (in_mode == 1 && (!in_scale || !in_shift)) return P2C_EINVAL;
    if (bwd_partials && (!Yprev || !prev_stat)) return P2C_EINVAL;
    if ((N & 3) || (K & 3)) return P2C_EALIGN;
    P2C_REQ_ALIGNED(dZ, lddz);
    P2C_REQ_ALIGNED(W, ldw);
    P2C_REQ_ALIGNED(X, ldx);
    P2C_REQ_ALIGNED(Yfwd, ldy);
    P2C_REQ_ALIGNED(coef, 0);
    hipStream_t s = (hipStream_t)stream;
#define P2C_BB(G_, I_)                                                                                                                      \
    return launch_bwd_both<G_, I_>(dZ, lddz, Yfwd, ldy, coef, pool_arg, pool_ns, X, ldx, in_scale, in_shift, W, ldw, dX, lddx, Yprev, ldyp, \
                                   prev_stat, bwd_partials, dW, lddw, M, N, K, s)
    if (grad_mode == 1) { if (in_mode == 0) P2C_BB(1, 0); P2C_BB(1, 1); }
    if (in_mode == 0) P2C_BB(2, 0);
    P2C_BB(2, 1);
#undef P2C_BB
}
