// fwd_pp3.hip -- the persistent forward of fwd_pp.hip on the bf16 matrix pipe AT FP32 ACCURACY ("bf16x3 split").
//
// Why: v_mfma_f32_32x32x2_f32 issues once per 64 cycles per SIMD (157 TFLOP/s dense) and the persistent fp32 kernels sit at 0.56-0.6 of
// that with nothing left to remove.  v_mfma_f32_32x32x16_bf16 does 8x the k-depth in HALF the cycles.  Every
// fp32 operand is split once, while its tile is staged, into three bf16 pieces  x = hi + mid + lo  (round-to-nearest each: 24+ significant
// bits) and a 32x32x16 block is accumulated from the six products  mm, hl, lh, hm, mh, hh  (smallest first) in the fp32 accumulator:
// 6 x 32 = 192 matrix-pipe cycles instead of 8 x 64 = 512 for the same block, and the dropped terms (ml, lm, ll) are <= 2^-24 of the
// product - tools/ubench/split_bf16.hip measures 2.7e-7 max / 2.2e-8 mean of sum|a_k b_k| against 2.4e-7 / 2.0e-8 for the fp32 MFMA.
// The parity tests are the proof that this is an fp32-accurate path, not a reduced-precision one (they run unchanged).
//
// Structure: as fwd_pp.hip - one 512-thread workgroup per CU, W in LDS once (as three bf16 planes), two halves of 4 waves one phase
// apart, everything outside the MFMA phase wave-local - with these differences:
//  * a half's tile is 32*WR rows with WR x WC = 4 waves, WC = columns/32 of the workgroup (4: 32 rows x 128 columns, 2: 64 x 64): one
//    32x32 block per wave, so three bf16 planes of W and of both tiles fit the 160 KB (156,672 B at K = 128);
//  * LDS rows are k-contiguous bf16 with a 16-byte pad: one ds_read_b128 per lane is the whole 8-deep operand of one MFMA
//    (lane (i, h) holds k = 16q + 8h .. +7 of row i); per k-step 6 reads feed 6 MFMAs;
//  * the split costs ~5.5 VALU per element at staging time (v_cvt_pk_bf16_f32 + shift/and + subtract, twice, + one more cvt), next
//    to the act_in transform that already runs there;
//  * the two halves run IN THE SAME PHASE (no one-phase offset; P2C_FWD3_LOCKSTEP=0 restores it for A/B runs).  The phase trace of the
//    offset form (tools/fused_trace.py --fwd, 128 -> 128, cycles per 32-row half tile): MFMA phase 2.1 k, wave-local phase 3.9 k, of which
//    the first ~45 instructions take 2.0 k - the wave whose partner on the SIMD streams MFMAs is granted about one issue slot per MFMA,
//    for the 32-cycle bf16 instruction as for the 64-cycle fp32 one - so the period was the SUM of the two phases (8.1 k per 64 rows).
//    In lockstep both waves of a SIMD share the matrix pipe (3.5 k for the later one) and then run their vector work side by side,
//    hiding each other's latencies: 6.8 k per 64 rows, the forward launches 5-13 % shorter, the step 4.34 -> 4.27 ms on one box.
#include "common.h"
#include "fwd_pp.h"
#include <stdlib.h>

#define P2C_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));

// two fp32 -> one dword of two bf16 (round to nearest even; gfx950 has the instruction, v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t p2c_pk_bf16(float a, float b)
{
    const v2f v = {a, b};
    const bf16x2 r = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ float p2c_bf16_lo(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float p2c_bf16_hi(uint32_t p) { return __builtin_bit_cast(float, p & 0xFFFF0000u); }

// x (4 consecutive k of one row) -> its three bf16 pieces, each 4 x bf16 = 8 bytes: x = hi + mid + lo up to 2^-26 |x|
__device__ __forceinline__ void p2c_split3(const v4f &x, v2u &hi, v2u &mid, v2u &lo)
{
    hi.x = p2c_pk_bf16(x.x, x.y);
    hi.y = p2c_pk_bf16(x.z, x.w);
    const float r0 = x.x - p2c_bf16_lo(hi.x), r1 = x.y - p2c_bf16_hi(hi.x), r2 = x.z - p2c_bf16_lo(hi.y), r3 = x.w - p2c_bf16_hi(hi.y);
    mid.x = p2c_pk_bf16(r0, r1);
    mid.y = p2c_pk_bf16(r2, r3);
    lo.x = p2c_pk_bf16(r0 - p2c_bf16_lo(mid.x), r1 - p2c_bf16_hi(mid.x));
    lo.y = p2c_pk_bf16(r2 - p2c_bf16_lo(mid.y), r3 - p2c_bf16_hi(mid.y));
}

// Register budget: as fwd_pp.hip (<= 160 VGPRs so that two waves per SIMD leave room for the sampling kernel of the forked stream).
template <int KP, int WC, int MODE, int EX, bool POOL = false>
__global__ void __launch_bounds__(512, 2) fwd_pp3_kernel(FwdPPArgs a)
{
    constexpr int WR = 4 / WC, BMH = 32 * WR, BN = 32 * WC;          // rows of a half's tile, columns of the workgroup
    constexpr int LDB = 2 * KP + 16;                                 // bytes of one LDS row of one piece
    constexpr int RPW = BMH / 4;                                     // tile rows a wave stages (8 or 16)
    constexpr int PL = RPW * LDB;                                    // bytes of one piece plane of a wave's region
    constexpr int REG0 = 3 * PL > 4096 ? 3 * PL : 4096;              // the region also parks the wave's 32 x 32 fp32 output block
    // region stride: rows 8..15 of a 16-row read group must land 128 B (mod 256) after rows 0..7 when a wave owns 8 rows
    constexpr int REG = RPW == 8 ? ((REG0 + 127) / 256) * 256 + 128 : ((REG0 + 255) / 256) * 256;
    constexpr int HB = 4 * REG;                                      // bytes of a half's buffer
    constexpr int WPL = BN * LDB;                                    // bytes of one piece plane of W
    constexpr int QW = KP / 4;                                       // float4 per tile row
    constexpr int UPW = RPW * QW / 64;                               // float4 units a lane stages per tile
    constexpr int NQ = KP / 16;                                      // k-steps
    static_assert(UPW >= 1 && (RPW * QW) % 64 == 0, "");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
    unsigned char *Wsb = smem3;                                      // [3][BN][LDB]
    unsigned char *Hb = Wsb + 3 * WPL;                               // [2 halves][4 waves][REG]
    float *We = reinterpret_cast<float *>(Hb + 2 * HB);              // [BN][4]      (EX)
    float *Xe = We + (EX ? BN * 4 : 0);                              // [2 halves][2 tile parities][BMH][4]   (EX)
    float *red = reinterpret_cast<float *>(Hb);                      // [2 halves][WR][2][BN]: aliases the tile buffers, used after the loop

    const int half = threadIdx.x >> 8, tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5, wm = wave / WC, wn = wave % WC;
    const int j0 = blockIdx.y * BN;
    const int ntiles = (a.M + BMH - 1) / BMH;
    const int nk = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int niter = (nk + 1) / 2;
    unsigned char *hb = Hb + half * HB;
    unsigned char *mine = hb + wave * REG;                           // this wave's region: tile rows RPW*wave .. +RPW-1
    float *xe2 = Xe + half * 2 * BMH * 4;
    auto tile_of = [&](int k) { return (int)blockIdx.x + k * (int)gridDim.x; };

    // ---- W (and the EX trailing columns) -> LDS as three bf16 planes, zero-padded to [BN][KP]
    for (int u = threadIdx.x; u < BN * QW; u += 512) {
        const int n = u / QW, kq = u % QW;
        v4f v = {0.f, 0.f, 0.f, 0.f};
        if (j0 + n < a.N && 4 * kq < a.K) v = *reinterpret_cast<const v4f *>(a.w + (size_t)(j0 + n) * a.ldw + 4 * kq);
        v2u h, m, l;
        p2c_split3(v, h, m, l);
        unsigned char *d = Wsb + n * LDB + 8 * kq;
        *reinterpret_cast<v2u *>(d) = h;
        *reinterpret_cast<v2u *>(d + WPL) = m;
        *reinterpret_cast<v2u *>(d + 2 * WPL) = l;
    }
    if (EX) {
        for (int n = threadIdx.x; n < BN; n += 512) {
            v4f v = {0.f, 0.f, 0.f, 0.f};
            if (j0 + n < a.N) v = *reinterpret_cast<const v4f *>(a.w + (size_t)(j0 + n) * a.ldw + a.K);
            *reinterpret_cast<v4f *>(&We[n * 4]) = v;
        }
    }

    // ---- per-lane constants of the staging map: unit u = lane + 64 i -> row RPW*wave + u / QW, float4 column u % QW
    const int kq = lane % QW;                                // the same for every i (64 % QW == 0)
    const bool kok = 4 * kq < a.K;
    v4f isc = {1.f, 1.f, 1.f, 1.f}, ish = {0.f, 0.f, 0.f, 0.f};
    if (MODE >= 1 && kok) {
        isc = *reinterpret_cast<const v4f *>(a.in_scale + 4 * kq);
        ish = *reinterpret_cast<const v4f *>(a.in_shift + 4 * kq);
    }
    uint32_t slo = 0, shi = 0;
    if (MODE == 3) { slo = a.seed[0]; shi = a.seed[1]; }
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
    const float *xlane = a.x + (size_t)(RPW * wave + lane / QW) * a.ldx + xcoloff;
    const size_t xstep = (size_t)(64 / QW) * a.ldx;
    auto gload = [&](int t) {                                // raw rows of tile t (clamped; masked when staged)
        const int m0 = t * BMH + RPW * wave;
        if (t * BMH + BMH <= a.M) {                          // uniform: only the last tile can be ragged
            const float *p = xlane + (size_t)t * BMH * a.ldx;
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
        if (EX && lane < RPW) rxe = *reinterpret_cast<const v4f *>(a.x + (size_t)min(m0 + lane, a.M - 1) * a.ldx + a.K);
    };
    auto stage = [&](int t, int par) {                       // act_in, zero outside [M, K], split, 8-byte LDS writes into this wave's rows
        const int m0 = t * BMH + RPW * wave;
        const bool full = t * BMH + BMH <= a.M && a.K >= KP; // uniform: nothing to mask
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
            if (MODE >= 1) {         // mul + add, NOT an fma: the backward kernels rebuild act_in(x) and its ReLU mask with the same two roundings
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
            v2u h, md, l;
            p2c_split3(v, h, md, l);
            unsigned char *d = mine + rl * LDB + 8 * kq;
            *reinterpret_cast<v2u *>(d) = h;
            *reinterpret_cast<v2u *>(d + PL) = md;
            *reinterpret_cast<v2u *>(d + 2 * PL) = l;
        }
        if (EX && lane < RPW) {
            const bool ok = m0 + lane < a.M;
            *reinterpret_cast<v4f *>(&xe2[par * BMH * 4 + (RPW * wave + lane) * 4]) = ok ? rxe : v4f{0.f, 0.f, 0.f, 0.f};
        }
    };

    // ---- per-lane constants of the MFMA / epilogue maps
    const int arow = wm * 32 + l31;                          // this lane's A row within the half's tile
    const unsigned char *Ap = hb + (arow / RPW) * REG + (arow % RPW) * LDB + 16 * lh;
    const unsigned char *Bp = Wsb + (wn * 32 + l31) * LDB + 16 * lh;
    const int col = j0 + wn * 32 + l31;
    const float bias = (a.bias && col < a.N) ? a.bias[col] : 0.f;
    v4f we = {0.f, 0.f, 0.f, 0.f};
    v2f s1v = {0.f, 0.f}, s2v = {0.f, 0.f};
    f32x16 acc;

    // ---- prologue: first tile of this half -> LDS, second -> registers
    gload(tile_of(half < nk ? half : 0));
    stage(half < nk ? tile_of(half) : ntiles, 0);            // tile index past the end: all rows masked to zero
    gload(tile_of(half + 2 < nk ? half + 2 : 0));
    __syncthreads();
    if (EX) we = *reinterpret_cast<const v4f *>(&We[(wn * 32 + l31) * 4]);
    if (half == 1 && !a.lockstep) P2C_LDS_BARRIER();         // run one phase behind half 0 (lockstep: both halves in the same phase)
    for (int it = 0; it < niter; ++it) {
        const int k = 2 * it + half;
        const bool valid = k < nk;                           // uniform within the half
        const int m0 = tile_of(valid ? k : 0) * BMH;
        // ================= MFMA phase =================
        P2C_TR(0);
        if (valid) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            bf16x8 ah, am, al, bh, bm, bl, ah2, am2, al2, bh2, bm2, bl2;
#define P2C_LD3(H_, M_, L_, P_, PLS_, Q_)                                              \
    do {                                                                               \
        H_ = *reinterpret_cast<const bf16x8 *>((P_) + 32 * (Q_));                      \
        M_ = *reinterpret_cast<const bf16x8 *>((P_) + (PLS_) + 32 * (Q_));            \
        L_ = *reinterpret_cast<const bf16x8 *>((P_) + 2 * (PLS_) + 32 * (Q_));        \
    } while (0)
#define P2C_MM6(AH_, AM_, AL_, BH_, BM_, BL_)                                         \
    do {                                                                               \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AM_, BM_, acc, 0, 0, 0);        \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH_, BL_, acc, 0, 0, 0);        \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AL_, BH_, acc, 0, 0, 0);        \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH_, BM_, acc, 0, 0, 0);        \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AM_, BH_, acc, 0, 0, 0);        \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH_, BH_, acc, 0, 0, 0);        \
    } while (0)
            P2C_LD3(ah, am, al, Ap, PL, 0);
            P2C_LD3(bh, bm, bl, Bp, WPL, 0);
#pragma unroll
            for (int q = 0; q < NQ; q += 2) {
                // the reads of k-step q+1 are issued before the MFMAs of k-step q (pinned: the scheduler would sink them)
                if (q + 1 < NQ) {
                    P2C_LD3(ah2, am2, al2, Ap, PL, q + 1);
                    P2C_LD3(bh2, bm2, bl2, Bp, WPL, q + 1);
                }
                __builtin_amdgcn_sched_barrier(0);
                P2C_MM6(ah, am, al, bh, bm, bl);
                __builtin_amdgcn_sched_barrier(0);
                if (q + 1 < NQ) {
                    if (q + 2 < NQ) {
                        P2C_LD3(ah, am, al, Ap, PL, q + 2);
                        P2C_LD3(bh, bm, bl, Bp, WPL, q + 2);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    P2C_MM6(ah2, am2, al2, bh2, bm2, bl2);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#undef P2C_LD3
#undef P2C_MM6
        }
        P2C_TR(1);
        P2C_LDS_BARRIER();
        P2C_TR(2);
        // ================= wave-local phase (the other half is in its MFMA phase) =================
        __builtin_amdgcn_s_setprio(1);
        asm volatile("" ::"v"(rx[UPW - 1]));                 // wait for the prefetch while only loads are outstanding (see fwd_pp.hip)
        if (EX) asm volatile("" ::"v"(rxe));
        if (valid) {
            float *out = reinterpret_cast<float *>(mine);    // [32 rows][32 columns] of this wave's block
            if (EX) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rf = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const v4f xv = *reinterpret_cast<const v4f *>(&xe2[(it & 1) * BMH * 4 + (wm * 32 + rf) * 4]);
                    acc[r] += (xv.x * we.x + xv.y * we.y) + (xv.z * we.z + xv.w * we.w);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const v2f v = {acc[r], acc[r + 1]};
                s1v += v;
                s2v = __builtin_elementwise_fma(v, v, s2v);
                const v2f ov = v + v2f{bias, bias};
                const int rf = (r & 3) + 8 * (r >> 2) + 4 * lh;        // r even: rows rf and rf+1
                out[rf * 32 + l31] = ov.x;
                out[(rf + 1) * 32 + l31] = ov.y;
            }
            P2C_TR(3);
            __builtin_amdgcn_wave_barrier();
            // whole 128-byte row pieces, 16 bytes per lane: all reads first, then the stores
            const int c4 = lane & 7, rf0 = lane >> 3;        // 8 float4 per block row, 8 rows per instruction
            const int ocol = j0 + wn * 32 + 4 * c4;
            v4f o[4];
            if (a.y != nullptr) {                            // (POOL: Y may be absent - its backward needs no Y, csrc/bwd_pool.hip)
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = *reinterpret_cast<const v4f *>(&out[(rf0 + 8 * i) * 32 + 4 * c4]);
            }
            P2C_TR(4);
            if (a.y != nullptr && ocol < a.N) {
                float *yp = a.y + (size_t)(m0 + wm * 32 + rf0) * a.ldy + ocol;
                if (m0 + BMH <= a.M) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) *reinterpret_cast<v4f *>(yp + (size_t)(8 * i) * a.ldy) = o[i];
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (m0 + wm * 32 + rf0 + 8 * i < a.M) *reinterpret_cast<v4f *>(yp + (size_t)(8 * i) * a.ldy) = o[i];
                }
            }
            if (POOL) {
                // per column the largest and the smallest pre-BN value of this 32-row block and their rows (see fwd_pp.hip): lane (c, hh)
                // scans rows 16 hh .. +15 of column c in ascending order (strict compares keep the first), the upper half-wave hands its
                // result to the lower one, which keeps its own on ties (lower rows)
                float vmax = -INFINITY, vmin = INFINITY;
                int imax = 0, imin = 0;
#pragma unroll 8
                for (int r = 0; r < 16; ++r) {
                    const float v = out[(16 * lh + r) * 32 + l31];
                    if (v > vmax) { vmax = v; imax = 16 * lh + r; }
                    if (v < vmin) { vmin = v; imin = 16 * lh + r; }
                }
                const float omax = __shfl_xor(vmax, 32), omin = __shfl_xor(vmin, 32);
                const int oimax = __shfl_xor(imax, 32), oimin = __shfl_xor(imin, 32);
                if (lh == 0) {
                    if (omax > vmax) { vmax = omax; imax = oimax; }
                    if (omin < vmin) { vmin = omin; imin = oimin; }
                    if (col < a.N) {
                        const int t32 = (m0 >> 5) + wm;                        // 32-row block index; its neighbourhood of 64 = t32 >> 1
                        const size_t po = (size_t)t32 * a.N + col;
                        const int rb = (t32 & 1) * 32;
                        a.pool_max[po] = vmax;
                        a.pool_min[po] = vmin;
                        a.pool_idx[po] = (rb + imax) | ((rb + imin) << 16);
                    }
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
    if (half == 0 && !a.lockstep) P2C_LDS_BARRIER();
    // ---- BatchNorm sums: every (half, row block) parks its column sums in its own LDS slot, summed in a fixed order, then one fp64
    //      atomic per column into this workgroup's slot row
    if (a.partials) {
        __syncthreads();                                     // `red` aliases the tile buffers
        const float u1 = s1v.x + s1v.y, u2 = s2v.x + s2v.y;
        const float t1 = u1 + __shfl_xor(u1, 32), t2 = u2 + __shfl_xor(u2, 32);
        if (lh == 0) {
            float *r = red + (half * WR + wm) * 2 * BN;
            r[wn * 32 + l31] = t1;
            r[BN + wn * 32 + l31] = t2;
        }
        __syncthreads();
        if (threadIdx.x < BN && j0 + threadIdx.x < a.N) {
            const int c = threadIdx.x;
            float q1 = 0.f, q2 = 0.f;
#pragma unroll
            for (int s = 0; s < 2 * WR; ++s) { q1 += red[s * 2 * BN + c]; q2 += red[s * 2 * BN + BN + c]; }
            double *o = a.partials + (size_t)(blockIdx.x % P2C_STAT_SLOTS) * 2 * a.N;
            atomicAdd(&o[j0 + c], (double)q1);
            atomicAdd(&o[a.N + j0 + c], (double)q2);
        }
    }
}

template <int KP, int WC, int MODE, int EX, bool POOL = false>
static int launch_pp3(const FwdPPArgs &a, hipStream_t s)
{
    constexpr int WR = 4 / WC, BMH = 32 * WR, BN = 32 * WC, LDB = 2 * KP + 16, RPW = BMH / 4, PL = RPW * LDB;
    constexpr int REG0 = 3 * PL > 4096 ? 3 * PL : 4096;
    constexpr int REG = RPW == 8 ? ((REG0 + 127) / 256) * 256 + 128 : ((REG0 + 255) / 256) * 256;
    const size_t lds = (size_t)3 * BN * LDB + 2 * 4 * REG + (EX ? (BN * 4 + 4 * BMH * 4) * sizeof(float) : 0);
    static_assert(3 * BN * LDB + 8 * REG + (EX ? (BN * 4 + 4 * BMH * 4) * 4 : 0) <= 160 * 1024, "LDS");
    const int gy = (a.N + BN - 1) / BN;
    const int ntiles = (a.M + BMH - 1) / BMH;
    int gx = 256 / gy;
    if (gx > (ntiles + 1) / 2) gx = (ntiles + 1) / 2;        // at least one tile per half
    if (gx < 1) gx = 1;
    (void)hipFuncSetAttribute((const void *)fwd_pp3_kernel<KP, WC, MODE, EX, POOL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((fwd_pp3_kernel<KP, WC, MODE, EX, POOL>), dim3(gx, gy), dim3(512), lds, s, a);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// same contract as p2c_fwd_pp_launch (fwd_pp.hip), which forwards here unless the fp32-MFMA path was asked for
int p2c_fwd_pp3_launch(const FwdPPArgs &a_in, int in_mode, hipStream_t s)
{
    FwdPPArgs a = a_in;
    a.lockstep = 1;          // both halves of the workgroup in the same phase: -5...13 % per launch against the one-phase offset
    const int K = a.K;          // EX columns already split off by the caller
    if (a.pool_max) {
        if (!a.pool_min || !a.pool_idx || !p2c_linear_fwd_pool_supported(a.M, a.N, K, in_mode, 64)) return P2C_EINVAL;
        if (K == 64) return launch_pp3<64, 4, 1, 0, true>(a, s);
        return launch_pp3<128, 4, 1, 0, true>(a, s);
    }
    const bool ex = a.Kfull == 132 && in_mode != 3;
    const int wc = a.N <= 64 ? 2 : 4;
#define P2C_PP(KP_, WC_, MODE_, EX_) return launch_pp3<KP_, WC_, MODE_, EX_>(a, s)
#define P2C_PPK(WC_, MODE_)                          \
    do {                                             \
        if (ex) P2C_PP(128, WC_, MODE_, 4);          \
        if (K <= 32) P2C_PP(32, WC_, MODE_, 0);      \
        if (K <= 64) P2C_PP(64, WC_, MODE_, 0);      \
        P2C_PP(128, WC_, MODE_, 0);                  \
    } while (0)
    if (in_mode == 0) { if (wc == 2) P2C_PPK(2, 0); P2C_PPK(4, 0); }
    if (in_mode == 1) { if (wc == 2) P2C_PPK(2, 1); P2C_PPK(4, 1); }
    if (in_mode == 4) {                  // folded first layer of 64 channels
        if (K != 64) return P2C_EINVAL;
        if (wc == 2) P2C_PP(64, 2, 4, 0);
        P2C_PP(64, 4, 4, 0);
    }
    if (in_mode == 3) {
        if (wc == 2) { if (K <= 64) P2C_PP(64, 2, 3, 0); P2C_PP(128, 2, 3, 0); }
        if (K <= 64) P2C_PP(64, 4, 3, 0);
        P2C_PP(128, 4, 3, 0);
    }
#undef P2C_PPK
#undef P2C_PP
    return P2C_EINVAL;
}
