// gemm_big.hip -- the large products of the sketch branch's implicit decoder (IGR/network.py:20-92: eight 512-wide layers over
// B x K x P = 262 144 rows, evaluated forward, backward and backward-of-backward by the with-sketch step, train_Point2Cyl.py:608-648) on
// v_mfma_f32_32x32x16_bf16 with the bf16 x 3 operand split of gemm.hip (six products, fp32-accurate), gfx950 only.
//
// Why a second GEMM: gemm.hip's 64 x 128 tile re-splits its W tile in EVERY workgroup (4096 times per layer) and its A tile four times,
// and on this chip VALU work does not hide behind the matrix pipe (a wave next to an MFMA stream gets about one issue slot per MFMA,
// DESIGN 3 "issue model"): the split arithmetic is paid in full - 1.25 ms for 137 GFLOP, 26 % of the split ceiling.  Here
//   * W is split ONCE per call by a 5 us pre-pass into three bf16 planes, tiled and swizzled exactly as the LDS image wants them
//     ([k-tile][plane][row][16 k], the 16-byte half h of row r stored at h ^ ((r >> 3) & 1)): a B tile is three contiguous 8 KB copies,
//     no arithmetic, no padding, conflict-free ds_read_b128 fragment reads;
//   * the tile is 128 x 256 (wave tile 64 x 128, 8 accumulators): 48 MFMAs per 18 fragment reads, and the A split (the only VALU work
//     left) is amortised over 256 columns instead of 128;
//   * the k-tile is ONE MFMA k-step (16) with two LDS stages of 36.9 KB: two workgroups per CU (240 registers, 73.7 KB each), one barrier
//     per tile; the other workgroup's MFMAs run while this one splits / stores / waits.
// Measured: forward 1.25 -> 0.75 ms, data gradient with the softplus derivative 1.29 -> 0.88 ms at 262 144 x 512 x 512.
// C[M, J] = A[M, Kd] . B[J, Kd]^T with B = W (forward: J = N, Kd = K) or B = W^T (data gradient: J = K, Kd = N).
#include "common.h"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 b_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b_bf16x2 __attribute__((ext_vector_type(2)));
typedef float b_v2f __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));       // a native vector: HIP's uint4 struct is copied by memcpy and stays in scratch

constexpr int BM = 128, BN = 256, BK = 16;                                // one MFMA k-step per tile: two LDS stages fit twice per CU
constexpr int A_PLANE = BM * BK, B_PLANE = BN * BK;                       // bf16 elements
constexpr int STAGE = 3 * A_PLANE + 3 * B_PLANE;                          // elements per stage (36 864 bytes)
constexpr int LDS_OPS = 2 * STAGE * 2;                                    // two stages: 73 728 bytes
constexpr int LDO = BN + 4;                                               // epilogue staging: 64 rows x LDO floats
constexpr int LDS_OUT = 64 * LDO * 4;
constexpr int LDS_BYTES = LDS_OPS > LDS_OUT ? LDS_OPS : LDS_OUT;

__device__ __forceinline__ uint32_t pk_bf16(float a, float b)
{
    const b_v2f v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, b_bf16x2));
}
__device__ __forceinline__ float bf_lo(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf_hi(uint32_t p) { return __builtin_bit_cast(float, p & 0xFFFF0000u); }
// x = hi + mid + lo, each rounded to nearest (the same split as gemm.hip's g_split_pair: the two kernels give the same planes)
__device__ __forceinline__ void split_pair(float a, float b, uint32_t &h, uint32_t &m, uint32_t &l)
{
    h = pk_bf16(a, b);
    const float ra = a - bf_lo(h), rb = b - bf_hi(h);
    m = pk_bf16(ra, rb);
    l = pk_bf16(ra - bf_lo(m), rb - bf_hi(m));
}

// Plane images are [rows][16 k] bf16, 32-byte rows; the 16-byte half h of row r is stored at h ^ ((r >> 3) & 1): the 16 lanes of a
// ds_read_b128 group (rows r .. r + 15, same half) then cover 16 distinct bank quads.
__device__ __forceinline__ int sw_half(int row, int h) { return row * BK + ((h ^ ((row >> 3) & 1)) << 3); }

// ---- pre-pass: W [R, C] fp32 -> planes [KT][3][Jpad][16] (bf16, swizzled).  transposed = 0: B[j][k] = W[j][k] (j < R, k < C);
// transposed = 1: B[j][k] = W[k][j] (j < C, k < R).  Rows / k beyond the matrix are zero.  One thread per (k-tile, row, 8-k half).
__global__ void __launch_bounds__(256) big_split_kernel(const float *__restrict__ W, int ldw, int R, int C, int transposed, uint16_t *__restrict__ planes,
                                                        int Jpad, int KT)
{
    const long long u = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)KT * Jpad * 2;
    if (u >= total) return;
    const int c = (int)(u & 1), j = (int)((u >> 1) % Jpad), kt = (int)((u >> 1) / Jpad);
    const int J = transposed ? C : R, Kd = transposed ? R : C;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = kt * BK + c * 8 + e;
        v[e] = (j < J && k < Kd) ? (transposed ? W[(size_t)k * ldw + j] : W[(size_t)j * ldw + k]) : 0.f;
    }
    uint32_t h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split_pair(v[2 * e], v[2 * e + 1], h[e], m[e], l[e]);
    uint16_t *o = planes + (size_t)kt * 3 * Jpad * BK + sw_half(j, c);
    const size_t ps = (size_t)Jpad * BK;
    *reinterpret_cast<uint4 *>(o) = make_uint4(h[0], h[1], h[2], h[3]);
    *reinterpret_cast<uint4 *>(o + ps) = make_uint4(m[0], m[1], m[2], m[3]);
    *reinterpret_cast<uint4 *>(o + 2 * ps) = make_uint4(l[0], l[1], l[2], l[3]);
}

struct BigEpi {
    float *out; int ldo;
    const float *bias;                              // EPI 0: + bias[col] (may be NULL)
    const float *spz; int ldspz; float beta, thr;   // EPI 1: * sigmoid(beta * Z) (1 where beta * Z > thr); spz NULL: plain product.  EPI 0 with beta > 0: softplus(. , beta, thr) of the result
    const float *add; int ldadd;                    // both: + add[row, col] after the transform above (may alias out); NULL: nothing
};

// EPI 0: forward (bias), EPI 1: data gradient (softplus derivative)
template <int EPI>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) big_gemm_kernel(const float *__restrict__ A, int lda, const uint16_t *__restrict__ planes, int Jpad, int KT,
                                                       BigEpi epi, int M, int J, int Kd, int nbx, int nby)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint16_t *Ap = reinterpret_cast<uint16_t *>(smem_raw);          // stage s: A planes at Ap + s * STAGE, B planes behind them
    float *so = reinterpret_cast<float *>(smem_raw);

    // XCD-aware tile order: consecutive workgroup ids go round the 8 XCDs, so ids i, i + 8, i + 16, ... share an L2.  The nby column blocks
    // of one row block get ids 8 apart: the A rows they share are fetched from HBM once.
    const int id = blockIdx.x, grp = id / (8 * nby), within = id - grp * (8 * nby);
    const int by = within >> 3, bx = grp * 8 + (within & 7);
    if (bx >= nbx) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int i0 = bx * BM, j0 = by * BN;
    const int il = lane & 31, kl = lane >> 5;

    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // A tile 128 x 16 fp32 = 512 float4: 2 per thread (unit u = tid + 256 i: row u >> 2, k quad u & 3).  B tile: 3 planes x 8 KB, contiguous in
    // the pre-split image: 2 x 16 bytes per thread and plane.  Two LDS stages: the loads of tile kt + 1 are issued before the MFMAs of tile kt,
    // split / stored into the other stage after them, one barrier per tile.
    float4 ra[2];
    u32x4 rb[6];
    const float *arow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) arow[i] = A + (size_t)min(i0 + ((tid + 256 * i) >> 2), M - 1) * lda + (tid & 3) * 4;
    const int akq = (tid & 3) * 4;
#define P2C_BIG_LOAD(KT_)                                                                                                      \
    do {                                                                                                                       \
        const int kt__ = (KT_);                                                                                                \
        const bool kin__ = kt__ * BK + akq < Kd;                                                                               \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                        \
            ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);                                                                           \
            if (kin__) ra[i] = *reinterpret_cast<const float4 *>(arow[i] + kt__ * BK);                                         \
        }                                                                                                                      \
        const u32x4 *src__ = reinterpret_cast<const u32x4 *>(planes + ((size_t)kt__ * 3 * Jpad + j0) * BK) + tid;              \
        _Pragma("unroll") for (int q = 0; q < 6; ++q) rb[q] = src__[(size_t)(q >> 1) * (Jpad * BK / 8) + 256 * (q & 1)];       \
    } while (0)
#define P2C_BIG_STORE(ST_)                                                                                                     \
    do {                                                                                                                       \
        uint16_t *As__ = Ap + (ST_) * STAGE, *Bs__ = As__ + 3 * A_PLANE;                                                       \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                        \
            const int u = tid + 256 * i, row = u >> 2, kq = u & 3;                                                             \
            uint32_t h0, m0, l0, h1, m1, l1;                                                                                   \
            split_pair(ra[i].x, ra[i].y, h0, m0, l0);                                                                          \
            split_pair(ra[i].z, ra[i].w, h1, m1, l1);                                                                          \
            uint16_t *o = As__ + sw_half(row, kq >> 1) + (kq & 1) * 4;                                                         \
            *reinterpret_cast<uint2 *>(o) = make_uint2(h0, h1);                                                                \
            *reinterpret_cast<uint2 *>(o + A_PLANE) = make_uint2(m0, m1);                                                      \
            *reinterpret_cast<uint2 *>(o + 2 * A_PLANE) = make_uint2(l0, l1);                                                  \
        }                                                                                                                      \
        _Pragma("unroll") for (int q = 0; q < 6; ++q) reinterpret_cast<u32x4 *>(Bs__ + (q >> 1) * B_PLANE)[tid + 256 * (q & 1)] = rb[q]; \
    } while (0)
    P2C_BIG_LOAD(0);
    P2C_BIG_STORE(0);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const bool more = kt + 1 < KT;
        P2C_BIG_LOAD(more ? kt + 1 : kt);                           // in flight under the MFMAs below (unconditional: a load under a branch
                                                                    // makes the 32 prefetch registers loop-carried copies, 32 v_mov per tile)
        __builtin_amdgcn_s_setprio(0);
        {
            const uint16_t *As = Ap + (kt & 1) * STAGE, *Bs = As + 3 * A_PLANE;
            b_bf16x8 a[2][3], b[4][3];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int row = wm * 64 + t * 32 + il;
#pragma unroll
                for (int p = 0; p < 3; ++p) a[t][p] = *reinterpret_cast<const b_bf16x8 *>(As + p * A_PLANE + sw_half(row, kl));
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row = wn * 128 + t * 32 + il;
#pragma unroll
                for (int p = 0; p < 3; ++p) b[t][p] = *reinterpret_cast<const b_bf16x8 *>(Bs + p * B_PLANE + sw_half(row, kl));
            }
            // smallest products first (as gemm.hip); consecutive MFMAs go to different accumulators
#define P2C_B3(PA_, PB_)                                                                                             \
    _Pragma("unroll") for (int ta = 0; ta < 2; ++ta) _Pragma("unroll") for (int tb = 0; tb < 4; ++tb)                 \
        acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ta][PA_], b[tb][PB_], acc[ta][tb], 0, 0, 0)
            P2C_B3(1, 1); P2C_B3(0, 2); P2C_B3(2, 0); P2C_B3(0, 1); P2C_B3(1, 0); P2C_B3(0, 0);
#undef P2C_B3
        }
        // The other workgroup's wave on this SIMD is (ideally) in its MFMA phase now, and an MFMA stream leaves its partner about one issue
        // slot per MFMA: the split / store phase asks for priority so that its ~200 instructions do not take four MFMA phases.
        __builtin_amdgcn_s_setprio(3);
        P2C_BIG_STORE((kt + 1) & 1);                                // after the last tile: a harmless re-store into the idle stage
        __syncthreads();
    }
    __builtin_amdgcn_s_setprio(0);
#undef P2C_BIG_LOAD
#undef P2C_BIG_STORE

    // ---- epilogue: 64 rows at a time through LDS (the operand images are dead), whole rows out as 16-byte pieces per lane
    const int col_l = lane & 31, rquad = lane >> 5;
    const int c4 = (tid & 63) * 4, col = j0 + c4;                   // this thread's 4 columns of every row it stores
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (EPI == 0 && epi.bias && col < J) {
#pragma unroll
        for (int c = 0; c < 4; ++c) bv[c] = epi.bias[col + c];
    }
    const bool has_z = EPI == 1 && epi.spz != nullptr;
#pragma unroll
    for (int ta = 0; ta < 2; ++ta) {
        // the pre-activations this pass multiplies by are requested BEFORE the tile goes through LDS, so their latency sits under the
        // barrier pair and the 64 staging stores instead of in front of every row's arithmetic
        f32x4 zr[16];
        if (has_z) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rl = (tid >> 6) + 4 * i;
                const int row = i0 + (rl >> 5) * 64 + ta * 32 + (rl & 31);
                zr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (row < M && col < J) zr[i] = *reinterpret_cast<const f32x4 *>(epi.spz + (size_t)row * epi.ldspz + col);
            }
        }
        __syncthreads();
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                so[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * rquad) * LDO + wn * 128 + tb * 32 + col_l] = acc[ta][tb][r];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int rl = (tid >> 6) + 4 * i;
            const int row = i0 + (rl >> 5) * 64 + ta * 32 + (rl & 31);      // staged row rl = wm * 32 + r  ->  tile row wm * 64 + ta * 32 + r
            if (row >= M || col >= J) continue;
            f32x4 v4 = *reinterpret_cast<const f32x4 *>(&so[rl * LDO + c4]);
            if (EPI == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) v4[c] += bv[c];
            } else if (has_z) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float bz = zr[i][c] * epi.beta;
                    if (!(bz > epi.thr)) {                                  // linear region of softplus: derivative 1
                        const float e = __expf(-fabsf(bz)), sg = 1.f / (1.f + e);
                        v4[c] *= bz >= 0.f ? sg : 1.f - sg;
                    }
                }
            }
            if (epi.add) {                  // the second product of a two-operand layer, or the other gradient this tensor receives: summed here instead
                const f32x4 ad = *reinterpret_cast<const f32x4 *>(epi.add + (size_t)row * epi.ldadd + col);      // of by a 3-stream add pass
#pragma unroll
                for (int c = 0; c < 4; ++c) v4[c] += ad[c];
            }
            if (EPI == 0 && epi.beta > 0.f) {       // inference: the layer's activation instead of its pre-activation (softplus.hip's MODE 0, term for term)
                const float inv_beta = 1.f / epi.beta;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float z = v4[c], bz = z * epi.beta;
                    const float e = __expf(-fabsf(bz));
                    const float l1p = e < 0.0078125f ? e * (1.f - e * (0.5f - e * 0.33333334f)) : __logf(1.f + e);
                    v4[c] = bz > epi.thr ? z : (fmaxf(bz, 0.f) + l1p) * inv_beta;
                }
            }
            *reinterpret_cast<f32x4 *>(epi.out + (size_t)row * epi.ldo + col) = v4;
        }
    }
}

static int big_launch(int epi_kind, const float *A, int lda, const float *W, int ldw, int R, int C, int transposed, const BigEpi &epi, int M, void *ws,
                      hipStream_t s)
{
    const int J = transposed ? C : R, Kd = transposed ? R : C;
    const int Jpad = p2c_cdiv(J, BN) * BN, KT = p2c_cdiv(Kd, BK);
    uint16_t *planes = reinterpret_cast<uint16_t *>(ws);
    hipLaunchKernelGGL(big_split_kernel, dim3(p2c_cdiv((long long)KT * Jpad * 2, 256)), dim3(256), 0, s, W, ldw, R, C, transposed, planes, Jpad, KT);
    const int nbx = p2c_cdiv(M, BM), nby = Jpad / BN;
    const int grid = p2c_cdiv(nbx, 8) * 8 * nby;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)big_gemm_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute((const void *)big_gemm_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
            return P2C_EINVAL;
        attr_done = true;
    }
    if (epi_kind == 0)
        hipLaunchKernelGGL(big_gemm_kernel<0>, dim3(grid), dim3(256), LDS_BYTES, s, A, lda, planes, Jpad, KT, epi, M, J, Kd, nbx, nby);
    else
        hipLaunchKernelGGL(big_gemm_kernel<1>, dim3(grid), dim3(256), LDS_BYTES, s, A, lda, planes, Jpad, KT, epi, M, J, Kd, nbx, nby);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

}  // namespace

// Shapes the big-tile kernels take (the decoder's: M >= 16 384 rows, both widths >= 128); everything else stays on gemm.hip.
extern "C" int p2c_linear_big_supported(int M, int N, int K)
{
    return (M >= 16384 && N >= 128 && K >= 128 && (N & 3) == 0 && (K & 3) == 0) ? 1 : 0;
}

// bytes of the split image of W [N, K] either way round (the larger of the two orientations)
extern "C" size_t p2c_linear_big_ws_bytes(int N, int K)
{
    const size_t a = (size_t)p2c_cdiv(N, BN) * BN * p2c_cdiv(K, BK) * BK, b = (size_t)p2c_cdiv(K, BN) * BN * p2c_cdiv(N, BK) * BK;
    return 3 * 2 * (a > b ? a : b);
}

// Y[M,N] = X[M,K] . W[N,K]^T + bias  (p2c_linear_fwd_f32 with in_mode 0 and no statistics, for the shapes above)
extern "C" int p2c_linear_fwd_big_f32(const float *X, int ldx, const float *W, int ldw, const float *bias, float *Y, int ldy, int M, int N, int K,
                                      void *ws, void *stream)
{
    if (!X || !W || !Y || !ws || !p2c_linear_big_supported(M, N, K)) return P2C_EINVAL;
    if ((ldx & 3) || (ldy & 3) || (((uintptr_t)X | (uintptr_t)Y | (uintptr_t)ws) & 15)) return P2C_EALIGN;
    BigEpi e{Y, ldy, bias, nullptr, 0, 0.f, 0.f, nullptr, 0};
    return big_launch(0, X, ldx, W, ldw, N, K, 0, e, M, ws, (hipStream_t)stream);
}

// Y = softplus(X . W^T + bias [+ add], beta, threshold): the layer's ACTIVATION, for inference (nobody differentiates: the pre-activation is not kept)
extern "C" int p2c_linear_fwd_big_sp_f32(const float *X, int ldx, const float *W, int ldw, const float *bias, const float *add, int ldadd, float beta,
                                         float threshold, float *Y, int ldy, int M, int N, int K, void *ws, void *stream)
{
    if (!X || !W || !Y || !ws || !(beta > 0.f) || !p2c_linear_big_supported(M, N, K)) return P2C_EINVAL;
    if ((ldx & 3) || (ldy & 3) || (((uintptr_t)X | (uintptr_t)Y | (uintptr_t)ws) & 15)) return P2C_EALIGN;
    if (add && ((ldadd & 3) || ((uintptr_t)add & 15))) return P2C_EALIGN;
    BigEpi e{Y, ldy, bias, nullptr, 0, beta, threshold, add, add ? ldadd : 0};
    return big_launch(0, X, ldx, W, ldw, N, K, 0, e, M, ws, (hipStream_t)stream);
}

// ... + add[M,N] (ldadd % 4 == 0, 16-byte aligned; may be Y itself: accumulate in place).  A layer fed by two operands (the decoder's skip
// layer: [h | input] . W^T = h . Wa^T + input . Wb^T) runs as two products, the second adding the first's result in its epilogue.
extern "C" int p2c_linear_fwd_big_add_f32(const float *X, int ldx, const float *W, int ldw, const float *bias, const float *add, int ldadd, float *Y,
                                          int ldy, int M, int N, int K, void *ws, void *stream)
{
    if (!X || !W || !Y || !ws || !add || !p2c_linear_big_supported(M, N, K)) return P2C_EINVAL;
    if ((ldx & 3) || (ldy & 3) || (ldadd & 3) || (((uintptr_t)X | (uintptr_t)Y | (uintptr_t)ws | (uintptr_t)add) & 15)) return P2C_EALIGN;
    BigEpi e{Y, ldy, bias, nullptr, 0, 0.f, 0.f, add, ldadd};
    return big_launch(0, X, ldx, W, ldw, N, K, 0, e, M, ws, (hipStream_t)stream);
}

// dX[M,K] = (dZ[M,N] . W[N,K]) (* sigmoid(beta * Z[M,K]) when Z is given: p2c_linear_bwd_data_sig_f32; Z NULL: p2c_linear_bwd_data_f32 in its
// plain gradient mode)
extern "C" int p2c_linear_bwd_data_big_f32(const float *dZ, int lddz, const float *W, int ldw, const float *Z, int ldz, float beta, float threshold,
                                           float *dX, int lddx, int M, int N, int K, void *ws, void *stream)
{
    if (!dZ || !W || !dX || !ws || !p2c_linear_big_supported(M, N, K) || (Z && beta <= 0.f)) return P2C_EINVAL;
    if ((lddz & 3) || (lddx & 3) || (Z && (ldz & 3)) || (((uintptr_t)dZ | (uintptr_t)dX | (uintptr_t)ws | (uintptr_t)Z) & 15)) return P2C_EALIGN;
    BigEpi e{dX, lddx, nullptr, Z, ldz, beta, threshold, nullptr, 0};
    return big_launch(1, dZ, lddz, W, ldw, N, K, 1, e, M, ws, (hipStream_t)stream);
}

// ... + add[M,K] after the (optional) softplus derivative: dX = (dZ . W) * sigmoid(beta Z) + add.  The double backward of the decoder gives
// every pre-activation two gradients (through the forward pass and through the first backward pass); the second arrives here.
extern "C" int p2c_linear_bwd_data_big_add_f32(const float *dZ, int lddz, const float *W, int ldw, const float *Z, int ldz, float beta, float threshold,
                                               const float *add, int ldadd, float *dX, int lddx, int M, int N, int K, void *ws, void *stream)
{
    if (!dZ || !W || !dX || !ws || !add || !p2c_linear_big_supported(M, N, K) || (Z && beta <= 0.f)) return P2C_EINVAL;
    if ((lddz & 3) || (lddx & 3) || (ldadd & 3) || (Z && (ldz & 3)) ||
        (((uintptr_t)dZ | (uintptr_t)dX | (uintptr_t)ws | (uintptr_t)Z | (uintptr_t)add) & 15))
        return P2C_EALIGN;
    BigEpi e{dX, lddx, nullptr, Z, ldz, beta, threshold, add, ldadd};
    return big_launch(1, dZ, lddz, W, ldw, N, K, 1, e, M, ws, (hipStream_t)stream);
}
