// gemm.hip -- the shared per-point MLP as fp32 MFMA GEMMs with fused prologues / epilogues (gfx950).
//
// One LDS-tiled kernel template, C[I,J] = sum_k A(i,k) * B(k,j), built on
// v_mfma_f32_32x32x2_f32 (exact fp32, bitwise an fmaf chain; 64 cycles per 32x32x2 block, so one wave
// per SIMD already saturates the matrix pipe and the LDS feed is 1 dword per operand per MFMA).
// The three GEMMs of a 1x1-conv layer are instances of it; what differs is how an operand ELEMENT is
// produced while its tile is loaded (the "operand" structs) and what the epilogue does:
//
//   forward      Y  = act_in(X) . W^T + b        A = act_in(X)[m,k]       B = W[n,k]       epi: +bias, store, BN partial sums
//   backward-data dX = dY . W                     A = dY(dZ,Y)[m,co]       B = W[co,ci]     epi: (*dropout mask), store
//   backward-wgt  dW = dY^T . act_in(X)           A = dY(dZ,Y)[m,co] (k=m) B = act_in(X)[m,ci] (k=m)   epi: atomicAdd (split over m)
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
// tile t (register prefetch).
#include "common.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GK 32   // k-tile

// ------------------------------------------------------------------------------------------------
// Operands: row-major [R, C] matrices with the channel dim contiguous; load4(r, c) yields 4 consecutive
// channels of row r AFTER the element transform, zero outside [R, C].
// ------------------------------------------------------------------------------------------------
struct OpPlain {
    const float *p; int ld; bool vec;
    __device__ __forceinline__ float4 load4(long long r, int c, long long R, int C) const
    {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r >= R || c >= C) return v;
        const float *q = p + r * ld + c;
        if (vec && c + 3 < C) return *reinterpret_cast<const float4 *>(q);
        v.x = q[0];
        if (c + 1 < C) v.y = q[1];
        if (c + 2 < C) v.z = q[2];
        if (c + 3 < C) v.w = q[3];
        return v;
    }
};

__device__ __forceinline__ float4 p2c_ld4(const float *q, int c, int C, bool vec)
{
    if (vec && c + 3 < C) return *reinterpret_cast<const float4 *>(q + c);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    v.x = q[c];
    if (c + 1 < C) v.y = q[c + 1];
    if (c + 2 < C) v.z = q[c + 2];
    if (c + 3 < C) v.w = q[c + 3];
    return v;
}

// act_in(X): mode 0 identity, 1 relu(scale*x+shift), 2 same * dropout mask * dscale
struct OpActIn {
    const float *p; int ld; bool vec;
    int mode; const float *scale; const float *shift; bool pvec;
    const uint8_t *mask; int ldmask; float dscale;
    __device__ __forceinline__ float4 load4(long long r, int c, long long R, int C) const
    {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r >= R || c >= C) return v;
        v = p2c_ld4(p + r * ld, c, C, vec);
        if (mode == 0) return v;
        const float4 s = p2c_ld4(scale, c, C, pvec), t = p2c_ld4(shift, c, C, pvec);
        v.x = fmaxf(s.x * v.x + t.x, 0.f);
        v.y = fmaxf(s.y * v.y + t.y, 0.f);
        v.z = fmaxf(s.z * v.z + t.z, 0.f);
        v.w = fmaxf(s.w * v.w + t.w, 0.f);
        if (mode == 2) {
            const uint8_t *m = mask + r * ldmask + c;
            v.x = m[0] ? v.x * dscale : 0.f;
            v.y = (c + 1 < C && m[1]) ? v.y * dscale : 0.f;
            v.z = (c + 2 < C && m[2]) ? v.z * dscale : 0.f;
            v.w = (c + 3 < C && m[3]) ? v.w * dscale : 0.f;
        }
        if (c + 1 >= C) v.y = 0.f;
        if (c + 2 >= C) v.z = 0.f;
        if (c + 3 >= C) v.w = 0.f;
        return v;
    }
};

// dY rebuilt from the upstream gradient dZ (w.r.t. relu(bn(Y))) and the saved pre-BN Y:
//   mode 0: dY = dZ;   mode 1: dY = gs*(dZ*[scale*Y+shift > 0]) + q*Y + p,  coef = [scale|shift|gs|q|p] x C
struct OpGrad {
    const float *dz; int lddz; bool vec;
    const float *y; int ldy; bool yvec;
    int mode; const float *coef; int Cc; bool pvec;
    __device__ __forceinline__ float4 load4(long long r, int c, long long R, int C) const
    {
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r >= R || c >= C) return g;
        g = p2c_ld4(dz + r * lddz, c, C, vec);
        if (mode == 0) return g;
        const float4 yy = p2c_ld4(y + r * ldy, c, C, yvec);
        const float4 s = p2c_ld4(coef, c, C, pvec), t = p2c_ld4(coef + Cc, c, C, pvec);
        const float4 gs = p2c_ld4(coef + 2 * Cc, c, C, pvec), q = p2c_ld4(coef + 3 * Cc, c, C, pvec),
                     pp = p2c_ld4(coef + 4 * Cc, c, C, pvec);
        float4 o;
        o.x = gs.x * ((s.x * yy.x + t.x > 0.f) ? g.x : 0.f) + q.x * yy.x + pp.x;
        o.y = gs.y * ((s.y * yy.y + t.y > 0.f) ? g.y : 0.f) + q.y * yy.y + pp.y;
        o.z = gs.z * ((s.z * yy.z + t.z > 0.f) ? g.z : 0.f) + q.z * yy.z + pp.z;
        o.w = gs.w * ((s.w * yy.w + t.w > 0.f) ? g.w : 0.f) + q.w * yy.w + pp.w;
        if (c + 1 >= C) o.y = 0.f;
        if (c + 2 >= C) o.z = 0.f;
        if (c + 3 >= C) o.w = 0.f;
        return o;
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
    static __device__ __forceinline__ void load(const Op &op, long long o0, long long Olim, long long k0, long long Klim, float4 (&regs)[UNITS])
    {
#pragma unroll
        for (int i = 0; i < UNITS; ++i) {
            const int u = threadIdx.x + 256 * i;
            if (KCONTIG) {
                const int nk = u >> 3, kq = u & 7;                     // 8 float4 per 32-wide k row
                regs[i] = op.load4(o0 + nk, (int)(k0 + kq * 4), Olim, (int)Klim);
            } else {
                const int kr = u / (ROWS / 4), c4 = u % (ROWS / 4);
                regs[i] = op.load4(k0 + kr, (int)(o0 + c4 * 4), Klim, (int)Olim);
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
// Epilogues.  C/D fragment of mfma 32x32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
// ------------------------------------------------------------------------------------------------
struct EpiFwd {
    float *Y; int ldy; const float *bias; float *partials;   // partials[tile][2][N] or NULL
};
struct EpiBwdData {
    float *dX; int lddx; const uint8_t *mask; int ldmask; float mscale;
};
struct EpiAtomic {
    float *dW; int lddw; float *dbias;
};

template <int TM, int TN, bool AK, bool BK_, class OpA, class OpB, class Epi>
__global__ void __launch_bounds__(256) gemm_kernel(OpA opA, OpB opB, Epi epi, long long I, long long J, long long K, long long k_per_split)
{
    constexpr int BM = 64 * TM, BN = 64 * TN;
    using TA = Tile<BM, AK>;
    using TB = Tile<BN, BK_>;
    __shared__ __attribute__((aligned(16))) float smem[GK * TA::LD + GK * TB::LD + 2 * BN + BM];
    float *As = smem, *Bs = smem + GK * TA::LD;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const long long i0 = (long long)blockIdx.x * BM, j0 = (long long)blockIdx.y * BN;
    const long long kbeg = (long long)blockIdx.z * k_per_split;
    const long long kend = min(K, kbeg + k_per_split);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float4 ra[TA::UNITS], rb[TB::UNITS];
    float dbias_acc = 0.f;
    if (kbeg < kend) {
        TA::load(opA, i0, I, kbeg, kend, ra);
        TB::load(opB, j0, J, kbeg, kend, rb);
    }
    for (long long k0 = kbeg; k0 < kend; k0 += GK) {
        __syncthreads();              // previous tile fully consumed
        TA::store(As, ra);
        TB::store(Bs, rb);
        __syncthreads();
        if (k0 + GK < kend) {         // prefetch the next tile while the matrix pipe works
            TA::load(opA, i0, I, k0 + GK, kend, ra);
            TB::load(opB, j0, J, k0 + GK, kend, rb);
        }
        if constexpr (std::is_same<Epi, EpiAtomic>::value) {
            if (epi.dbias && blockIdx.y == 0 && tid < BM) {
                float s = 0.f;
#pragma unroll 8
                for (int kk = 0; kk < GK; ++kk) s += As[kk * TA::LD + tid];
                dbias_acc += s;
            }
        }
        const int kl = lane >> 5, il = lane & 31;
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
    if constexpr (std::is_same<Epi, EpiFwd>::value) {
        float *sstat = smem + GK * TA::LD + GK * TB::LD;      // [2][BN] per-wm partial, combined below
        __syncthreads();
        if (tid < 2 * BN) sstat[tid] = 0.f;
        __syncthreads();
#pragma unroll
        for (int tb = 0; tb < TN; ++tb) {
            const long long col = j0 + wn * (TN * 32) + tb * 32 + col_l;
            const float bv = (epi.bias && col < J) ? epi.bias[col] : 0.f;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long row = i0 + wm * (TM * 32) + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * rquad;
                    const float v = acc[ta][tb][r];
                    if (row < I && col < J) {
                        epi.Y[row * epi.ldy + col] = v + bv;
                        s1 += v;
                        s2 += v * v;
                    }
                }
            if (epi.partials) {
                s1 += __shfl_xor(s1, 32);
                s2 += __shfl_xor(s2, 32);
                if (rquad == 0) {
                    atomicAdd(&sstat[wn * (TN * 32) + tb * 32 + col_l], s1);          // LDS atomics: 2 waves (wm) per column
                    atomicAdd(&sstat[BN + wn * (TN * 32) + tb * 32 + col_l], s2);
                }
            }
        }
        if (epi.partials) {
            __syncthreads();
            if (tid < BN && j0 + tid < J) {
                float *o = epi.partials + (size_t)blockIdx.x * 2 * J;
                o[j0 + tid] = sstat[tid];
                o[J + j0 + tid] = sstat[BN + tid];
            }
        }
    } else if constexpr (std::is_same<Epi, EpiBwdData>::value) {
#pragma unroll
        for (int tb = 0; tb < TN; ++tb) {
            const long long col = j0 + wn * (TN * 32) + tb * 32 + col_l;
#pragma unroll
            for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long row = i0 + wm * (TM * 32) + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * rquad;
                    if (row < I && col < J) {
                        float v = acc[ta][tb][r];
                        if (epi.mask) v = epi.mask[row * epi.ldmask + col] ? v * epi.mscale : 0.f;
                        epi.dX[row * epi.lddx + col] = v;
                    }
                }
        }
    } else {
#pragma unroll
        for (int tb = 0; tb < TN; ++tb) {
            const long long col = j0 + wn * (TN * 32) + tb * 32 + col_l;
#pragma unroll
            for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long row = i0 + wm * (TM * 32) + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * rquad;
                    if (row < I && col < J) atomicAdd(&epi.dW[row * epi.lddw + col], acc[ta][tb][r]);
                }
        }
        if (epi.dbias && blockIdx.y == 0 && tid < BM && i0 + tid < I) atomicAdd(&epi.dbias[i0 + tid], dbias_acc);
    }
}

static inline bool al16(const void *p) { return ((uintptr_t)p & 15) == 0; }

extern "C" int p2c_linear_stat_tiles(int M) { return (M + P2C_STAT_TILE_M - 1) / P2C_STAT_TILE_M; }

extern "C" int p2c_linear_fwd_f32(const float *X, int ldx, const float *W, int ldw, const float *bias, float *Y, int ldy, int M, int N,
                                  int K, int in_mode, const float *in_scale, const float *in_shift, const uint8_t *drop_mask,
                                  int ldmask, float drop_scale, float *stat_partials, void *stream)
{
    if (!X || !W || !Y || M <= 0 || N <= 0 || K <= 0 || in_mode < 0 || in_mode > 2) return P2C_EINVAL;
    if (in_mode >= 1 && (!in_scale || !in_shift)) return P2C_EINVAL;
    if (in_mode == 2 && !drop_mask) return P2C_EINVAL;
    OpActIn a{X, ldx, al16(X) && ldx % 4 == 0, in_mode, in_scale, in_shift, al16(in_scale) && al16(in_shift), drop_mask, ldmask, drop_scale};
    OpPlain b{W, ldw, al16(W) && ldw % 4 == 0};
    EpiFwd e{Y, ldy, bias, stat_partials};
    hipStream_t s = (hipStream_t)stream;
    const long long kps = ((long long)K + GK - 1) / GK * GK;
    if (N > 64) {
        dim3 grid(p2c_cdiv(M, 128), p2c_cdiv(N, 128), 1);
        hipLaunchKernelGGL((gemm_kernel<2, 2, true, true, OpActIn, OpPlain, EpiFwd>), grid, dim3(256), 0, s, a, b, e, (long long)M,
                           (long long)N, (long long)K, kps);
    } else {
        dim3 grid(p2c_cdiv(M, 128), 1, 1);
        hipLaunchKernelGGL((gemm_kernel<2, 1, true, true, OpActIn, OpPlain, EpiFwd>), grid, dim3(256), 0, s, a, b, e, (long long)M,
                           (long long)N, (long long)K, kps);
    }
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

extern "C" int p2c_linear_bwd_data_f32(const float *dZ, int lddz, const float *Yfwd, int ldy, int grad_mode, const float *coef,
                                       const float *W, int ldw, float *dX, int lddx, int M, int N, int K, const uint8_t *out_mask,
                                       int ldmask, float out_mask_scale, void *stream)
{
    // layer: Y[M,N] = in[M,K] . W[N,K]^T ; here the GEMM is dX[M,K] = dY[M,N] . W[N,K]
    if (!dZ || !W || !dX || M <= 0 || N <= 0 || K <= 0 || grad_mode < 0 || grad_mode > 1) return P2C_EINVAL;
    if (grad_mode == 1 && (!Yfwd || !coef)) return P2C_EINVAL;
    OpGrad a{dZ, lddz, al16(dZ) && lddz % 4 == 0, Yfwd, ldy, al16(Yfwd) && ldy % 4 == 0, grad_mode, coef, N, al16(coef) && N % 4 == 0};
    OpPlain b{W, ldw, al16(W) && ldw % 4 == 0};
    EpiBwdData e{dX, lddx, out_mask, ldmask, out_mask_scale};
    hipStream_t s = (hipStream_t)stream;
    const long long kps = ((long long)N + GK - 1) / GK * GK;
    if (K > 64) {
        dim3 grid(p2c_cdiv(M, 128), p2c_cdiv(K, 128), 1);
        hipLaunchKernelGGL((gemm_kernel<2, 2, true, false, OpGrad, OpPlain, EpiBwdData>), grid, dim3(256), 0, s, a, b, e, (long long)M,
                           (long long)K, (long long)N, kps);
    } else {
        dim3 grid(p2c_cdiv(M, 128), 1, 1);
        hipLaunchKernelGGL((gemm_kernel<2, 1, true, false, OpGrad, OpPlain, EpiBwdData>), grid, dim3(256), 0, s, a, b, e, (long long)M,
                           (long long)K, (long long)N, kps);
    }
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

extern "C" int p2c_linear_bwd_weight_f32(const float *dZ, int lddz, const float *Yfwd, int ldy, int grad_mode, const float *coef,
                                         const float *X, int ldx, int in_mode, const float *in_scale, const float *in_shift,
                                         const uint8_t *drop_mask, int ldmask, float drop_scale, float *dW, int lddw, float *dbias,
                                         int M, int N, int K, void *stream)
{
    // dW[N,K] += sum_m dY[m,N]^T act_in(X)[m,K]: GEMM with I=N (co), J=K (ci), reduction over the M rows
    if (!dZ || !X || !dW || M <= 0 || N <= 0 || K <= 0) return P2C_EINVAL;
    if (grad_mode == 1 && (!Yfwd || !coef)) return P2C_EINVAL;
    if (in_mode >= 1 && (!in_scale || !in_shift)) return P2C_EINVAL;
    OpGrad a{dZ, lddz, al16(dZ) && lddz % 4 == 0, Yfwd, ldy, al16(Yfwd) && ldy % 4 == 0, grad_mode, coef, N, al16(coef) && N % 4 == 0};
    OpActIn b{X, ldx, al16(X) && ldx % 4 == 0, in_mode, in_scale, in_shift, al16(in_scale) && al16(in_shift), drop_mask, ldmask, drop_scale};
    EpiAtomic e{dW, lddw, dbias};
    hipStream_t s = (hipStream_t)stream;
    const int ti = N > 64 ? p2c_cdiv(N, 128) : 1, tj = K > 64 ? p2c_cdiv(K, 128) : 1;
    long long ktiles = ((long long)M + GK - 1) / GK;
    long long splits = 2048 / ((long long)ti * tj);
    if (splits < 1) splits = 1;
    if (splits > (ktiles + 7) / 8) splits = (ktiles + 7) / 8;
    const long long kps = (ktiles + splits - 1) / splits * GK;
    splits = ((long long)M + kps - 1) / kps;
    dim3 grid(ti, tj, (unsigned)splits);
#define P2C_BW(TM_, TN_)                                                                                                              \
    hipLaunchKernelGGL((gemm_kernel<TM_, TN_, false, false, OpGrad, OpActIn, EpiAtomic>), grid, dim3(256), 0, s, a, b, e, (long long)N, \
                       (long long)K, (long long)M, kps)
    if (N > 64 && K > 64) P2C_BW(2, 2);
    else if (N > 64) P2C_BW(2, 1);
    else if (K > 64) P2C_BW(1, 2);
    else P2C_BW(1, 1);
#undef P2C_BW
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}
