// heads.hip -- backward of a NARROW linear layer on many rows in one pass: the per-point prediction heads (the reference's
// fc2 convolutions on the 128-channel point features, models/pointnet_extrusion.py:56-61 / :94-101 -> 19 outputs, padded to 20) at
// B x N = 262,144 rows.  The generic kernels run it as two GEMMs (gemm.hip: dW = dZ^T . act(X), then dX = dZ . W with the sums of the
// BatchNorm below in its epilogue): X is read by the first, its pre-activation again by the second - 163 us per step for 2.7 GFLOP.
// Here a workgroup takes 64-row tiles and, per tile, reads dZ (64 x Co) and the pre-BatchNorm input Y (64 x 128) ONCE:
//   x    = dropout(relu(scale * y + shift))           (the layer's input as the forward rebuilt it: same two roundings, same hashed mask)
//   dW  += dZ^T . x                                    (accumulated in registers over the workgroup's tiles; per-XCD copy at the end)
//   dX   = (dZ . W) * keep * dscale                    (stored)
//   s1  += relu' * dX,  s2 += relu' * dX * xhat        (the sums the BatchNorm below needs: fp64 slot rows)
//   db  += sum_rows dZ
// fp32 matrix instructions (v_mfma_f32_32x32x2_f32): Co <= 32 makes both products a small fraction of the kernel, the kernel is bound by
// the 289 MB it moves (Y, dZ in; dX out).  Algorithmic bytes per row: 4 * (Co + 2 * 128).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define HB_CI 128
#define HB_BM 64
#define HB_LDY (HB_CI + 4)
#define HB_LDZ 33

struct HeadsBwdArgs {
    const float *dz; int lddz;          // [M, Co]
    const float *y; int ldy;            // [M, 128] pre-BatchNorm input of the layer
    const float *stat;                  // [4, 128]: scale | shift | mean | invstd of that BatchNorm
    const uint32_t *seed; uint32_t thr; float dscale;     // hashed dropout on the layer's input (seed NULL: none)
    const float *w; int ldw;            // [Co, 128]
    float *dx; int lddx;                // [M, 128]
    float *dw; int lddw; long long dw_slot_stride;        // 8 copies [Co, 128], accumulated (zeroed by the caller)
    float *dbias;                       // [Co], accumulated
    double *partials;                   // [P2C_STAT_SLOTS][2][128], accumulated
    int M, Co;
};

__global__ void __launch_bounds__(256, 2) heads_bwd_kernel(HeadsBwdArgs a)
{
    __shared__ float Ys[HB_BM * HB_LDY];         // raw pre-BatchNorm rows
    __shared__ float Xs[HB_BM * HB_LDY];         // the layer's input
    __shared__ float Zs[HB_BM * HB_LDZ];         // dZ, columns Co..31 zero
    __shared__ unsigned char Kb[HB_BM * 32];     // dropout keep flags of the tile: bit j of byte [row][c / 4] = element (row, 4 * (c / 4) + j)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int il = lane & 31, kl = lane >> 5;
    const int cb = wave * 32;                                            // this wave's 32 input channels
    const int Co = a.Co, KS = (Co + 1) / 2;                              // k-steps of the dX product
    const int ntiles = (a.M + HB_BM - 1) / HB_BM;
    // W fragments of the dX product: B[k = co][j = ci] -> lane (j = il, k = 2 * s + kl)
    float wf[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int co = 2 * s + kl;
        wf[s] = co < Co ? a.w[(size_t)co * a.ldw + cb + il] : 0.f;
    }
    const float psc = a.stat[cb + il], psh = a.stat[HB_CI + cb + il], pmu = a.stat[2 * HB_CI + cb + il], pis = a.stat[3 * HB_CI + cb + il];
    const uint32_t slo = a.seed ? a.seed[0] : 0u, shi = a.seed ? a.seed[1] : 0u;
    // load mapping: thread -> (row = tid / 32 + 8 j, 4 channels at 4 * (tid % 32))
    const int lr = tid >> 5, lc = (tid & 31) * 4;
    const float4 lsc = *reinterpret_cast<const float4 *>(a.stat + lc), lsh = *reinterpret_cast<const float4 *>(a.stat + HB_CI + lc);
    float4 ry[8];
    float rz[8];                                                         // dZ tile: 64 x 32 slots, thread -> (row = tid / 32 + 8 j, col = tid % 32)
    auto gload = [&](int t) {
        const int m0 = t * HB_BM;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = min(m0 + lr + 8 * j, a.M - 1);
            ry[j] = *reinterpret_cast<const float4 *>(a.y + (size_t)row * a.ldy + lc);
            rz[j] = ((tid & 31) < Co && m0 + lr + 8 * j < a.M) ? a.dz[(size_t)row * a.lddz + (tid & 31)] : 0.f;
        }
    };
    f32x16 accW;
#pragma unroll
    for (int r = 0; r < 16; ++r) accW[r] = 0.f;
    float s1 = 0.f, s2 = 0.f, db = 0.f;
    double d1 = 0.0, d2 = 0.0;
    int t = blockIdx.x;
    if (t < ntiles) gload(t);
    for (; t < ntiles; t += gridDim.x) {
        const int m0 = t * HB_BM;
        __syncthreads();                                                 // the previous tile's LDS image is dead
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int rl = lr + 8 * j;
            const float4 v = ry[j];
            float4 x;
            x.x = fmaxf(lsc.x * v.x + lsh.x, 0.f); x.y = fmaxf(lsc.y * v.y + lsh.y, 0.f);
            x.z = fmaxf(lsc.z * v.z + lsh.z, 0.f); x.w = fmaxf(lsc.w * v.w + lsh.w, 0.f);
            if (a.seed) {
                const uint32_t e = (uint32_t)(m0 + rl) * (uint32_t)HB_CI + (uint32_t)lc;
                const uint32_t hq = p2c_hash32(slo, shi, e >> 2);            // e is a multiple of 4: one hash for the four elements
                const bool k0 = p2c_keep4(hq, 0, a.thr), k1 = p2c_keep4(hq, 1, a.thr), k2 = p2c_keep4(hq, 2, a.thr), k3 = p2c_keep4(hq, 3, a.thr);
                x.x = k0 ? x.x * a.dscale : 0.f;
                x.y = k1 ? x.y * a.dscale : 0.f;
                x.z = k2 ? x.z * a.dscale : 0.f;
                x.w = k3 ? x.w * a.dscale : 0.f;
                Kb[rl * 32 + (tid & 31)] = (unsigned char)((k0 ? 1 : 0) | (k1 ? 2 : 0) | (k2 ? 4 : 0) | (k3 ? 8 : 0));     // the dX epilogue reads it back
            }
            if (m0 + rl >= a.M) x = float4{0.f, 0.f, 0.f, 0.f};          // rows past M contribute nothing to dW
            *reinterpret_cast<float4 *>(&Ys[rl * HB_LDY + lc]) = v;
            *reinterpret_cast<float4 *>(&Xs[rl * HB_LDY + lc]) = x;
            Zs[rl * HB_LDZ + (tid & 31)] = rz[j];
        }
        __syncthreads();
        if (t + (int)gridDim.x < ntiles) gload(t + gridDim.x);           // next tile's rows in flight under the products
        // ---- dbias: column sums of the dZ tile
        if (tid < 32) {
            float s = 0.f;
#pragma unroll 8
            for (int r = 0; r < HB_BM; ++r) s += Zs[r * HB_LDZ + tid];
            db += s;
        }
        // ---- dW[co, ci] += sum_rows dZ[row, co] * x[row, ci]:  A[i = co][k = row], B[k = row][j = ci]
#pragma unroll 8
        for (int kk = 0; kk < HB_BM; kk += 2) {
            const float av = Zs[(kk + kl) * HB_LDZ + il];
            const float bv = Xs[(kk + kl) * HB_LDY + cb + il];
            accW = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, accW, 0, 0, 0);
        }
        // ---- dX[row, ci] = sum_co dZ[row, co] * W[co, ci]:  A[i = row][k = co], B[k = co][j = ci] (registers)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (s < KS) {
                    const float av = Zs[(rb * 32 + il) * HB_LDZ + 2 * s + kl];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wf[s], acc, 0, 0, 0);
                }
            }
            const int col = cb + il;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl;
                const int row = m0 + rl;
                float v = acc[r];
                if (a.seed) v = ((Kb[rl * 32 + (col >> 2)] >> (col & 3)) & 1) ? v * a.dscale : 0.f;
                if (row < a.M) {
                    a.dx[(size_t)row * a.lddx + col] = v;
                    const float yp = Ys[rl * HB_LDY + col];
                    const float g = (psc * yp + psh > 0.f) ? v : 0.f;     // the forward's two roundings
                    s1 += g;
                    s2 += g * ((yp - pmu) * pis);
                }
            }
        }
        d1 += (double)s1; d2 += (double)s2;                              // fp32 within a tile, fp64 across tiles
        s1 = s2 = 0.f;
    }
    // ---------------- flush
    {
        float *dws = a.dw + (size_t)(blockIdx.x & 7) * a.dw_slot_stride;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = (r & 3) + 8 * (r >> 2) + 4 * kl;
            if (co < Co) atomicAdd(&dws[(size_t)co * a.lddw + cb + il], accW[r]);
        }
    }
    {
        const double u1 = d1 + __shfl_xor(d1, 32), u2 = d2 + __shfl_xor(d2, 32);
        if (kl == 0) {
            double *o = a.partials + (size_t)(blockIdx.x % P2C_STAT_SLOTS) * 2 * HB_CI;
            atomicAdd(&o[cb + il], u1);
            atomicAdd(&o[HB_CI + cb + il], u2);
        }
    }
    if (a.dbias && tid < Co) atomicAdd(&a.dbias[tid], db);
}

extern "C" int p2c_linear_bwd_narrow_supported(int M, int Co, int Ci, int in_mode)
{
    return (Ci == HB_CI && Co >= 1 && Co <= 32 && M >= 4096 && (in_mode == 1 || in_mode == 3)) ? 1 : 0;
}

// dW (8 per-XCD copies, accumulated), dbias (accumulated), dX and the BatchNorm-backward sums of the layer below (fp64 slot rows,
// accumulated) of Z = dropout(relu(bn(Y))) W^T + b from dZ, in one pass.  in_mode 1: no dropout (seed NULL), 3: hashed dropout with
// keep-scale `dscale` (seed -> the int64 counter the forward used).  Finish with p2c_bn_bwd_finalize_sum_f32.
extern "C" int p2c_linear_bwd_narrow_f32(const float *dZ, int lddz, const float *Y, int ldy, const float *stat, const void *seed, float dscale,
                                         const float *W, int ldw, float *dX, int lddx, float *dW8, int lddw, long long dw_slot_stride, float *dbias,
                                         double *partials, int M, int Co, int Ci, void *stream)
{
    if (!dZ || !Y || !stat || !W || !dX || !dW8 || !partials || M <= 0) return P2C_EINVAL;
    if (!p2c_linear_bwd_narrow_supported(M, Co, Ci, seed ? 3 : 1)) return P2C_EINVAL;
    if ((ldy & 3) || ((uintptr_t)Y & 15) || ((uintptr_t)stat & 15)) return P2C_EALIGN;
    if (seed && !p2c_drop_scale_representable(dscale)) return P2C_EINVAL;      // hashed dropout: p must be a multiple of 1/256 (common.h)
    HeadsBwdArgs a{dZ, lddz, Y, ldy, stat, (const uint32_t *)seed, seed ? p2c_drop_threshold(dscale) : 0u, dscale, W, ldw, dX, lddx, dW8, lddw,
                   dw_slot_stride, dbias, partials, M, Co};
    const int ntiles = (M + HB_BM - 1) / HB_BM;
    const int grid = ntiles < 512 ? ntiles : 512;                        // two workgroups per CU: one's loads under the other's products
    hipLaunchKernelGGL(heads_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}
