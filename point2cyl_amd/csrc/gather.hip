// gather.hip -- grouping gather (+centring, concat) and 3-NN interpolation, forward and backward.
// HBM-bound row gathers: each wave moves whole feature rows (contiguous, coalesced); the backward
// passes are scatter-adds with fp32 atomics (several groups share a source point).
#include "common.h"
#include <type_traits>

// out row r=(b,s,j): [xyz[b,idx]-new_xyz[b,s] | feats[b,idx,0:D] | 0...]  (pointnet_util.py:128-139)
__global__ void __launch_bounds__(256) group_gather_kernel(const float *__restrict__ xyz, const float *__restrict__ feats, int ldf,
                                                           const float *__restrict__ new_xyz, const int32_t *__restrict__ idx,
                                                           int N, int S, int ns, int D, long long rows, float *__restrict__ out,
                                                           int ldo, int xyz_last)
{
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long nw = (long long)gridDim.x * 4;
    for (long long r = wave; r < rows; r += nw) {
        const long long g = r / ns;              // (b,s)
        const int b = (int)(g / S);
        const int src = idx[r];
        const float *p = xyz + ((size_t)b * N + src) * 3;
        const float *c = new_xyz + (size_t)g * 3;
        const float *f = feats ? feats + ((size_t)b * N + src) * ldf : nullptr;
        float *o = out + (size_t)r * ldo;
        const int xo = xyz_last ? D : 0, fo = xyz_last ? 0 : 3;      // column offsets of the xyz / feature parts
        for (int col = lane; col < ldo; col += 64) {
            float v = 0.f;
            if (col >= xo && col < xo + 3) v = p[col - xo] - c[col - xo];
            else if (col >= fo && col < fo + D) v = f[col - fo];
            o[col] = v;
        }
    }
}

// D == 0 fast path: one thread per row, a single 16-byte store
__global__ void __launch_bounds__(256) group_gather_xyz_kernel(const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                               const int32_t *__restrict__ idx, int N, int S, int ns,
                                                               long long rows, float4 *__restrict__ out)
{
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const long long g = r / ns;
    const int b = (int)(g / S);
    const float *p = xyz + ((size_t)b * N + idx[r]) * 3;
    const float *c = new_xyz + (size_t)g * 3;
    out[r] = make_float4(p[0] - c[0], p[1] - c[1], p[2] - c[2], 0.f);
}

extern "C" int p2c_group_gather_f32(const float *xyz, const float *feats, int ldf, const float *new_xyz, const int32_t *idx, int B,
                                    int N, int S, int nsample, int D, float *out, int ldo, int xyz_last, void *stream)
{
    if (!xyz || !new_xyz || !idx || !out || B <= 0 || D < 0 || (D > 0 && !feats) || ldo < 3 + D) return P2C_EINVAL;
    if (ldo % 4) return P2C_EALIGN;
    const long long rows = (long long)B * S * nsample;
    hipStream_t s = (hipStream_t)stream;
    if (D == 0 && ldo == 4) {
        hipLaunchKernelGGL(group_gather_xyz_kernel, dim3(p2c_cdiv(rows, 256)), dim3(256), 0, s, xyz, new_xyz, idx, N, S, nsample, rows,
                           reinterpret_cast<float4 *>(out));
    } else {
        const int blocks = (int)min((long long)p2c_cdiv(rows, 4), 16384LL);
        hipLaunchKernelGGL(group_gather_kernel, dim3(blocks), dim3(256), 0, s, xyz, feats, ldf, new_xyz, idx, N, S, nsample, D, rows, out, ldo, xyz_last);
    }
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

__global__ void __launch_bounds__(256) group_gather_bwd_kernel(const float *__restrict__ dout, int ldo, const int32_t *__restrict__ idx,
                                                               int N, int S, int ns, int D, long long rows, float *__restrict__ dfeats,
                                                               int ldf, int coff)
{
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long nw = (long long)gridDim.x * 4;
    for (long long r = wave; r < rows; r += nw) {
        const int b = (int)(r / ((long long)S * ns));
        float *d = dfeats + ((size_t)b * N + idx[r]) * ldf;
        const float *g = dout + (size_t)r * ldo + coff;
        for (int c = lane; c < D; c += 64) atomicAdd(d + c, g[c]);
    }
}

extern "C" int p2c_group_gather_bwd_f32(const float *dout, int ldo, const int32_t *idx, int B, int N, int S, int nsample, int D,
                                        float *dfeats, int ldf, int xyz_last, void *stream)
{
    if (!dout || !idx || !dfeats || D <= 0) return P2C_EINVAL;
    const long long rows = (long long)B * S * nsample;
    const int blocks = (int)min((long long)p2c_cdiv(rows, 4), 16384LL);
    hipLaunchKernelGGL(group_gather_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dout, ldo, idx, N, S, nsample, D, rows,
                       dfeats, ldf, xyz_last ? 0 : 3);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// out[b,n,c] = (f0*w0 + f1*w1) + f2*w2   (pointnet_util.py:308: sum over the 3 neighbours in order)
__global__ void __launch_bounds__(256) three_interp_kernel(const float *__restrict__ feats, int ldf, const int32_t *__restrict__ idx,
                                                           const float *__restrict__ w, int N, int S, int C, long long rows,
                                                           float *__restrict__ out, int ldo)
{
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long nw = (long long)gridDim.x * 4;
    for (long long r = wave; r < rows; r += nw) {
        const int b = (int)(r / N);
        const int i0 = idx[r * 3 + 0], i1 = idx[r * 3 + 1], i2 = idx[r * 3 + 2];
        const float w0 = w[r * 3 + 0], w1 = w[r * 3 + 1], w2 = w[r * 3 + 2];
        const float *f0 = feats + ((size_t)b * S + i0) * ldf, *f1 = feats + ((size_t)b * S + i1) * ldf,
                    *f2 = feats + ((size_t)b * S + i2) * ldf;
        float *o = out + (size_t)r * ldo;
        for (int c = lane; c < C; c += 64) o[c] = (f0[c] * w0 + f1[c] * w1) + f2[c] * w2;
    }
}

extern "C" int p2c_three_interp_f32(const float *feats, int ldf, const int32_t *idx, const float *weight, int B, int N, int S, int C,
                                    float *out, int ldo, void *stream)
{
    if (!feats || !idx || !weight || !out || C <= 0) return P2C_EINVAL;
    const long long rows = (long long)B * N;
    const int blocks = (int)min((long long)p2c_cdiv(rows, 4), 16384LL);
    hipLaunchKernelGGL(three_interp_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, feats, ldf, idx, weight, N, S, C, rows, out, ldo);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// The same with the row's skip features in front: out[r] = [skip[r, :Cs] | interpolated (C) | zeros up to width] - the input of a
// feature-propagation level (pointnet_util.py:308-312) written by ONE kernel (the strided copy of the skip block was a launch of its own).
__global__ void __launch_bounds__(256) three_interp_skip_kernel(const float *__restrict__ feats, int ldf, const int32_t *__restrict__ idx,
                                                                const float *__restrict__ w, int N, int S, int C, long long rows,
                                                                const float *__restrict__ skip, int ldskip, int Cs, float *__restrict__ out, int ldo,
                                                                int width)
{
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long nw = (long long)gridDim.x * 4;
    for (long long r = wave; r < rows; r += nw) {
        const int b = (int)(r / N);
        const int i0 = idx[r * 3 + 0], i1 = idx[r * 3 + 1], i2 = idx[r * 3 + 2];
        const float w0 = w[r * 3 + 0], w1 = w[r * 3 + 1], w2 = w[r * 3 + 2];
        const float *f0 = feats + ((size_t)b * S + i0) * ldf, *f1 = feats + ((size_t)b * S + i1) * ldf,
                    *f2 = feats + ((size_t)b * S + i2) * ldf;
        const float *sk = skip + (size_t)r * ldskip;
        float *o = out + (size_t)r * ldo;
        for (int c = lane; c < Cs; c += 64) o[c] = sk[c];
        for (int c = lane; c < C; c += 64) o[Cs + c] = (f0[c] * w0 + f1[c] * w1) + f2[c] * w2;
        for (int c = Cs + C + lane; c < width; c += 64) o[c] = 0.f;
    }
}

extern "C" int p2c_three_interp_skip_f32(const float *feats, int ldf, const int32_t *idx, const float *weight, int B, int N, int S, int C,
                                         const float *skip, int ldskip, int Cskip, float *out, int ldo, int width, void *stream)
{
    if (!feats || !idx || !weight || !out || !skip || C <= 0 || Cskip <= 0 || width < Cskip + C || ldo < width) return P2C_EINVAL;
    const long long rows = (long long)B * N;
    const int blocks = (int)min((long long)p2c_cdiv(rows, 4), 16384LL);
    hipLaunchKernelGGL(three_interp_skip_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, feats, ldf, idx, weight, N, S, C, rows, skip,
                       ldskip, Cskip, out, ldo, width);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

__global__ void __launch_bounds__(256) three_interp_bwd_kernel(const float *__restrict__ dout, int ldo, const int32_t *__restrict__ idx,
                                                               const float *__restrict__ w, int N, int S, int C, long long rows,
                                                               float *__restrict__ dfeats, int ldf)
{
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long nw = (long long)gridDim.x * 4;
    for (long long r = wave; r < rows; r += nw) {
        const int b = (int)(r / N);
        const float *g = dout + (size_t)r * ldo;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float *d = dfeats + ((size_t)b * S + idx[r * 3 + j]) * ldf;
            const float wj = w[r * 3 + j];
            for (int c = lane; c < C; c += 64) atomicAdd(d + c, g[c] * wj);
        }
    }
}

extern "C" int p2c_three_interp_bwd_f32(const float *dout, int ldo, const int32_t *idx, const float *weight, int B, int N, int S,
                                        int C, float *dfeats, int ldf, void *stream)
{
    if (!dout || !idx || !weight || !dfeats || C <= 0) return P2C_EINVAL;
    const long long rows = (long long)B * N;
    const int blocks = (int)min((long long)p2c_cdiv(rows, 4), 16384LL);
    hipLaunchKernelGGL(three_interp_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dout, ldo, idx, weight, N, S, C, rows,
                       dfeats, ldf);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// ------------------------------------------------------------------------------------------------
// Gather-formulated backward of the two gathers.  The scatter targets are known before the step starts (they depend
// only on the coordinates), so the inverse map "target row -> list of entries that read it" is built once per batch
// (p2c_build_csr_i32, a counting sort per cloud in LDS; part of the geometry that can be prefetched) and the backward
// becomes a gather: one wave per target row sums  w[e] * src[row(e), :]  over its entries with coalesced row reads.
// No read-modify-write traffic, no 100 M fp32 atomics per step.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) build_csr_kernel(const int32_t *__restrict__ idx, const float *__restrict__ w, int E, int ediv, int T,
                                                         int32_t *__restrict__ offsets, int32_t *__restrict__ entries,
                                                         float *__restrict__ wsorted)
{
    extern __shared__ int cnt[];                 // [T] counts -> cursors, then [T+1] offsets
    int *off = cnt + T;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int32_t *ib = idx + (size_t)b * E;
    for (int t = tid; t < T; t += 1024) cnt[t] = 0;
    __syncthreads();
    for (int e = tid; e < E; e += 1024) {
        const int t = ib[e];
        if (t >= 0 && t < T) atomicAdd(&cnt[t], 1);
    }
    __syncthreads();
    // exclusive scan of the T counts by the whole workgroup: thread i owns the chunk [i*per, (i+1)*per), a wave scan of the chunk sums,
    // the 16 wave totals through LDS.  (Thread 0 alone took 60 us for the 8192 targets of the first level - on the forked stream, where
    // this kernel then sat next to FP1's forward kernels for that long.)
    {
        __shared__ int wtot[16];
        const int per = (T + 1023) / 1024, t0 = tid * per, t1 = min(T, t0 + per);
        int sum = 0;
        for (int t = t0; t < t1; ++t) sum += cnt[t];
        int inc = sum;                            // inclusive scan over the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(inc, o);
            if ((tid & 63) >= o) inc += v;
        }
        if ((tid & 63) == 63) wtot[tid >> 6] = inc;
        __syncthreads();
        int base = 0;
        for (int wv = 0; wv < (tid >> 6); ++wv) base += wtot[wv];
        int run = base + inc - sum;               // exclusive prefix of this thread's chunk
        for (int t = t0; t < t1; ++t) { off[t] = run; run += cnt[t]; }
        if (tid == 1023) off[T] = base + inc;
    }
    __syncthreads();
    for (int t = tid; t <= T; t += 1024) offsets[(size_t)b * (T + 1) + t] = off[t];
    for (int t = tid; t < T; t += 1024) cnt[t] = off[t];
    __syncthreads();
    for (int e = tid; e < E; e += 1024) {
        const int t = ib[e];
        if (t >= 0 && t < T) {
            const size_t k = (size_t)b * E + atomicAdd(&cnt[t], 1);
            entries[k] = e / ediv;               // the source ROW of the entry (what the gather needs), not the entry id
            if (w) wsorted[k] = w[(size_t)b * E + e];
        }
    }
}

extern "C" int p2c_build_csr_i32(const int32_t *idx, const float *w, int B, int E, int ediv, int T, int32_t *offsets, int32_t *rows,
                                 float *wsorted, void *stream)
{
    int32_t *entries = rows;
    if (!idx || !offsets || !entries || B <= 0 || E <= 0 || T <= 0 || T > 16000 || ediv <= 0 || (w && !wsorted)) return P2C_EINVAL;
    const size_t lds = (size_t)(2 * T + 1) * sizeof(int);
    (void)hipFuncSetAttribute((const void *)build_csr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(build_csr_kernel, dim3(B), dim3(1024), lds, (hipStream_t)stream, idx, w, E, ediv, T, offsets, entries, wsorted);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

template <int CPL>
__global__ void __launch_bounds__(256) csr_gather_kernel(const float *__restrict__ src, int lds_, int coff, const int32_t *__restrict__ offsets,
                                                         const int32_t *__restrict__ entries, const float *__restrict__ w, int E,
                                                         int rows_b, int T, int C, long long total, float *__restrict__ out, int ldo)
{
    const int lane = threadIdx.x & 63;
    const long long wv = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wv >= total) return;
    const int b = (int)(wv / T), t = (int)(wv - (long long)b * T);
    const int k0 = offsets[(size_t)b * (T + 1) + t], k1 = offsets[(size_t)b * (T + 1) + t + 1];
    const int32_t *eb = entries + (size_t)b * E;
    const float *wb = w ? w + (size_t)b * E : nullptr;
    float acc[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) acc[i] = 0.f;
    int k = k0;
    constexpr int UB = 8;                        // independent row reads in flight per lane
    for (; k + UB <= k1; k += UB) {
        float we[UB]; const float *row[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            we[u] = wb ? wb[k + u] : 1.f;
            row[u] = src + ((size_t)b * rows_b + eb[k + u]) * lds_ + coff;
        }
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int c = lane + 64 * i;
            if (c < C) {
                float v[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) v[u] = row[u][c];
#pragma unroll
                for (int u = 0; u < UB; ++u) acc[i] += we[u] * v[u];
            }
        }
    }
    for (; k < k1; ++k) {
        const int e = eb[k];
        const float we = wb ? wb[k] : 1.f;
        const float *row = src + ((size_t)b * rows_b + e) * lds_ + coff;
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int c = lane + 64 * i;
            if (c < C) acc[i] += we * row[c];
        }
    }
    float *o = out + ((size_t)b * T + t) * ldo;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int c = lane + 64 * i;
        if (c < C) o[c] = acc[i];
    }
}

// out[b,t,0:C] = sum_{k in bucket(b,t)} wsorted[b,k] * src[b*rows_b + rows[b,k], coff:coff+C]   (wsorted NULL = 1; overwrites out)
extern "C" int p2c_csr_gather_f32(const float *src, int ld_src, int coff, const int32_t *offsets, const int32_t *entries, const float *w,
                                  int B, int E, int rows_b, int T, int C, float *out, int ldo, void *stream)
{
    if (!src || !offsets || !entries || !out || B <= 0 || E <= 0 || T <= 0 || C <= 0 || C > 256) return P2C_EINVAL;
    const long long total = (long long)B * T;
    dim3 grid(p2c_cdiv(total, 4));
    hipStream_t s = (hipStream_t)stream;
#define P2C_CG(CPL_) hipLaunchKernelGGL(csr_gather_kernel<CPL_>, grid, dim3(256), 0, s, src, ld_src, coff, offsets, entries, w, E, rows_b, T, C, total, out, ldo)
    if (C <= 64) P2C_CG(1);
    else if (C <= 128) P2C_CG(2);
    else P2C_CG(4);
#undef P2C_CG
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// ------------------------------------------------------------------------------------------------
// "Linear before the gather".  A 1x1-conv layer applied to gathered rows commutes with the gather (both are linear):
//     interp(F) . W^T = interp(F . W^T),      [F[idx] | dxyz] . [Wf | Wx]^T = (F . Wf^T)[idx] + dxyz . Wx^T
// so the GEMM runs on the SPARSE set (FP1: 512 instead of 8192 rows per cloud, SA2: 512 points instead of 128 x 64 grouped
// rows: 16x fewer FLOPs) and the dense pre-BN tensor Y0 is produced by a gather that also adds the bias and the BatchNorm
// sums.  Backward: dY0 (the ReLU+BN backward of the stored gradient, rebuilt per element) is reduced to the sparse rows by
// the CSR gather, then two small GEMMs give dW and dF.
// ------------------------------------------------------------------------------------------------
template <int CPL>
__global__ void __launch_bounds__(256) three_interp_stats_kernel(const float *__restrict__ feats, int ldf, const int32_t *__restrict__ idx,
                                                                 const float *__restrict__ w, int N, int S, int C, long long rows,
                                                                 const float *__restrict__ bias, float *__restrict__ out, int ldo,
                                                                 double *__restrict__ slots, int xcd_bpc)
{
    constexpr int RPW = 16, U = 4;                // rows per wave; rows in flight
    __shared__ float red[2][4][64 * CPL];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // xcd_bpc != 0 (= workgroups per cloud): cloud b is served by XCD b % 8 only (workgroup ids go round the XCDs), so the 262 KB of sparse
    // rows its 8192 dense rows read three at a time stay in ONE L2 - 4 clouds = 1 MB per XCD instead of all 32 = 8.4 MB in each of them
    // (PMC: 152 MB fetched for 15 MB of sparse rows + indices).
    int blk = blockIdx.x;
    if (xcd_bpc) {
        const int xcd = blk & 7, slot = blk >> 3, round = slot / xcd_bpc;
        blk = (xcd + 8 * round) * xcd_bpc + (slot - round * xcd_bpc);
    }
    const long long r0 = ((long long)blk * 4 + wave) * RPW;          // uniform
    // (Lane l owning CPL CONSECUTIVE channels with one 8 / 16-byte access per row - half the memory instructions - measured SLOWER:
    // 54.7 us against 49.4 for the strided dword form at 128 channels.)
    float bv[CPL], s1[CPL], s2[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int c = lane + 64 * i;
        bv[i] = (bias && c < C) ? bias[c] : 0.f;
        s1[i] = s2[i] = 0.f;
    }
    // The 16 rows' neighbour rows and weights in one shot: lane l < 48 holds entry l of the 48 as (sparse row number, weight), fetched back
    // with v_readlane in the row loop - uniform row pointers (scalar base + lane offset loads), no dependent index -> feature load chain, U
    // rows in flight.  ONE division per wave for the cloud of its first row (the loop had a 64-bit r / N per row: with the ds_bpermute
    // broadcasts and the per-lane 64-bit addresses that followed from them, 160 instructions per two rows, half of them bookkeeping).
    const long long nleft = rows - r0;
    const int nrow = (int)(nleft < RPW ? (nleft > 0 ? nleft : 0) : RPW);
    const long long b0 = r0 / N;
    const int rem0 = (int)(r0 - b0 * N);
    int mysrc = 0; float myw = 0.f;
    if (lane < 3 * nrow) {
        const int o = rem0 + lane / 3;                              // row of this entry, counted from the first row's cloud
        const long long b = b0 + (o >= N ? o / N : 0);
        mysrc = (int)(b * S) + idx[r0 * 3 + lane];
        myw = w[r0 * 3 + lane];
    }
    auto rows_step = [&](int rr, auto exact) {
        constexpr bool EXACT = decltype(exact)::value;          // all U rows exist and C == 64 * CPL: nothing conditional between the loads and the stores
        const float *f[U][3]; float ww[U][3];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = EXACT ? rr + u : min(rr + u, nrow - 1);          // (a slot past the end repeats the last row: loaded, neither summed nor stored)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                f[u][j] = feats + (size_t)__builtin_amdgcn_readlane(mysrc, 3 * row + j) * ldf;
                ww[u][j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(myw), 3 * row + j));
            }
        }
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int c = lane + 64 * i;
            if (EXACT || c < C) {
                float a[U][3];
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int j = 0; j < 3; ++j) a[u][j] = f[u][j][c];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (EXACT || rr + u < nrow) {                   // uniform
                        const float v = (a[u][0] * ww[u][0] + a[u][1] * ww[u][1]) + a[u][2] * ww[u][2];
                        s1[i] += v;
                        s2[i] += v * v;
                        out[(size_t)(r0 + rr + u) * ldo + c] = v + bv[i];
                    }
                }
            }
        }
    };
    if (nrow == RPW && C == 64 * CPL) {
        for (int rr = 0; rr < RPW; rr += U) rows_step(rr, std::true_type{});
    } else {
        for (int rr = 0; rr < nrow; rr += U) rows_step(rr, std::false_type{});
    }
    if (!slots) return;
#pragma unroll
    for (int i = 0; i < CPL; ++i) { red[0][wave][lane + 64 * i] = s1[i]; red[1][wave][lane + 64 * i] = s2[i]; }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        double *o = slots + (size_t)(blk % P2C_STAT_SLOTS) * 2 * C;
        atomicAdd(&o[c], (double)((red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c])));
        atomicAdd(&o[C + c], (double)((red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c])));
    }
}

// out[b,n,:] = interp(feats)[b,n,:] + bias; BatchNorm sums of the bias-free value into stat_slots (NULL: none)
extern "C" int p2c_three_interp_bias_stats_f32(const float *feats, int ldf, const int32_t *idx, const float *weight, int B, int N, int S, int C,
                                               const float *bias, float *out, int ldo, double *stat_slots, void *stream)
{
    if (!feats || !idx || !weight || !out || C <= 0 || C > 256) return P2C_EINVAL;
    const long long rows = (long long)B * N;
    const int blocks = p2c_cdiv(rows, 64);
    hipStream_t s = (hipStream_t)stream;
    const bool xcd_on = true;          // cloud b served by XCD b % 8 (its rows share one L2): csr_gather_bn fetch 742 -> 543 MB, DESIGN.md 3
    const int xcd_bpc = (xcd_on && B % 8 == 0 && N % 64 == 0) ? N / 64 : 0;
#define P2C_TI(CPL_) hipLaunchKernelGGL(three_interp_stats_kernel<CPL_>, dim3(blocks), dim3(256), 0, s, feats, ldf, idx, weight, N, S, C, rows, bias, out, ldo, stat_slots, xcd_bpc)
    if (C <= 64) P2C_TI(1);
    else if (C <= 128) P2C_TI(2);
    else P2C_TI(4);
#undef P2C_TI
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// CSR gather of dY = gs*(dZ*[scale*Y+shift > 0]) + q*Y + p (coef = [scale|shift|gs|q|p] x C, as everywhere): the ReLU +
// train-BatchNorm backward of a stored gradient, rebuilt per element and reduced to the target rows in one pass.
template <int CPL>
__global__ void __launch_bounds__(256) csr_gather_bn_kernel(const float *__restrict__ dz, int lddz, const float *__restrict__ y, int ldy,
                                                            const float *__restrict__ coef, const int32_t *__restrict__ offsets,
                                                            const int32_t *__restrict__ entries, const float *__restrict__ w, int E, int rows_b,
                                                            int T, int C, long long total, float *__restrict__ out, int ldo, int xcd_clouds, int nsplit)
{
    const int lane = threadIdx.x & 63;
    // XCD-aware: workgroups go round the 8 XCDs (blockIdx.x % 8), and every dense row is read by ~3 target rows of the SAME cloud - with the
    // targets of a cloud spread over all XCDs each of those reads missed its own L2 (742 MB fetched for a 268 MB stream).  xcd_clouds != 0:
    // cloud b is served by XCD b % 8 only, all its targets resident there at about the same time, so the second and third read hit that L2.
    // nsplit = 2 (xcd_clouds only): the channels go in two halves, ALL targets of a cloud for the first half before any for the second, so the
    // rows a cloud's targets share (dz + y of one cloud: 8.4 MB at 128 channels, twice an XCD's 4 MB L2) are a 4.2 MB working set per pass.
    // (A wave serves ONE target row of ~4 entries: its prologue is half of what it executes.  The (cloud, row, channel half) of a workgroup
    // come from a 2-D grid without a division - x = xcd + 8 * (half * blocks_per_cloud + block), y = round of eight clouds; the hardware
    // deals workgroups to the XCDs by their linear id, and gridDim.x is a multiple of 8 - and the wave index is made uniform, so that the
    // row's offsets are scalar loads.)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int b, t, h = 0;
    if (xcd_clouds) {
        const int xcd = (int)(blockIdx.x & 7), r = (int)(blockIdx.x >> 3), bpc = T / 4;           // blocks per cloud (and channel half)
        h = r >= bpc ? 1 : 0;                                                                    // (nsplit <= 2)
        b = xcd + 8 * (int)blockIdx.y;
        t = (r - h * bpc) * 4 + wave;
    } else {
        const long long wv = (long long)blockIdx.x * 4 + wave;
        if (wv >= total) return;
        b = (int)(wv / T); t = (int)(wv - (long long)b * T);
    }
    const int Cs = C / nsplit, c0 = h * Cs;                  // this wave's channels: [c0, c0 + Cs)
    const int k0 = offsets[(size_t)b * (T + 1) + t], k1 = offsets[(size_t)b * (T + 1) + t + 1];
    const int32_t *eb = entries + (size_t)b * E;
    const float *wb = w ? w + (size_t)b * E : nullptr;
    float acc[CPL], csc[CPL], csh[CPL], cgs[CPL], cq[CPL], cp[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int c = c0 + min(lane + 64 * i, Cs - 1);
        acc[i] = 0.f;
        csc[i] = coef[c]; csh[i] = coef[C + c]; cgs[i] = coef[2 * C + c]; cq[i] = coef[3 * C + c]; cp[i] = coef[4 * C + c];
    }
    constexpr int UB = 8;                         // (dz, y) row pairs in flight per lane
    // 64 entries at a time: lane l fetches entry kc + l (row id and weight, one coalesced load each) and the loops below read them back with
    // v_readlane - no scalar load sits between two batches of row loads (as in group_linear_bwd_kernel below).
    for (int kc = k0; kc < k1; kc += 64) {
        const int n = min(64, k1 - kc);
        const int kl = kc + min(lane, n - 1);
        const int le = eb[kl];
        const float lw = wb ? wb[kl] : 1.f;
        int k = 0;
        for (; k + UB <= n; k += UB) {
            float we[UB]; size_t ro[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                we[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lw), k + u));
                ro[u] = (size_t)b * rows_b + __builtin_amdgcn_readlane(le, k + u);
            }
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                const int c = c0 + lane + 64 * i;
                if (c < c0 + Cs) {
                    float g[UB], yy[UB];
#pragma unroll
                    for (int u = 0; u < UB; ++u) { g[u] = dz[ro[u] * lddz + c]; yy[u] = y[ro[u] * ldy + c]; }
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        const float d = __builtin_fmaf(cgs[i], (csc[i] * yy[u] + csh[i] > 0.f) ? g[u] : 0.f, __builtin_fmaf(cq[i], yy[u], cp[i]));
                        acc[i] += we[u] * d;
                    }
                }
            }
        }
        for (; k < n; ++k) {
            const float we = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lw), k));
            const size_t ro = (size_t)b * rows_b + __builtin_amdgcn_readlane(le, k);
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                const int c = c0 + lane + 64 * i;
                if (c < c0 + Cs) {
                    const float g = dz[ro * lddz + c], yy = y[ro * ldy + c];
                    acc[i] += we * __builtin_fmaf(cgs[i], (csc[i] * yy + csh[i] > 0.f) ? g : 0.f, __builtin_fmaf(cq[i], yy, cp[i]));
                }
            }
        }
    }
    float *o = out + ((size_t)b * T + t) * ldo;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int c = c0 + lane + 64 * i;
        if (c < c0 + Cs) o[c] = acc[i];
    }
}

extern "C" int p2c_csr_gather_bn_f32(const float *dz, int lddz, const float *y, int ldy, const float *coef, const int32_t *offsets,
                                     const int32_t *rows, const float *wsorted, int B, int E, int rows_b, int T, int C, float *out, int ldo,
                                     void *stream)
{
    if (!dz || !y || !coef || !offsets || !rows || !out || B <= 0 || E <= 0 || T <= 0 || C <= 0 || C > 256) return P2C_EINVAL;
    const long long total = (long long)B * T;
    hipStream_t s = (hipStream_t)stream;
    const bool xcd_on = true, split_on = true;      // cloud -> XCD mapping; channel halves (4.2 MB working set per XCD): 118 -> 91 us, DESIGN.md 3
    const int xcd_clouds = (xcd_on && B % 8 == 0 && T % 4 == 0) ? 1 : 0;
    const int nsplit = (xcd_clouds && split_on && C == 128) ? 2 : 1;
    dim3 grid(p2c_cdiv(total, 4) * nsplit);
    if (xcd_clouds) grid = dim3(8 * nsplit * (T / 4), B / 8);
    const int Cs = C / nsplit;
#define P2C_CGB(CPL_) hipLaunchKernelGGL(csr_gather_bn_kernel<CPL_>, grid, dim3(256), 0, s, dz, lddz, y, ldy, coef, offsets, rows, wsorted, E, rows_b, T, C, total, out, ldo, xcd_clouds, nsplit)
    if (Cs <= 64) P2C_CGB(1);
    else if (Cs <= 128) P2C_CGB(2);
    else P2C_CGB(4);
#undef P2C_CGB
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// Grouped layer with point features (SA2): Y0[(b,s,j), :] = G[b, idx[b,s,j], :] + Wx . (xyz[b,idx] - new_xyz[b,s]) + bias, where
// G = F . Wf^T was computed on the N points (W = [Wx | Wf] in the reference's [xyz | features] column order).
// Wx: [C,4] row-major (4th column unused).  BatchNorm sums of the bias-free value into stat_slots (NULL: none).
template <int CPL>
__global__ void __launch_bounds__(256) group_linear_stats_kernel(const float *__restrict__ Gf, int ldg, const float *__restrict__ xyz,
                                                                 const float *__restrict__ new_xyz, const int32_t *__restrict__ idx,
                                                                 const float *__restrict__ Wx, const float *__restrict__ bias, int N, int S, int ns,
                                                                 int C, long long rows, float *__restrict__ out, int ldo,
                                                                 double *__restrict__ slots, int xcd_bpc)
{
    constexpr int RPW = 16, U = 4;                // rows per wave; rows in flight
    __shared__ float red[2][4][64 * CPL];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int blk = blockIdx.x;                         // xcd_bpc != 0: cloud b on XCD b % 8, as in three_interp_stats_kernel
    if (xcd_bpc) {
        const int xcd = blk & 7, slot = blk >> 3, round = slot / xcd_bpc;
        blk = (xcd + 8 * round) * xcd_bpc + (slot - round * xcd_bpc);
    }
    const long long r0 = ((long long)blk * 4 + wave) * RPW;          // uniform
    float bv[CPL], s1[CPL], s2[CPL], wx0[CPL], wx1[CPL], wx2[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int c = min(lane + 64 * i, C - 1);
        bv[i] = bias ? bias[c] : 0.f;
        wx0[i] = Wx[c * 4 + 0]; wx1[i] = Wx[c * 4 + 1]; wx2[i] = Wx[c * 4 + 2];
        s1[i] = s2[i] = 0.f;
    }
    // lane l < 16 resolves row r0 + l (source point, relative coordinates) so the row loop has no dependent load chain; the group and the
    // cloud of the wave's FIRST row by two divisions per wave, the lanes' from there by compares (the per-lane 64-bit r / ns and grp / S,
    // and the ds_bpermute broadcasts of what they produced, were a third of the kernel's instructions)
    const long long nleft = rows - r0;
    const int nrow = (int)(nleft < RPW ? (nleft > 0 ? nleft : 0) : RPW);
    const long long g0 = r0 / ns, b0 = g0 / S;
    const int grem = (int)(r0 - g0 * ns), srem = (int)(g0 - b0 * S);
    int myrow_lo = 0, myrow_hi = 0; float mdx = 0.f, mdy = 0.f, mdz = 0.f;
    if (lane < nrow) {
        const int og = grem + lane, dg = og >= ns ? og / ns : 0;       // groups past the first row's
        const long long grp = g0 + dg;
        const int ob = srem + dg;
        const long long b = b0 + (ob >= S ? ob / S : 0);
        const int p = idx[r0 + lane];
        const float *pp = xyz + ((size_t)b * N + p) * 3, *cc = new_xyz + (size_t)grp * 3;
        mdx = pp[0] - cc[0]; mdy = pp[1] - cc[1]; mdz = pp[2] - cc[2];
        const long long myrow = b * N + p;
        myrow_lo = (int)(myrow & 0xffffffffLL); myrow_hi = (int)(myrow >> 32);
    }
    auto rows_step = [&](int rr, auto exact) {
        constexpr bool EXACT = decltype(exact)::value;          // all U rows exist and C == 64 * CPL: nothing conditional between the loads and the stores
        const float *g[U]; float dx[U], dy[U], dz[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int src = EXACT ? rr + u : min(rr + u, nrow - 1);          // (a slot past the end repeats the last row: loaded, neither summed nor stored)
            const unsigned long long row = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(myrow_hi, src) << 32) | (unsigned)__builtin_amdgcn_readlane(myrow_lo, src);
            g[u] = Gf + (size_t)row * ldg;
            dx[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mdx), src));
            dy[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mdy), src));
            dz[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mdz), src));
        }
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int c = lane + 64 * i;
            if (EXACT || c < C) {
                float gv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) gv[u] = g[u][c];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (EXACT || rr + u < nrow) {                   // uniform
                        const float v = gv[u] + __builtin_fmaf(wx2[i], dz[u], __builtin_fmaf(wx1[i], dy[u], wx0[i] * dx[u]));
                        s1[i] += v;
                        s2[i] += v * v;
                        out[(size_t)(r0 + rr + u) * ldo + c] = v + bv[i];
                    }
                }
            }
        }
    };
    if (nrow == RPW && C == 64 * CPL) {
        for (int rr = 0; rr < RPW; rr += U) rows_step(rr, std::true_type{});
    } else {
        for (int rr = 0; rr < nrow; rr += U) rows_step(rr, std::false_type{});
    }
    if (!slots) return;
#pragma unroll
    for (int i = 0; i < CPL; ++i) { red[0][wave][lane + 64 * i] = s1[i]; red[1][wave][lane + 64 * i] = s2[i]; }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        double *o = slots + (size_t)(blk % P2C_STAT_SLOTS) * 2 * C;
        atomicAdd(&o[c], (double)((red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c])));
        atomicAdd(&o[C + c], (double)((red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c])));
    }
}

extern "C" int p2c_group_linear_bias_stats_f32(const float *G, int ldg, const float *xyz, const float *new_xyz, const int32_t *idx, const float *Wx,
                                               const float *bias, int B, int N, int S, int nsample, int C, float *out, int ldo,
                                               double *stat_slots, void *stream)
{
    if (!G || !xyz || !new_xyz || !idx || !Wx || !out || B <= 0 || N <= 0 || S <= 0 || nsample <= 0 || C <= 0 || C > 256) return P2C_EINVAL;
    const long long rows = (long long)B * S * nsample;
    const int blocks = p2c_cdiv(rows, 64);
    hipStream_t s = (hipStream_t)stream;
    const bool xcd_on = true;          // cloud b served by XCD b % 8 (its rows share one L2): csr_gather_bn fetch 742 -> 543 MB, DESIGN.md 3
    const long long rows_b = (long long)S * nsample;
    const int xcd_bpc = (xcd_on && B % 8 == 0 && rows_b % 64 == 0) ? (int)(rows_b / 64) : 0;
#define P2C_GL(CPL_) hipLaunchKernelGGL(group_linear_stats_kernel<CPL_>, dim3(blocks), dim3(256), 0, s, G, ldg, xyz, new_xyz, idx, Wx, bias, N, S, nsample, C, rows, out, ldo, stat_slots, xcd_bpc)
    if (C <= 64) P2C_GL(1);
    else if (C <= 128) P2C_GL(2);
    else P2C_GL(4);
#undef P2C_GL
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// Backward of the above: dG[b,p,:] = sum of dY0 over the grouped rows that read point p (CSR, unit weights), and
// dWx[c, 0..2] = sum over ALL rows of dY0[row,c] * (xyz[b,p] - new_xyz[b,row/ns]) accumulated on the way (every grouped row
// belongs to exactly one point) into dwx_slots [P2C_STAT_SLOTS][3][C] (fp64, zeroed by the caller).
// Balanced over ENTRIES, not points: ball-query padding repeats the first neighbour, so a few points are read by hundreds of
// grouped rows (one wave - or one workgroup - per point left a 0.3 - 1 ms tail).  Every wave takes EPW consecutive entries of
// the point-sorted list, runs a segmented sum (the point changes where k reaches offsets[t+1]) and flushes each segment into dG
// (zeroed by the caller): plain stores for the points that lie wholly inside the wave's range, one atomicAdd per channel for the
// (at most two) points its ends cut.
template <int CPL, int EPW>
__global__ void __launch_bounds__(256) group_linear_bwd_kernel(const float *__restrict__ dz, int lddz, const float *__restrict__ y, int ldy,
                                                               const float *__restrict__ coef, const int32_t *__restrict__ offsets,
                                                               const int32_t *__restrict__ entries, const float *__restrict__ xyz,
                                                               const float *__restrict__ new_xyz, int B, int N, int S, int ns, int C, int wpc,
                                                               float *__restrict__ dG, int ldo, double *__restrict__ dwx_slots)
{
    constexpr int UB = 4;            // 64 entries per wave: ~4 whole points (plain stores) + 2 cut ones (atomics)
    __shared__ float red[3][4][64 * CPL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int E = S * ns;
    float csc[CPL], csh[CPL], cgs[CPL], cq[CPL], cp[CPL], a0[CPL], a1[CPL], a2[CPL], acc[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int c = min(lane + 64 * i, C - 1);
        csc[i] = coef[c]; csh[i] = coef[C + c]; cgs[i] = coef[2 * C + c]; cq[i] = coef[3 * C + c]; cp[i] = coef[4 * C + c];
        a0[i] = a1[i] = a2[i] = acc[i] = 0.f;
    }
    const long long gw = (long long)blockIdx.x * 4 + wave;      // global wave id: cloud b, entry range [kb, kend)
    const int b = (int)(gw / wpc);
    if (b < B) {
        const int32_t *off = offsets + (size_t)b * (N + 1);
        const int32_t *eb = entries + (size_t)b * E;
        const int kb = (int)(gw - (long long)b * wpc) * EPW, kend = min(min(E, kb + EPW), off[N]);     // off[N] = number of valid entries
        if (kb < kend) {
            // Per-LANE prologue, so that the entry loop below carries no dependent scalar loads: lane l (and l + 64 when EPW = 128) owns
            // entry kb + l - its grouped row, the point that reads it (binary search of the offsets, all lanes at once), that point's entry
            // range and the relative coordinate.  The loop fetches them with v_readlane and has only the row loads left in flight.
            constexpr int EL = (EPW + 63) / 64;
            int le[EL], lt[EL], llo[EL], lhi[EL]; float rx[EL], ry[EL], rz[EL];
#pragma unroll
            for (int j = 0; j < EL; ++j) {
                const int kk = min(kb + lane + 64 * j, kend - 1);
                int lo = 0, hi = N;                             // point t with off[t] <= kk < off[t+1]
                while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (off[mid] <= kk) lo = mid; else hi = mid; }
                const int e = eb[kk];
                le[j] = e; lt[j] = lo; llo[j] = off[lo]; lhi[j] = off[lo + 1];
                const float *pp = xyz + ((size_t)b * N + lo) * 3, *cc = new_xyz + ((size_t)b * S + e / ns) * 3;
                rx[j] = pp[0] - cc[0]; ry[j] = pp[1] - cc[1]; rz[j] = pp[2] - cc[2];
            }
            int cur = -1, cur_lo = 0, cur_hi = 0;               // owner of acc[] (wave-uniform) and its entry range
            auto flush = [&]() {
                float *o = dG + ((size_t)b * N + cur) * ldo;
                // a point whose whole entry range lies inside this wave's range is nobody else's: plain stores.  Only the (at most two) points
                // cut by the range's ends are shared with the neighbouring waves and need the atomic.  (With one atomic per channel and
                // segment the kernel issued 4.2 M fp32 atomics per launch and ran at THEIR rate, not the memory system's.)
                const bool whole = cur_lo >= kb && cur_hi <= kend;              // uniform
#pragma unroll
                for (int i = 0; i < CPL; ++i) {
                    const int c = lane + 64 * i;
                    if (c < C) { if (whole) o[c] = acc[i]; else atomicAdd(o + c, acc[i]); }
                    acc[i] = 0.f;
                }
            };
            auto pick_i = [&](const int (&v)[EL], int idx) { int r = __builtin_amdgcn_readlane(v[0], idx & 63);
                                                             if (EL > 1 && idx >= 64) r = __builtin_amdgcn_readlane(v[EL - 1], idx & 63);
                                                             return r; };
            auto pick_f = [&](const float (&v)[EL], int idx) { int r = __builtin_amdgcn_readlane(__float_as_int(v[0]), idx & 63);
                                                               if (EL > 1 && idx >= 64) r = __builtin_amdgcn_readlane(__float_as_int(v[EL - 1]), idx & 63);
                                                               return __int_as_float(r); };
            const int cnt = kend - kb;
            for (int k = 0; k < cnt; k += UB) {
                size_t ro[UB]; float dx[UB], dy[UB], dzz[UB]; int tt[UB], tlo[UB], thi[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int idx = min(k + u, cnt - 1);        // uniform
                    tt[u] = pick_i(lt, idx); tlo[u] = pick_i(llo, idx); thi[u] = pick_i(lhi, idx);
                    ro[u] = (size_t)b * E + pick_i(le, idx);
                    dx[u] = pick_f(rx, idx); dy[u] = pick_f(ry, idx); dzz[u] = pick_f(rz, idx);
                }
                float d[UB][CPL];
#pragma unroll
                for (int i = 0; i < CPL; ++i) {
                    const int c = min(lane + 64 * i, C - 1);
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        const float g = dz[ro[u] * lddz + c], yy = y[ro[u] * ldy + c];
                        d[u][i] = __builtin_fmaf(cgs[i], (csc[i] * yy + csh[i] > 0.f) ? g : 0.f, __builtin_fmaf(cq[i], yy, cp[i]));
                    }
                }
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    if (k + u < cnt) {                          // uniform
                        if (tt[u] != cur) {
                            if (cur >= 0) flush();
                            cur = tt[u]; cur_lo = tlo[u]; cur_hi = thi[u];
                        }
#pragma unroll
                        for (int i = 0; i < CPL; ++i) {
                            acc[i] += d[u][i];
                            a0[i] = __builtin_fmaf(d[u][i], dx[u], a0[i]);
                            a1[i] = __builtin_fmaf(d[u][i], dy[u], a1[i]);
                            a2[i] = __builtin_fmaf(d[u][i], dzz[u], a2[i]);
                        }
                    }
                }
            }
            if (cur >= 0) flush();
        }
    }
#pragma unroll
    for (int i = 0; i < CPL; ++i) { red[0][wave][lane + 64 * i] = a0[i]; red[1][wave][lane + 64 * i] = a1[i]; red[2][wave][lane + 64 * i] = a2[i]; }
    __syncthreads();
    for (int u = threadIdx.x; u < 3 * C; u += 256) {
        const int e = u / C, c = u - e * C;
        double *o = dwx_slots + (size_t)(blockIdx.x % P2C_STAT_SLOTS) * 3 * C;
        atomicAdd(&o[u], (double)((red[e][0][c] + red[e][1][c]) + (red[e][2][c] + red[e][3][c])));
    }
}

extern "C" int p2c_group_linear_bwd_f32(const float *dz, int lddz, const float *y, int ldy, const float *coef, const int32_t *offsets,
                                        const int32_t *rows, const float *xyz, const float *new_xyz, int B, int N, int S, int nsample, int C,
                                        float *dG, int ldo, double *dwx_slots, void *stream)
{
    if (!dz || !y || !coef || !offsets || !rows || !xyz || !new_xyz || !dG || !dwx_slots || B <= 0 || N <= 0 || S <= 0 || nsample <= 0 ||
        C <= 0 || C > 256)
        return P2C_EINVAL;
    constexpr int EPW = 64;                                           // measured: 32 -> 69 us, 64 -> 57 us, 128 -> 88 us (SA2, 1M entries)
    const int wpc = p2c_cdiv((long long)S * nsample, EPW);            // waves per cloud, EPW entries each
    const int blocks = p2c_cdiv((long long)B * wpc, 4);
    hipStream_t s = (hipStream_t)stream;
#define P2C_GLB(CPL_) hipLaunchKernelGGL((group_linear_bwd_kernel<CPL_, EPW>), dim3(blocks), dim3(256), 0, s, dz, lddz, y, ldy, coef, offsets, rows, xyz, new_xyz, B, N, S, nsample, C, wpc, dG, ldo, dwx_slots)
    if (C <= 64) P2C_GLB(1);
    else if (C <= 128) P2C_GLB(2);
    else P2C_GLB(4);
#undef P2C_GLB
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}
