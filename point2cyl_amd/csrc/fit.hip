// fit.hip -- differentiable extrusion-cylinder fitting on gfx950.
//
// estimate_extrusion_axis (data_utils.py:99-177) in the reference materialises two N x N diagonal
// matrices per segment and runs LAPACK syev per 3x3.  Here: ONE streaming pass over (X, Wb, Wc) per cloud
// accumulates, for every segment k, the 6+6 unique entries of B^T B = sum wb^2 x x^T and
// C^T C = sum wc^2 x x^T (76 B/point of HBM traffic, nothing else), followed in the same kernel by a
// cyclic-Jacobi eigen-solve in fp64 registers that returns the eigenvector of the smallest signed
// eigenvalue of B^T B/sb^2 - C^T C/sc^2 (the reference's v[:,:,0] of an ascending symeig).
// Thread (g,k) of the 1024-thread workgroup owns segment k and the point slice g, g+G, g+2G, ... so a wave
// reads whole 32-byte weight rows and every lane is busy for any K.
#include "common.h"

#define FIT_THREADS 1024
#define FIT_MAXK 16

#include "eigh3.h"

template <int THREADS>      // 1024: one workgroup per CU (few clouds: each gets a whole CU); 512: two per CU, one streams while the other is in its
                            // serial tail (LDS reduction, fp64 eigen-solve on K threads) - the fitting-only path runs 1250 clouds
__global__ void __launch_bounds__(THREADS) axis_kernel(const float *__restrict__ X, const float *__restrict__ Wb,
                                                           const float *__restrict__ Wc, const int64_t *__restrict__ bb_gt,
                                                           const int64_t *__restrict__ inst_gt, int normalize, int N, int K,
                                                           float *__restrict__ axis_out, float *__restrict__ eig_out,
                                                           double *__restrict__ axis64_out)
{
    __shared__ float red[14][THREADS];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int G = THREADS / K;                // point slices
    const int g = tid / K, k = tid - g * K;
    float acc[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) acc[i] = 0.f;
    if (g < G) {
        const float *x = X + (size_t)b * N * 3;
        const float *wb = Wb + (size_t)b * N * K, *wc = Wc + (size_t)b * N * K;
#pragma unroll 8
        for (int n = g; n < N; n += G) {          // 8 iterations' loads in flight: the kernel streams 76 B per point and nothing else
            const float x0 = x[n * 3 + 0], x1 = x[n * 3 + 1], x2 = x[n * 3 + 2];
            const float b_ = wb[(size_t)n * K + k], c_ = wc[(size_t)n * K + k];
            const float b2 = b_ * b_, c2 = c_ * c_;
            const float p00 = x0 * x0, p01 = x0 * x1, p02 = x0 * x2, p11 = x1 * x1, p12 = x1 * x2, p22 = x2 * x2;
            acc[0] += b2 * p00; acc[1] += b2 * p01; acc[2] += b2 * p02; acc[3] += b2 * p11; acc[4] += b2 * p12; acc[5] += b2 * p22;
            acc[6] += c2 * p00; acc[7] += c2 * p01; acc[8] += c2 * p02; acc[9] += c2 * p11; acc[10] += c2 * p12; acc[11] += c2 * p22;
            if (normalize) {
                const bool mine = inst_gt[(size_t)b * N + n] == k;
                const int64_t bbv = bb_gt[(size_t)b * N + n];
                acc[12] += (mine && bbv == 0) ? 1.f : 0.f;
                acc[13] += (mine && bbv == 1) ? 1.f : 0.f;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 14; ++i) red[i][tid] = acc[i];
    __syncthreads();
    // K*14 threads: fp64 sum over the G slices of one (k, entry).  The sums stay in fp64 from here on (difference of the two scatter
    // matrices, Jacobi sweeps): the fitted axis is then as close to the exact one as its fp32 inputs allow, which is what the eval
    // metric (an acos next to its clamp, eval.py:398) needs; the reference's fp32 bmm is one rounding further away from it
    __shared__ double tot[14][FIT_MAXK];
    const bool reducer = tid < K * 14;
    const int rk = tid / 14, re = tid - rk * 14;
    if (reducer) {
        double rs = 0.0;
        for (int gg = 0; gg < G; ++gg) rs += (double)red[re][gg * K + rk];
        tot[re][rk] = rs;
    }
    __syncthreads();
    if (tid >= K) return;
    double isb2 = 1.0, isc2 = 1.0;
    if (normalize) {
        const float sb = sqrtf((float)tot[12][tid]) + 1.0f, sc = sqrtf((float)tot[13][tid]) + 1.0f;   // data_utils.py:139-160
        isb2 = 1.0 / ((double)sb * (double)sb);
        isc2 = 1.0 / ((double)sc * (double)sc);
    }
    double a[6], lam[3], v[3][3];
    for (int e = 0; e < 6; ++e) a[e] = tot[e][tid] * isb2 - tot[6 + e][tid] * isc2;
    p2c_eigh3(a, lam, v);
    // canonical sign: largest-magnitude component positive
    int big = 0;
    if (fabs(v[0][1]) > fabs(v[0][big])) big = 1;
    if (fabs(v[0][2]) > fabs(v[0][big])) big = 2;
    const double sgn = v[0][big] < 0 ? -1.0 : 1.0;
    float *o = axis_out + ((size_t)b * K + tid) * 3;
    o[0] = (float)(sgn * v[0][0]); o[1] = (float)(sgn * v[0][1]); o[2] = (float)(sgn * v[0][2]);
    if (axis64_out) {           // the unit eigenvector as the fp64 Jacobi left it: the eval metric (eval.py:398-405) is an acos next to its clamp
        double *o64 = axis64_out + ((size_t)b * K + tid) * 3;
        o64[0] = sgn * v[0][0]; o64[1] = sgn * v[0][1]; o64[2] = sgn * v[0][2];
    }
    if (eig_out) {
        float *e = eig_out + ((size_t)b * K + tid) * 12;
        e[0] = (float)lam[0]; e[1] = (float)lam[1]; e[2] = (float)lam[2];
        e[3] = (float)v[1][0]; e[4] = (float)v[1][1]; e[5] = (float)v[1][2];
        e[6] = (float)v[2][0]; e[7] = (float)v[2][1]; e[8] = (float)v[2][2];
        e[9] = (float)isb2; e[10] = (float)isc2; e[11] = (float)sgn;
    }
}

extern "C" int p2c_extrusion_axis_f32(const float *X, const float *Wb, const float *Wc, const int64_t *bb_gt, const int64_t *inst_gt,
                                      int normalize, int B, int N, int K, float *axis_out, float *eig_out, double *axis64_out, void *stream)
{
    if (!X || !Wb || !Wc || !axis_out || B <= 0 || N <= 0 || K <= 0 || K > FIT_MAXK) return P2C_EINVAL;
    if (normalize && (!bb_gt || !inst_gt)) return P2C_EINVAL;
    if (B >= 1024)
        hipLaunchKernelGGL(axis_kernel<256>, dim3(B), dim3(256), 0, (hipStream_t)stream, X, Wb, Wc, bb_gt, inst_gt, normalize, N, K, axis_out, eig_out, axis64_out);
    else if (B >= 512)
        hipLaunchKernelGGL(axis_kernel<512>, dim3(B), dim3(512), 0, (hipStream_t)stream, X, Wb, Wc, bb_gt, inst_gt, normalize, N, K, axis_out, eig_out, axis64_out);
    else
        hipLaunchKernelGGL(axis_kernel<FIT_THREADS>, dim3(B), dim3(FIT_THREADS), 0, (hipStream_t)stream, X, Wb, Wc, bb_gt, inst_gt, normalize, N, K,
                           axis_out, eig_out, axis64_out);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// Backward.  With a = sgn*v0, g = sgn*dL/da:  dL/dM = sum_{j=1,2} (v_j.g)/(lam0-lam_j) v_j v0^T; only its
// symmetric part Gs matters because M is built symmetrically:  M = sum_n (wb^2/sb^2 - wc^2/sc^2) x x^T
//   dL/dwb_n =  2 wb_n/sb^2 (x^T Gs x),  dL/dwc_n = -2 wc_n/sc^2 (x^T Gs x),  dL/dx_n = sum_k alpha_nk 2 Gs_k x.
// A zero upstream gradient yields exactly zero (masked segments), also for degenerate spectra.
__global__ void __launch_bounds__(256) axis_bwd_kernel(const float *__restrict__ daxis, const float *__restrict__ axis,
                                                       const float *__restrict__ eig, const float *__restrict__ X,
                                                       const float *__restrict__ Wb, const float *__restrict__ Wc, int N, int K,
                                                       float *__restrict__ dX, float *__restrict__ dWb, float *__restrict__ dWc)
{
    __shared__ float sG[FIT_MAXK][8];   // Gs (6), 1/sb^2, 1/sc^2
    const int b = blockIdx.y;
    if (threadIdx.x < K) {
        const int k = threadIdx.x;
        const float *e = eig + ((size_t)b * K + k) * 12;
        const float *a = axis + ((size_t)b * K + k) * 3, *da = daxis + ((size_t)b * K + k) * 3;
        const double sgn = e[11];
        const double v0[3] = {sgn * a[0], sgn * a[1], sgn * a[2]};
        const double g[3] = {sgn * da[0], sgn * da[1], sgn * da[2]};
        double Gm[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        if (g[0] != 0.0 || g[1] != 0.0 || g[2] != 0.0) {
            for (int j = 1; j <= 2; ++j) {
                const double vj[3] = {e[3 * j], e[3 * j + 1], e[3 * j + 2]};
                const double cj = (vj[0] * g[0] + vj[1] * g[1] + vj[2] * g[2]) / ((double)e[0] - (double)e[j]);
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) Gm[r][c] += cj * vj[r] * v0[c];
            }
        }
        sG[k][0] = (float)Gm[0][0];
        sG[k][1] = (float)(0.5 * (Gm[0][1] + Gm[1][0]));
        sG[k][2] = (float)(0.5 * (Gm[0][2] + Gm[2][0]));
        sG[k][3] = (float)Gm[1][1];
        sG[k][4] = (float)(0.5 * (Gm[1][2] + Gm[2][1]));
        sG[k][5] = (float)Gm[2][2];
        sG[k][6] = e[9];
        sG[k][7] = e[10];
    }
    __syncthreads();
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const size_t pn = (size_t)b * N + n;
    const float x0 = X[pn * 3 + 0], x1 = X[pn * 3 + 1], x2 = X[pn * 3 + 2];
    float d0 = 0.f, d1 = 0.f, d2 = 0.f;
    for (int k = 0; k < K; ++k) {
        const float g00 = sG[k][0], g01 = sG[k][1], g02 = sG[k][2], g11 = sG[k][3], g12 = sG[k][4], g22 = sG[k][5];
        const float gx0 = g00 * x0 + g01 * x1 + g02 * x2, gx1 = g01 * x0 + g11 * x1 + g12 * x2, gx2 = g02 * x0 + g12 * x1 + g22 * x2;
        const float quad = x0 * gx0 + x1 * gx1 + x2 * gx2;
        const float wb = Wb[pn * K + k], wc = Wc[pn * K + k];
        dWb[pn * K + k] = 2.f * wb * sG[k][6] * quad;
        dWc[pn * K + k] = -2.f * wc * sG[k][7] * quad;
        const float alpha = 2.f * (wb * wb * sG[k][6] - wc * wc * sG[k][7]);
        d0 += alpha * gx0; d1 += alpha * gx1; d2 += alpha * gx2;
    }
    dX[pn * 3 + 0] = d0; dX[pn * 3 + 1] = d1; dX[pn * 3 + 2] = d2;
}

extern "C" int p2c_extrusion_axis_bwd_f32(const float *daxis, const float *axis, const float *eig, const float *X, const float *Wb,
                                          const float *Wc, int B, int N, int K, float *dX, float *dWb, float *dWc, void *stream)
{
    if (!daxis || !axis || !eig || !X || !Wb || !Wc || !dX || !dWb || !dWc || K <= 0 || K > FIT_MAXK) return P2C_EINVAL;
    hipLaunchKernelGGL(axis_bwd_kernel, dim3(p2c_cdiv(N, 256), B), dim3(256), 0, (hipStream_t)stream, daxis, axis, eig, X, Wb, Wc, N, K, dX, dWb,
                       dWc);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// ------------------------------------------------------------------------------------------------
// Centres.  MODE 0: c[b,k] = (1/N) sum_n W[b,n,k] p[b,n]  (data_utils.py:253-266, a mean over N).
//           MODE 1: hard centroids from labels (eval.py:409-436): mean of points with label == k;
//                   found = count > 1 (a single point counts as "not found", zeros).
// Same (slice, k) thread mapping as the axis kernel; one workgroup per cloud.
// ------------------------------------------------------------------------------------------------
template <int MODE, int THREADS>      // THREADS as in axis_kernel: 1024 for a few clouds, 256 when there are enough to keep four per CU busy
__global__ void __launch_bounds__(THREADS) centers_kernel(const float *__restrict__ W, const int64_t *__restrict__ label,
                                                          const float *__restrict__ P, int N, int K, float *__restrict__ out,
                                                          float *__restrict__ found)
{
    __shared__ float red[4][THREADS];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int G = THREADS / K;
    const int g = tid / K, k = tid - g * K;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, cnt = 0.f;
    if (g < G) {
        const float *p = P + (size_t)b * N * 3;
        for (int n = g; n < N; n += G) {
            float w;
            if (MODE == 0) w = W[((size_t)b * N + n) * K + k];
            else w = (label[(size_t)b * N + n] == k) ? 1.f : 0.f;
            a0 += w * p[n * 3 + 0]; a1 += w * p[n * 3 + 1]; a2 += w * p[n * 3 + 2];
            cnt += w;
        }
    }
    red[0][tid] = a0; red[1][tid] = a1; red[2][tid] = a2; red[3][tid] = cnt;
    __syncthreads();
    const bool reducer = tid < K * 4;
    const int rk = tid >> 2, re = tid & 3;
    double rs = 0.0;
    if (reducer)
        for (int gg = 0; gg < G; ++gg) rs += (double)red[re][gg * K + rk];
    __syncthreads();
    if (reducer) red[re][rk] = (float)rs;
    __syncthreads();
    if (tid < K) {
        float *o = out + ((size_t)b * K + tid) * 3;
        if (MODE == 0) {
            const float inv = 1.0f / (float)N;
            o[0] = red[0][tid] * inv; o[1] = red[1][tid] * inv; o[2] = red[2][tid] * inv;
        } else {
            const float c = red[3][tid];
            const bool ok = c > 1.f;
            o[0] = ok ? red[0][tid] / c : 0.f; o[1] = ok ? red[1][tid] / c : 0.f; o[2] = ok ? red[2][tid] / c : 0.f;
            found[(size_t)b * K + tid] = ok ? 1.f : 0.f;
        }
    }
}

// Hard centroids, one thread per POINT (K compile-time): label and coordinates are read once (20 B per point) instead of once per
// segment by the (slice, k) mapping above; per-thread bins over k, LDS tree over the threads.  Same result (sums of the same fp32
// values in a different order; fp64 across threads).
template <int KK, int THREADS>
__global__ void __launch_bounds__(THREADS) centroids_by_point_kernel(const int64_t *__restrict__ label, const float *__restrict__ P, int N,
                                                                    float *__restrict__ out, float *__restrict__ found)
{
    __shared__ float red[4 * KK][THREADS];
    const int b = blockIdx.x, tid = threadIdx.x;
    float acc[KK][4];
#pragma unroll
    for (int k = 0; k < KK; ++k) acc[k][0] = acc[k][1] = acc[k][2] = acc[k][3] = 0.f;
    const float *p = P + (size_t)b * N * 3;
    const int64_t *lab = label + (size_t)b * N;
#pragma unroll 4
    for (int n = tid; n < N; n += THREADS) {
        const int l = (int)lab[n];
        const float x = p[n * 3 + 0], y = p[n * 3 + 1], z = p[n * 3 + 2];
#pragma unroll
        for (int k = 0; k < KK; ++k) {
            const float w = l == k ? 1.f : 0.f;
            acc[k][0] += w * x; acc[k][1] += w * y; acc[k][2] += w * z; acc[k][3] += w;
        }
    }
#pragma unroll
    for (int k = 0; k < KK; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[k * 4 + e][tid] = acc[k][e];
    __syncthreads();
    // 4*KK rows of THREADS values: 8 threads per row (fp64), then lane 0 of each group finishes
    __shared__ double part[4 * KK][8];
    const int row = tid / 8, sub = tid % 8;
    if (row < 4 * KK) {
        double rs = 0.0;
        for (int t = sub; t < THREADS; t += 8) rs += (double)red[row][t];
        part[row][sub] = rs;
    }
    __syncthreads();
    if (tid < KK) {
        double v[4];
        for (int e = 0; e < 4; ++e) {
            double rs = 0.0;
            for (int q = 0; q < 8; ++q) rs += part[tid * 4 + e][q];
            v[e] = rs;
        }
        const float c = (float)v[3];
        const bool ok = c > 1.f;
        float *o = out + ((size_t)b * KK + tid) * 3;
        o[0] = ok ? (float)v[0] / c : 0.f; o[1] = ok ? (float)v[1] / c : 0.f; o[2] = ok ? (float)v[2] / c : 0.f;
        found[(size_t)b * KK + tid] = ok ? 1.f : 0.f;
    }
}

extern "C" int p2c_extrusion_centers_f32(const float *W, const float *P, int B, int N, int K, float *centers_out, void *stream)
{
    if (!W || !P || !centers_out || K <= 0 || K > FIT_MAXK) return P2C_EINVAL;
    if (B >= 1024)
        hipLaunchKernelGGL((centers_kernel<0, 256>), dim3(B), dim3(256), 0, (hipStream_t)stream, W, (const int64_t *)nullptr, P, N, K, centers_out,
                           (float *)nullptr);
    else
        hipLaunchKernelGGL((centers_kernel<0, FIT_THREADS>), dim3(B), dim3(FIT_THREADS), 0, (hipStream_t)stream, W, (const int64_t *)nullptr, P, N, K,
                           centers_out, (float *)nullptr);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

extern "C" int p2c_segment_centroids_f32(const float *P, const int64_t *label, int B, int N, int K, float *centroids_out,
                                         float *found_out, void *stream)
{
    if (!P || !label || !centroids_out || !found_out || K <= 0 || K > FIT_MAXK) return P2C_EINVAL;
    if (K == 8 && B >= 256)
        hipLaunchKernelGGL((centroids_by_point_kernel<8, 256>), dim3(B), dim3(256), 0, (hipStream_t)stream, label, P, N, centroids_out, found_out);
    else if (B >= 1024)
        hipLaunchKernelGGL((centers_kernel<1, 256>), dim3(B), dim3(256), 0, (hipStream_t)stream, (const float *)nullptr, label, P, N, K,
                           centroids_out, found_out);
    else
        hipLaunchKernelGGL((centers_kernel<1, FIT_THREADS>), dim3(B), dim3(FIT_THREADS), 0, (hipStream_t)stream, (const float *)nullptr, label, P, N, K,
                           centroids_out, found_out);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// dW[b,n,k] = (1/N) dc[b,k,:] . p[b,n,:]
__global__ void __launch_bounds__(256) centers_bwd_kernel(const float *__restrict__ dC, const float *__restrict__ P, int N, int K,
                                                          float *__restrict__ dW)
{
    const int b = blockIdx.y;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)N * K) return;
    const int n = (int)(e / K), k = (int)(e - (long long)n * K);
    const float *p = P + ((size_t)b * N + n) * 3, *d = dC + ((size_t)b * K + k) * 3;
    dW[(size_t)b * N * K + e] = (d[0] * p[0] + d[1] * p[1] + d[2] * p[2]) / (float)N;
}

extern "C" int p2c_extrusion_centers_bwd_f32(const float *dcenters, const float *P, int B, int N, int K, float *dW, void *stream)
{
    if (!dcenters || !P || !dW) return P2C_EINVAL;
    hipLaunchKernelGGL(centers_bwd_kernel, dim3(p2c_cdiv((long long)N * K, 256), B), dim3(256), 0, (hipStream_t)stream, dcenters, P, N, K, dW);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// ------------------------------------------------------------------------------------------------
// Extents (data_utils.py:1650-1730).  One workgroup per CLOUD builds the K ascending lists of barrel points in LDS in
// a single pass over the labels (ballot prefix sums, every wave owns a contiguous range of points so the lists come
// out in index order with two barriers in total), then projects the S sampled points of every segment (the caller's
// randint draws index those lists) on the axis and takes min / max.  A second tiny kernel applies the reference's
// batch-level rule (:1671: a segment with <= 1 barrel point in the WHOLE batch is skipped, extents stay 0) and its
// per-cloud rule (:1690: <= 1 point => projected points are zeros).
// (First version: one workgroup per (cloud, segment), each rescanning all N labels with four barriers per 256 points -
//  1.09 ms for 1250 clouds x 8 segments; this one reads every label once.)
// ------------------------------------------------------------------------------------------------
#ifdef P2C_FIT_TRACE       // tools/fit_trace.py: shader-clock stamps of workgroup 0 at the phase boundaries
__device__ unsigned long long p2c_fit_stamps[40];
extern "C" int p2c_fit_trace_read(void *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(p2c_fit_stamps), sizeof(p2c_fit_stamps)) == hipSuccess ? 0 : 1; }
#ifndef P2C_FIT_TRACE_WG
#define P2C_FIT_TRACE_WG 0            // (-DP2C_FIT_TRACE_WG=1000: a workgroup of a later round, when the CUs no longer stream in lockstep)
#endif
#define FIT_TR(i) do { if (blockIdx.x == P2C_FIT_TRACE_WG && threadIdx.x == 0) p2c_fit_stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define FIT_TR(i) do { } while (0)
#endif

#define EXT_MAXN 32768
#define EXT_THREADS 1024
#define EXT_WAVES (EXT_THREADS / 64)
#define EXT_MAXCH 8                       // 64-point chunks a wave keeps in registers per sweep: N <= EXT_WAVES*64*EXT_MAXCH per sweep

// passes 1-2 of both kernels below: the K ascending lists of barrel points of cloud b -> list[start[k] .. start[k+1])
// key_of(n) -> segment of barrel point n, -1 for every other point
template <int WAVES = EXT_WAVES, class KeyF>
__device__ __forceinline__ void ext_build_lists_by(KeyF key_of, int N, int K, int *list, int (*wcnt)[FIT_MAXK], int *start)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = ((N + WAVES - 1) / WAVES + 63) / 64 * 64;     // points per wave, a multiple of 64
    const int n_begin = wave * per, n_end = min(N, n_begin + per);
    // pass 1: key of every point of this wave's range (-1 = not a barrel point of any segment); per-segment counts
    int key[EXT_MAXCH];
    int cntk[FIT_MAXK];
#pragma unroll
    for (int k = 0; k < FIT_MAXK; ++k) cntk[k] = 0;
    const int nch = (max(n_end - n_begin, 0) + 63) / 64;
    for (int c0 = 0; c0 < nch; c0 += EXT_MAXCH) {
#pragma unroll
        for (int c = 0; c < EXT_MAXCH; ++c) {
            const int n = n_begin + (c0 + c) * 64 + lane;
            int kk = -1;
            if (c0 + c < nch && n < n_end) kk = key_of(n);
            if (c0 == 0) key[c] = kk;                  // the first sweep stays in registers for pass 2
#pragma unroll
            for (int k = 0; k < FIT_MAXK; ++k)
                if (k < K) cntk[k] += __popcll(__ballot(kk == k));
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < FIT_MAXK; ++k)
            if (k < K) wcnt[wave][k] = cntk[k];
    }
    FIT_TR(11);
    __syncthreads();
    FIT_TR(12);
    // counts -> offsets, in place and in parallel (a single thread walking the K x 16 table, then every thread its K x wave prefix,
    // were two chains of dependent LDS reads: 10 us of the 80 a cloud takes in the one-pass kernel)
    int mine = 0, before = 0;
    const int pw = tid / K, pk = tid - pw * K;                 // thread (wave pw, segment pk) of the table
    if (tid < WAVES * K) {
        mine = wcnt[pw][pk];
        for (int w = 0; w < pw; ++w) before += wcnt[w][pk];
    }
    __syncthreads();
    if (tid < WAVES * K) {
        wcnt[pw][pk] = before;                                 // now: barrel points of segment pk in the waves before pw
        if (pw == WAVES - 1) start[pk + 1] = before + mine;    // segment totals, prefix-summed below
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        start[0] = 0;
        for (int k = 0; k < K; ++k) { run += start[k + 1]; start[k + 1] = run; }
    }
    __syncthreads();
    FIT_TR(13);
    // pass 2: ascending lists
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    int base[FIT_MAXK];
#pragma unroll
    for (int k = 0; k < FIT_MAXK; ++k) base[k] = k < K ? start[k] + wcnt[wave][k] : 0;
    for (int c0 = 0; c0 < nch; c0 += EXT_MAXCH) {
#pragma unroll
        for (int c = 0; c < EXT_MAXCH; ++c) {
            const int n = n_begin + (c0 + c) * 64 + lane;
            int kk = -1;
            if (c0 == 0) {
                kk = key[c];
            } else if (c0 + c < nch && n < n_end) {
                kk = key_of(n);
            }
            // the slot of this lane's point: ONE store per 64 points (a store under `if (kk == k)` inside the segment loop was eight
            // exec-masked stores per chunk: 11 k of a cloud's cycles, tools/fit_trace.py)
            int pos = 0;
#pragma unroll
            for (int k = 0; k < FIT_MAXK; ++k) {
                if (k < K) {
                    const unsigned long long m = __ballot(kk == k);
                    pos = kk == k ? base[k] + (int)__popcll(m & lt_mask) : pos;
                    base[k] += __popcll(m);
                }
            }
            if (kk >= 0) list[pos] = n;
        }
    }
    __syncthreads();
}

// Inclusive scan over the 64 lanes of PACKED counters (fields that never carry into each other): Hillis-Steele inside each row of 16 with
// four row_shr steps, then the two row broadcasts - six v_add_u32_dpp.
__device__ __forceinline__ unsigned p2c_wave_incl_scan_u32(unsigned v)
{
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);      // row_shr:1
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);      // row_shr:2
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);      // row_shr:4
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);      // row_shr:8
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);      // row_bcast:15 -> rows 1, 3
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);      // row_bcast:31 -> rows 2, 3
    return v;
}

// The same lists from ONE pass of per-lane work: a point's key becomes a one-hot BYTE counter (four keys per 32-bit word, NW words), the
// inclusive scan of those words over the wave gives every lane the number of equal keys at or below it (<= 64: a byte) and lane 63 the
// chunk's totals; the totals of the earlier chunks ride in uniform registers as 16-bit fields.  A lane so knows its point's slot relative
// to (segment start + the earlier waves' points) before the barrier, and the placement after it is two table reads and a store - against
// K ballots with a masked population count and a select per 64 points in BOTH passes (the placement pass was 9 k of a cloud's 95 k cycles,
// tools/fit_trace.py).  One sweep only: N <= WAVES * 64 * EXT_MAXCH, keys < 4 * NW.
template <int WAVES, int NW, class KeyF>
__device__ __forceinline__ void ext_build_lists_scan(KeyF key_of, int N, int K, int *list, int (*wcnt)[FIT_MAXK], int *start)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = ((N + WAVES - 1) / WAVES + 63) / 64 * 64;     // points per wave, a multiple of 64
    const int n_begin = wave * per, n_end = min(N, n_begin + per);
    const int nch = (max(n_end - n_begin, 0) + 63) / 64;
    int key[EXT_MAXCH], prel[EXT_MAXCH];
    unsigned run_e[NW], run_o[NW];             // uniform: keys 4w / 4w+2 and 4w+1 / 4w+3 of this wave so far, 16 bits each
#pragma unroll
    for (int w = 0; w < NW; ++w) run_e[w] = run_o[w] = 0u;
#pragma unroll
    for (int c = 0; c < EXT_MAXCH; ++c) {
        const int n = n_begin + c * 64 + lane;
        int kk = -1;
        if (c < nch && n < n_end) kk = key_of(n);
        key[c] = kk;
        const unsigned one = kk >= 0 ? 1u << ((kk & 3) * 8) : 0u;
        const int kw = kk >> 2;                 // (-1 for no key: matches no word)
        unsigned mine = 0u, before = 0u;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const unsigned v = p2c_wave_incl_scan_u32(kw == w ? one : 0u);
            const unsigned tot = (unsigned)__builtin_amdgcn_readlane((int)v, 63);
            if (kw == w) {
                mine = v;
                before = (kk & 1) ? run_o[w] : run_e[w];
            }
            run_e[w] += tot & 0x00ff00ffu;
            run_o[w] += (tot >> 8) & 0x00ff00ffu;
        }
        prel[c] = (int)((before >> ((kk & 2) * 8)) & 0xffffu) + (int)((mine >> ((kk & 3) * 8)) & 0xffu) - 1;
    }
    {
        unsigned r = 0u;                        // lane k: this wave's count of key k
#pragma unroll
        for (int w = 0; w < NW; ++w)
            if ((lane >> 2) == w) r = (lane & 1) ? run_o[w] : run_e[w];
        if (lane < K) wcnt[wave][lane] = (int)((r >> ((lane & 2) * 8)) & 0xffffu);
    }
    FIT_TR(11);
    __syncthreads();
    FIT_TR(12);
    int mine = 0, before = 0;
    const int pw = tid / K, pk = tid - pw * K;                 // thread (wave pw, segment pk) of the table
    if (tid < WAVES * K) {
        mine = wcnt[pw][pk];
        for (int w = 0; w < pw; ++w) before += wcnt[w][pk];
    }
    __syncthreads();
    if (tid < WAVES * K) {
        wcnt[pw][pk] = before;                                 // now: barrel points of segment pk in the waves before pw
        if (pw == WAVES - 1) start[pk + 1] = before + mine;    // segment totals, prefix-summed below
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        start[0] = 0;
        for (int k = 0; k < K; ++k) { run += start[k + 1]; start[k + 1] = run; }
    }
    __syncthreads();
    FIT_TR(13);
#pragma unroll
    for (int c = 0; c < EXT_MAXCH; ++c) {
        const int kk = key[c];
        if (kk >= 0) list[start[kk] + wcnt[wave][kk] + prel[c]] = n_begin + c * 64 + lane;
    }
    __syncthreads();
}

template <int WAVES = EXT_WAVES, class KeyF>
__device__ __forceinline__ void ext_build_lists_any(KeyF key_of, int N, int K, int *list, int (*wcnt)[FIT_MAXK], int *start)
{
    if (N <= WAVES * 64 * EXT_MAXCH) {          // (uniform)
        if (K <= 8) ext_build_lists_scan<WAVES, 2>(key_of, N, K, list, wcnt, start);
        else ext_build_lists_scan<WAVES, 4>(key_of, N, K, list, wcnt, start);
    } else {
        ext_build_lists_by<WAVES>(key_of, N, K, list, wcnt, start);
    }
}

__device__ __forceinline__ void ext_build_lists(const int64_t *__restrict__ sg, const int64_t *__restrict__ bl, int N, int K, int *list,
                                                int (*wcnt)[FIT_MAXK], int *start)
{
    ext_build_lists_any([&](int n) { const int64_t sv = sg[n]; return (bl[n] == 0 && sv >= 0 && sv < K) ? (int)sv : -1; }, N, K, list, wcnt, start);
}

__global__ void __launch_bounds__(EXT_THREADS) extents_kernel(const float *__restrict__ P, const int64_t *__restrict__ seg,
                                                              const int64_t *__restrict__ bb, const float *__restrict__ axes,
                                                              const float *__restrict__ centers, const int64_t *__restrict__ rand_idx, int N, int K,
                                                              int S, float *__restrict__ ext_tmp, int *__restrict__ counts)
{
    extern __shared__ int list[];                       // N ints, partitioned by segment
    __shared__ int wcnt[EXT_WAVES][FIT_MAXK];           // barrel points of segment k in wave w's range
    __shared__ int start[FIT_MAXK + 1];
    __shared__ float rmin[2 * EXT_WAVES], rmax[2 * EXT_WAVES];       // one slot per (segment, chunk) task: K * nch <= max(K, 16)
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    ext_build_lists(seg + (size_t)b * N, bb + (size_t)b * N, N, K, list, wcnt, start);
    // pass 3: project the samples of every segment.  The K x S samples are dealt to the 16 waves as (segment, chunk) tasks - for
    // K = 8 two waves per segment - so that the min / max of a task is a wave reduction and the whole pass needs ONE barrier
    // (a loop over the segments with a workgroup-wide reduction each cost 2 K barriers: 0.22 ms for 1250 clouds).
    {
        const int nch = K <= EXT_WAVES ? EXT_WAVES / K : 1;                 // chunks per segment
        const int per_chunk = (S + nch - 1) / nch;
        for (int t = wave; t < K * nch; t += EXT_WAVES) {
            const int k = t / nch, ch = t - k * nch;
            const int cnt = start[k + 1] - start[k];
            const float *a = axes + ((size_t)b * K + k) * 3, *c = centers + ((size_t)b * K + k) * 3;
            const float a0 = a[0], a1 = a[1], a2 = a[2], c0_ = c[0], c1_ = c[1], c2_ = c[2];
            const int64_t *ri = rand_idx + ((size_t)b * K + k) * S;
            const int *lk = list + start[k];
            float lo = INFINITY, hi = -INFINITY;
            const int s_end = min(S, (ch + 1) * per_chunk);
            constexpr int RU = 8;                       // draws, then points, of 8 x 64 samples in flight (one dependent chain per 64 before)
            for (int s0 = ch * per_chunk; s0 < s_end; s0 += 64 * RU) {
                int r[RU];
#pragma unroll
                for (int u = 0; u < RU; ++u) { const int s = s0 + u * 64 + lane; r[u] = (s < s_end && cnt > 1) ? (int)ri[s] : 0; }
                float px[RU], py[RU], pz[RU];
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    px[u] = py[u] = pz[u] = 0.f;
                    if (cnt > 1 && s0 + u * 64 + lane < s_end) {
                        const float *pp = P + ((size_t)b * N + lk[r[u]]) * 3;
                        px[u] = pp[0]; py[u] = pp[1]; pz[u] = pp[2];
                    }
                }
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    if (s0 + u * 64 + lane < s_end) {
                        const float dx = px[u] - c0_, dy = py[u] - c1_, dz = pz[u] - c2_;
                        const float tt = __builtin_fmaf(dz, a2, __builtin_fmaf(dy, a1, dx * a0));
                        lo = fminf(lo, tt); hi = fmaxf(hi, tt);
                    }
                }
            }
            for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
            if (lane == 0) { rmin[t] = lo; rmax[t] = hi; }
        }
        __syncthreads();
        if (tid < K) {
            float l2 = rmin[tid * nch], h2 = rmax[tid * nch];
            for (int ch = 1; ch < nch; ++ch) { l2 = fminf(l2, rmin[tid * nch + ch]); h2 = fmaxf(h2, rmax[tid * nch + ch]); }
            ext_tmp[((size_t)b * K + tid) * 2 + 0] = l2;
            ext_tmp[((size_t)b * K + tid) * 2 + 1] = h2;
            counts[b * K + tid] = start[tid + 1] - start[tid];
        }
    }
}

// one workgroup per segment k: batch-wide count (the :1671 rule), then the (K,B,2) layout and the found mask
__global__ void __launch_bounds__(256) extents_finish_kernel(const float *__restrict__ ext_tmp, const int *__restrict__ counts, int B, int K,
                                                             float *__restrict__ extents, float *__restrict__ found)
{
    __shared__ long long part[4];
    const int k = blockIdx.x, tid = threadIdx.x;
    long long t = 0;
    for (int b = tid; b < B; b += 256) t += counts[b * K + k];
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    if ((tid & 63) == 0) part[tid >> 6] = t;
    __syncthreads();
    const bool seg_ok = part[0] + part[1] + part[2] + part[3] > 1;
    for (int b = tid; b < B; b += 256) {
        extents[((size_t)k * B + b) * 2 + 0] = seg_ok ? ext_tmp[((size_t)b * K + k) * 2 + 0] : 0.f;
        extents[((size_t)k * B + b) * 2 + 1] = seg_ok ? ext_tmp[((size_t)b * K + k) * 2 + 1] : 0.f;
        found[(size_t)b * K + k] = (seg_ok && counts[b * K + k] > 1) ? 1.f : 0.f;
    }
}

extern "C" size_t p2c_extents_ws_bytes(int B, int K) { return (size_t)B * K * (2 * sizeof(float) + sizeof(int)); }

extern "C" int p2c_extrusion_extents_f32(const float *P, const int64_t *seg, const int64_t *bb, const float *axes, const float *centers,
                                         const int64_t *rand_idx, int B, int N, int K, int S, float *extents_out, float *found_out,
                                         void *ws, void *stream)
{
    if (!P || !seg || !bb || !axes || !centers || !rand_idx || !extents_out || !found_out || !ws || N > EXT_MAXN || K <= 0 || K > FIT_MAXK || S <= 0)
        return P2C_EINVAL;
    float *ext_tmp = (float *)ws;
    int *counts = (int *)(ext_tmp + (size_t)B * K * 2);
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)N * sizeof(int);
    (void)hipFuncSetAttribute((const void *)extents_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(extents_kernel, dim3(B), dim3(EXT_THREADS), lds, s, P, seg, bb, axes, centers, rand_idx, N, K, S, ext_tmp, counts);
    hipLaunchKernelGGL(extents_finish_kernel, dim3(K), dim3(256), 0, s, ext_tmp, counts, B, K, extents_out, found_out);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// ------------------------------------------------------------------------------------------------
// The fitting-only path of eval.py in ONE pass over a cloud (BASELINE configs[3]: axis :397 -> hard centroids :409-436 -> extents
// data_utils.py:1650-1730 on the fitted axes / centroids): the three kernels above read the same cloud three times (normals + weights,
// points + labels, labels + points again).  One workgroup per cloud:
//   phase 1  streams (X, Wb, Wc, P, seg, bb) once - thread (slice g, segment k) as in axis_kernel - into the 12 scatter sums, the 2 counts of
//            the normalised variant and the centroid sums (3 + count) of segment k; the thread of segment 0 parks the point and its
//            barrel key in LDS on the way (96 KB + 8 KB at N = 8192);
//   phase 2  fp64 reduction (wave shuffles, then over the 16 waves), eigen-solve + centroid on K threads -> axes, centroids (global + LDS);
//   phase 3  barrel lists from the LDS keys, projection of the S samples per segment from the LDS points -> min / max.
// extents_finish_kernel applies the batch-level rules afterwards, as for the separate kernel.
// K a power of two <= 8 and 3N floats + N bytes + N ints within the LDS: other shapes take the three separate kernels.
// ------------------------------------------------------------------------------------------------

// THREADS 1024 / PLDS: one workgroup per CU, the cloud's points parked in LDS.  THREADS 512 / !PLDS: two workgroups per CU (one streams
// while the other is in its serial phases), the projection gathers its points from global memory (L2 / MALL: the workgroup has just read them).
// HARD: the memberships are IMPLIED by the labels (Wb[n,k] = [seg == k and bb == 0], Wc[n,k] = [seg == k and bb == 1]: pre-segmented clouds,
// eval.py's --use_gt_segmentation --use_gt_bb operands, BASELINE configs[3]) and are not read: 40 B per point instead of 104 (SURVEY 8(d):
// "16 B of labels if one-hot is implied").  With no per-(point, segment) operand left a lane takes a whole POINT (LPP = K/4 lanes per point
// for K segments, 2 segments each): one wave instruction then fetches 32-64 distinct points instead of 8 (eight times fewer memory
// instructions per byte: the (slice, segment) mapping moves 32 useful bytes per wave instruction through a CU's one 64 B/clk vector
// memory path), and the segment sums are selected by the label with 0/1 factors - the same products, another summation order.
typedef float fit_v2f __attribute__((ext_vector_type(2)));

// the value of lane m of this lane's group of LPP consecutive lanes (LPP 1, 2 or 4: inside a quad -> one v_mov_b32_dpp quad_perm)
template <int LPP>
__device__ __forceinline__ int fit_group_bcast(int v, int m)
{
    // (every lane reads a valid lane of its own quad: nothing is taken from `old`, so it is left undefined - mov_dpp - and the
    // instruction needs no copy of its source first)
    if constexpr (LPP == 1) return v;
    else if constexpr (LPP == 2) return m == 0 ? __builtin_amdgcn_mov_dpp(v, 0xA0, 0xF, 0xF, true)        // quad_perm [0,0,2,2]
                                               : __builtin_amdgcn_mov_dpp(v, 0xF5, 0xF, 0xF, true);       // quad_perm [1,1,3,3]
    else return m == 0 ? __builtin_amdgcn_mov_dpp(v, 0x00, 0xF, 0xF, true) : m == 1 ? __builtin_amdgcn_mov_dpp(v, 0x55, 0xF, 0xF, true)
              : m == 2 ? __builtin_amdgcn_mov_dpp(v, 0xAA, 0xF, 0xF, true) : __builtin_amdgcn_mov_dpp(v, 0xFF, 0xF, 0xF, true);
}
template <int LPP>
__device__ __forceinline__ float fit_group_bcast(float v, int m) { return __builtin_bit_cast(float, fit_group_bcast<LPP>(__builtin_bit_cast(int, v), m)); }

struct __attribute__((packed, aligned(4))) fit_f3 { float x, y, z; };          // 12-byte load (global_load_dwordx3) of a 4-byte aligned triple

template <int KK, int THREADS, bool PLDS, bool HARD = false>
__global__ void __launch_bounds__(THREADS, 4) fit_fused_kernel(const float *__restrict__ X, const float *__restrict__ Wb, const float *__restrict__ Wc,
                                                                const float *__restrict__ P, const int64_t *__restrict__ seg,
                                                                const int64_t *__restrict__ bb, const int64_t *__restrict__ rand_idx, int normalize,
                                                                int N, int S, float *__restrict__ axis_out, float *__restrict__ cen_out,
                                                                float *__restrict__ cfound_out, float *__restrict__ ext_tmp, int *__restrict__ counts,
                                                                double *__restrict__ axis64_out)
{
    constexpr int NA = 18;                               // 12 scatter sums, 2 counts (normalize), centroid x y z, count
    constexpr int WAVES = THREADS / 64, G = THREADS / KK;                  // point slices
    extern __shared__ __attribute__((aligned(16))) int dyn[];          // phase 2 views it as double[]
    const int NL = max(N, WAVES * KK * NA * 2);      // ints: the lists of phase 3 / the per-wave fp64 sums of phase 2
    int *list = dyn;
    float *Ps = reinterpret_cast<float *>(dyn + NL);     // [3N] (PLDS)
    signed char *keyb = reinterpret_cast<signed char *>(Ps + (PLDS ? 3 * (size_t)N : 0));    // [N]
    __shared__ double tot[NA][KK];
    __shared__ int wcnt[WAVES][FIT_MAXK];
    __shared__ int start[FIT_MAXK + 1];
    __shared__ float rmin[2 * WAVES], rmax[2 * WAVES];
    __shared__ float axs[KK][3], cns[KK][3];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = tid / KK, k = tid % KK;
    // ---------------- phase 1
    FIT_TR(0);
    // HARD: SPL segments per lane (2: 36 accumulators - four per lane spilled at the 128 registers a 1024-thread workgroup has), LPP lanes per point
    constexpr int SPL = HARD ? (KK >= 2 ? 2 : 1) : 1, LPP = HARD ? KK / SPL : 1;
    float acc[HARD ? 1 : NA];
    fit_v2f hacc2[SPL][NA / 2];                          // HARD: the 18 sums of a segment as nine pairs (scalar v_fma_f32 in this build: no packed fp32, point2cyl_amd/build.py)
    if constexpr (!HARD) {
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = 0.f;
    } else {
#pragma unroll
        for (int j = 0; j < SPL; ++j)
#pragma unroll
            for (int i = 0; i < NA / 2; ++i) hacc2[j][i] = fit_v2f{0.f, 0.f};
    }
    if constexpr (HARD) {
        const float *x = X + (size_t)b * N * 3, *pp = P + (size_t)b * N * 3;
        const int *sg = reinterpret_cast<const int *>(seg + (size_t)b * N), *bl = reinterpret_cast<const int *>(bb + (size_t)b * N);
        const int hsel = tid % LPP, k0 = hsel * SPL;
        // Every lane loads its OWN point - consecutive lanes, consecutive points: a wave instruction fetches 64 distinct points - and parks it;
        // the LPP lanes of a group then take the group's points one after the other, each through a quad broadcast (DPP) of the owner's
        // registers, and sum them into their SPL segments.  (Before: the LPP lanes of a group all loaded the same point - four times the lane
        // bytes through the CU's one vector memory path, 1.3 MB per cloud at 64 B/clk = 20 k of the phase's 30 - 46 k cycles - and parked it
        // four times.)  U points per lane are REQUESTED before the first is used, with no control flow between the loads and their uses
        // (clamped indices instead of a bounds branch: with a branch the compiler drained the memory counter in every iteration and the
        // phase ran at the memory LATENCY, round 5's trace); a clamped slot repeats point N - 1: parked again (same values), not summed.
        // (Two or four register stages with the next stage's loads in flight while one is summed - software pipelining pinned with
        // sched_barrier - measured and dropped: the stream got no shorter, 21 - 35 k cycles per wave either way, and at the 128 registers
        // of a 1024-thread workgroup the longer live ranges pushed the projection loop into scratch: 1.0 ms per pass.)
        // (U = 1, 2, 3: the same 35 k cycles to the last wave's end - 5.7 k vector instructions per SIMD at ~6 cycles each set the phase, not the
        // depth of the loads in flight; the per-wave ends step up in issue-priority order, oldest first.  That staircase is what hides the
        // loads: with s_setprio falling with a wave's progress the four waves of a SIMD advance together, wait for their second step's
        // points together, and the last wave ends at 50 k cycles instead of 35 k - measured, dropped.  tools/ubench/valu_rate.hip: a lone
        // wave issues one vector instruction per 5 clocks; four waves one v_pk_fma_f32 / quad-broadcast v_mov_b32_dpp per 4.2, one
        // v_fma_f32 per 2.5 - the 710 instructions per 16 (point, lane) steps price the phase at 22 k, the tail waves' lone running at 35 k)
        constexpr int U = 4;
        fit_v2f q4 = {0.f, 1.f};                   // (p2, 1): the count rides with the centroid's third sum; only .x is rewritten per point
        for (int n0 = tid; n0 < N; n0 += U * THREADS) {
            float xq[U][3], pq[U][3];
            int svq[U], bvq[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int n = n0 + u * THREADS, nc = n < N ? n : N - 1;
                // (unsigned 32-bit byte offsets from the cloud's uniform base: one VGPR per address and the scalar-base form of the load)
                const unsigned o12 = (unsigned)nc * 12u, o8 = (unsigned)nc * 8u;
                const fit_f3 xv = *reinterpret_cast<const fit_f3 *>(reinterpret_cast<const char *>(x) + o12);
                const fit_f3 pv = *reinterpret_cast<const fit_f3 *>(reinterpret_cast<const char *>(pp) + o12);
                xq[u][0] = xv.x; xq[u][1] = xv.y; xq[u][2] = xv.z;
                pq[u][0] = pv.x; pq[u][1] = pv.y; pq[u][2] = pv.z;
                svq[u] = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(sg) + o8);
                bvq[u] = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(bl) + o8);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int n = n0 + u * THREADS, nc = n < N ? n : N - 1;
                const int sv_own = svq[u], bv_own = bvq[u];
                if (PLDS) { Ps[nc * 3 + 0] = pq[u][0]; Ps[nc * 3 + 1] = pq[u][1]; Ps[nc * 3 + 2] = pq[u][2]; }
                keyb[nc] = (signed char)((bv_own == 0 && sv_own >= 0 && sv_own < KK) ? (int)sv_own : -1);
                const int sv_sum = n < N ? sv_own : -2;                  // (a clamped slot belongs to no segment)
#pragma unroll
                for (int m = 0; m < LPP; ++m) {
                    // the point of lane m of this group, in every lane of the group
                    // (pairs: xa = (x0, x1), xb = (x2, x1) give the six products (x0 x0, x0 x1), (x0 x2, x1 x1), (x1 x2, x2 x2); written for
                    // v_pk_mul_f32 / v_pk_fma_f32 with half selects - measured 0.1845 ms that way - and compiled to scalar multiplies and
                    // FMAs since the library is built without packed fp32 instructions: 0.1876 ms, see point2cyl_amd/build.py for why)
                    const fit_v2f xa = {fit_group_bcast<LPP>(xq[u][0], m), fit_group_bcast<LPP>(xq[u][1], m)};
                    const fit_v2f xb = {fit_group_bcast<LPP>(xq[u][2], m), fit_group_bcast<LPP>(xq[u][1], m)};
                    const fit_v2f q3 = {fit_group_bcast<LPP>(pq[u][0], m), fit_group_bcast<LPP>(pq[u][1], m)};
                    q4.x = fit_group_bcast<LPP>(pq[u][2], m);
                    const int sv = fit_group_bcast<LPP>(sv_sum, m), bv = fit_group_bcast<LPP>(bv_own, m);
                    const fit_v2f q0 = xa.xx * xa, q1 = xa * xb, q2 = xb.yx * xb.xx;
#pragma unroll
                    for (int j = 0; j < SPL; ++j) {
                        const bool mine = sv == k0 + j;
                        const float w = mine ? 1.f : 0.f;
                        const fit_v2f vbc = {(mine && bv == 0) ? 1.f : 0.f, (mine && bv == 1) ? 1.f : 0.f};
                        fit_v2f *a_ = hacc2[j];
                        // the factors are 0 or 1: every product is exact, so a fused multiply-add rounds exactly like the multiply + add of
                        // the general route (the file is built with -ffp-contract=off); nine pair operations per (point, segment): the
                        // pair (12, 13) is one packed add of (b2, c2) - summed whether or not `normalize` reads them -, the count 17 rides in
                        // the pair (p2, 1) of the centroid sums
                        const fit_v2f vb = vbc.xx, vc = vbc.yy, vw = {w, w};
                        a_[0] = __builtin_elementwise_fma(vb, q0, a_[0]); a_[1] = __builtin_elementwise_fma(vb, q1, a_[1]); a_[2] = __builtin_elementwise_fma(vb, q2, a_[2]);
                        a_[3] = __builtin_elementwise_fma(vc, q0, a_[3]); a_[4] = __builtin_elementwise_fma(vc, q1, a_[4]); a_[5] = __builtin_elementwise_fma(vc, q2, a_[5]);
                        a_[6] += vbc;
                        a_[7] = __builtin_elementwise_fma(vw, q3, a_[7]); a_[8] = __builtin_elementwise_fma(vw, q4, a_[8]);
                    }
                }
            }
        }
    } else
    {
        const float *x = X + (size_t)b * N * 3, *pp = P + (size_t)b * N * 3;
        const float *wb = Wb + (size_t)b * N * KK, *wc = Wc + (size_t)b * N * KK;
        // the labels' low words (little-endian int64 holding small non-negative values or -1): half the registers per point in flight, and
        // a CU's streaming rate is set by the bytes it keeps in flight (16 waves x 8 points x 104 B per unrolled step)
        const int *sg = reinterpret_cast<const int *>(seg + (size_t)b * N), *bl = reinterpret_cast<const int *>(bb + (size_t)b * N);
        // (as in the labels-implied loop: U points per thread requested before the first is used, nothing conditional in between)
        constexpr int U = 4;
        for (int n0 = g; n0 < N; n0 += U * G) {
            float xq[U][3], pq[U][3], bq[U], cq[U];
            int svq[U], bvq[U], nq[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int n = n0 + u * G, nc = n < N ? n : N - 1;
                nq[u] = nc;
                xq[u][0] = x[nc * 3 + 0]; xq[u][1] = x[nc * 3 + 1]; xq[u][2] = x[nc * 3 + 2];
                pq[u][0] = pp[nc * 3 + 0]; pq[u][1] = pp[nc * 3 + 1]; pq[u][2] = pp[nc * 3 + 2];
                bq[u] = wb[(size_t)nc * KK + k]; cq[u] = wc[(size_t)nc * KK + k];
                svq[u] = sg[2 * nc]; bvq[u] = bl[2 * nc];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int n = nq[u];
                const bool live = n0 + u * G < N;
                const float x0 = xq[u][0], x1 = xq[u][1], x2 = xq[u][2];
                const float p0 = pq[u][0], p1 = pq[u][1], p2 = pq[u][2];
                const float b_ = live ? bq[u] : 0.f, c_ = live ? cq[u] : 0.f;
                const int sv = svq[u], bv = bvq[u];
                const float b2 = b_ * b_, c2 = c_ * c_;
                const float p00 = x0 * x0, p01 = x0 * x1, p02 = x0 * x2, p11 = x1 * x1, p12 = x1 * x2, p22 = x2 * x2;
                acc[0] += b2 * p00; acc[1] += b2 * p01; acc[2] += b2 * p02; acc[3] += b2 * p11; acc[4] += b2 * p12; acc[5] += b2 * p22;
                acc[6] += c2 * p00; acc[7] += c2 * p01; acc[8] += c2 * p02; acc[9] += c2 * p11; acc[10] += c2 * p12; acc[11] += c2 * p22;
                const bool mine = live && sv == k;
                if (normalize) {
                    acc[12] += (mine && bv == 0) ? 1.f : 0.f;
                    acc[13] += (mine && bv == 1) ? 1.f : 0.f;
                }
                const float w = mine ? 1.f : 0.f;
                acc[14] += w * p0; acc[15] += w * p1; acc[16] += w * p2; acc[17] += w;
                if (PLDS) { Ps[n * 3 + 0] = p0; Ps[n * 3 + 1] = p1; Ps[n * 3 + 2] = p2; }        // (every segment lane of the point: same address, same value)
                keyb[n] = (signed char)((bv == 0 && sv >= 0 && sv < KK) ? (int)sv : -1);
            }
        }
    }
    FIT_TR(1);
#ifdef P2C_FIT_TRACE
    if (blockIdx.x == P2C_FIT_TRACE_WG && lane == 0) p2c_fit_stamps[16 + wave] = __builtin_readcyclecounter();
#endif
    // the sample draws of this wave's first (segment, chunk) task: requested now, consumed after the reduction, the eigen-solve and the list
    // build (one dependent global load per 64 samples inside the projection loop was a third of the separate kernel's time)
    constexpr int nch = WAVES / KK, RU = 16;
    const int per_chunk = (S + nch - 1) / nch;
    int r0[RU];
    {
        const int kq = wave / nch, ch = wave - kq * nch;
        const int64_t *ri = rand_idx + ((size_t)b * KK + kq) * S;
        const int s_beg = ch * per_chunk, s_end = min(S, (ch + 1) * per_chunk);
#pragma unroll
        for (int u = 0; u < RU; ++u) { const int s = s_beg + u * 64 + lane; r0[u] = s < s_end ? (int)ri[s] : 0; }
    }
    // ---------------- phase 2: the 64/K slices of a wave in fp32, fp64 from there on (see axis_kernel)
    {
        double *wsum = reinterpret_cast<double *>(list);                  // [WAVES][KK][NA]
        if constexpr (HARD) {
            __syncthreads();                                              // (the keys / points parked above share no LDS with wsum; the lists do)
            FIT_TR(8);
#pragma unroll
            for (int j = 0; j < SPL; ++j)
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    const float v = p2c_wave_class_sum_f32<LPP>((i & 1) ? hacc2[j][i >> 1].y : hacc2[j][i >> 1].x);      // the lanes of this segment group in the wave: lane % LPP
                    if (lane < LPP) wsum[(wave * KK + lane * SPL + j) * NA + i] = (double)v;
                }
        } else {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            float v = acc[i];
#pragma unroll
            for (int o = KK; o < 64; o <<= 1) v += __shfl_xor(v, o);      // the lanes of segment k in this wave: lane % KK == k
            if (lane < KK) wsum[(wave * KK + lane) * NA + i] = (double)v;
        }
        }
        FIT_TR(9);
        __syncthreads();
        FIT_TR(10);
        if (tid < KK * NA) {
            const int rk = tid / NA, re = tid - rk * NA;
            double rs = 0.0;
            for (int w = 0; w < WAVES; ++w) rs += wsum[(w * KK + rk) * NA + re];
            tot[re][rk] = rs;
        }
        __syncthreads();
        FIT_TR(2);
        if (tid < KK) {
            double isb2 = 1.0, isc2 = 1.0;
            if (normalize) {
                const float sb = sqrtf((float)tot[12][tid]) + 1.0f, sc = sqrtf((float)tot[13][tid]) + 1.0f;   // data_utils.py:139-160
                isb2 = 1.0 / ((double)sb * (double)sb);
                isc2 = 1.0 / ((double)sc * (double)sc);
            }
            double a[6], lam[3], v[3][3];
            for (int e = 0; e < 6; ++e) a[e] = tot[e][tid] * isb2 - tot[6 + e][tid] * isc2;
            p2c_eigh3(a, lam, v);
            int big = 0;
            if (fabs(v[0][1]) > fabs(v[0][big])) big = 1;
            if (fabs(v[0][2]) > fabs(v[0][big])) big = 2;
            const double sgn = v[0][big] < 0 ? -1.0 : 1.0;
            float *o = axis_out + ((size_t)b * KK + tid) * 3;
            for (int e = 0; e < 3; ++e) { const float av = (float)(sgn * v[0][e]); o[e] = av; axs[tid][e] = av; }
            if (axis64_out)
                for (int e = 0; e < 3; ++e) axis64_out[((size_t)b * KK + tid) * 3 + e] = sgn * v[0][e];
            const float c = (float)tot[17][tid];                          // hard centroid: as centroids_by_point_kernel
            const bool ok = c > 1.f;
            float *oc = cen_out + ((size_t)b * KK + tid) * 3;
            for (int e = 0; e < 3; ++e) { const float cv = ok ? (float)tot[14 + e][tid] / c : 0.f; oc[e] = cv; cns[tid][e] = cv; }
            cfound_out[(size_t)b * KK + tid] = ok ? 1.f : 0.f;
        }
        __syncthreads();
    }
    // ---------------- phase 3 (extents_kernel, on the LDS copies)
    FIT_TR(3);
    if (N <= WAVES * 64 * EXT_MAXCH) ext_build_lists_scan<WAVES, (KK <= 4 ? 1 : 2)>([&](int n) { return (int)keyb[n]; }, N, KK, list, wcnt, start);
    else ext_build_lists_by<WAVES>([&](int n) { return (int)keyb[n]; }, N, KK, list, wcnt, start);
    FIT_TR(4);
    {
        for (int t = wave; t < KK * nch; t += WAVES) {
            const int kq = t / nch, ch = t - kq * nch;
            const int cnt = start[kq + 1] - start[kq];
            const float a0 = axs[kq][0], a1 = axs[kq][1], a2 = axs[kq][2], c0_ = cns[kq][0], c1_ = cns[kq][1], c2_ = cns[kq][2];
            const int64_t *ri = rand_idx + ((size_t)b * KK + kq) * S;
            const int *lk = list + start[kq];
            float lo = INFINITY, hi = -INFINITY;
            const int s_beg = ch * per_chunk, s_end = min(S, (ch + 1) * per_chunk);
            for (int s0 = s_beg; s0 < s_end; s0 += 64 * RU) {
                int r[RU];
                if (t == wave && s0 == s_beg) {
#pragma unroll
                    for (int u = 0; u < RU; ++u) r[u] = r0[u];
                } else {
#pragma unroll
                    for (int u = 0; u < RU; ++u) { const int s = s0 + u * 64 + lane; r[u] = s < s_end ? (int)ri[s] : 0; }
                }
                float qx[RU], qy[RU], qz[RU];
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    qx[u] = qy[u] = qz[u] = 0.f;
                    if (cnt > 1 && s0 + u * 64 + lane < s_end) {
                        const int n = lk[r[u]];
                        const float *pg = PLDS ? Ps + n * 3 : P + ((size_t)b * N + n) * 3;
                        qx[u] = pg[0]; qy[u] = pg[1]; qz[u] = pg[2];
                    }
                }
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    const int s = s0 + u * 64 + lane;
                    if (s < s_end) {
                        const float px = qx[u], py = qy[u], pz = qz[u];
                        const float dx = px - c0_, dy = py - c1_, dz = pz - c2_;
                        const float tt = __builtin_fmaf(dz, a2, __builtin_fmaf(dy, a1, dx * a0));
                        lo = fminf(lo, tt); hi = fmaxf(hi, tt);
                    }
                }
            }
            for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
            if (lane == 0) { rmin[t] = lo; rmax[t] = hi; }
        }
        __syncthreads();
        if (tid < KK) {
            float l2 = rmin[tid * nch], h2 = rmax[tid * nch];
            for (int ch = 1; ch < nch; ++ch) { l2 = fminf(l2, rmin[tid * nch + ch]); h2 = fmaxf(h2, rmax[tid * nch + ch]); }
            ext_tmp[((size_t)b * KK + tid) * 2 + 0] = l2;
            ext_tmp[((size_t)b * KK + tid) * 2 + 1] = h2;
            counts[b * KK + tid] = start[tid + 1] - start[tid];
        }
        FIT_TR(5);
    }
}

static size_t fit_fused_lds(int N, int K, int waves, bool plds)
{
    const size_t nl = (size_t)N > (size_t)waves * K * 18 * 2 ? (size_t)N : (size_t)waves * K * 18 * 2;
    return nl * sizeof(int) + (size_t)N * ((plds ? 3 * sizeof(float) : 0) + 1);
}

extern "C" int p2c_fit_fused_supported(int N, int K, int S)
{
    const bool kpow = K == 1 || K == 2 || K == 4 || K == 8;
    return (kpow && S > 0 && N > 0 && (N % 4) == 0 && fit_fused_lds(N, K, EXT_WAVES, true) <= 140 * 1024) ? 1 : 0;
}

// axes (B,K,3), centroids (B,K,3) + their found mask (B,K), extents (K,B,2) + found mask (B,K); ws: p2c_extents_ws_bytes(B, K)
extern "C" int p2c_fit_fused_f32(const float *X, const float *Wb, const float *Wc, const int64_t *bb_gt, const int64_t *inst_gt, int normalize,
                                 const float *P, const int64_t *rand_idx, int B, int N, int K, int S, float *axis_out, float *centroids_out,
                                 float *cfound_out, float *extents_out, float *found_out, double *axis64_out, void *ws, void *stream)
{
    if (!X || !bb_gt || !inst_gt || !P || !rand_idx || !axis_out || !centroids_out || !cfound_out || !extents_out || !found_out || !ws ||
        B <= 0 || !p2c_fit_fused_supported(N, K, S))
        return P2C_EINVAL;
    if ((Wb == nullptr) != (Wc == nullptr)) return P2C_EINVAL;
    const bool hard = Wb == nullptr;            // memberships implied by the labels: not read (fit_fused_kernel<.., HARD>)
    float *ext_tmp = (float *)ws;
    int *counts = (int *)(ext_tmp + (size_t)B * K * 2);
    hipStream_t s = (hipStream_t)stream;
    // P2C_FIT_VARIANT=1: two half-size workgroups per CU, one's serial phases under the other's streaming, points gathered from global memory
    const char *fv = getenv("P2C_FIT_VARIANT");
    const int forced = fv ? atoi(fv) : -1;
    const bool half = forced == 1;          // (measured at 1250 clouds: 0.45 ms against 0.40 for the one-workgroup form - the gathers cost what the overlap gains)
#define P2C_FF(KK_, TH_, PLDS_)                                                                                                         \
    do {                                                                                                                                \
        const size_t lds = fit_fused_lds(N, K, TH_ / 64, PLDS_);                                                                        \
        (void)hipFuncSetAttribute((const void *)fit_fused_kernel<KK_, TH_, PLDS_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);       \
        hipLaunchKernelGGL((fit_fused_kernel<KK_, TH_, PLDS_>), dim3(B), dim3(TH_), lds, s, X, Wb, Wc, P, inst_gt, bb_gt, rand_idx, normalize, N, S,    \
                           axis_out, centroids_out, cfound_out, ext_tmp, counts, axis64_out);                                          \
    } while (0)
#define P2C_FFH_(KK_, TH_, PLDS_)                                                                                                       \
    do {                                                                                                                                \
        const size_t lds = fit_fused_lds(N, K, TH_ / 64, PLDS_);                                                                        \
        (void)hipFuncSetAttribute((const void *)fit_fused_kernel<KK_, TH_, PLDS_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
        hipLaunchKernelGGL((fit_fused_kernel<KK_, TH_, PLDS_, true>), dim3(B), dim3(TH_), lds, s, X, Wb, Wc, P, inst_gt, bb_gt, rand_idx, normalize, N, \
                           S, axis_out, centroids_out, cfound_out, ext_tmp, counts, axis64_out);                                       \
    } while (0)
#define P2C_FFH(KK_) do { if (half) P2C_FFH_(KK_, 512, false); else P2C_FFH_(KK_, 1024, true); } while (0)
#define P2C_FFK(KK_) do { if (hard) P2C_FFH(KK_); else if (half) P2C_FF(KK_, 512, false); else P2C_FF(KK_, 1024, true); } while (0)
    if (K == 8) P2C_FFK(8);
    else if (K == 4) P2C_FFK(4);
    else if (K == 2) P2C_FFK(2);
    else P2C_FFK(1);
#undef P2C_FFK
#undef P2C_FFH
#undef P2C_FFH_
#undef P2C_FF
    hipLaunchKernelGGL(extents_finish_kernel, dim3(K), dim3(256), 0, s, ext_tmp, counts, B, K, extents_out, found_out);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// ------------------------------------------------------------------------------------------------
// Sketch projection (data_utils.py:1014-1146; :1149 = the same + found mask; :1284 = every point of the cloud for every
// segment, no sampling -> all_points).  Same structure as the extents: one workgroup per cloud builds the barrel lists,
// then per segment turns the S sampled points AND normals by the matrix that takes the segment's axis onto z, keeps x, y,
// subtracts the turned centre from the points and takes the largest radius (the sketch's scale).
//   * the matrix is the reference's: rotation vector cross(axis, z) * acos(axis.z) - the cross product is NOT normalised
//     there, so its length is angle*sin(angle); kept - through torchgeometry 0.1.2's angle_axis_to_rotation_matrix
//     (not installed anywhere here: its published algorithm is restated, Rodrigues with axis aa/(|aa| + 1e-6), first-order
//     form for |aa|^2 <= 1e-6; parity unpinned at that boundary, see oracle/ref_torch.py);
//   * points are ROW vectors times the matrix (:1110);
//   * a cloud with <= 1 barrel point of the segment keeps zero samples that still go through the centring (rows =
//     -centre_turned, :1054 + :1131) and gets scale 1 (:1143); a segment with <= 1 barrel point in the whole batch is
//     skipped: zeros, scale 1 (:1043) - applied by the finish kernel.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void sketch_axis_rotation(float ax, float ay, float az, float (&R)[3][3])
{
    R[0][0] = 1.f; R[0][1] = 0.f; R[0][2] = 0.f; R[1][0] = 0.f; R[1][1] = 1.f; R[1][2] = 0.f; R[2][0] = 0.f; R[2][1] = 0.f; R[2][2] = 1.f;
    const float ang = acosf(az);
    if (!(ang > 1.0e-6f)) return;                      // g_zero_tol; NaN compares false: identity
    const float rx = ay * ang, ry = -ax * ang, rz = 0.f;       // cross((ax,ay,az), (0,0,1)) * angle
    const float th2 = rx * rx + ry * ry + rz * rz;
    if (th2 > 1.0e-6f) {
        const float th = sqrtf(th2), inv = 1.f / (th + 1.0e-6f);
        const float wx = rx * inv, wy = ry * inv, wz = rz * inv, c = cosf(th), s = sinf(th), oc = 1.f - c;
        R[0][0] = c + wx * wx * oc;      R[0][1] = wx * wy * oc - wz * s; R[0][2] = wy * s + wx * wz * oc;
        R[1][0] = wz * s + wx * wy * oc; R[1][1] = c + wy * wy * oc;      R[1][2] = -wx * s + wy * wz * oc;
        R[2][0] = -wy * s + wx * wz * oc; R[2][1] = wx * s + wy * wz * oc; R[2][2] = c + wz * wz * oc;
    } else {
        R[0][1] = -rz; R[0][2] = ry; R[1][0] = rz; R[1][2] = -rx; R[2][0] = -ry; R[2][1] = rx;
    }
}

__global__ void __launch_bounds__(EXT_THREADS) sketch_project_kernel(const float *__restrict__ P, const float *__restrict__ X,
                                                                     const int64_t *__restrict__ seg, const int64_t *__restrict__ bb,
                                                                     const float *__restrict__ axes, const float *__restrict__ centers,
                                                                     const int64_t *__restrict__ rand_idx, int B, int N, int K, int S, int all_points,
                                                                     float *__restrict__ Pp, float *__restrict__ Xp, float *__restrict__ scale_tmp,
                                                                     int *__restrict__ counts)
{
    extern __shared__ int list[];                       // N ints, partitioned by segment
    __shared__ int wcnt[EXT_WAVES][FIT_MAXK];
    __shared__ int start[FIT_MAXK + 1];
    __shared__ float rmax[EXT_WAVES];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (!all_points) ext_build_lists(seg + (size_t)b * N, bb + (size_t)b * N, N, K, list, wcnt, start);
    for (int k = 0; k < K; ++k) {
        const int cnt = all_points ? N : start[k + 1] - start[k];
        const float *a = axes + ((size_t)b * K + k) * 3, *c = centers + ((size_t)b * K + k) * 3;
        float R[3][3];
        sketch_axis_rotation(a[0], a[1], a[2], R);
        const float c0 = c[0], c1 = c[1], c2 = c[2];
        const float cqx = c0 * R[0][0] + c1 * R[1][0] + c2 * R[2][0], cqy = c0 * R[0][1] + c1 * R[1][1] + c2 * R[2][1];
        const int64_t *ri = rand_idx ? rand_idx + ((size_t)b * K + k) * S : nullptr;
        const int *lk = list + (all_points ? 0 : start[k]);
        float2 *po = reinterpret_cast<float2 *>(Pp) + ((size_t)k * B + b) * S, *xo = reinterpret_cast<float2 *>(Xp) + ((size_t)k * B + b) * S;
        float r2max = 0.f;
        for (int s = tid; s < S; s += EXT_THREADS) {
            float px = 0.f, py = 0.f, pz = 0.f, nx = 0.f, ny = 0.f, nz = 0.f;
            if (cnt > 1) {
                const int n = all_points ? s : lk[(int)ri[s]];
                const float *pp = P + ((size_t)b * N + n) * 3, *xx = X + ((size_t)b * N + n) * 3;
                px = pp[0]; py = pp[1]; pz = pp[2];
                nx = xx[0]; ny = xx[1]; nz = xx[2];
            }
            const float qx = (px * R[0][0] + py * R[1][0] + pz * R[2][0]) - cqx, qy = (px * R[0][1] + py * R[1][1] + pz * R[2][1]) - cqy;
            po[s] = make_float2(qx, qy);
            xo[s] = make_float2(nx * R[0][0] + ny * R[1][0] + nz * R[2][0], nx * R[0][1] + ny * R[1][1] + nz * R[2][1]);
            r2max = fmaxf(r2max, qx * qx + qy * qy);
        }
        for (int o = 32; o > 0; o >>= 1) r2max = fmaxf(r2max, __shfl_xor(r2max, o));
        if (lane == 0) rmax[wave] = r2max;
        __syncthreads();
        if (tid == 0) {
            float m = rmax[0];
            for (int w = 1; w < EXT_WAVES; ++w) m = fmaxf(m, rmax[w]);
            scale_tmp[(size_t)b * K + k] = sqrtf(m);
            counts[b * K + k] = cnt;
        }
        __syncthreads();
    }
}

// one workgroup per segment k: the batch-wide rule (:1043) and the scale / found conventions (:1143)
__global__ void __launch_bounds__(256) sketch_finish_kernel(const float *__restrict__ scale_tmp, const int *__restrict__ counts, int B, int K, int S,
                                                            float *__restrict__ Pp, float *__restrict__ Xp, float *__restrict__ scales,
                                                            float *__restrict__ found)
{
    __shared__ long long part[4];
    const int k = blockIdx.x, tid = threadIdx.x;
    long long t = 0;
    for (int b = tid; b < B; b += 256) t += counts[b * K + k];
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    if ((tid & 63) == 0) part[tid >> 6] = t;
    __syncthreads();
    const bool seg_ok = part[0] + part[1] + part[2] + part[3] > 1;
    for (int b = tid; b < B; b += 256) {
        const bool f = seg_ok && counts[b * K + k] > 1;
        scales[(size_t)k * B + b] = f ? scale_tmp[(size_t)b * K + k] : 1.f;
        found[(size_t)b * K + k] = f ? 1.f : 0.f;
    }
    if (!seg_ok) {
        const size_t n = (size_t)B * S * 2, o = (size_t)k * n;
        for (size_t i = tid; i < n; i += 256) { Pp[o + i] = 0.f; Xp[o + i] = 0.f; }
    }
}

extern "C" int p2c_sketch_projection_f32(const float *P, const float *X, const int64_t *seg, const int64_t *bb, const float *axes,
                                         const float *centers, const int64_t *rand_idx, int B, int N, int K, int S, int all_points,
                                         float *P_proj, float *X_proj, float *scales_out, float *found_out, void *ws, void *stream)
{
    if (!P || !X || !axes || !centers || !P_proj || !X_proj || !scales_out || !found_out || !ws || B <= 0 || N <= 0 || N > EXT_MAXN || K <= 0 ||
        K > FIT_MAXK || S <= 0)
        return P2C_EINVAL;
    if (all_points ? S != N : (!seg || !bb || !rand_idx)) return P2C_EINVAL;
    if (((uintptr_t)P_proj & 7) || ((uintptr_t)X_proj & 7)) return P2C_EALIGN;
    float *scale_tmp = (float *)ws;
    int *counts = (int *)(scale_tmp + (size_t)B * K * 2);
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)N * sizeof(int);
    (void)hipFuncSetAttribute((const void *)sketch_project_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(sketch_project_kernel, dim3(B), dim3(EXT_THREADS), lds, s, P, X, seg, bb, axes, centers, rand_idx, B, N, K, S, all_points,
                       P_proj, X_proj, scale_tmp, counts);
    hipLaunchKernelGGL(sketch_finish_kernel, dim3(K), dim3(256), 0, s, scale_tmp, counts, B, K, S, P_proj, X_proj, scales_out, found_out);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}
