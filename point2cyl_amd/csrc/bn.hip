// bn.hip -- the small kernels around the GEMMs: BatchNorm statistics -> affine, the materialised
// relu(bn(.)) where a consumer needs it, max-pool over the neighbour axis (fwd/bwd), and the
// per-channel reductions of the ReLU+BatchNorm backward.  All HBM-bound streaming / reductions; lanes
// run along the (contiguous) channel axis so every wave access is a coalesced row segment.
#include "common.h"

// ------------------------------------------------------------------------------------------------
// BatchNorm finalize.  partials[tile][0][C] = sum(acc), [tile][1][C] = sum(acc^2) with acc = y - bias
// (the GEMM epilogue accumulates before the bias add; var is bias-free, mean gets the bias back here).
// Block = 64 channels x 16 tile-lanes, fp64 accumulation.  torch semantics: biased variance normalises,
// unbiased variance feeds running_var; running = (1-momentum)*running + momentum*batch.
// ------------------------------------------------------------------------------------------------
// Sum the P2C_STAT_SLOTS fp64 rows of one channel with 16 threads (4 rows each: 8 independent loads), combined through LDS.
// Block = 16 channels x 16 slot-lanes (a finalisation is a latency chain of one small workgroup per 16 channels: with 64 channels x 4 lanes
// a thread walked 16 rows); returns the totals to the slot-lane-0 thread of each channel (threadIdx.x < 16).
#define FIN_CH 16
__device__ __forceinline__ void p2c_sum_slots(const double *__restrict__ slots, int n_slots, int C, int c, bool active, double &s1, double &s2)
{
    __shared__ double red[2][16][FIN_CH];
    const int cx = threadIdx.x & (FIN_CH - 1), ty = threadIdx.x >> 4;
    double a = 0.0, b = 0.0;
    if (active && c < C) {
#pragma unroll 4
        for (int t = ty; t < n_slots; t += 16) {
            a += slots[(size_t)t * 2 * C + c];
            b += slots[(size_t)t * 2 * C + C + c];
        }
    }
    red[0][ty][cx] = a;
    red[1][ty][cx] = b;
    __syncthreads();
    s1 = s2 = 0.0;
    if (ty == 0) {
#pragma unroll
        for (int t = 0; t < 16; t += 4) {
            s1 += (red[0][t][cx] + red[0][t + 1][cx]) + (red[0][t + 2][cx] + red[0][t + 3][cx]);
            s2 += (red[1][t][cx] + red[1][t + 1][cx]) + (red[1][t + 2][cx] + red[1][t + 3][cx]);
        }
    }
}

__global__ void __launch_bounds__(256) bn_finalize_kernel(const double *__restrict__ partials, int n_tiles, int C, long long count,
                                                           const float *__restrict__ bias, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, float eps, float momentum, int training,
                                                           float *__restrict__ running_mean, float *__restrict__ running_var,
                                                           float *__restrict__ scale, float *__restrict__ shift,
                                                           float *__restrict__ mean_out, float *__restrict__ invstd_out)
{
    const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1));
    double s1 = 0.0, s2 = 0.0;
    p2c_sum_slots(partials, n_tiles, C, c, training != 0, s1, s2);
    if (c >= C || threadIdx.x >= FIN_CH) return;
    float mean, invstd;
    if (training) {
        const double m0 = s1 / (double)count;
        double var = s2 / (double)count - m0 * m0;
        if (var < 0.0) var = 0.0;
        const double m = m0 + (bias ? (double)bias[c] : 0.0);
        mean = (float)m;
        invstd = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) {
            const double unb = count > 1 ? var * (double)count / (double)(count - 1) : var;
            running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * m);
            running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unb);
        }
    } else {
        mean = running_mean[c];
        invstd = 1.0f / sqrtf(running_var[c] + eps);
    }
    const float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - mean * sc;
    if (mean_out) mean_out[c] = mean;
    if (invstd_out) invstd_out[c] = invstd;
}

extern "C" int p2c_bn_finalize_f32(const double *stat_slots, int C, long long count, const float *bias, const float *gamma,
                                   const float *beta, float eps, float momentum, int training, float *running_mean, float *running_var,
                                   float *scale, float *shift, float *mean, float *invstd, void *stream)
{
    if (C <= 0 || !gamma || !beta || !scale || !shift) return P2C_EINVAL;
    if (training && (!stat_slots || count <= 0)) return P2C_EINVAL;
    if (!training && (!running_mean || !running_var)) return P2C_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(p2c_cdiv(C, FIN_CH)), dim3(256), 0, s, stat_slots, P2C_STAT_SLOTS, C, count, bias,
                       gamma, beta, eps, momentum, training, running_mean, running_var, scale, shift, mean, invstd);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// ------------------------------------------------------------------------------------------------
// Eval-mode BatchNorm of EVERY layer of a forward in one launch: (scale, shift, mean, invstd) from the running statistics, one workgroup
// per layer.  An inference forward otherwise spends 17 one-workgroup launches (7 us each as graph nodes, 0.125 ms of a 1.17 ms forward,
// profiles/r06_forward_modes.log) on 6k channels.  The table is a device array of rows; the kernel reads the LIVE parameters, so a captured
// graph stays right when the weights change in place.  Same arithmetic as bn_finalize_kernel's eval branch.
// ------------------------------------------------------------------------------------------------
struct P2cBnEvalRow {
    const float *gamma, *beta, *running_mean, *running_var;
    float *st;          // (4, C): scale, shift, mean, invstd
    int C;
    float eps;
};
static_assert(sizeof(P2cBnEvalRow) == 48, "host packs 48-byte rows (ops.BNEvalStage)");

__global__ void __launch_bounds__(256) bn_eval_affine_batch_kernel(const P2cBnEvalRow *__restrict__ rows)
{
    const P2cBnEvalRow r = rows[blockIdx.x];
    for (int c = threadIdx.x; c < r.C; c += 256) {
        const float mean = r.running_mean[c];
        const float invstd = 1.0f / sqrtf(r.running_var[c] + r.eps);
        const float sc = r.gamma[c] * invstd;
        r.st[c] = sc;
        r.st[r.C + c] = r.beta[c] - mean * sc;
        r.st[2 * r.C + c] = mean;
        r.st[3 * r.C + c] = invstd;
    }
}

extern "C" int p2c_bn_eval_affine_batch_f32(const void *rows, int n_rows, void *stream)
{
    if (n_rows < 0 || (n_rows > 0 && !rows)) return P2C_EINVAL;
    if (n_rows == 0) return P2C_OK;
    hipLaunchKernelGGL(bn_eval_affine_batch_kernel, dim3(n_rows), dim3(256), 0, (hipStream_t)stream, (const P2cBnEvalRow *)rows);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// ------------------------------------------------------------------------------------------------
// Z = relu(scale*Y + shift)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) bn_relu_apply_kernel(const float *__restrict__ Y, int ldy, const float *__restrict__ scale,
                                                            const float *__restrict__ shift, long long M, int C, float *__restrict__ Z,
                                                            int ldz)
{
    const long long total = M * C;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long r = e / C;
        const int c = (int)(e - r * C);
        Z[r * ldz + c] = fmaxf(scale[c] * Y[r * ldy + c] + shift[c], 0.f);
    }
}

extern "C" int p2c_bn_relu_apply_f32(const float *Y, int ldy, const float *scale, const float *shift, int M, int C, float *Z, int ldz,
                                     void *stream)
{
    if (!Y || !scale || !shift || !Z || M <= 0 || C <= 0) return P2C_EINVAL;
    const int blocks = (int)min((long long)p2c_cdiv((long long)M * C, 256), 8192LL);
    hipLaunchKernelGGL(bn_relu_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, Y, ldy, scale, shift, (long long)M, C, Z, ldz);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// ------------------------------------------------------------------------------------------------
// max over the neighbour axis of relu(bn(Y)) (pointnet_util.py:205).  One thread per (group, channel).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) maxpool_bnrelu_kernel(const float *__restrict__ Y, int ldy, const float *__restrict__ scale,
                                                             const float *__restrict__ shift, int G, int ns, int C,
                                                             float *__restrict__ out, int ldo, int32_t *__restrict__ arg,
                                                             float *__restrict__ ywin)
{
    const int c = blockIdx.y * blockDim.x + threadIdx.x;
    const int g = blockIdx.x;
    if (c >= C) return;
    const float sc = scale[c], sh = shift[c];
    const float *y = Y + (size_t)g * ns * ldy + c;
    float best = -INFINITY, yb = 0.f;
    int bj = 0;
#pragma unroll 8
    for (int j = 0; j < ns; ++j) {
        const float yv = y[(size_t)j * ldy];
        const float z = fmaxf(sc * yv + sh, 0.f);
        if (z > best) { best = z; bj = j; yb = yv; }
    }
    out[(size_t)g * ldo + c] = best;
    arg[(size_t)g * C + c] = bj;
    if (ywin) ywin[(size_t)g * C + c] = yb;      // pre-BN value of the winner: all the backward reduction needs
}

extern "C" int p2c_maxpool_bnrelu_f32(const float *Y, int ldy, const float *scale, const float *shift, int G, int ns, int C, float *out,
                                      int ldo, int32_t *arg, float *ywin, void *stream)
{
    if (!Y || !scale || !shift || !out || !arg || G <= 0 || ns <= 0 || C <= 0) return P2C_EINVAL;
    const int bx = C >= 256 ? 256 : ((C + 63) & ~63);
    hipLaunchKernelGGL(maxpool_bnrelu_kernel, dim3(G, p2c_cdiv(C, bx)), dim3(bx), 0, (hipStream_t)stream, Y, ldy, scale, shift, G, ns, C, out,
                       ldo, arg, ywin);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// The same result from the per-half-group extremes the pooled forward emitted (fwd_pp.hip, POOL): relu(scale*y + shift) is monotone
// in y, increasing for scale >= 0 and decreasing for scale < 0, so the group's maximum is taken by its largest (smallest) pre-BN
// value.  First row wins among equal pre-BN values, like the scan above (which breaks ties on the POST-activation value: two
// different pre-BN values that round to the same activation - or are both clamped to 0 - may resolve to another row here; the
// pooled value is the same, and for a clamped winner the gradient is zero either way).
__global__ void __launch_bounds__(256) pool_select_kernel(const float *__restrict__ pmax, const float *__restrict__ pmin,
                                                          const int32_t *__restrict__ pidx, const float *__restrict__ scale,
                                                          const float *__restrict__ shift, int G, int C, float *__restrict__ out, int ldo,
                                                          int32_t *__restrict__ arg, float *__restrict__ ywin)
{
    const int c = blockIdx.y * blockDim.x + threadIdx.x;
    const int g = blockIdx.x;
    if (c >= C) return;
    const float sc = scale[c], sh = shift[c];
    const size_t o0 = ((size_t)2 * g) * C + c, o1 = o0 + C;
    float yb;
    int bj;
    if (sc >= 0.f) {
        const float v0 = pmax[o0], v1 = pmax[o1];
        const bool second = v1 > v0;
        yb = second ? v1 : v0;
        bj = (second ? pidx[o1] : pidx[o0]) & 0xffff;
    } else {
        const float v0 = pmin[o0], v1 = pmin[o1];
        const bool second = v1 < v0;
        yb = second ? v1 : v0;
        bj = ((second ? pidx[o1] : pidx[o0]) >> 16) & 0xffff;
    }
    out[(size_t)g * ldo + c] = fmaxf(sc * yb + sh, 0.f);
    arg[(size_t)g * C + c] = bj;
    if (ywin) ywin[(size_t)g * C + c] = yb;
}

extern "C" int p2c_pool_select_f32(const float *pool_max, const float *pool_min, const int32_t *pool_idx, const float *scale,
                                   const float *shift, int G, int C, float *out, int ldo, int32_t *arg, float *ywin, void *stream)
{
    if (!pool_max || !pool_min || !pool_idx || !scale || !shift || !out || !arg || G <= 0 || C <= 0) return P2C_EINVAL;
    const int bx = C >= 256 ? 256 : ((C + 63) & ~63);
    hipLaunchKernelGGL(pool_select_kernel, dim3(G, p2c_cdiv(C, bx)), dim3(bx), 0, (hipStream_t)stream, pool_max, pool_min, pool_idx, scale,
                       shift, G, C, out, ldo, arg, ywin);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

__global__ void __launch_bounds__(256) maxpool_bwd_kernel(const float *__restrict__ dout, int ldo, const int32_t *__restrict__ arg, int G,
                                                          int ns, int C, float *__restrict__ dZ, int ldz)
{
    const int c = blockIdx.y * blockDim.x + threadIdx.x;
    const int g = blockIdx.x;
    if (c >= C) return;
    const float gv = dout[(size_t)g * ldo + c];
    const int a = arg[(size_t)g * C + c];
    float *d = dZ + (size_t)g * ns * ldz + c;
#pragma unroll 8
    for (int j = 0; j < ns; ++j) d[(size_t)j * ldz] = (j == a) ? gv : 0.f;
}

extern "C" int p2c_maxpool_bwd_f32(const float *dout, int ldo, const int32_t *arg, int G, int ns, int C, float *dZ, int ldz, void *stream)
{
    if (!dout || !arg || !dZ || G <= 0 || ns <= 0 || C <= 0) return P2C_EINVAL;
    const int bx = C >= 256 ? 256 : ((C + 63) & ~63);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(G, p2c_cdiv(C, bx)), dim3(bx), 0, (hipStream_t)stream, dout, ldo, arg, G, ns, C, dZ, ldz);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// ------------------------------------------------------------------------------------------------
// ReLU + train-mode BatchNorm backward, reduction part.
//   g = dZ * [scale*Y+shift > 0];  s1[c] = sum_m g;  s2[c] = sum_m g * (Y-mean)*invstd
// pass 1: block = 64 channels x 4 row-lanes over a chunk of BWD_ROWS rows -> ws[chunk][2][C]
// pass 2: fp64 sum over chunks; dgamma = s2, dbeta = s1; coef = {scale, shift, gs, q, p} with
//         gs = gamma*invstd, q = -gs*invstd*s2/M, p = -gs*s1/M - q*mean   (so dY = gs*g + q*Y + p)
// ------------------------------------------------------------------------------------------------
#define BWD_ROWS 512

__global__ void __launch_bounds__(256) bn_bwd_partial_kernel(const float *__restrict__ dZ, int lddz, const float *__restrict__ Y, int ldy,
                                                             const float *__restrict__ scale, const float *__restrict__ shift,
                                                             const float *__restrict__ mean, const float *__restrict__ invstd,
                                                             long long M, int C, int rows_per_chunk, double *__restrict__ ws)
{
    __shared__ float red[2][4][64];
    const int cx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cx;
    const long long r0 = (long long)blockIdx.x * rows_per_chunk;
    const long long r1 = min(M, r0 + rows_per_chunk);
    float s1 = 0.f, s2 = 0.f;
    if (c < C) {
        const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
#pragma unroll 4
        for (long long r = r0 + ty; r < r1; r += 4) {
            const float y = Y[r * ldy + c];
            const float g = (sc * y + sh > 0.f) ? dZ[r * lddz + c] : 0.f;
            s1 += g;
            s2 += g * ((y - mu) * is);
        }
    }
    red[0][ty][cx] = s1;
    red[1][ty][cx] = s2;
    __syncthreads();
    if (ty == 0 && c < C) {
        double *o = ws + (size_t)(blockIdx.x % P2C_STAT_SLOTS) * 2 * C;
        atomicAdd(&o[c], (double)((red[0][0][cx] + red[0][1][cx]) + (red[0][2][cx] + red[0][3][cx])));
        atomicAdd(&o[C + c], (double)((red[1][0][cx] + red[1][1][cx]) + (red[1][2][cx] + red[1][3][cx])));
    }
}

__global__ void __launch_bounds__(256) bn_bwd_finalize_kernel(const double *__restrict__ ws, int n_chunks, int C, long long M,
                                                             const float *__restrict__ scale, const float *__restrict__ shift,
                                                             const float *__restrict__ mean, const float *__restrict__ invstd,
                                                             const float *__restrict__ gamma, float *__restrict__ dgamma,
                                                             float *__restrict__ dbeta, float *__restrict__ coef)
{
    const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1));
    double s1 = 0.0, s2 = 0.0;
    p2c_sum_slots(ws, n_chunks, C, c, true, s1, s2);
    if (c >= C || threadIdx.x >= FIN_CH) return;
    if (dgamma) dgamma[c] = (float)s2;
    if (dbeta) dbeta[c] = (float)s1;
    const double is = (double)invstd[c], gs = (double)gamma[c] * is;
    const double q = -gs * is * s2 / (double)M;
    const double p = -gs * s1 / (double)M - q * (double)mean[c];
    coef[c] = scale[c];
    coef[C + c] = shift[c];
    coef[2 * C + c] = (float)gs;
    coef[3 * C + c] = (float)q;
    coef[4 * C + c] = (float)p;
}

// Same reduction when the layer feeds the max-pool: dZ is non-zero only at the winner rows, so the sums run over
// the G x C pooled gradients (1/ns of the rows) and read Y at the winners only.
#define POOL_ROWS 64     // G is 1/ns of the rows: small chunks, or a 16k-group layer is reduced by 8 workgroups
__global__ void __launch_bounds__(256) pool_bwd_partial_kernel(const float *__restrict__ dout, int ldo, const float *__restrict__ ywin,
                                                               const float *__restrict__ stat, int G, int C, double *__restrict__ ws)
{
    __shared__ float red[2][4][64];
    const int cx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cx;
    const int g0 = blockIdx.x * POOL_ROWS, g1 = min(G, g0 + POOL_ROWS);
    float s1 = 0.f, s2 = 0.f;
    if (c < C) {
        const float sc = stat[c], sh = stat[C + c], mu = stat[2 * C + c], is = stat[3 * C + c];
#pragma unroll 4
        for (int g = g0 + ty; g < g1; g += 4) {
            const float y = ywin[(size_t)g * C + c];
            const float gr = (sc * y + sh > 0.f) ? dout[(size_t)g * ldo + c] : 0.f;
            s1 += gr;
            s2 += gr * ((y - mu) * is);
        }
    }
    red[0][ty][cx] = s1;
    red[1][ty][cx] = s2;
    __syncthreads();
    if (ty == 0 && c < C) {
        double *o = ws + (size_t)(blockIdx.x % P2C_STAT_SLOTS) * 2 * C;
        atomicAdd(&o[c], (double)((red[0][0][cx] + red[0][1][cx]) + (red[0][2][cx] + red[0][3][cx])));
        atomicAdd(&o[C + c], (double)((red[1][0][cx] + red[1][1][cx]) + (red[1][2][cx] + red[1][3][cx])));
    }
}

extern "C" int p2c_maxpool_bn_bwd_stats_f32(const float *dout, int ldo, const float *ywin, const float *stat, const float *gamma, int G,
                                            int ns, int C, float *dgamma, float *dbeta, float *coef_out, double *slots, void *stream)
{
    if (!dout || !ywin || !stat || !gamma || !coef_out || !slots || G <= 0 || ns <= 0 || C <= 0) return P2C_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int chunks = p2c_cdiv(G, POOL_ROWS);
    hipLaunchKernelGGL(pool_bwd_partial_kernel, dim3(chunks, p2c_cdiv(C, 64)), dim3(256), 0, s, dout, ldo, ywin, stat, G, C, slots);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(p2c_cdiv(C, FIN_CH)), dim3(256), 0, s, (const double *)slots, P2C_STAT_SLOTS, C,
                       (long long)G * ns, stat, stat + C, stat + 2 * C, stat + 3 * C, gamma, dgamma, dbeta, coef_out);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// ------------------------------------------------------------------------------------------------
// Folded first layer.  A stack whose first layer has <= 4 input channels (the grouped relative coordinates of SA1:
// [dx, dy, dz, 0], 1 M rows at config 1) never materialises that layer's output Y0 = X0 W0^T + b0 (268 MB written once and
// read three times): its train-mode BatchNorm statistics follow from the input moments,
//     mean0 = W0 mu + b0,   var0[c] = w_c^T Cov(x) w_c,
// and the layers that consume Y0 rebuild it from the 16-byte input row while they stage their operand tiles.
// p2c_input_moments_f32: out[0:4] = sum_m x, out[4:14] = upper triangle of sum_m x x^T (00 01 02 03 11 12 13 22 23 33), fp64,
// ACCUMULATED into out (zero it first).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) input_moments_kernel(const float *__restrict__ X, int ld, long long M, double *__restrict__ out)
{
    __shared__ double red[4][14];
    double a[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) a[i] = 0.0;
    for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long long)gridDim.x * 256) {
        const float4 v = *reinterpret_cast<const float4 *>(X + m * ld);
        const double x0 = v.x, x1 = v.y, x2 = v.z, x3 = v.w;
        a[0] += x0; a[1] += x1; a[2] += x2; a[3] += x3;
        a[4] += x0 * x0; a[5] += x0 * x1; a[6] += x0 * x2; a[7] += x0 * x3;
        a[8] += x1 * x1; a[9] += x1 * x2; a[10] += x1 * x3;
        a[11] += x2 * x2; a[12] += x2 * x3; a[13] += x3 * x3;
    }
#pragma unroll
    for (int i = 0; i < 14; ++i) {
        const double v = p2c_wave_sum_f64(a[i]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 14) atomicAdd(&out[threadIdx.x], (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

extern "C" int p2c_input_moments_f32(const float *X, int ld, long long M, double *out, void *stream)
{
    if (!X || !out || M <= 0 || ld < 4 || (ld & 3) || ((uintptr_t)X & 15)) return P2C_EINVAL;
    const int blocks = (int)min((long long)p2c_cdiv(M, 256 * 8), 1024LL);
    hipLaunchKernelGGL(input_moments_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, X, ld, M, out);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// scale / shift / mean / invstd (and the running statistics) of the folded layer from the input moments; W0 is [C,4] row-major.
__global__ void bn_finalize_affine_kernel(const double *__restrict__ mom, long long M, const float *__restrict__ W0, const float *__restrict__ b0,
                                          const float *__restrict__ gamma, const float *__restrict__ beta, float eps, float momentum,
                                          float *__restrict__ running_mean, float *__restrict__ running_var, int C, float *__restrict__ st)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double inv = 1.0 / (double)M;
    double mu[4], cov[4][4];
    for (int i = 0; i < 4; ++i) mu[i] = mom[i] * inv;
    int t = 4;
    for (int i = 0; i < 4; ++i)
        for (int j = i; j < 4; ++j) {
            const double v = mom[t++] * inv - mu[i] * mu[j];
            cov[i][j] = v;
            cov[j][i] = v;
        }
    double w[4];
    for (int i = 0; i < 4; ++i) w[i] = (double)W0[c * 4 + i];
    double m = b0 ? (double)b0[c] : 0.0, var = 0.0;
    for (int i = 0; i < 4; ++i) {
        m += w[i] * mu[i];
        for (int j = 0; j < 4; ++j) var += w[i] * w[j] * cov[i][j];
    }
    if (var < 0.0) var = 0.0;
    const float mean = (float)m, invstd = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
        running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * m);
        running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unb);
    }
    const float sc = gamma[c] * invstd;
    st[c] = sc;
    st[C + c] = beta[c] - mean * sc;
    st[2 * C + c] = mean;
    st[3 * C + c] = invstd;
}

extern "C" int p2c_bn_finalize_affine_f32(const double *moments, long long M, const float *W0, const float *b0, const float *gamma,
                                          const float *beta, float eps, float momentum, float *running_mean, float *running_var, int C,
                                          float *stat, void *stream)
{
    if (!moments || !W0 || !gamma || !beta || !stat || M <= 0 || C <= 0) return P2C_EINVAL;
    hipLaunchKernelGGL(bn_finalize_affine_kernel, dim3(p2c_cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, moments, M, W0, b0, gamma, beta, eps,
                       momentum, running_mean, running_var, C, stat);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// Backward of the folded layer from the 5 column sums the consumer's fused backward left in partials5 [slots][5][C]:
//   dgamma = s2, dbeta = s1, and with dY0 = gs*g + q*Y0 + p (gs, q, p as in bn_bwd_finalize):
//   dW0[c, e] = sum_m dY0[m,c] x0[m,e] = gs*G[c,e] + q*(sum_e' W0[c,e'] S2[e',e] + b0[c] S1[e]) + p*S1[e]
// (S1, S2 = the input moments).  The bias in front of a train-mode BatchNorm has an exactly zero gradient.
__device__ __forceinline__ void p2c_sum_copies(const float *__restrict__ src, long long stride, int copies, float *__restrict__ out, long long n,
                                               long long blk);       // (defined below)
__global__ void __launch_bounds__(256) fold0_bwd_finalize_kernel(const double *__restrict__ part, const double *__restrict__ mom, long long M,
                                          const float *__restrict__ W0, const float *__restrict__ b0, const float *__restrict__ stat,
                                          const float *__restrict__ gamma, int C, float *__restrict__ dgamma, float *__restrict__ dbeta,
                                          float *__restrict__ dW0, int lddw0, int fin_blocks, const float *__restrict__ src, long long stride,
                                          int copies, float *__restrict__ out, long long n)
{
    if ((int)blockIdx.x >= fin_blocks) {           // extra workgroups: the per-XCD copies of the NEXT layer's dW summed (p2c_sum_copies_f32)
        p2c_sum_copies(src, stride, copies, out, n, (long long)blockIdx.x - fin_blocks);
        return;
    }
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C || threadIdx.x >= 64) return;
    double sv[5] = {0, 0, 0, 0, 0};
#pragma unroll 16
    for (int t = 0; t < P2C_STAT_SLOTS; ++t)       // 80 independent loads in flight per round: one wave does all of this, latency is all it costs
#pragma unroll
        for (int q = 0; q < 5; ++q) sv[q] += part[((size_t)t * 5 + q) * C + c];
    const double s1 = sv[0], s2 = sv[1];
    dgamma[c] = (float)s2;
    dbeta[c] = (float)s1;
    const double mean = (double)stat[2 * C + c], is = (double)stat[3 * C + c], gs = (double)gamma[c] * is;
    const double q = -gs * is * s2 / (double)M;
    const double p = -gs * s1 / (double)M - q * mean;
    double S2[4][4];
    int t = 4;
    for (int i = 0; i < 4; ++i)
        for (int j = i; j < 4; ++j) { S2[i][j] = mom[t]; S2[j][i] = mom[t]; ++t; }
    const double bb = b0 ? (double)b0[c] : 0.0;
    for (int e = 0; e < 4; ++e) {
        double yx = bb * mom[e];
        for (int k = 0; k < 4; ++k) yx += (double)W0[c * 4 + k] * S2[k][e];
        const double G = e < 3 ? sv[2 + e] : 0.0;
        if (e < lddw0) dW0[c * lddw0 + e] = (float)(gs * G + q * yx + p * mom[e]);
    }
}

extern "C" int p2c_fold0_bwd_finalize_f32(const double *partials5, const double *moments, long long M, const float *W0, const float *b0,
                                          const float *stat0, const float *gamma0, int C0, float *dgamma0, float *dbeta0, float *dW0,
                                          void *stream)
{
    if (!partials5 || !moments || !W0 || !stat0 || !gamma0 || !dgamma0 || !dbeta0 || !dW0 || C0 <= 0 || M <= 0) return P2C_EINVAL;
    hipLaunchKernelGGL(fold0_bwd_finalize_kernel, dim3(p2c_cdiv(C0, 64)), dim3(64), 0, (hipStream_t)stream, partials5, moments, M, W0, b0, stat0,
                       gamma0, C0, dgamma0, dbeta0, dW0, 4, p2c_cdiv(C0, 64), (const float *)nullptr, 0LL, 0, (float *)nullptr, 0LL);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// The same with dW0 written [C0, lddw0] (lddw0 = 3: the parameter's own (C0, 3, 1, 1) layout, no slicing copy afterwards) and, as extra
// workgroups of the launch, the per-XCD copies of the consumer layer's weight gradient summed: out[i] = sum_c src[c * stride + i], i < n.
extern "C" int p2c_fold0_bwd_finalize_sum_f32(const double *partials5, const double *moments, long long M, const float *W0, const float *b0,
                                              const float *stat0, const float *gamma0, int C0, float *dgamma0, float *dbeta0, float *dW0, int lddw0,
                                              const float *src, long long stride, int copies, float *out, long long n, void *stream)
{
    if (!partials5 || !moments || !W0 || !stat0 || !gamma0 || !dgamma0 || !dbeta0 || !dW0 || C0 <= 0 || M <= 0 || lddw0 < 1 || lddw0 > 4)
        return P2C_EINVAL;
    if (!src || !out || copies <= 0 || n <= 0) return P2C_EINVAL;
    const int fin = p2c_cdiv(C0, 64);
    hipLaunchKernelGGL(fold0_bwd_finalize_kernel, dim3(fin + (int)p2c_cdiv(n, 1024)), dim3(256), 0, (hipStream_t)stream, partials5, moments, M, W0, b0,
                       stat0, gamma0, C0, dgamma0, dbeta0, dW0, lddw0, fin, src, stride, copies, out, n);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// Weight gradient of a grouped first layer in the PARAMETER's layout [Co, 3 + Cf] = [coordinate part | feature part] (pointnet_util.py:137
// column order): the coordinate part is the sum of the P2C_STAT_SLOTS fp64 rows p2c_group_linear_bwd_f32 accumulated (dwx [slots][3][Co]),
// the feature part the first Cf columns of the GEMM's dW [Co, lddw] - one launch instead of a reduction, a transpose / cast and a concatenation.
__global__ void __launch_bounds__(256) group_weight_grad_kernel(const double *__restrict__ dwx, int Cs, const float *__restrict__ dW, int lddw, int Co,
                                                                int Cf, float *__restrict__ out)
{
    const int c = blockIdx.x;
    const int ldo = 3 + Cf;
    if (threadIdx.x < 3) {
        double a = 0.0;
#pragma unroll 8
        for (int t = 0; t < P2C_STAT_SLOTS; ++t) a += dwx[((size_t)t * 3 + threadIdx.x) * Cs + c];
        out[(size_t)c * ldo + threadIdx.x] = (float)a;
    }
    for (int k = threadIdx.x; k < Cf; k += 256) out[(size_t)c * ldo + 3 + k] = dW[(size_t)c * lddw + k];
}

extern "C" int p2c_group_weight_grad_f32(const double *dwx_slots, int Cs, const float *dW, int lddw, int Co, int Cf, float *out, void *stream)
{
    if (!dwx_slots || !dW || !out || Co <= 0 || Cf <= 0 || Cs < Co || lddw < Cf) return P2C_EINVAL;
    hipLaunchKernelGGL(group_weight_grad_kernel, dim3(Co), dim3(256), 0, (hipStream_t)stream, dwx_slots, Cs, dW, lddw, Co, Cf, out);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// out[g, c] = sum over the rpg rows of group g of dY[m, c],  dY = gs*(dZ*[scale*Y+shift>0]) + q*Y + p  (coef [5][C]):
// the gradient of a per-group additive term (p2c_linear_fwd_gbias_f32).  One workgroup per (group, 64 channels).
__global__ void __launch_bounds__(256) group_colsum_bn_kernel(const float *__restrict__ dz, int lddz, const float *__restrict__ y, int ldy,
                                                              const float *__restrict__ coef, int rpg, int C, float *__restrict__ out, int ldo)
{
    __shared__ float red[4][64];
    const int cx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cx, g = blockIdx.x;
    float s = 0.f;
    if (c < C) {
        const float sc = coef[c], sh = coef[C + c], gs = coef[2 * C + c], q = coef[3 * C + c], p = coef[4 * C + c];
        for (int r = ty; r < rpg; r += 4) {
            const size_t m = (size_t)g * rpg + r;
            const float yy = y[m * ldy + c], gg = dz[m * lddz + c];
            s += __builtin_fmaf(gs, (sc * yy + sh > 0.f) ? gg : 0.f, __builtin_fmaf(q, yy, p));
        }
    }
    red[ty][cx] = s;
    __syncthreads();
    if (ty == 0 && c < C) out[(size_t)g * ldo + c] = (red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx]);
}

extern "C" int p2c_group_colsum_bn_f32(const float *dz, int lddz, const float *y, int ldy, const float *coef, int G, int rows_per_group, int C,
                                       float *out, int ldo, void *stream)
{
    if (!dz || !y || !coef || !out || G <= 0 || rows_per_group <= 0 || C <= 0) return P2C_EINVAL;
    hipLaunchKernelGGL(group_colsum_bn_kernel, dim3(G, p2c_cdiv(C, 64)), dim3(256), 0, (hipStream_t)stream, dz, lddz, y, ldy, coef, rows_per_group,
                       C, out, ldo);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

extern "C" size_t p2c_stat_slots_bytes(int C) { return (size_t)P2C_STAT_SLOTS * 2 * (size_t)C * sizeof(double); }

// The per-XCD copies of a weight gradient (csrc/bwd_fused3.hip flushes dW into the copy of the workgroup's XCD) summed into one matrix,
// copy 0 first: out[i] = ((c0 + c1) + ... ) - alone, or as extra workgroups of the finalize launch that follows a fused backward kernel
// anyway (a reduction launch of its own per layer was 8 x 6 us of a 4.2 ms step).
__device__ __forceinline__ void p2c_sum_copies(const float *__restrict__ src, long long stride, int copies, float *__restrict__ out, long long n,
                                               long long blk)
{
    const long long i = (blk * 256 + threadIdx.x) * 4;
    if (i + 3 < n && ((stride | (long long)((uintptr_t)src >> 2) | (long long)((uintptr_t)out >> 2)) & 3) == 0) {
        float4 a = *reinterpret_cast<const float4 *>(src + i);
        for (int c = 1; c < copies; ++c) {
            const float4 v = *reinterpret_cast<const float4 *>(src + c * stride + i);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        *reinterpret_cast<float4 *>(out + i) = a;
    } else {
        for (long long j = i; j < min(i + 4, n); ++j) {
            float a = src[j];
            for (int c = 1; c < copies; ++c) a += src[c * stride + j];
            out[j] = a;
        }
    }
}

__global__ void __launch_bounds__(256) sum_copies_kernel(const float *__restrict__ src, long long stride, int copies, float *__restrict__ out, long long n)
{
    p2c_sum_copies(src, stride, copies, out, n, blockIdx.x);
}

extern "C" int p2c_sum_copies_f32(const float *src, long long stride, int copies, float *out, long long n, void *stream)
{
    if (!src || !out || copies <= 0 || n <= 0) return P2C_EINVAL;
    hipLaunchKernelGGL(sum_copies_kernel, dim3(p2c_cdiv(n, 1024)), dim3(256), 0, (hipStream_t)stream, src, stride, copies, out, n);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

__global__ void __launch_bounds__(256) bn_bwd_finalize_sum_kernel(const double *__restrict__ ws, int n_chunks, int C, long long M,
                                                                 const float *__restrict__ stat, const float *__restrict__ gamma,
                                                                 float *__restrict__ dgamma, float *__restrict__ dbeta, float *__restrict__ coef,
                                                                 int fin_blocks, const float *__restrict__ src, long long stride, int copies,
                                                                 float *__restrict__ out, long long n)
{
    if ((int)blockIdx.x >= fin_blocks) {
        p2c_sum_copies(src, stride, copies, out, n, (long long)blockIdx.x - fin_blocks);
        return;
    }
    const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1));
    double s1 = 0.0, s2 = 0.0;
    p2c_sum_slots(ws, n_chunks, C, c, true, s1, s2);
    if (c >= C || threadIdx.x >= FIN_CH) return;
    if (dgamma) dgamma[c] = (float)s2;
    if (dbeta) dbeta[c] = (float)s1;
    const double is = (double)stat[3 * C + c], gs = (double)gamma[c] * is;
    const double q = -gs * is * s2 / (double)M;
    const double p = -gs * s1 / (double)M - q * (double)stat[2 * C + c];
    coef[c] = stat[c];
    coef[C + c] = stat[C + c];
    coef[2 * C + c] = (float)gs;
    coef[3 * C + c] = (float)q;
    coef[4 * C + c] = (float)p;
}

// p2c_bn_bwd_finalize_f32 + p2c_sum_copies_f32 in one launch
extern "C" int p2c_bn_bwd_finalize_sum_f32(const double *slots, int C, long long M, const float *stat, const float *gamma, float *dgamma,
                                           float *dbeta, float *coef_out, const float *src, long long stride, int copies, float *out, long long n,
                                           void *stream)
{
    if (!slots || !stat || !gamma || !coef_out || C <= 0 || !src || !out || copies <= 0 || n <= 0) return P2C_EINVAL;
    const int fin = p2c_cdiv(C, FIN_CH);
    hipLaunchKernelGGL(bn_bwd_finalize_sum_kernel, dim3(fin + p2c_cdiv(n, 1024)), dim3(256), 0, (hipStream_t)stream, slots, P2C_STAT_SLOTS, C, M, stat,
                       gamma, dgamma, dbeta, coef_out, fin, src, stride, copies, out, n);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// finalize from per-tile partials produced elsewhere (the fused backward-data epilogue): stat = [scale|shift|mean|invstd] x C
extern "C" int p2c_bn_bwd_finalize_f32(const double *slots, int C, long long M, const float *stat, const float *gamma, float *dgamma,
                                       float *dbeta, float *coef_out, void *stream)
{
    if (!slots || !stat || !gamma || !coef_out || C <= 0) return P2C_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(p2c_cdiv(C, FIN_CH)), dim3(256), 0, s, slots, P2C_STAT_SLOTS, C, M, stat, stat + C,
                       stat + 2 * C, stat + 3 * C, gamma, dgamma, dbeta, coef_out);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

extern "C" int p2c_bn_relu_bwd_stats_f32(const float *dZ, int lddz, const float *Y, int ldy, const float *scale, const float *shift,
                                         const float *mean, const float *invstd, const float *gamma, int M, int C, float *dgamma,
                                         float *dbeta, float *coef_out, double *slots, void *stream)
{
    if (!dZ || !Y || !scale || !shift || !mean || !invstd || !gamma || !coef_out || !slots || M <= 0 || C <= 0) return P2C_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    // rows per workgroup: 512 for the long layers, down to 32 for the deep small-M ones (4096 rows in 512-row chunks are
    // 8 x C/64 workgroups, each a 128-step serial loop: 65 us for a 4 MB reduction)
    int rows = BWD_ROWS;
    while (rows > 32 && (long long)p2c_cdiv(M, rows) * p2c_cdiv(C, 64) < 1024) rows >>= 1;
    const int chunks = p2c_cdiv(M, rows);
    hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(chunks, p2c_cdiv(C, 64)), dim3(256), 0, s, dZ, lddz, Y, ldy, scale, shift, mean, invstd,
                       (long long)M, C, rows, slots);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(p2c_cdiv(C, FIN_CH)), dim3(256), 0, s, (const double *)slots, P2C_STAT_SLOTS, C, (long long)M, scale, shift,
                       mean, invstd, gamma, dgamma, dbeta, coef_out);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// ------------------------------------------------------------------------------------------------
// Adam over a LIST of tensors in one launch (train_Point2Cyl_without_sketch.py:204, :369 torch.optim.Adam with default
// betas / eps, no weight decay, no amsgrad).  torch's multi-tensor path gives every 65536-element chunk to ONE block
// (1.4 M parameters in 123 tensors -> ~140 blocks, 2 launches, ~88 us per step); here a block takes 1024 elements.
//   tab  [n_tensors][4] int64: param, grad, exp_avg, exp_avg_sq pointers;  numel [n_tensors] int64
//   chunks [n_chunks][2] int32: (tensor, first element / 1024)
// Update rule, in torch's order of operations:
//   m = m + (g - m)*(1-b1);  v = v*b2 + g*g*(1-b2);  p = p - (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) adam_multi_kernel(const long long *__restrict__ tab, const long long *__restrict__ numel,
                                                         const int32_t *__restrict__ chunks, float step_size, float b1, float b2,
                                                         float eps, float inv_sqrt_bc2)
{
    const int t = chunks[2 * blockIdx.x], c = chunks[2 * blockIdx.x + 1];
    float *p = reinterpret_cast<float *>(tab[4 * t + 0]);
    const float *g = reinterpret_cast<const float *>(tab[4 * t + 1]);
    float *m = reinterpret_cast<float *>(tab[4 * t + 2]);
    float *v = reinterpret_cast<float *>(tab[4 * t + 3]);
    const long long n = numel[t];
    const long long i0 = (long long)c * 1024 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long i = i0 + 256 * k;
        if (i < n) {
            const float gi = g[i];
            const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
            const float vi = v[i] * b2 + gi * gi * (1.0f - b2);
            m[i] = mi;
            v[i] = vi;
            p[i] = p[i] - step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
        }
    }
}

extern "C" int p2c_adam_multi_f32(const long long *table, const long long *numel, const int32_t *chunks, int n_chunks, float lr, float beta1,
                                  float beta2, float eps, long long step, void *stream)
{
    if (!table || !numel || !chunks || n_chunks <= 0 || step <= 0) return P2C_EINVAL;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_multi_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, table, numel, chunks, (float)((double)lr / bc1), beta1,
                       beta2, eps, (float)(1.0 / sqrt(bc2)));
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}


// ------------------------------------------------------------------------------------------------
// Batched 2-D copies.  A training step prepares a dozen small operands from the parameters - 3 -> 4 / 259 -> 260 / 19 -> 20 zero-padded
// weights, the [xyz | features] -> [features | xyz] column order of the grouped layers, the column blocks of a weight that multiply
// different inputs - each with its own torch copy / cat launch (~4 us of the stream apiece, profiles/r04_torch_glue_ops.log).  The
// sources and destinations never move, so ONE launch driven by a device-resident table does all of them.
// table: n entries {src, dst, rows, cols, ld_src, ld_dst}; grid (n, row slices).
// ------------------------------------------------------------------------------------------------
struct P2cCopy2D { const float *src; float *dst; int rows, cols, lds, ldd; };
__global__ void __launch_bounds__(256) copy2d_batch_kernel(const P2cCopy2D *__restrict__ tab)
{
    const P2cCopy2D d = tab[blockIdx.x];
    for (int r = blockIdx.y; r < d.rows; r += gridDim.y)
        for (int c = threadIdx.x; c < d.cols; c += 256) d.dst[(size_t)r * d.ldd + c] = d.src[(size_t)r * d.lds + c];
}

extern "C" int p2c_copy2d_batch_f32(const void *table, int n, void *stream)
{
    if (!table || n <= 0) return P2C_EINVAL;
    hipLaunchKernelGGL(copy2d_batch_kernel, dim3(n, 32), dim3(256), 0, (hipStream_t)stream, (const P2cCopy2D *)table);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// The same launch also advances the step's device-resident 64-bit counters (the 17 num_batches_tracked of the BatchNorms, += 1 each, and
// the dropout hash's seed, += its golden-ratio stride): a multi-tensor add and a scalar add launch per forward otherwise (~5 us each).
// counters (device memory): n_counters entries of struct { long long *ptr; long long inc; }; two's-complement wrap-around like torch's int64 add.
struct P2cCounter { unsigned long long *ptr; unsigned long long inc; };
__global__ void __launch_bounds__(256) copy2d_batch_inc_kernel(const P2cCopy2D *__restrict__ tab, int n, const P2cCounter *__restrict__ ctab, int nc)
{
    if ((int)blockIdx.x == n) {
        if (blockIdx.y == 0)
            for (int i = threadIdx.x; i < nc; i += 256) *ctab[i].ptr += ctab[i].inc;
        return;
    }
    const P2cCopy2D d = tab[blockIdx.x];
    for (int r = blockIdx.y; r < d.rows; r += gridDim.y)
        for (int c = threadIdx.x; c < d.cols; c += 256) d.dst[(size_t)r * d.ldd + c] = d.src[(size_t)r * d.lds + c];
}

extern "C" int p2c_copy2d_batch_inc_f32(const void *table, int n, const void *counters, int n_counters, void *stream)
{
    if (n < 0 || n_counters < 0 || (n > 0 && !table) || (n_counters > 0 && !counters) || n + n_counters == 0) return P2C_EINVAL;
    hipLaunchKernelGGL(copy2d_batch_inc_kernel, dim3(n + (n_counters > 0 ? 1 : 0), n > 0 ? 32 : 1), dim3(256), 0, (hipStream_t)stream,
                       (const P2cCopy2D *)table, n, (const P2cCounter *)counters, n_counters);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// ------------------------------------------------------------------------------------------------
// n flat copies (contiguous buffers of 4-byte words) in ONE launch, the descriptors passed BY VALUE in the kernel arguments - a captured
// graph node then holds them, no device table to keep in step with tensors that are re-allocated by every capture.  Used for the hand-over
// of the prefetched geometry (graph.py): torch's multi-tensor copy moves those ~40 MB in two launches of ~21 us (one block per 64 K
// elements of a tensor); here every block takes 4096-word chunks of whatever tensor its index falls into, 16 bytes per lane.
// ------------------------------------------------------------------------------------------------
#define P2C_FLAT_MAX 32
struct P2cFlatBatch {
    const uint32_t *src[P2C_FLAT_MAX];
    uint32_t *dst[P2C_FLAT_MAX];
    long long words[P2C_FLAT_MAX];
    int chunk0[P2C_FLAT_MAX + 1];          // first chunk of entry e (4096 words per chunk), chunk0[n] = total
    int n;
};
__global__ void __launch_bounds__(256) copy_flat_batch_kernel(P2cFlatBatch d)
{
    for (int blk = blockIdx.x; blk < d.chunk0[d.n]; blk += gridDim.x) {
        int e = 0;
        while (e + 1 < d.n && blk >= d.chunk0[e + 1]) ++e;                       // uniform, <= 32 steps over kernel-argument (scalar) memory
        const long long w0 = (long long)(blk - d.chunk0[e]) * 4096, nw = min(4096LL, d.words[e] - w0);
        const uint32_t *s = d.src[e] + w0;
        uint32_t *o = d.dst[e] + w0;
        if (((((uintptr_t)s) | ((uintptr_t)o)) & 15) == 0) {
            const int nv = (int)(nw >> 2);
            for (int i = threadIdx.x; i < nv; i += 256) reinterpret_cast<uint4 *>(o)[i] = reinterpret_cast<const uint4 *>(s)[i];
            for (int i = (nv << 2) + threadIdx.x; i < nw; i += 256) o[i] = s[i];
        } else {
            for (int i = threadIdx.x; i < nw; i += 256) o[i] = s[i];
        }
    }
}

// srcs / dsts / nbytes: HOST arrays of n device pointers / byte counts (multiples of 4).  Any n (split into launches of 32).
extern "C" int p2c_copy_flat_batch(const void *const *srcs, void *const *dsts, const long long *nbytes, int n, void *stream)
{
    if (!srcs || !dsts || !nbytes || n <= 0) return P2C_EINVAL;
    for (int i = 0; i < n; ++i)
        if (!srcs[i] || !dsts[i] || nbytes[i] < 0 || (nbytes[i] & 3) || (((uintptr_t)srcs[i] | (uintptr_t)dsts[i]) & 3)) return P2C_EINVAL;
    for (int base = 0; base < n; base += P2C_FLAT_MAX) {
        P2cFlatBatch d;
        d.n = n - base < P2C_FLAT_MAX ? n - base : P2C_FLAT_MAX;
        long long chunks = 0;
        for (int i = 0; i < d.n; ++i) {
            d.src[i] = (const uint32_t *)srcs[base + i]; d.dst[i] = (uint32_t *)dsts[base + i]; d.words[i] = nbytes[base + i] / 4;
            d.chunk0[i] = (int)chunks;
            chunks += (d.words[i] + 4095) / 4096;
        }
        if (chunks > 0x7fffffffLL) return P2C_EINVAL;
        d.chunk0[d.n] = (int)chunks;
        for (int i = d.n; i < P2C_FLAT_MAX; ++i) { d.src[i] = nullptr; d.dst[i] = nullptr; d.words[i] = 0; d.chunk0[i + 1] = (int)chunks; }
        if (chunks == 0) continue;
        const int grid = chunks < 4096 ? (int)chunks : 4096;
        hipLaunchKernelGGL(copy_flat_batch_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, d);
    }
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}
