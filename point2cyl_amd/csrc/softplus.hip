// softplus.hip -- the activation of the sketch branch's implicit decoder (IGR/network.py:58-59: nn.Softplus(beta=100)) and the two
// derivatives of it that a training step needs.  The decoder is differentiated twice (d f / d point feeds the eikonal / normal
// losses, train_Point2Cyl.py:619-646), so next to h = softplus(z) the step evaluates u*s(z) in the gradient pass and, in the
// backward of THAT, g*s(z) and g*u*beta*s(z)(1 - s(z)) (s = sigmoid(beta z)).  torch runs these as chains of elementwise
// launches over [rows, 512] tensors (0.5 - 1.2 GB each at the trainer's sizes); here each is one pass: 16-byte loads, one
// exp per element, HBM-bound.  Same branch structure as torch's kernels: beta*z > threshold is the linear region (derivative 1,
// second derivative 0).
#include "common.h"

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sp_sigmoid(float bz) { const float e = __expf(-fabsf(bz)); const float s = 1.f / (1.f + e); return bz >= 0.f ? s : 1.f - s; }

// MODE 0: h = softplus(z)   1: o = u * s(z)   2: du = g * s(z), dz = g * u * beta * s (1 - s)
//      3: du = g * s(z), dz = g * a * beta * (1 - s)  with a = P * s(z) given instead of P (u carries a)
template <int MODE>
__device__ __forceinline__ void softplus_elem(float z, float u, float g, float beta, float inv_beta, float thr, float &a, float &b)
{
    const float bz = z * beta;
    const bool lin = bz > thr;
    b = 0.f;
    if (MODE == 0) {
        const float e = __expf(-fabsf(bz));          // in (0, 1]
        // log1p(e): three terms of the series below 2^-7 (error < e^4/4 < 1e-9 relative), the hardware log above it
        const float l1p = e < 0.0078125f ? e * (1.f - e * (0.5f - e * 0.33333334f)) : __logf(1.f + e);
        a = lin ? z : (fmaxf(bz, 0.f) + l1p) * inv_beta;
    } else {
        const float s = lin ? 1.f : sp_sigmoid(bz);
        if (MODE == 1) {
            a = u * s;
        } else {
            a = g * s;
            b = lin ? 0.f : g * u * beta * (MODE == 3 ? 1.f : s) * (1.f - s);
        }
    }
}

template <int MODE>
__global__ void __launch_bounds__(256) softplus_kernel(const float *__restrict__ z, const float *__restrict__ u, const float *__restrict__ g,
                                                       float *__restrict__ o0, float *__restrict__ o1, size_t n, float beta, float thr)
{
    const float inv_beta = 1.f / beta;
    const size_t n4 = n / 4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const v4f zz = reinterpret_cast<const v4f *>(z)[i];
        v4f a = {0.f, 0.f, 0.f, 0.f}, b = a, uu = a, gg = a;
        if (MODE >= 1) uu = reinterpret_cast<const v4f *>(u)[i];
        if (MODE >= 2) gg = reinterpret_cast<const v4f *>(g)[i];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float av, bv;
            softplus_elem<MODE>(zz[c], uu[c], gg[c], beta, inv_beta, thr, av, bv);
            a[c] = av; b[c] = bv;
        }
        reinterpret_cast<v4f *>(o0)[i] = a;
        if (MODE >= 2) reinterpret_cast<v4f *>(o1)[i] = b;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {          // the last n % 4 elements
        const size_t i = n4 * 4 + threadIdx.x;
        float av, bv;
        softplus_elem<MODE>(z[i], MODE >= 1 ? u[i] : 0.f, MODE >= 2 ? g[i] : 0.f, beta, inv_beta, thr, av, bv);
        o0[i] = av;
        if (MODE >= 2) o1[i] = bv;
    }
}

static int softplus_launch(int mode, const float *z, const float *u, const float *g, float *o0, float *o1, long long n, float beta, float thr, void *stream)
{
    if (!z || !o0 || n <= 0 || beta <= 0.f || (mode >= 1 && !u) || (mode >= 2 && (!g || !o1))) return P2C_EINVAL;
    if (((uintptr_t)z | (uintptr_t)o0 | (uintptr_t)u | (uintptr_t)g | (uintptr_t)o1) & 15) return P2C_EALIGN;
    const size_t n4 = (size_t)n / 4;
    const int grid = (int)(n4 < (size_t)256 * 2048 ? (n4 + 255) / 256 + 1 : 2048);
    hipStream_t s = (hipStream_t)stream;
    if (mode == 0) hipLaunchKernelGGL(softplus_kernel<0>, dim3(grid), dim3(256), 0, s, z, u, g, o0, o1, (size_t)n, beta, thr);
    else if (mode == 1) hipLaunchKernelGGL(softplus_kernel<1>, dim3(grid), dim3(256), 0, s, z, u, g, o0, o1, (size_t)n, beta, thr);
    else if (mode == 2) hipLaunchKernelGGL(softplus_kernel<2>, dim3(grid), dim3(256), 0, s, z, u, g, o0, o1, (size_t)n, beta, thr);
    else hipLaunchKernelGGL(softplus_kernel<3>, dim3(grid), dim3(256), 0, s, z, u, g, o0, o1, (size_t)n, beta, thr);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

extern "C" int p2c_softplus_fwd_f32(const float *z, float *h, long long n, float beta, float threshold, void *stream)
{
    return softplus_launch(0, z, nullptr, nullptr, h, nullptr, n, beta, threshold, stream);
}
extern "C" int p2c_softplus_bwd_f32(const float *u, const float *z, float *out, long long n, float beta, float threshold, void *stream)
{
    return softplus_launch(1, z, u, nullptr, out, nullptr, n, beta, threshold, stream);
}
extern "C" int p2c_softplus_bwd_bwd_f32(const float *g, const float *u, const float *z, float *du, float *dz, long long n, float beta, float threshold,
                                        void *stream)
{
    return softplus_launch(2, z, u, g, du, dz, n, beta, threshold, stream);
}
// backward of a = (A W) * s(z) (p2c_linear_bwd_data_sig_f32) w.r.t. the product and z, from a itself: t = g * s(z) (what goes back
// through the product) and dz = g * a * beta * (1 - s(z))
extern "C" int p2c_softplus_sig_bwd_f32(const float *g, const float *a, const float *z, float *t, float *dz, long long n, float beta, float threshold,
                                        void *stream)
{
    return softplus_launch(3, z, a, g, t, dz, n, beta, threshold, stream);
}

// ------------------------------------------------------------------------------------------------
// Row-structured variants for the ends of the decoder (its last layer has ONE output, its input gradient is used in TWO columns): the
// products there are outer products / dot products with a single weight row, so they ride on the activation passes instead of being
// [M x 4]-padded GEMMs over 0.5 GB operands.  z, q, e, outputs: [M, K] contiguous, K % 4 == 0; one wave per row, 16 bytes per lane.
//   ROWMODE 0  out[m]   = softplus(z[m,:]) . w + bias                          (forward of the last layer; the activation is never stored)
//   ROWMODE 1  o[m,c]   = gs(m) * w[c] * s(z[m,c]) (+ q[m,c])                  (gradient of the last layer's input: gs = 1 or g[m])
//   ROWMODE 2  E[m,c]   = (Ein[m,c]) + ga[m,0] wa[c] + ga[m,1] wb[c];  t = E s(z),  dz = E e beta (1 - s)      (sig_bwd with a rank-2 term)
// ------------------------------------------------------------------------------------------------
struct RowArgs {
    const float *z; const float *w; const float *wb; const float *g; int ldg; const float *q; const float *e;
    float *o0; float *o1; long long M; int K; float beta, thr; const float *bias;
};

template <int ROWMODE>
__global__ void __launch_bounds__(256) softplus_row_kernel(RowArgs a)
{
    const int lane = threadIdx.x & 63;
    const long long wave0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long long)gridDim.x * 4;
    const int K4 = a.K / 4;
    const float inv_beta = 1.f / a.beta;
    for (long long m = wave0; m < a.M; m += nw) {
        const v4f *zr = reinterpret_cast<const v4f *>(a.z + (size_t)m * a.K);
        float dot = 0.f;
        float g0 = 1.f, g1 = 0.f;
        if (ROWMODE == 1 && a.g) g0 = a.g[(size_t)m * a.ldg];
        if (ROWMODE == 2) { g0 = a.g[(size_t)m * a.ldg]; g1 = a.g[(size_t)m * a.ldg + 1]; }
        for (int c4 = lane; c4 < K4; c4 += 64) {
            const v4f zz = zr[c4];
            const v4f ww = reinterpret_cast<const v4f *>(a.w)[c4];
            if (ROWMODE == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float h, unused;
                    softplus_elem<0>(zz[c], 0.f, 0.f, a.beta, inv_beta, a.thr, h, unused);
                    dot += h * ww[c];
                }
            } else if (ROWMODE == 1) {
                v4f o;
                v4f qq = {0.f, 0.f, 0.f, 0.f};
                if (a.q) qq = reinterpret_cast<const v4f *>(a.q + (size_t)m * a.K)[c4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float bz = zz[c] * a.beta;
                    const float s = bz > a.thr ? 1.f : sp_sigmoid(bz);
                    o[c] = g0 * ww[c] * s + qq[c];
                }
                reinterpret_cast<v4f *>(a.o0 + (size_t)m * a.K)[c4] = o;
            } else {
                const v4f wb = reinterpret_cast<const v4f *>(a.wb)[c4];
                const v4f ee = reinterpret_cast<const v4f *>(a.e + (size_t)m * a.K)[c4];
                v4f Ein = {0.f, 0.f, 0.f, 0.f};
                if (a.q) Ein = reinterpret_cast<const v4f *>(a.q + (size_t)m * a.K)[c4];
                v4f t, dz;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float E = Ein[c] + (g0 * ww[c] + g1 * wb[c]);
                    float tv, dv;
                    softplus_elem<3>(zz[c], ee[c], E, a.beta, inv_beta, a.thr, tv, dv);
                    t[c] = tv; dz[c] = dv;
                }
                reinterpret_cast<v4f *>(a.o0 + (size_t)m * a.K)[c4] = t;
                reinterpret_cast<v4f *>(a.o1 + (size_t)m * a.K)[c4] = dz;
            }
        }
        if (ROWMODE == 0) {
            dot = p2c_wave_sum_f32(dot);
            if (lane == 0) a.o0[m] = dot + (a.bias ? a.bias[0] : 0.f);
        }
    }
}

static int softplus_row_launch(int mode, const RowArgs &a, void *stream)
{
    if (!a.z || !a.w || !a.o0 || a.M <= 0 || a.K <= 0 || (a.K & 3) || a.beta <= 0.f) return P2C_EINVAL;
    if (((uintptr_t)a.z | (uintptr_t)a.w | (uintptr_t)a.wb | (uintptr_t)a.q | (uintptr_t)a.e | (uintptr_t)a.o1) & 15) return P2C_EALIGN;
    if (mode != 0 && ((uintptr_t)a.o0 & 15)) return P2C_EALIGN;
    const long long blocks = (a.M + 3) / 4;
    const int grid = (int)(blocks < 8192 ? blocks : 8192);
    hipStream_t s = (hipStream_t)stream;
    if (mode == 0) hipLaunchKernelGGL(softplus_row_kernel<0>, dim3(grid), dim3(256), 0, s, a);
    else if (mode == 1) hipLaunchKernelGGL(softplus_row_kernel<1>, dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(softplus_row_kernel<2>, dim3(grid), dim3(256), 0, s, a);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// out[m] = softplus(z[m,:K]) . w[:K] + bias[0] (bias: device scalar or NULL)
extern "C" int p2c_softplus_dot_f32(const float *z, const float *w, const float *bias, float *out, long long M, int K, float beta, float threshold,
                                    void *stream)
{
    RowArgs a{z, w, nullptr, nullptr, 0, nullptr, nullptr, out, nullptr, M, K, beta, threshold, bias};
    return softplus_row_launch(0, a, stream);
}
// o[m,c] = (g ? g[m*ldg] : 1) * w[c] * sigmoid(beta z[m,c]) + (q ? q[m,c] : 0)
extern "C" int p2c_softplus_row_bwd_f32(const float *g, int ldg, const float *w, const float *z, const float *q, float *o, long long M, int K, float beta,
                                        float threshold, void *stream)
{
    RowArgs a{z, w, nullptr, g, ldg, q, nullptr, o, nullptr, M, K, beta, threshold, nullptr};
    return softplus_row_launch(1, a, stream);
}
// E = (Ein ? Ein : 0) + ga[m*ldg] wa[c] + ga[m*ldg+1] wb[c];  t = E * s(z),  dz = E * e * beta * (1 - s(z))   (p2c_softplus_sig_bwd_f32 with a rank-2 term)
extern "C" int p2c_softplus_sig_bwd_rank2_f32(const float *ga, int ldg, const float *wa, const float *wb, const float *Ein, const float *e, const float *z,
                                              float *t, float *dz, long long M, int K, float beta, float threshold, void *stream)
{
    if (!ga || !wb || !e || !t || !dz) return P2C_EINVAL;
    RowArgs a{z, wa, wb, ga, ldg, Ein, e, t, dz, M, K, beta, threshold, nullptr};
    return softplus_row_launch(2, a, stream);
}
