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
