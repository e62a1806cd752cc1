// loss.hip -- the three training losses of Point2Cyl-without-sketch fused into two passes over the head output.
//
// The reference evaluates them as ~60 small torch ops (losses.py:90-143, :317-351; the base/barrel block inline in
// train_Point2Cyl_without_sketch.py:283-307) on (B,N,2K) / (B,N,K) tensors.  Here the head output `heads`
// [M = B*N, ld] (columns: 3 normal components, then 2K segmentation logits) is read twice:
//   pass 1 (loss_reduce_kernel): per cloud, the sums every loss needs -- for the mIoU  dot[k] = sum_n 1[gt=k] Wr[n,k],
//           sumW[k] = sum_n Wr[n,k], cnt[k] = sum_n 1[gt=k]  (Wr = W reordered by the Hungarian matching), the
//           normal term sum_n (1 - |X.n_gt|) and the base/barrel term sum_n sum_k q_k ce_k;
//   finalize (one small block): the three scalars and the per-(b,k) mIoU gradient coefficients;
//   pass 2 (loss_grad_kernel): d total / d heads, written in the layout of `heads` (no slicing / cat kernels).
// Per point (all in registers): p = softmax(logits[2K]); W[k] = p[2k] + p[2k+1] (:254-265); X = x/max(|x|,1e-12) (:247);
// q = softmax_k(mask_k * W[match_k]) (:289-290); ce_k = logaddexp(l[2k], l[2k+1]) - l[2k + bb]  (:295-303, the reference
// gathers the RAW logits of column k for the k-th reordered segment; sum_k ce_k*q_k does not depend on its sort).
#include "common.h"

#define LOSS_MAXK 8

struct LossArgs {
    const float *heads; int ld; int xoff; int woff;      // X at cols [xoff, xoff+3), logits at [woff, woff+2K)
    const float *ngt;                                    // [M,3]
    const int64_t *igt, *bbgt;                           // [M]
    const int64_t *match; const uint8_t *mask;           // [B,K]
    int B, N, K;
    float w_seg, w_normal, w_bb;
    double *acc;                                         // [B][3K+2] zeroed: dot[K] | sumW[K] | cnt[K] | normal | bb
    float *out;                                          // [4]: total, normal, miou, bb
    float *coef;                                         // [B][2K]: a_bk (on the gt point) | c_bk (on every point)
    float *dheads;                                       // [M, ld]
    const float *gscale;                                 // upstream d / d total (device scalar) the gradient is multiplied by; NULL = 1
};

struct PointEval {
    float p[2 * LOSS_MAXK], W[LOSS_MAXK], q[LOSS_MAXK], ce[LOSS_MAXK], sb[LOSS_MAXK];   // sb = sigmoid-like P(barrel | pair k)
    float X[3], inv_norm, cosv, bbsum;
};

template <int K>
__device__ __forceinline__ void eval_point(const LossArgs &a, size_t m, int b, int bb, PointEval &e)
{
    const float *h = a.heads + m * a.ld;
    float l[2 * K], mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 2 * K; ++j) { l[j] = h[a.woff + j]; mx = fmaxf(mx, l[j]); }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 2 * K; ++j) { e.p[j] = expf(l[j] - mx); s += e.p[j]; }
    const float inv = 1.f / s;
#pragma unroll
    for (int j = 0; j < 2 * K; ++j) e.p[j] *= inv;
#pragma unroll
    for (int k = 0; k < K; ++k) e.W[k] = e.p[2 * k] + e.p[2 * k + 1];
    // q = softmax over k of mask_k * W[match_k]
    float u[K], qs = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int mk = (int)a.match[b * K + k];
        float wr = 0.f;
#pragma unroll
        for (int j = 0; j < K; ++j) wr = (j == mk) ? e.W[j] : wr;
        u[k] = a.mask[b * K + k] ? wr : 0.f;
        e.q[k] = expf(u[k] - 1.f);          // u in [0,1]: shift by the upper bound
        qs += e.q[k];
    }
    const float qinv = 1.f / qs;
    e.bbsum = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        e.q[k] *= qinv;
        const float lb = l[2 * k], lc = l[2 * k + 1];
        const float hi = fmaxf(lb, lc), lo = fminf(lb, lc);
        const float lse = hi + log1pf(expf(lo - hi));
        e.ce[k] = lse - (bb == 0 ? lb : lc);
        e.sb[k] = expf(lb - lse);
        e.bbsum += e.q[k] * e.ce[k];
    }
    const float x0 = h[a.xoff], x1 = h[a.xoff + 1], x2 = h[a.xoff + 2];
    const float nrm = fmaxf(sqrtf(x0 * x0 + x1 * x1 + x2 * x2), 1e-12f);
    e.inv_norm = 1.f / nrm;
    e.X[0] = x0 * e.inv_norm; e.X[1] = x1 * e.inv_norm; e.X[2] = x2 * e.inv_norm;
    const float *g = a.ngt + m * 3;
    e.cosv = e.X[0] * g[0] + e.X[1] * g[1] + e.X[2] * g[2];
}

// ---- pass 1: block = 256 points of one cloud; per-wave shuffle reduce, then fp64 atomics per cloud ----------------
template <int K>
__global__ void __launch_bounds__(256) loss_reduce_kernel(LossArgs a)
{
    const int b = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
    float acc[3 * K + 2];
#pragma unroll
    for (int i = 0; i < 3 * K + 2; ++i) acc[i] = 0.f;
    if (n < a.N) {
        const size_t m = (size_t)b * a.N + n;
        const int lab = (int)a.igt[m], bb = (int)a.bbgt[m];
        PointEval e;
        eval_point<K>(a, m, b, bb, e);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int mk = (int)a.match[b * K + k];
            float wr = 0.f;
#pragma unroll
            for (int j = 0; j < K; ++j) wr = (j == mk) ? e.W[j] : wr;
            acc[k] = (lab == k) ? wr : 0.f;
            acc[K + k] = wr;
            acc[2 * K + k] = (lab == k) ? 1.f : 0.f;
        }
        acc[3 * K] = 1.f - fabsf(e.cosv);
        acc[3 * K + 1] = e.bbsum;
    }
    __shared__ float red[4][3 * K + 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 3 * K + 2; ++i) {
        const float v = p2c_wave_sum_f32(acc[i]);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 3 * K + 2) {
        const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        atomicAdd(&a.acc[(size_t)b * (3 * K + 2) + threadIdx.x], (double)v);
    }
}

// ---- finalize: one block, thread b handles cloud b ----------------------------------------------------------------
template <int K>
__global__ void __launch_bounds__(256) loss_finalize_kernel(LossArgs a)
{
    __shared__ double sm[256], sn[256], sbb[256];
    double miou = 0.0, nrm = 0.0, bbl = 0.0;
    for (int b = threadIdx.x; b < a.B; b += 256) {
        const double *c = a.acc + (size_t)b * (3 * K + 2);
        int nv = 0;
        for (int k = 0; k < K; ++k) nv += a.mask[b * K + k] ? 1 : 0;      // mask_gt == the matching mask (k < max(I_gt)+1)
        double s = 0.0;
        for (int k = 0; k < K; ++k) {
            const double dot = c[k], den = c[2 * K + k] + c[K + k] - dot + 1e-10;      // losses.py:99-101
            const bool valid = a.mask[b * K + k] != 0;
            if (valid) s += 1.0 - dot / den;
            const double g = (valid && nv > 0) ? (double)a.w_seg / ((double)a.B * nv) : 0.0;
            a.coef[(size_t)b * 2 * K + k] = (float)(-g * (den + dot) / (den * den));   // d/d dot  (through dot and den)
            a.coef[(size_t)b * 2 * K + K + k] = (float)(g * dot / (den * den));        // d/d sumW
        }
        miou += nv > 0 ? s / nv : 0.0;
        nrm += c[3 * K] / a.N;
        bbl += c[3 * K + 1] / a.N;
    }
    sm[threadIdx.x] = miou; sn[threadIdx.x] = nrm; sbb[threadIdx.x] = bbl;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = 0, n = 0, q = 0;
        for (int i = 0; i < 256; ++i) { m += sm[i]; n += sn[i]; q += sbb[i]; }
        m /= a.B; n /= a.B; q /= a.B;
        a.out[1] = (float)n; a.out[2] = (float)m; a.out[3] = (float)q;
        a.out[0] = (float)(a.w_seg * m + a.w_normal * n + a.w_bb * q);
    }
}

// ---- pass 2: gradient w.r.t. the head output ------------------------------------------------------------------------
template <int K>
__global__ void __launch_bounds__(256) loss_grad_kernel(LossArgs a)
{
    const int b = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
    if (n >= a.N) return;
    const size_t m = (size_t)b * a.N + n;
    const int lab = (int)a.igt[m], bb = (int)a.bbgt[m];
    PointEval e;
    eval_point<K>(a, m, b, bb, e);
    const float gsc = a.gscale ? a.gscale[0] : 1.f;
    const float cpt = gsc / ((float)a.B * (float)a.N);      // the normal and base/barrel terms are linear in cpt, the mIoU term takes gsc itself
    // d total / d W[j]  (j = original column)
    float dW[K];
#pragma unroll
    for (int j = 0; j < K; ++j) dW[j] = 0.f;
    // base/barrel: dq_k = ce_k * c ; du_k = q_k (dq_k - sum_j q_j dq_j) ; dW[match_k] += mask_k du_k
    const float cb = a.w_bb * cpt;
    float qd = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) qd += e.q[k] * e.ce[k];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int mk = (int)a.match[b * K + k];
        const float du = a.mask[b * K + k] ? cb * e.q[k] * (e.ce[k] - qd) : 0.f;
        const float dmi = gsc * (a.coef[(size_t)b * 2 * K + K + k] + ((lab == k) ? a.coef[(size_t)b * 2 * K + k] : 0.f));   // mIoU
#pragma unroll
        for (int j = 0; j < K; ++j) dW[j] += (j == mk) ? (du + dmi) : 0.f;
    }
    // softmax over the 2K logits: dp[2k] = dp[2k+1] = dW[k];  dl_j = p_j (dp_j - sum_i p_i dp_i)  + direct ce terms
    float pd = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) pd += e.W[k] * dW[k];
    float *o = a.dheads + m * a.ld;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float dce = cb * e.q[k];                       // d total / d ce_k
        o[a.woff + 2 * k] = e.p[2 * k] * (dW[k] - pd) + dce * (e.sb[k] - (bb == 0 ? 1.f : 0.f));
        o[a.woff + 2 * k + 1] = e.p[2 * k + 1] * (dW[k] - pd) + dce * ((1.f - e.sb[k]) - (bb == 0 ? 0.f : 1.f));
    }
    // normal: L = c (1 - |X.g|) ; dX = -c sign(X.g) g ; dx = (dX - (dX.X) X) / |x|
    const float *g = a.ngt + m * 3;
    const float cn = -a.w_normal * cpt * (e.cosv > 0.f ? 1.f : (e.cosv < 0.f ? -1.f : 0.f));
    const float d0 = cn * g[0], d1 = cn * g[1], d2 = cn * g[2];
    const float dd = d0 * e.X[0] + d1 * e.X[1] + d2 * e.X[2];
    o[a.xoff + 0] = (d0 - dd * e.X[0]) * e.inv_norm;
    o[a.xoff + 1] = (d1 - dd * e.X[1]) * e.inv_norm;
    o[a.xoff + 2] = (d2 - dd * e.X[2]) * e.inv_norm;
    for (int c = 0; c < a.ld; ++c)
        if ((c < a.xoff || c >= a.xoff + 3) && (c < a.woff || c >= a.woff + 2 * K)) o[c] = 0.f;   // padding columns
}

extern "C" size_t p2c_seg_losses_ws_bytes(int B, int K) { return (size_t)B * (3 * K + 2) * sizeof(double) + (size_t)B * 2 * K * sizeof(float); }

// losses.py:317-351 (compute_all_losses, collapse=True) + the base/barrel block of train…:283-307, forward AND gradient.
// ws: zeroed p2c_seg_losses_ws_bytes(B,K).  out[4] = {total, normal, miou, bb}.  K = 1 ... 8 (the reference's default is 8; its --K is free).
extern "C" int p2c_seg_losses_f32(const float *heads, int ld, int xoff, int woff, const float *normals_gt, const int64_t *I_gt,
                                  const int64_t *bb_gt, const int64_t *match, const uint8_t *mask, int B, int N, int K, float w_seg,
                                  float w_normal, float w_bb, float *out, float *dheads, void *ws, void *stream)
{
    // dheads == NULL: the two forward launches only; the gradient then comes from p2c_seg_losses_grad_f32 with the same ws
    if (!heads || !normals_gt || !I_gt || !bb_gt || !match || !mask || !out || !ws || B <= 0 || N <= 0) return P2C_EINVAL;
    if (K < 1 || K > 8) return P2C_EINVAL;
    LossArgs a{heads, ld, xoff, woff, normals_gt, I_gt, bb_gt, match, mask, B, N, K, w_seg, w_normal, w_bb, (double *)ws, out,
               (float *)((char *)ws + (size_t)B * (3 * K + 2) * sizeof(double)), dheads, nullptr};
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(p2c_cdiv(N, 256), B);
#define P2C_LK(K_)                                                                       \
    do {                                                                                 \
        hipLaunchKernelGGL(loss_reduce_kernel<K_>, grid, dim3(256), 0, s, a);            \
        hipLaunchKernelGGL(loss_finalize_kernel<K_>, dim3(1), dim3(256), 0, s, a);       \
        if (dheads) hipLaunchKernelGGL(loss_grad_kernel<K_>, grid, dim3(256), 0, s, a);  \
    } while (0)
    switch (K) {
    case 1: P2C_LK(1); break; case 2: P2C_LK(2); break; case 3: P2C_LK(3); break; case 4: P2C_LK(4); break;
    case 5: P2C_LK(5); break; case 6: P2C_LK(6); break; case 7: P2C_LK(7); break; default: P2C_LK(8); break;
    }
#undef P2C_LK
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// The gradient pass on its own: d total / d heads times the upstream gradient *gscale (a device scalar, NULL = 1), from the sums and
// coefficients p2c_seg_losses_f32 left in ws.  Lets the caller's autograd node produce the scaled gradient in its backward in ONE launch
// instead of keeping an unscaled copy from the forward and multiplying 21 MB by a scalar.
extern "C" int p2c_seg_losses_grad_f32(const float *heads, int ld, int xoff, int woff, const float *normals_gt, const int64_t *I_gt,
                                       const int64_t *bb_gt, const int64_t *match, const uint8_t *mask, int B, int N, int K, float w_seg,
                                       float w_normal, float w_bb, const float *gscale, float *dheads, void *ws, void *stream)
{
    if (!heads || !normals_gt || !I_gt || !bb_gt || !match || !mask || !dheads || !ws || B <= 0 || N <= 0) return P2C_EINVAL;
    if (K < 1 || K > 8) return P2C_EINVAL;
    LossArgs a{heads, ld, xoff, woff, normals_gt, I_gt, bb_gt, match, mask, B, N, K, w_seg, w_normal, w_bb, (double *)ws, nullptr,
               (float *)((char *)ws + (size_t)B * (3 * K + 2) * sizeof(double)), dheads, gscale};
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(p2c_cdiv(N, 256), B);
#define P2C_LG(K_) hipLaunchKernelGGL(loss_grad_kernel<K_>, grid, dim3(256), 0, s, a)
    switch (K) {
    case 1: P2C_LG(1); break; case 2: P2C_LG(2); break; case 3: P2C_LG(3); break; case 4: P2C_LG(4); break;
    case 5: P2C_LG(5); break; case 6: P2C_LG(6); break; case 7: P2C_LG(7); break; default: P2C_LG(8); break;
    }
#undef P2C_LG
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// ------------------------------------------------------------------------------------------------
// Head post-processing for the fitting losses (train_Point2Cyl_without_sketch.py:247-265, :319-325, :342-344): from the raw head
// output of a point - 3 normal components and 2K segmentation logits - the unit normal X = x / max(|x|, 1e-12) (F.normalize), the
// softmax over the 2K logits, and the barrel / base probabilities of the segments in MATCHED order
// (torch.gather(W_barrel, 2, matching_indices) etc.): what estimate_extrusion_axis / estimate_extrusion_centers consume.  One pass
// forward, one pass backward (normalize and softmax Jacobians, the gather's scatter - matching_indices repeats column 0 for the
// unmatched slots, so contributions are summed) instead of ~25 torch launches over [B*N, 2K] tensors.
// ------------------------------------------------------------------------------------------------
#define HP_MAXK 16
__global__ void __launch_bounds__(256) head_post_kernel(const float *__restrict__ heads, int ld, int xoff, int woff, const int64_t *__restrict__ match,
                                                        int B, int N, int K, float *__restrict__ X, float *__restrict__ Wb, float *__restrict__ Wc)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= (size_t)B * N) return;
    const int b = (int)(p / N);
    const float *h = heads + p * ld;
    const float x0 = h[xoff], x1 = h[xoff + 1], x2 = h[xoff + 2];
    const float inv = 1.f / fmaxf(sqrtf(x0 * x0 + x1 * x1 + x2 * x2), 1e-12f);
    X[p * 3 + 0] = x0 * inv; X[p * 3 + 1] = x1 * inv; X[p * 3 + 2] = x2 * inv;
    float l[2 * HP_MAXK], mx = -INFINITY, sum = 0.f;
#pragma unroll
    for (int j = 0; j < 2 * HP_MAXK; ++j) if (j < 2 * K) { l[j] = h[woff + j]; mx = fmaxf(mx, l[j]); }
#pragma unroll
    for (int j = 0; j < 2 * HP_MAXK; ++j) if (j < 2 * K) { l[j] = expf(l[j] - mx); sum += l[j]; }
    const float is = 1.f / sum;
    const int64_t *m = match + (size_t)b * K;
    for (int k = 0; k < K; ++k) {
        const int c = (int)m[k];
        float pb = 0.f, pc = 0.f;
#pragma unroll
        for (int j = 0; j < HP_MAXK; ++j) if (j == c) { pb = l[2 * j]; pc = l[2 * j + 1]; }
        Wb[p * K + k] = pb * is;
        Wc[p * K + k] = pc * is;
    }
}

__global__ void __launch_bounds__(256) head_post_bwd_kernel(const float *__restrict__ heads, int ld, int xoff, int woff, const int64_t *__restrict__ match,
                                                            int B, int N, int K, const float *__restrict__ dX, const float *__restrict__ dWb,
                                                            const float *__restrict__ dWc, float *__restrict__ dheads, int ldd)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= (size_t)B * N) return;
    const int b = (int)(p / N);
    const float *h = heads + p * ld;
    float *dh = dheads + p * ldd;
    for (int j = 0; j < ldd; ++j) dh[j] = 0.f;
    // normalize: X = x / n  ->  dx = (dX - X (X . dX)) / n   (n clamped at 1e-12: there the map is x / 1e-12, dx = dX / 1e-12)
    const float x0 = h[xoff], x1 = h[xoff + 1], x2 = h[xoff + 2];
    const float nrm = sqrtf(x0 * x0 + x1 * x1 + x2 * x2);
    const float g0 = dX ? dX[p * 3 + 0] : 0.f, g1 = dX ? dX[p * 3 + 1] : 0.f, g2 = dX ? dX[p * 3 + 2] : 0.f;
    if (nrm > 1e-12f) {
        const float inv = 1.f / nrm, u0 = x0 * inv, u1 = x1 * inv, u2 = x2 * inv, dot = u0 * g0 + u1 * g1 + u2 * g2;
        dh[xoff] = (g0 - u0 * dot) * inv; dh[xoff + 1] = (g1 - u1 * dot) * inv; dh[xoff + 2] = (g2 - u2 * dot) * inv;
    } else {
        dh[xoff] = g0 * 1e12f; dh[xoff + 1] = g1 * 1e12f; dh[xoff + 2] = g2 * 1e12f;
    }
    // softmax + matched gather
    float l[2 * HP_MAXK], dp[2 * HP_MAXK], mx = -INFINITY, sum = 0.f;
#pragma unroll
    for (int j = 0; j < 2 * HP_MAXK; ++j) { dp[j] = 0.f; if (j < 2 * K) { l[j] = h[woff + j]; mx = fmaxf(mx, l[j]); } }
#pragma unroll
    for (int j = 0; j < 2 * HP_MAXK; ++j) if (j < 2 * K) { l[j] = expf(l[j] - mx); sum += l[j]; }
    const float is = 1.f / sum;
    const int64_t *m = match + (size_t)b * K;
    for (int k = 0; k < K; ++k) {
        const int c = (int)m[k];
        const float gb = dWb ? dWb[p * K + k] : 0.f, gc = dWc ? dWc[p * K + k] : 0.f;
#pragma unroll
        for (int j = 0; j < HP_MAXK; ++j) if (j == c) { dp[2 * j] += gb; dp[2 * j + 1] += gc; }
    }
    float pd = 0.f;
#pragma unroll
    for (int j = 0; j < 2 * HP_MAXK; ++j) if (j < 2 * K) { l[j] *= is; pd += l[j] * dp[j]; }
#pragma unroll
    for (int j = 0; j < 2 * HP_MAXK; ++j) if (j < 2 * K) dh[woff + j] = l[j] * (dp[j] - pd);
}

extern "C" int p2c_head_post_f32(const float *heads, int ld, int xoff, int woff, const int64_t *match, int B, int N, int K, float *X, float *Wb,
                                 float *Wc, void *stream)
{
    if (!heads || !match || !X || !Wb || !Wc || B <= 0 || N <= 0 || K <= 0 || K > HP_MAXK || xoff < 0 || woff < 0 || ld < xoff + 3 || ld < woff + 2 * K)
        return P2C_EINVAL;
    const size_t n = (size_t)B * N;
    hipLaunchKernelGGL(head_post_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, heads, ld, xoff, woff, match, B, N, K, X,
                       Wb, Wc);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

extern "C" int p2c_head_post_bwd_f32(const float *heads, int ld, int xoff, int woff, const int64_t *match, int B, int N, int K, const float *dX,
                                     const float *dWb, const float *dWc, float *dheads, int ldd, void *stream)
{
    if (!heads || !match || !dheads || B <= 0 || N <= 0 || K <= 0 || K > HP_MAXK || xoff < 0 || woff < 0 || ld < xoff + 3 || ld < woff + 2 * K ||
        ldd < xoff + 3 || ldd < woff + 2 * K)
        return P2C_EINVAL;
    const size_t n = (size_t)B * N;
    hipLaunchKernelGGL(head_post_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, heads, ld, xoff, woff, match, B, N, K,
                       dX, dWb, dWc, dheads, ldd);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// ------------------------------------------------------------------------------------------------
// compute_all_losses on its OWN inputs (losses.py:317-351 with collapse=True): the reference's trainer hands it the softmaxed membership
// W [B,N,K] and the unit normals X [B,N,3] it has already formed in torch (train…:246-271, :280), so the drop-in of that function cannot
// start from the logits like p2c_seg_losses_f32.  Same three steps on (W, X): per-cloud sums (dot, sumW, cnt, normal term), a small
// finalisation (the two means, the per-(b,k) mIoU gradient coefficients), one pass that writes the UNWEIGHTED gradients
// d miou / d W and d normal / d X (the caller scales them by its multipliers and the upstream gradient).
// ------------------------------------------------------------------------------------------------
struct AllLossArgs {
    const float *W, *X, *ngt;                            // [B,N,K], [B,N,3], [B,N,3]
    const int64_t *igt, *match; const uint8_t *mask;     // [B,N]; [B,K]; [B,K]
    int B, N;
    double *acc;                                         // [B][3K+1] zeroed: dot | sumW | cnt | normal
    float *out;                                          // [2]: normal mean, miou mean
    float *coef;                                         // [B][2K]
    float *dW, *dX;
};

template <int K>
__global__ void __launch_bounds__(256) all_losses_reduce_kernel(AllLossArgs a)
{
    const int b = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
    float acc[3 * K + 1];
#pragma unroll
    for (int i = 0; i < 3 * K + 1; ++i) acc[i] = 0.f;
    if (n < a.N) {
        const size_t m = (size_t)b * a.N + n;
        const int lab = (int)a.igt[m];
        float w[K];
#pragma unroll
        for (int j = 0; j < K; ++j) w[j] = a.W[m * K + j];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int mk = (int)a.match[b * K + k];
            float wr = 0.f;
#pragma unroll
            for (int j = 0; j < K; ++j) wr = (j == mk) ? w[j] : wr;
            acc[k] = (lab == k) ? wr : 0.f;
            acc[K + k] = wr;
            acc[2 * K + k] = (lab == k) ? 1.f : 0.f;
        }
        const float *x = a.X + m * 3, *g = a.ngt + m * 3;
        acc[3 * K] = 1.f - fabsf(x[0] * g[0] + x[1] * g[1] + x[2] * g[2]);
    }
    __shared__ float red[4][3 * K + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 3 * K + 1; ++i) {
        const float v = p2c_wave_sum_f32(acc[i]);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 3 * K + 1) {
        const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        atomicAdd(&a.acc[(size_t)b * (3 * K + 1) + threadIdx.x], (double)v);
    }
}

template <int K>
__global__ void __launch_bounds__(256) all_losses_finalize_kernel(AllLossArgs a)
{
    __shared__ double sm[256], sn[256];
    double miou = 0.0, nrm = 0.0;
    for (int b = threadIdx.x; b < a.B; b += 256) {
        const double *c = a.acc + (size_t)b * (3 * K + 1);
        int nv = 0;
        for (int k = 0; k < K; ++k) nv += a.mask[b * K + k] ? 1 : 0;
        double s = 0.0;
        for (int k = 0; k < K; ++k) {
            const double dot = c[k], den = c[2 * K + k] + c[K + k] - dot + 1e-10;      // losses.py:99-101
            const bool valid = a.mask[b * K + k] != 0;
            if (valid) s += 1.0 - dot / den;
            const double g = (valid && nv > 0) ? 1.0 / ((double)a.B * nv) : 0.0;
            a.coef[(size_t)b * 2 * K + k] = (float)(-g * (den + dot) / (den * den));
            a.coef[(size_t)b * 2 * K + K + k] = (float)(g * dot / (den * den));
        }
        miou += nv > 0 ? s / nv : 0.0;
        nrm += c[3 * K] / a.N;
    }
    sm[threadIdx.x] = miou; sn[threadIdx.x] = nrm;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = 0, n = 0;
        for (int i = 0; i < 256; ++i) { m += sm[i]; n += sn[i]; }
        a.out[0] = (float)(n / a.B); a.out[1] = (float)(m / a.B);
    }
}

template <int K>
__global__ void __launch_bounds__(256) all_losses_grad_kernel(AllLossArgs a)
{
    const int b = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
    if (n >= a.N) return;
    const size_t m = (size_t)b * a.N + n;
    const int lab = (int)a.igt[m];
    float dW[K];
#pragma unroll
    for (int j = 0; j < K; ++j) dW[j] = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int mk = (int)a.match[b * K + k];
        const float d = a.coef[(size_t)b * 2 * K + K + k] + ((lab == k) ? a.coef[(size_t)b * 2 * K + k] : 0.f);
#pragma unroll
        for (int j = 0; j < K; ++j) dW[j] += (j == mk) ? d : 0.f;           // unmatched slots repeat a column: contributions add (gather's backward)
    }
#pragma unroll
    for (int j = 0; j < K; ++j) a.dW[m * K + j] = dW[j];
    const float *x = a.X + m * 3, *g = a.ngt + m * 3;
    const float c = x[0] * g[0] + x[1] * g[1] + x[2] * g[2];
    const float cn = -(c > 0.f ? 1.f : (c < 0.f ? -1.f : 0.f)) / ((float)a.B * (float)a.N);
    a.dX[m * 3 + 0] = cn * g[0]; a.dX[m * 3 + 1] = cn * g[1]; a.dX[m * 3 + 2] = cn * g[2];
}

extern "C" size_t p2c_all_losses_ws_bytes(int B, int K) { return (size_t)B * (3 * K + 1) * sizeof(double) + (size_t)B * 2 * K * sizeof(float); }

extern "C" int p2c_all_losses_f32(const float *W, const float *X, const float *normals_gt, const int64_t *I_gt, const int64_t *match,
                                  const uint8_t *mask, int B, int N, int K, float *out2, float *dW, float *dX, void *ws, void *stream)
{
    if (!W || !X || !normals_gt || !I_gt || !match || !mask || !out2 || !dW || !dX || !ws || B <= 0 || N <= 0) return P2C_EINVAL;
    if (K < 1 || K > 8) return P2C_EINVAL;
    AllLossArgs a{W, X, normals_gt, I_gt, match, mask, B, N, (double *)ws, out2,
                  (float *)((char *)ws + (size_t)B * (3 * K + 1) * sizeof(double)), dW, dX};
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(p2c_cdiv(N, 256), B);
#define P2C_ALK(K_)                                                                          \
    do {                                                                                     \
        hipLaunchKernelGGL(all_losses_reduce_kernel<K_>, grid, dim3(256), 0, s, a);          \
        hipLaunchKernelGGL(all_losses_finalize_kernel<K_>, dim3(1), dim3(256), 0, s, a);     \
        hipLaunchKernelGGL(all_losses_grad_kernel<K_>, grid, dim3(256), 0, s, a);            \
    } while (0)
    switch (K) {
    case 1: P2C_ALK(1); break; case 2: P2C_ALK(2); break; case 3: P2C_ALK(3); break; case 4: P2C_ALK(4); break;
    case 5: P2C_ALK(5); break; case 6: P2C_ALK(6); break; case 7: P2C_ALK(7); break; default: P2C_ALK(8); break;
    }
#undef P2C_ALK
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// =============================================================================================
// The two fitting terms of the full loss set on their (B, K) operands - extrusion-axis loss  mean_b masked-mean_k (1 - |E . A_gt|)
// (losses.py:127-143 with angle_diff = False, :83-88; train...:326-332) and centre loss  mean_b masked-mean_k |c - c_gt|^2 (:342-353) -
// forward AND gradient in one launch of one workgroup: as torch expressions they are ~36 launches on 256-element tensors forward and as
// many backward, a dependent chain of 4-5 us nodes in the replayed step.  mask [B,K] = k < number of ground-truth instances of cloud b
// (what the matching returns); a cloud without instances contributes 0.  dE / dC = the gradients of the WEIGHTED terms w.r.t. the
// fitted axes / centres (the caller scales them by the upstream gradient of the total).  E or C may be NULL (term switched off: 0).
// =============================================================================================
__global__ void __launch_bounds__(256) fit_terms_kernel(const float *__restrict__ E, const float *__restrict__ A, const float *__restrict__ C,
                                                        const float *__restrict__ Cg, const uint8_t *__restrict__ mask, int B, int K,
                                                        float w_ext, float w_cen, float *__restrict__ out2, float *__restrict__ dE,
                                                        float *__restrict__ dC)
{
    __shared__ float s_ext[256], s_cen[256], s_n[256];
    __shared__ double acc[2];
    const int tid = threadIdx.x;
    if (tid < 2) acc[tid] = 0.0;
    __syncthreads();
    const int BK = B * K;
    const int P = (256 / K) * K;                            // entries per pass: whole clouds only, so a cloud's K entries lie in one pass
    for (int base = 0; base < BK; base += P) {
        const int e = base + tid;
        const bool ok = tid < P && e < BK;
        const bool m = ok && mask[e] != 0;
        float ext = 0.f, cen = 0.f, sgn = 0.f;
        float dx = 0.f, dy = 0.f, dz = 0.f;
        if (ok && E) {
            const float d = E[e * 3] * A[e * 3] + E[e * 3 + 1] * A[e * 3 + 1] + E[e * 3 + 2] * A[e * 3 + 2];
            ext = 1.f - fabsf(d);
            sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        }
        if (ok && C) {
            dx = C[e * 3] - Cg[e * 3]; dy = C[e * 3 + 1] - Cg[e * 3 + 1]; dz = C[e * 3 + 2] - Cg[e * 3 + 2];
            cen = dx * dx + dy * dy + dz * dz;
        }
        s_ext[tid] = m ? ext : 0.f;
        s_cen[tid] = m ? cen : 0.f;
        s_n[tid] = m ? 1.f : 0.f;
        __syncthreads();
        const int k0 = tid - (tid % K);
        float n = 0.f, se = 0.f, sc = 0.f;
        for (int k = 0; k < K; ++k) { n += s_n[k0 + k]; se += s_ext[k0 + k]; sc += s_cen[k0 + k]; }
        const float inv = n > 0.f ? 1.f / n : 0.f;          // reduce_mean_masked_instance: 0 for a cloud without instances
        if (ok && (tid % K) == 0) {
            atomicAdd(&acc[0], (double)(se * inv));
            atomicAdd(&acc[1], (double)(sc * inv));
        }
        const float g = m ? inv / (float)B : 0.f;
        if (ok && E && dE) {
            const float c = -sgn * g * w_ext;
            dE[e * 3] = c * A[e * 3]; dE[e * 3 + 1] = c * A[e * 3 + 1]; dE[e * 3 + 2] = c * A[e * 3 + 2];
        }
        if (ok && C && dC) {
            const float c = 2.f * g * w_cen;
            dC[e * 3] = c * dx; dC[e * 3 + 1] = c * dy; dC[e * 3 + 2] = c * dz;
        }
        __syncthreads();
    }
    if (tid == 0) {
        out2[0] = E ? (float)(acc[0] / (double)B) * w_ext : 0.f;
        out2[1] = C ? (float)(acc[1] / (double)B) * w_cen : 0.f;
    }
}

extern "C" int p2c_fit_terms_f32(const float *E_AX, const float *gt_axes, const float *centers, const float *gt_centers, const uint8_t *mask,
                                 int B, int K, float w_ext, float w_center, float *out2, float *dE, float *dC, void *stream)
{
    if (!mask || !out2 || B <= 0 || K <= 0 || K > 256) return P2C_EINVAL;
    if ((E_AX && !gt_axes) || (centers && !gt_centers)) return P2C_EINVAL;
    hipLaunchKernelGGL(fit_terms_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, E_AX, gt_axes, centers, gt_centers, mask, B, K, w_ext,
                       w_center, out2, dE, dC);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}
