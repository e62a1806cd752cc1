// metrics.hip -- the evaluation metrics of one batch (eval.py:270-446, default operand choice: no --use_gt_*) in TWO launches.
//
// The reference (and the torch-op mirror in point2cyl_amd/eval.py:eval_metrics) evaluates the chain as ~90 separate kernels:
// F.normalize, softmax over the 2K logits, barrel / base split, hard one-hot encoding with the null-column rule
// (losses.py:55-68), Hungarian matching (losses.py:22-52), segmentation IoU (:106-109), normal angle error (:146-159), base/barrel
// accuracy (eval.py:339-342), the axis fit on the matched SOFT barrel / base memberships (data_utils.py:99-177, eval.py:386-397), its
// angle error (eval.py:398-405) and the hard per-segment centroids (eval.py:409-446).  Every one of them is a reduction over the
// points of a cloud of quantities that are functions of ONE point's head outputs - and the matching only permutes COLUMNS.  So:
//
//   eval_sums_kernel    (B x split workgroups): one pass over the heads / points / labels; per predicted column k the 6+6 scatter-matrix
//                       sums of p_barrel^2 x x^T and p_base^2 x x^T, the soft column mass, the hard-label count and coordinate sums, the
//                       (gt label x predicted label) confusion counts; per cloud the angle sum, the base/barrel hits, the largest label.
//   eval_finish_kernel  (B workgroups of one wave): partial sums added in split order (deterministic), null columns, IoU costs (exact:
//                       counts), the wave solver of assign.hip, then per MATCHED segment the eigen-solve (fp64 Jacobi, eigh3.h), the
//                       centroid and the two masked means.
//
// Arithmetic: the per-point part follows torch's kernels operation for operation (softmax = exp(x - max) / sum with the 16-lane
// butterfly's summation order; W = p_barrel + p_base; first-index argmax), so the hard labels - and with them the matching, mIoU and
// base/barrel accuracy, which are ratios of COUNTS - equal the torch chain's.  Floating-point sums over the points (angle mean, scatter
// matrices, centroids) are accumulated in fp32 over <= 32 points per thread and in fp64 from there: they agree with the torch chain to
// rounding (tests/test_gpu_flows.py), not bit for bit - no two summation orders do.
#include "common.h"
#include "eigh3.h"
#include "lsa.h"

#define EV_THREADS 256
#define EV_SPLIT 8
#define EV_NACC 28            // per (slice, column) accumulators: Sb[6] Sc[6] colsum cnt cen[3] conf[K+1 <= 9] nb nc
#define EV_WS(KK) (EV_NACC * (KK) + 3)      // doubles per (cloud, part): the table + angle sum, base/barrel hits, max label

template <int KK>
__global__ void __launch_bounds__(EV_THREADS) eval_sums_kernel(const float *__restrict__ heads, int ld, int xoff, int woff,
                                                               const float *__restrict__ pcs, const float *__restrict__ gtn,
                                                               const int64_t *__restrict__ inst, const float *__restrict__ bbf, int N, int split,
                                                               float pi_f, double *__restrict__ ws)
{
    constexpr int G = EV_THREADS / KK;                 // point slices of phase B
    constexpr int PER = EV_THREADS / G;                // points of a chunk each slice owns (= KK)
    static_assert(KK + 1 + 19 <= EV_NACC, "");
    // one LDS block: the chunk's staged points (phase A -> B), re-used for the final reduction rows
    constexpr int STAGE = EV_THREADS * (3 * KK + 9 + 3), REDW = EV_NACC * (EV_THREADS + 1);
    __shared__ float smem[STAGE > REDW ? STAGE : REDW];
    float (*s_b2)[KK] = (float (*)[KK])smem;
    float (*s_c2)[KK] = (float (*)[KK])(smem + EV_THREADS * KK);
    float (*s_w)[KK] = (float (*)[KK])(smem + 2 * EV_THREADS * KK);
    float (*s_pp)[6] = (float (*)[6])(smem + 3 * EV_THREADS * KK);
    float (*s_pc)[3] = (float (*)[3])(smem + 3 * EV_THREADS * KK + 6 * EV_THREADS);
    int *s_pred = (int *)(smem + 3 * EV_THREADS * KK + 9 * EV_THREADS), *s_gt = s_pred + EV_THREADS, *s_bb = s_gt + EV_THREADS;
    float (*red)[EV_THREADS + 1] = (float (*)[EV_THREADS + 1])smem;
    __shared__ double s_ang[EV_THREADS / 64];
    __shared__ int s_hit[EV_THREADS / 64], s_max[EV_THREADS / 64];
    const int b = blockIdx.x / split, part = blockIdx.x % split, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (N + split - 1) / split, n0 = part * per, n1 = min(N, n0 + per);
    const int g = tid / KK, k = tid - g * KK;
    float acc[EV_NACC];
#pragma unroll
    for (int i = 0; i < EV_NACC; ++i) acc[i] = 0.f;
    float ang_sum = 0.f;
    int hits = 0, mx = -1;
    for (int c0 = n0; c0 < n1; c0 += EV_THREADS) {
        const int n = c0 + tid;
        // ---- phase A: this thread's point
        if (n < n1) {
            const size_t pn = (size_t)b * N + n;
            const float *row = heads + pn * ld;
            const float h0 = row[xoff], h1 = row[xoff + 1], h2 = row[xoff + 2];
            const float nrm = sqrtf(h0 * h0 + h1 * h1 + h2 * h2);                 // F.normalize(p=2, eps=1e-12): v / max(|v|, eps)  (eval.py:270)
            const float den = fmaxf(nrm, 1e-12f);
            const float x0 = h0 / den, x1 = h1 / den, x2 = h2 / den;
            const float g0 = gtn[pn * 3], g1 = gtn[pn * 3 + 1], g2 = gtn[pn * 3 + 2];
            float cs = fabsf(x0 * g0 + x1 * g1 + x2 * g2);                        // losses.py:146-159
            cs = fminf(fmaxf(cs, -1.0f + 1e-6f), 1.0f - 1e-6f);
            ang_sum += acosf(cs) * 180.0f / pi_f;
            float e[2 * KK], m = -INFINITY;
#pragma unroll
            for (int j = 0; j < 2 * KK; ++j) { e[j] = row[woff + j]; m = fmaxf(m, e[j]); }
#pragma unroll
            for (int j = 0; j < 2 * KK; ++j) e[j] = expf(e[j] - m);
            // the sum in the order of torch's softmax_warp_forward (one element per lane of a 2K-lane group, xor butterfly from the
            // widest offset down): s[j] += s[j ^ off]
            float s[2 * KK];
#pragma unroll
            for (int j = 0; j < 2 * KK; ++j) s[j] = e[j];
#pragma unroll
            for (int off = KK; off > 0; off >>= 1) {
                float t[2 * KK];
#pragma unroll
                for (int j = 0; j < 2 * KK; ++j) t[j] = s[j] + s[j ^ off];
#pragma unroll
                for (int j = 0; j < 2 * KK; ++j) s[j] = t[j];
            }
            const float sum = s[0];
            float bb0 = 0.f, bb1 = 0.f, best = -INFINITY;
            int pred = 0;
#pragma unroll
            for (int q = 0; q < KK; ++q) {
                const float pb = e[2 * q] / sum, pc_ = e[2 * q + 1] / sum;         // eval.py:278-286
                const float w = pb + pc_;                                          // :289
                // :297-300, in index order.  (The base/barrel decision bb1 > bb0 of a point whose two sums agree to an ulp depends on the
                // summation order of W_barrel.sum(-1): against the torch chain on the device ~2 of 8.4 M untrained-network points fall
                // the other way - 2e-7 of the accuracy; a four-accumulator order, tried, was further off.)
                bb0 += pb; bb1 += pc_;
                if (w > best) { best = w; pred = q; }                              // argmax: first maximum (losses.py:60)
                s_b2[tid][q] = pb * pb; s_c2[tid][q] = pc_ * pc_; s_w[tid][q] = w;
            }
            const int pbb = bb1 > bb0 ? 1 : 0;                                     // eval.py:340
            const int gbb = (int)(long long)bbf[pn];                               // :257 / .to(long)
            hits += pbb == gbb ? 1 : 0;                                            // :342
            const int gl = (int)inst[pn];
            mx = max(mx, gl);
            s_pred[tid] = pred; s_gt[tid] = gl; s_bb[tid] = gbb;
            s_pp[tid][0] = x0 * x0; s_pp[tid][1] = x0 * x1; s_pp[tid][2] = x0 * x2;
            s_pp[tid][3] = x1 * x1; s_pp[tid][4] = x1 * x2; s_pp[tid][5] = x2 * x2;
            s_pc[tid][0] = pcs[pn * 3]; s_pc[tid][1] = pcs[pn * 3 + 1]; s_pc[tid][2] = pcs[pn * 3 + 2];
        }
        __syncthreads();
        // ---- phase B: thread (slice g, column k) adds the chunk's points g, g + G, ...
        const int cnt = min(EV_THREADS, n1 - c0);
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int j = g + i * G;
            if (j < cnt) {
                const float b2 = s_b2[j][k], c2 = s_c2[j][k];
#pragma unroll
                for (int q = 0; q < 6; ++q) { const float pp = s_pp[j][q]; acc[q] += b2 * pp; acc[6 + q] += c2 * pp; }
                acc[12] += s_w[j][k];
                const int pred = s_pred[j], gl = s_gt[j];
                const float mine = pred == k ? 1.f : 0.f;
                acc[13] += mine;
                acc[14] += mine * s_pc[j][0]; acc[15] += mine * s_pc[j][1]; acc[16] += mine * s_pc[j][2];
                const int r = gl < 0 ? KK : gl;                                    // -1 = background: the eye(n_gt + 1) row the cost drops (losses.py:38-42)
#pragma unroll
                for (int q = 0; q <= KK; ++q) acc[17 + q] += (pred == k && r == q) ? 1.f : 0.f;
                acc[17 + KK + 1] += (gl == k && s_bb[j] == 0) ? 1.f : 0.f;         // data_utils.py:133-160 (normalize)
                acc[17 + KK + 2] += (gl == k && s_bb[j] == 1) ? 1.f : 0.f;
            }
        }
        __syncthreads();
    }
    // ---- workgroup totals: fp64 over the G slices of every (entry, column); the scalars over the waves
#pragma unroll
    for (int i = 0; i < EV_NACC; ++i) red[i][tid] = acc[i];
    const double a64 = p2c_wave_sum_f64((double)ang_sum);
    int h = hits;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o);
    mx = p2c_wave_max_i32(mx);
    if (lane == 0) { s_ang[wave] = a64; s_hit[wave] = h; s_max[wave] = mx; }
    __syncthreads();
    double *out = ws + ((size_t)b * split + part) * EV_WS(KK);
    for (int t = tid; t < EV_NACC * KK; t += EV_THREADS) {
        const int e = t / KK, q = t - e * KK;
        double rs = 0.0;
        for (int gg = 0; gg < G; ++gg) rs += (double)red[e][gg * KK + q];
        out[t] = rs;
    }
    if (tid == 0) {
        double a = 0.0;
        int hh = 0, mm = -1;
        for (int w = 0; w < EV_THREADS / 64; ++w) { a += s_ang[w]; hh += s_hit[w]; mm = max(mm, s_max[w]); }
        out[EV_NACC * KK] = a; out[EV_NACC * KK + 1] = (double)hh; out[EV_NACC * KK + 2] = (double)mm;
    }
}

// One wave per cloud.  out5 [5][B] fp64: mIoU, normal angle error (deg), base/barrel accuracy, extrusion angle error (deg), centroid
// difference - the rows of eval.py's report (:690-715).  Optional per-segment outputs for callers / tests: match [B,K] i64, mask [B,K] u8,
// axis64 [B,K,3] f64 (unit, canonical sign; zeros for unmatched rows), cen [B,K,3] f32, found [B,K] f32.
template <int KK>
__global__ void __launch_bounds__(64) eval_finish_kernel(const double *__restrict__ ws, int split, int N, int B, int normalize, float null_thr,
                                                         double pi_d, const float *__restrict__ gt_axes, const float *__restrict__ gt_cen,
                                                         double *__restrict__ out5, int64_t *__restrict__ match_out, uint8_t *__restrict__ mask_out,
                                                         double *__restrict__ axis64_out, float *__restrict__ cen_out, float *__restrict__ found_out)
{
    constexpr int NW = EV_WS(KK);
    __shared__ double tot[NW];
    __shared__ double cost[HM_MAXK * HM_MAXK];
    __shared__ int col4row[HM_MAXK + 1];
    __shared__ float s_iou[KK], s_ext[KK], s_cd[KK];
    const int b = blockIdx.x, lane = threadIdx.x;
    for (int t = lane; t < NW; t += 64) {
        const bool is_max = t == NW - 1;
        double s = is_max ? -1.0 : 0.0;
        for (int q = 0; q < split; ++q) {
            const double v = ws[((size_t)b * split + q) * NW + t];
            s = is_max ? fmax(s, v) : s + v;
        }
        tot[t] = s;
    }
    __syncthreads();
    auto T = [&](int e, int q) { return tot[e * KK + q]; };
    const int n_gt = (int)tot[NW - 1] + 1;                                         // losses.py:36
    const int nr = max(0, min(n_gt, KK));
    // hard encoding with the null-column rule (losses.py:62-66): a column whose SOFT mass is below thr * N is emptied
    auto is_null = [&](int q) { return (float)T(12, q) < null_thr; };
    if (lane < nr * KK) {
        const int r = lane / KK, q = lane - r * KK;
        const bool nul = is_null(q);
        const float dot = nul ? 0.f : (float)T(17 + r, q);
        const float col = nul ? 0.f : (float)T(13, q);
        float rc = 0.f;                                                            // points of gt label r = its confusion row, all columns
        for (int j = 0; j < KK; ++j) rc += (float)T(17 + r, j);
        const float den = (rc + col) - dot;                                        // losses.py:40
        cost[r * KK + q] = -(double)(dot / fmaxf(den, 1e-10f));                    // :41, :43
    }
    __syncthreads();
    if (nr > 0) p2c_lsa_min_wave(cost, nr, KK, col4row);
    __syncthreads();
    if (lane < KK) {
        const int gseg = lane;
        const bool on = gseg < nr;
        const int col = on ? col4row[gseg] : 0;
        if (match_out) match_out[(size_t)b * KK + gseg] = col;
        if (mask_out) mask_out[(size_t)b * KK + gseg] = on ? 1 : 0;
        float iou = 0.f, ext = 0.f, cd = 0.f;
        double ax[3] = {0.0, 0.0, 0.0};
        float cen[3] = {0.f, 0.f, 0.f};
        bool found = false;
        if (on) {
            const bool nul = is_null(col);
            // compute_segmentation_iou (losses.py:90-109): 1 - (1 - dot / (union + 1e-10))
            const float dot = nul ? 0.f : (float)T(17 + gseg, col);
            const float cpred = nul ? 0.f : (float)T(13, col);
            float rc = 0.f;
            for (int j = 0; j < KK; ++j) rc += (float)T(17 + gseg, j);
            const float uni = (rc + cpred) - dot;
            const float loss = 1.0f - dot / (uni + 1e-10f);
            iou = 1.0f - loss;
            // axis: smallest signed eigenvalue of B^T B / sb^2 - C^T C / sc^2 on the matched column's soft memberships (data_utils.py:162-171)
            double isb2 = 1.0, isc2 = 1.0;
            if (normalize) {
                const float sb = sqrtf((float)T(17 + KK + 1, gseg)) + 1.0f, sc = sqrtf((float)T(17 + KK + 2, gseg)) + 1.0f;
                isb2 = 1.0 / ((double)sb * (double)sb);
                isc2 = 1.0 / ((double)sc * (double)sc);
            }
            double a[6], lam[3], v[3][3];
            for (int e = 0; e < 6; ++e) a[e] = T(e, col) * isb2 - T(6 + e, col) * isc2;
            p2c_eigh3(a, lam, v);
            int big = 0;
            if (fabs(v[0][1]) > fabs(v[0][big])) big = 1;
            if (fabs(v[0][2]) > fabs(v[0][big])) big = 2;
            const double sgn = v[0][big] < 0 ? -1.0 : 1.0;
            ax[0] = sgn * v[0][0]; ax[1] = sgn * v[0][1]; ax[2] = sgn * v[0][2];
            const float *ga = gt_axes + ((size_t)b * KK + gseg) * 3;
            double cs = fabs(ax[0] * (double)ga[0] + ax[1] * (double)ga[1] + ax[2] * (double)ga[2]);     // eval.py:398 in fp64 (eval.py of this package)
            cs = fmin(fmax(cs, -1.0 + 1e-6), 1.0 - 1e-6);
            ext = (float)(acos(cs) * 180.0 / pi_d);
            // hard centroid of the points whose (non-null) label is the matched column (eval.py:409-436)
            const double c = nul ? 0.0 : T(13, col);
            found = c > 1.0;
            if (found) { cen[0] = (float)(T(14, col) / c); cen[1] = (float)(T(15, col) / c); cen[2] = (float)(T(16, col) / c); }
            const float *gc = gt_cen + ((size_t)b * KK + gseg) * 3;
            const float d0 = cen[0] - gc[0], d1 = cen[1] - gc[1], d2 = cen[2] - gc[2];
            cd = d0 * d0 + d1 * d1 + d2 * d2;                                       // eval.py:439
        }
        s_iou[gseg] = iou; s_ext[gseg] = ext; s_cd[gseg] = cd;
        if (axis64_out) { double *o = axis64_out + ((size_t)b * KK + gseg) * 3; o[0] = ax[0]; o[1] = ax[1]; o[2] = ax[2]; }
        if (cen_out) { float *o = cen_out + ((size_t)b * KK + gseg) * 3; o[0] = cen[0]; o[1] = cen[1]; o[2] = cen[2]; }
        if (found_out) found_out[(size_t)b * KK + gseg] = found ? 1.f : 0.f;
    }
    __syncthreads();
    if (lane == 0) {
        double si = 0.0, se = 0.0, sc = 0.0;
        for (int q = 0; q < nr; ++q) { si += (double)s_iou[q]; se += (double)s_ext[q]; sc += (double)s_cd[q]; }
        const double n = (double)nr;
        out5[0 * (size_t)B + b] = (double)(float)(si / n);                          // sum(mask * iou) / sum(mask): 0 / 0 = NaN without instances, as upstream
        out5[1 * (size_t)B + b] = (double)(float)(tot[NW - 3] / (double)N);
        out5[2 * (size_t)B + b] = (double)((float)tot[NW - 2] / (float)N);
        out5[3 * (size_t)B + b] = nr > 0 ? (double)(float)(se / n) : 0.0;          // reduce_mean_masked_instance (losses.py:83-88): 0 when empty
        out5[4 * (size_t)B + b] = nr > 0 ? (double)(float)(sc / n) : 0.0;
    }
}

extern "C" size_t p2c_eval_metrics_ws_bytes(int B, int K) { return (size_t)B * EV_SPLIT * (EV_NACC * (size_t)K + 3) * sizeof(double); }

extern "C" int p2c_eval_metrics_supported(int K) { return K == 8 || K == 4 || K == 2; }

template <int KK>
static int eval_metrics_launch(const float *heads, int ld, int xoff, int woff, const float *pcs, const float *gtn, const int64_t *inst, const float *bbf,
                               const float *gt_axes, const float *gt_cen, int normalize, float null_thr, double pi_d, int B, int N, double *out5,
                               int64_t *match_out, uint8_t *mask_out, double *axis64_out, float *cen_out, float *found_out, double *ws, hipStream_t st)
{
    const int split = N >= 1024 ? EV_SPLIT : 1;
    hipLaunchKernelGGL(eval_sums_kernel<KK>, dim3(B * split), dim3(EV_THREADS), 0, st, heads, ld, xoff, woff, pcs, gtn, inst, bbf, N, split, (float)pi_d, ws);
    hipLaunchKernelGGL(eval_finish_kernel<KK>, dim3(B), dim3(64), 0, st, (const double *)ws, split, N, B, normalize, null_thr, pi_d, gt_axes, gt_cen, out5,
                       match_out, mask_out, axis64_out, cen_out, found_out);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

extern "C" int p2c_eval_metrics_f32(const float *heads, int ld, int xoff, int woff, const float *pcs, const float *gt_normals, const int64_t *gt_inst,
                                    const float *gt_bb, const float *gt_axes, const float *gt_centers, int normalize, float null_thr, double pi,
                                    int B, int N, int K, double *out5, int64_t *match_out, uint8_t *mask_out, double *axis64_out, float *cen_out,
                                    float *found_out, void *ws, void *stream)
{
    if (!heads || !pcs || !gt_normals || !gt_inst || !gt_bb || !gt_axes || !gt_centers || !out5 || !ws || B <= 0 || N <= 0) return P2C_EINVAL;
    if (xoff < 0 || woff < 0 || ld < xoff + 3 || ld < woff + 2 * K || !(pi > 3.0 && pi < 3.3)) return P2C_EINVAL;
    hipStream_t st = (hipStream_t)stream;
#define P2C_EV(KK) return eval_metrics_launch<KK>(heads, ld, xoff, woff, pcs, gt_normals, gt_inst, gt_bb, gt_axes, gt_centers, normalize, null_thr, pi, B, N, \
                                                  out5, match_out, mask_out, axis64_out, cen_out, found_out, (double *)ws, st)
    if (K == 8) P2C_EV(8);
    if (K == 4) P2C_EV(4);
    if (K == 2) P2C_EV(2);
#undef P2C_EV
    return P2C_EINVAL;
}
