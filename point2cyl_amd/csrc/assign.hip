// assign.hip -- hungarian_matching (losses.py:22-52) without leaving the device.
//
// The reference builds, per sample, an IoU cost matrix (K' x K) with torch.mm, copies it to the host and
// calls scipy.optimize.linear_sum_assignment there (one blocking .cpu() per sample).  Here one workgroup
// per sample reduces the (K'+1) x K intersection / union sums deterministically and a single lane runs
// the same shortest-augmenting-path solver scipy uses (Crouse 2016, restated in oracle/p2c_oracle.c and
// checked against scipy on tie-heavy inputs), so the matching is available to the next kernel with no
// host round trip.
#include "common.h"

#include "lsa.h"

// One workgroup of 1024 threads per sample.  Thread (g, k) owns column k of W and the point slice
// g, g+G, ...; it adds W[n,k] into ITS OWN LDS row at slot label(n) (slot K' = background / -1 rows are
// skipped, slot `any` collects the plain column sum), so there are no atomics and the result is
// deterministic.  A second phase sums the G rows per (label, k) and lane 0 solves the assignment.
#define HM_THREADS 1024

// ld/woff/from_logits: W may be given as the raw head output (row stride ld, 2K logits starting at column woff); the
// per-point softmax and the barrel+base pair sums W[k] = p[2k] + p[2k+1] (train…:254-265) are then formed on the fly.
__global__ void __launch_bounds__(HM_THREADS) hungarian_kernel(const float *__restrict__ W, int ld, int woff, int from_logits,
                                                              const int64_t *__restrict__ I_gt, int N, int K,
                                                              int64_t *__restrict__ match_out, uint8_t *__restrict__ mask_out)
{
    extern __shared__ float sacc[];                 // [HM_THREADS][2K+1] private rows: K label sums | column sum | K label counts
    __shared__ int smax[HM_THREADS / 64];
    __shared__ double cost[HM_MAXK * HM_MAXK];
    __shared__ int col4row[HM_MAXK + 1];
    __shared__ float tot[(HM_MAXK + 1) * HM_MAXK], cnt[HM_MAXK];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t *lab = I_gt + (size_t)b * N;
    const float *w = W + (size_t)b * N * ld;
    const int G = HM_THREADS / K, g = tid / K, k = tid - g * K;
    const int RS = 2 * K + 1;
    float *mine = sacc + (size_t)tid * RS;
    for (int i = 0; i < RS; ++i) mine[i] = 0.f;
    int mx = -1;
    if (g < G) {
        for (int n = g; n < N; n += G) {
            const int label = (int)lab[n];
            mx = max(mx, label);
            float v;
            if (from_logits) {
                const float *logit = w + (size_t)n * ld + woff;
                float top = -INFINITY, sum = 0.f;
                for (int j = 0; j < 2 * K; ++j) top = fmaxf(top, logit[j]);
                for (int j = 0; j < 2 * K; ++j) sum += expf(logit[j] - top);
                v = (expf(logit[2 * k] - top) + expf(logit[2 * k + 1] - top)) / sum;
            } else {
                v = w[(size_t)n * ld + k];
            }
            mine[K] += v;
            if (label >= 0 && label < K) {       // labels >= K are a contract violation the host wrappers reject (ops.hungarian, train.ResidentDataset)
                mine[label] += v;
                if (k == 0) mine[K + 1 + label] += 1.f;
            }
        }
    }
    mx = p2c_wave_max_i32(mx);
    if (lane == 0) smax[wave] = mx;
    __syncthreads();
    int n_gt = -1;
    for (int i = 0; i < HM_THREADS / 64; ++i) n_gt = max(n_gt, smax[i]);
    n_gt += 1;                                       // losses.py:36
    // fixed-order sums over the G slices: (K+1)*K intersections / column sums, then K label counts
    if (tid < (K + 1) * K) {
        const int r = tid / K, q = tid - r * K;      // r = label slot (K = column sum), q = column
        float s = 0.f;
        for (int gg = 0; gg < G; ++gg) s += sacc[(size_t)(gg * K + q) * RS + r];
        tot[r * K + q] = s;
    } else if (tid < (K + 1) * K + K) {
        const int l = tid - (K + 1) * K;
        float s = 0.f;
        for (int gg = 0; gg < G; ++gg) s += sacc[(size_t)(gg * K) * RS + K + 1 + l];
        cnt[l] = s;
    }
    __syncthreads();
    const int nr = min(n_gt, K);
    if (tid < nr * K) {
        const int r = tid / K, q = tid % K;
        const float dot = tot[r * K + q], col = tot[K * K + q], rc = cnt[r];
        const float den = (rc + col) - dot;                       // :40
        const float iou = dot / fmaxf(den, 1e-10f);               // :41
        cost[r * K + q] = -(double)iou;                           // :43 maximise
    }
    __syncthreads();
    if (tid < 64) {
        if (nr > 0) p2c_lsa_min_wave(cost, nr, K, col4row);
        __builtin_amdgcn_wave_barrier();
        if (tid < K) {
            match_out[(size_t)b * K + tid] = tid < nr ? (int64_t)col4row[tid] : 0;   // rest stays 0 (:30)
            mask_out[(size_t)b * K + tid] = tid < nr ? 1 : 0;                        // :47
        }
    }
}

// From-logits variant for K == 8: one thread per point slice; the point's softmax is evaluated ONCE (16 exps) and its
// 8 pair sums W[k] are added to the (label, k) accumulators held in registers with compile-time-unrolled selects;
// a wave-shuffle + LDS reduction yields the (K+1) x K sums and the K label counts.
// A cloud is split over `split` workgroups (B workgroups alone leave 7/8 of the chip idle and the point loop is
// latency-bound); each writes its 80 partial sums and a second tiny kernel adds them in split order (deterministic)
// and solves the assignment.  (A "last workgroup finishes" single-kernel version was 3x SLOWER: the agent-scope
// fence it needs writes back the whole dirty L2 of the XCD, which at this point holds the head GEMM's output.)
#define HL_THREADS 256
__global__ void __launch_bounds__(HL_THREADS) hungarian_logits8_kernel(const float *__restrict__ heads, int ld, int woff,
                                                                      const int64_t *__restrict__ I_gt, int N, int split, float *__restrict__ ws)
{
    constexpr int K = 8, NA = (K + 1) * K + K, NP = NA + 1;          // 72 sums + 8 counts (+ max label)
    __shared__ float red[HL_THREADS / 64][NA];
    __shared__ int smax[HL_THREADS / 64];
    __shared__ float tot[NP];
    const int b = blockIdx.x / split, part = blockIdx.x % split, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t *lab = I_gt + (size_t)b * N;
    const float *h = heads + (size_t)b * N * ld + woff;
    const int per = (N + split - 1) / split, n0 = part * per, n1 = min(N, n0 + per);
    float acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = 0.f;
    int mx = -1;
    for (int n = n0 + tid; n < n1; n += HL_THREADS) {
        const int l = (int)lab[n];
        mx = max(mx, l);
        const float *lg = h + (size_t)n * ld;
        float e[2 * K], m = -INFINITY, sum = 0.f;
#pragma unroll
        for (int j = 0; j < 2 * K; ++j) { e[j] = lg[j]; m = fmaxf(m, e[j]); }
#pragma unroll
        for (int j = 0; j < 2 * K; ++j) { e[j] = expf(e[j] - m); sum += e[j]; }
        const float inv = 1.f / sum;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float w = (e[2 * k] + e[2 * k + 1]) * inv;
            acc[K * K + k] += w;                                   // column sum
#pragma unroll
            for (int r = 0; r < K; ++r) acc[r * K + k] += (l == r) ? w : 0.f;
        }
#pragma unroll
        for (int r = 0; r < K; ++r) acc[(K + 1) * K + r] += (l == r) ? 1.f : 0.f;
    }
    mx = p2c_wave_max_i32(mx);
    if (lane == 0) smax[wave] = mx;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const float v = p2c_wave_sum_f32(acc[i]);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (tid < NA) {
        float s = 0.f;
        for (int w = 0; w < HL_THREADS / 64; ++w) s += red[w][tid];
        tot[tid] = s;
    }
    if (tid == NA) {
        int m2 = -1;
        for (int i = 0; i < HL_THREADS / 64; ++i) m2 = max(m2, smax[i]);
        tot[NA] = (float)m2;
    }
    __syncthreads();
    if (tid < NP) ws[((size_t)b * split + part) * NP + tid] = tot[tid];
}

__global__ void __launch_bounds__(128) hungarian_finish8_kernel(const float *__restrict__ ws, int split, int64_t *__restrict__ match_out,
                                                               uint8_t *__restrict__ mask_out)
{
    constexpr int K = 8, NA = (K + 1) * K + K, NP = NA + 1;
    __shared__ double cost[HM_MAXK * HM_MAXK];
    __shared__ int col4row[HM_MAXK + 1];
    __shared__ float tot[NP];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid < NP) {
        float s = tid < NA ? 0.f : -1.f;
        for (int q = 0; q < split; ++q) {
            const float v = ws[((size_t)b * split + q) * NP + tid];
            s = tid < NA ? s + v : fmaxf(s, v);
        }
        tot[tid] = s;
    }
    __syncthreads();
    const int n_gt = (int)tot[NA] + 1;
    const int nr = min(n_gt, K);
    if (tid < nr * K) {
        const int r = tid / K, q = tid % K;
        const float dot = tot[r * K + q], col = tot[K * K + q], rc = tot[(K + 1) * K + r];
        const float den = (rc + col) - dot;
        cost[r * K + q] = -(double)(dot / fmaxf(den, 1e-10f));
    }
    __syncthreads();
    if (tid < 64) {
        if (nr > 0) p2c_lsa_min_wave(cost, nr, K, col4row);
        __builtin_amdgcn_wave_barrier();
        if (tid < K) {
            match_out[(size_t)b * K + tid] = tid < nr ? (int64_t)col4row[tid] : 0;
            mask_out[(size_t)b * K + tid] = tid < nr ? 1 : 0;
        }
    }
}

extern "C" int p2c_hungarian_f32(const float *W, const int64_t *I_gt, int B, int N, int K, int64_t *match_out, uint8_t *mask_out,
                                 void *stream)
{
    if (!W || !I_gt || !match_out || !mask_out || B <= 0 || N <= 0 || K <= 0 || K > HM_MAXK) return P2C_EINVAL;
    const size_t lds = (size_t)HM_THREADS * (2 * K + 1) * sizeof(float);
    (void)hipFuncSetAttribute((const void *)hungarian_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(hungarian_kernel, dim3(B), dim3(HM_THREADS), lds, (hipStream_t)stream, W, K, 0, 0, I_gt, N, K, match_out, mask_out);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

#define HL_SPLIT 8
extern "C" size_t p2c_hungarian_ws_bytes(int B) { return (size_t)B * HL_SPLIT * 81 * sizeof(float); }

extern "C" int p2c_hungarian_logits_f32(const float *heads, int ld, int woff, const int64_t *I_gt, int B, int N, int K, int64_t *match_out,
                                        uint8_t *mask_out, void *ws, void *stream)
{
    if (!heads || !I_gt || !match_out || !mask_out || B <= 0 || N <= 0 || K <= 0 || K > HM_MAXK || ld < woff + 2 * K) return P2C_EINVAL;
    if (K == 8 && ws) {
        const int split = N >= 2048 ? HL_SPLIT : 1;
        hipLaunchKernelGGL(hungarian_logits8_kernel, dim3(B * split), dim3(HL_THREADS), 0, (hipStream_t)stream, heads, ld, woff, I_gt, N, split,
                           (float *)ws);
        hipLaunchKernelGGL(hungarian_finish8_kernel, dim3(B), dim3(128), 0, (hipStream_t)stream, (const float *)ws, split, match_out, mask_out);
        P2C_LAUNCH_CHECK();
        return P2C_OK;
    }
    const size_t lds = (size_t)HM_THREADS * (2 * K + 1) * sizeof(float);
    (void)hipFuncSetAttribute((const void *)hungarian_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(hungarian_kernel, dim3(B), dim3(HM_THREADS), lds, (hipStream_t)stream, heads, ld, woff, 1, I_gt, N, K, match_out, mask_out);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

// scipy.optimize.linear_sum_assignment (minimisation; call site losses.py:43) for a batch of small dense problems on the device:
// cost [P, nr, nc] fp64, nr <= nc <= 15 -> col4row [P, nr] int32.  One wave per problem (the solver of the matching kernels above);
// solver 0 runs the single-lane restatement instead (kept as the in-library cross-check of the wave version).
__global__ void __launch_bounds__(64) lsa_kernel(const double *__restrict__ cost, int nr, int nc, int32_t *__restrict__ out, int solver)
{
    __shared__ double c[HM_MAXK * HM_MAXK];
    __shared__ int col4row[HM_MAXK + 1];
    const int p = blockIdx.x, tid = threadIdx.x;
    for (int e = tid; e < nr * nc; e += 64) c[e] = cost[(size_t)p * nr * nc + e];
    __syncthreads();
    if (solver == 0) { if (tid == 0) p2c_lsa_min(c, nr, nc, col4row); }
    else p2c_lsa_min_wave(c, nr, nc, col4row);
    __syncthreads();
    if (tid < nr) out[(size_t)p * nr + tid] = col4row[tid];
}

extern "C" int p2c_linear_sum_assignment_f64(const double *cost, int n_problems, int nr, int nc, int32_t *col4row_out, int solver, void *stream)
{
    if (!cost || !col4row_out || n_problems <= 0 || nr <= 0 || nc < nr || nc > HM_MAXK || solver < 0 || solver > 1) return P2C_EINVAL;
    hipLaunchKernelGGL(lsa_kernel, dim3(n_problems), dim3(64), 0, (hipStream_t)stream, cost, nr, nc, col4row_out, solver);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

extern "C" int p2c_abi_version(void) { return 2; }
extern "C" const char *p2c_build_arch(void) { return "gfx950"; }
