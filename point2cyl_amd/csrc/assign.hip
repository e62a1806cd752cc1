// assign.hip -- hungarian_matching (losses.py:22-52) without leaving the device.
//
// The reference builds, per sample, an IoU cost matrix (K' x K) with torch.mm, copies it to the host and
// calls scipy.optimize.linear_sum_assignment there (one blocking .cpu() per sample).  Here one workgroup
// per sample reduces the (K'+1) x K intersection / union sums deterministically and a single lane runs
// the same shortest-augmenting-path solver scipy uses (Crouse 2016, restated in oracle/p2c_oracle.c and
// checked against scipy on tie-heavy inputs), so the matching is available to the next kernel with no
// host round trip.
#include "common.h"

#define HM_MAXK 15

// Called by ONE thread of a workgroup.  The solver's state is indexed dynamically, which in private arrays means scratch
// memory (a ~1 us round trip per access: the 8x8 solve took 90 us); it lives in LDS instead.  col4row: LDS, >= nr ints.
__device__ void p2c_lsa_min(const double *cost, int nr, int nc, int *col4row)
{
    __shared__ double u[HM_MAXK + 1], v[HM_MAXK + 1], spc[HM_MAXK + 1];
    __shared__ int path[HM_MAXK + 1], row4col[HM_MAXK + 1], remaining[HM_MAXK + 1];
    __shared__ bool SR[HM_MAXK + 1], SC[HM_MAXK + 1];
    for (int i = 0; i < nr; ++i) { u[i] = 0.0; col4row[i] = -1; }
    for (int j = 0; j < nc; ++j) { v[j] = 0.0; row4col[j] = -1; path[j] = -1; }
    for (int cur = 0; cur < nr; ++cur) {
        double minVal = 0.0;
        int num_remaining = nc;
        for (int it = 0; it < nc; ++it) remaining[it] = nc - it - 1;
        for (int i = 0; i < nr; ++i) SR[i] = false;
        for (int j = 0; j < nc; ++j) { SC[j] = false; spc[j] = INFINITY; }
        int sink = -1, i = cur;
        while (sink == -1) {
            int index = -1;
            double lowest = INFINITY;
            SR[i] = true;
            for (int it = 0; it < num_remaining; ++it) {
                const int j = remaining[it];
                const double r = minVal + cost[i * nc + j] - u[i] - v[j];
                if (r < spc[j]) { path[j] = i; spc[j] = r; }
                if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) { lowest = spc[j]; index = it; }
            }
            minVal = lowest;
            if (index < 0) return;            // infeasible (cannot happen for finite costs)
            const int j = remaining[index];
            if (row4col[j] == -1) sink = j; else i = row4col[j];
            SC[j] = true;
            remaining[index] = remaining[--num_remaining];
        }
        u[cur] += minVal;
        for (int r = 0; r < nr; ++r)
            if (SR[r] && r != cur) u[r] += minVal - spc[col4row[r]];
        for (int j = 0; j < nc; ++j)
            if (SC[j]) v[j] -= minVal - spc[j];
        int j = sink;
        for (;;) {
            const int r = path[j];
            row4col[j] = r;
            const int t = col4row[r]; col4row[r] = j; j = t;
            if (r == cur) break;
        }
    }
}

// The same solver run by ONE WAVE: lane j owns column j (v, shortest path cost, predecessor, assigned row, position in the
// `remaining` list), lane i owns row i (u, assigned column, "in the tree" flag); the sequential version's scans over the remaining
// columns become 16-lane reductions.  Decision for decision the same as p2c_lsa_min, including its tie rule - the scan takes a
// column when it is strictly cheaper, or equally cheap and unassigned, so among the cheapest columns it ends on the LAST unassigned
// one in list order, else on the first - and the swap-with-last removal that defines that order; all arithmetic in the same order
// in fp64.  Twice as fast as the single-lane version (whose every step is a dependent LDS round trip); a variant that publishes
// the columns in LDS and lets every lane rescan them was slower than both.
// Called by all 64 lanes of one wave (converged), nr <= nc <= HM_MAXK (<= 15: one DPP row).  cost, col4row: LDS.
// Cross-lane traffic of the solver: DPP inside the row of 16 lanes and v_readlane for the wave-uniform picks.  __shfl / __shfl_xor go
// through ds_bpermute (an LDS crossbar round trip, ~100 cycles each, a dozen dependent ones per step of the search): the 8 x 8 problems
// of a training step took 26 us that way.
template <int CTRL>
__device__ __forceinline__ double p2c_dpp_f64(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = p2c_dpp<CTRL>((int)(b & 0xffffffffll)), hi = p2c_dpp<CTRL>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double p2c_row16_min_f64(double v)     // every lane of the row ends with the row's minimum
{
    v = fmin(v, p2c_dpp_f64<0xB1>(v));    // quad_perm [1,0,3,2]
    v = fmin(v, p2c_dpp_f64<0x4E>(v));    // quad_perm [2,3,0,1]
    v = fmin(v, p2c_dpp_f64<0x141>(v));   // row_half_mirror
    v = fmin(v, p2c_dpp_f64<0x140>(v));   // row_mirror
    return v;
}
__device__ __forceinline__ int p2c_row16_min_i32(int v)
{
    v = min(v, p2c_dpp<0xB1>(v));
    v = min(v, p2c_dpp<0x4E>(v));
    v = min(v, p2c_dpp<0x141>(v));
    v = min(v, p2c_dpp<0x140>(v));
    return v;
}
__device__ __forceinline__ int p2c_readlane_i32(int v, int lane_uniform) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(lane_uniform)); }
__device__ __forceinline__ double p2c_readlane_f64(double v, int lane_uniform)
{
    const int l = __builtin_amdgcn_readfirstlane(lane_uniform);
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ void p2c_lsa_min_wave(const double *cost, int nr, int nc, int *col4row)
{
    const int lane = threadIdx.x & 63;
    double u = 0.0, v = 0.0, spc = INFINITY;
    int path = -1, row4col = -1, c4r = -1;
    for (int cur = 0; cur < nr; ++cur) {
        double minVal = 0.0;
        int num_remaining = nc;
        int pos = lane < nc ? nc - 1 - lane : -1;            // remaining[it] = nc - it - 1
        bool SR = false, SC = false;
        spc = INFINITY;
        int sink = -1, i = cur;
        while (sink == -1) {
            if (lane == i) SR = true;
            const double ui = p2c_readlane_f64(u, i);
            const bool active = pos >= 0;
            if (active) {
                const double r = minVal + cost[i * nc + lane] - ui - v;
                if (r < spc) { path = i; spc = r; }
            }
            double m = p2c_readlane_f64(p2c_row16_min_f64(active ? spc : INFINITY), 0);
            if (!(m < INFINITY)) return;                     // infeasible (cannot happen for finite costs)
            const bool eq = active && spc == m;
            int ku = (eq && row4col == -1) ? pos : -1;       // last unassigned among the cheapest ...
            int ka = eq ? pos : 0x7fffffff;                  // ... else the first of them
            ku = __builtin_amdgcn_readlane(p2c_row16_max_i32(ku), 0);
            ka = __builtin_amdgcn_readlane(p2c_row16_min_i32(ka), 0);
            const int psel = ku >= 0 ? ku : ka;
            const int jsel = __ffsll((long long)(__ballot(active && pos == psel) & 0xFFFFull)) - 1;
            minVal = m;
            const int rc = p2c_readlane_i32(row4col, jsel);
            if (rc == -1) sink = jsel; else i = rc;
            if (lane == jsel) SC = true;
            // remaining[index] = remaining[--num_remaining]
            --num_remaining;
            if (pos == num_remaining && lane != jsel) pos = psel;
            if (lane == jsel) pos = -1;
        }
        // dual updates (rows of the tree other than cur read the path cost of their assigned column)
        const double spc_c = __shfl(spc, c4r >= 0 ? c4r : 0, 64);        // per-lane index: a real permute
        if (lane == cur) u += minVal;
        else if (SR && lane < nr) u += minVal - spc_c;
        if (SC) v -= minVal - spc;
        // augment along the predecessors
        int j = sink;
        for (;;) {
            const int r = p2c_readlane_i32(path, j);
            if (lane == j) row4col = r;
            const int t = p2c_readlane_i32(c4r, r);
            if (lane == r) c4r = j;
            j = t;
            if (r == cur) break;
        }
    }
    if (lane < nr) col4row[lane] = c4r;
}

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
