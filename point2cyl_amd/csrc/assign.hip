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

__device__ void p2c_lsa_min(const double *cost, int nr, int nc, int *col4row)
{
    double u[HM_MAXK + 1], v[HM_MAXK + 1], spc[HM_MAXK + 1];
    int path[HM_MAXK + 1], row4col[HM_MAXK + 1], remaining[HM_MAXK + 1];
    bool SR[HM_MAXK + 1], SC[HM_MAXK + 1];
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

// thread (slice, k', k): k' in [0, K] where row K collects the plain column sums; slices split the points.
__global__ void __launch_bounds__(256) hungarian_kernel(const float *__restrict__ W, const int64_t *__restrict__ I_gt, int N, int K,
                                                        int64_t *__restrict__ match_out, uint8_t *__restrict__ mask_out)
{
    __shared__ float part[256];
    __shared__ float cnt_part[256];
    __shared__ int smax[4];
    __shared__ double cost[HM_MAXK * HM_MAXK];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t *lab = I_gt + (size_t)b * N;
    const float *w = W + (size_t)b * N * K;
    // n_gt = max(I_gt)+1  (losses.py:36)
    int mx = -1;
    for (int n = tid; n < N; n += 256) mx = max(mx, (int)lab[n]);
    mx = p2c_wave_max_i32(mx);
    if (lane == 0) smax[wave] = mx;
    __syncthreads();
    const int n_gt = max(max(smax[0], smax[1]), max(smax[2], smax[3])) + 1;
    const int pairs = (K + 1) * K;
    const int slices = 256 / pairs;
    const int sl = tid / pairs, pr = tid - sl * pairs;
    const int kp = pr / K, k = pr - kp * K;
    float s = 0.f, c = 0.f;
    if (sl < slices) {
        const int per = (N + slices - 1) / slices;
        const int n1 = min(N, (sl + 1) * per);
        for (int n = sl * per; n < n1; ++n) {
            const int l = (int)lab[n];
            const bool hit = (kp == K) || (l == kp);
            if (hit) { s += w[(size_t)n * K + k]; c += 1.f; }
        }
    }
    part[tid] = s;
    cnt_part[tid] = c;
    __syncthreads();
    if (tid == 0) {
        // dot[k'][k], colsum[k], rowcount[k'] in fp32 like the reference's torch.mm / torch.sum
        for (int r = 0; r < n_gt && r < K; ++r)
            for (int q = 0; q < K; ++q) {
                float dot = 0.f, col = 0.f, rc = 0.f;
                for (int z = 0; z < slices; ++z) {
                    dot += part[z * pairs + r * K + q];
                    col += part[z * pairs + K * K + q];
                    rc += cnt_part[z * pairs + r * K + 0];
                }
                const float den = (rc + col) - dot;                       // :40
                const float iou = dot / fmaxf(den, 1e-10f);               // :41
                cost[r * K + q] = -(double)iou;                           // :43 maximise
            }
        int col4row[HM_MAXK + 1];
        const int nr = min(n_gt, K);
        if (nr > 0) p2c_lsa_min(cost, nr, K, col4row);
        for (int q = 0; q < K; ++q) {
            match_out[(size_t)b * K + q] = q < nr ? (int64_t)col4row[q] : 0;   // rest stays 0 (:30)
            mask_out[(size_t)b * K + q] = q < nr ? 1 : 0;                      // :47
        }
    }
}

extern "C" int p2c_hungarian_f32(const float *W, const int64_t *I_gt, int B, int N, int K, int64_t *match_out, uint8_t *mask_out,
                                 void *stream)
{
    if (!W || !I_gt || !match_out || !mask_out || B <= 0 || N <= 0 || K <= 0 || K > HM_MAXK || (K + 1) * K > 256) return P2C_EINVAL;
    hipLaunchKernelGGL(hungarian_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, W, I_gt, N, K, match_out, mask_out);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}

extern "C" int p2c_abi_version(void) { return 1; }
extern "C" const char *p2c_build_arch(void) { return "gfx950"; }
