// lsa.h -- scipy.optimize.linear_sum_assignment's solver (Crouse 2016; call site losses.py:43) as device functions, shared by the
// matching kernels (assign.hip) and the fused evaluation metrics (metrics.hip).  No relocatable device code in this build: every
// translation unit that includes this header gets its own copy (static).
#pragma once
#include "common.h"

#define HM_MAXK 15

// Called by ONE thread of a workgroup.  The solver's state is indexed dynamically, which in private arrays means scratch
// memory (a ~1 us round trip per access: the 8x8 solve took 90 us); it lives in LDS instead.  col4row: LDS, >= nr ints.
static __device__ void p2c_lsa_min(const double *cost, int nr, int nc, int *col4row)
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
static __device__ void p2c_lsa_min_wave(const double *cost, int nr, int nc, int *col4row)
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
