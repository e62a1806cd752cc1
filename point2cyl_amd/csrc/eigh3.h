// eigh3.h -- cyclic-Jacobi eigen-solve of a symmetric 3x3 in fp64 (the reference's torch.symeig per segment, data_utils.py:170), shared by
// the fitting kernels (fit.hip) and the fused evaluation metrics (metrics.hip).
#pragma once
#include "common.h"

// 1 / a and 1 / sqrt(a) in fp64 from the hardware seeds (v_rcp_f64 / v_rsq_f64: ~2^-23 relative) and two Newton steps each: full fp64
// accuracy to an ulp or two in ~8 instructions, against ~40 for the IEEE-exact division / square root the compiler expands `/` and sqrt()
// into (scaling, fix-up, special cases).  The Jacobi rotation only needs c^2 + s^2 = 1 to rounding - the ANGLE's last bits just move the
// next sweep's off-diagonal by 1e-16 of itself - so exact rounding buys nothing there; the eight solves of a cloud were 13 k of its ~95 k
// cycles (tools/fit_trace.py).  Arguments: finite, |a| >= 1e-280 (rcp), a >= 1 (rsq) - guaranteed at the call sites below.
static __device__ __forceinline__ double p2c_rcp64(double a)
{
    double r = __builtin_amdgcn_rcp(a);
    double e = __builtin_fma(-a, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-a, r, 1.0);
    return __builtin_fma(r, e, r);
}
static __device__ __forceinline__ double p2c_rsqrt64(double a)
{
    double y = __builtin_amdgcn_rsq(a);
    const double h = 0.5 * a;
    double e = __builtin_fma(-h * y, y, 0.5);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-h * y, y, 0.5);
    return __builtin_fma(y, e, y);
}

// cyclic Jacobi for a symmetric 3x3 (fp64).  a = {a00,a01,a02,a11,a12,a22}; out: lam[3] ascending, v[3][3]
// (v[j] = j-th eigenvector).  Zero matrix -> identity eigenvectors (LAPACK's answer too).
static __device__ void p2c_eigh3(const double a_in[6], double lam[3], double v[3][3])
{
    double A[3][3] = {{a_in[0], a_in[1], a_in[2]}, {a_in[1], a_in[3], a_in[4]}, {a_in[2], a_in[4], a_in[5]}};
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        const double diag = fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]);
        if (off <= 1e-18 * diag || off == 0.0) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                // (the entries are fp64 sums of fp32 products: zero or >= 1e-90 in magnitude - nothing between; theta stays below 1e100)
                if (!(fabs(A[p][q]) > 1e-280)) continue;
                const double theta = (A[q][q] - A[p][p]) * p2c_rcp64(2.0 * A[p][q]);
                const double th2 = theta * theta + 1.0;
                const double t = (theta >= 0 ? 1.0 : -1.0) * p2c_rcp64(fabs(theta) + th2 * p2c_rsqrt64(th2));
                const double c = p2c_rsqrt64(t * t + 1.0), s = t * c;
                for (int r = 0; r < 3; ++r) {   // A <- A J
                    const double arp = A[r][p], arq = A[r][q];
                    A[r][p] = c * arp - s * arq;
                    A[r][q] = s * arp + c * arq;
                }
                for (int r = 0; r < 3; ++r) {   // A <- J^T A
                    const double apr = A[p][r], aqr = A[q][r];
                    A[p][r] = c * apr - s * aqr;
                    A[q][r] = s * apr + c * aqr;
                }
                for (int r = 0; r < 3; ++r) {
                    const double vrp = V[r][p], vrq = V[r][q];
                    V[r][p] = c * vrp - s * vrq;
                    V[r][q] = s * vrp + c * vrq;
                }
            }
    }
    int o[3] = {0, 1, 2};
    double d[3] = {A[0][0], A[1][1], A[2][2]};
    // stable ascending sort of 3
    if (d[o[1]] < d[o[0]]) { int t = o[0]; o[0] = o[1]; o[1] = t; }
    if (d[o[2]] < d[o[1]]) { int t = o[1]; o[1] = o[2]; o[2] = t; }
    if (d[o[1]] < d[o[0]]) { int t = o[0]; o[0] = o[1]; o[1] = t; }
    for (int j = 0; j < 3; ++j) {
        lam[j] = d[o[j]];
        for (int r = 0; r < 3; ++r) v[j][r] = V[r][o[j]];
    }
}
