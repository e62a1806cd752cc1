// fwd_pp.h -- arguments of the persistent forward kernel (fwd_pp.hip), launched from p2c_linear_fwd_f32 (gemm.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct FwdPPArgs {
    const float *x; int ldx;
    const float *w; int ldw;
    const float *bias;
    float *y; int ldy;
    int M, N, K;                          // N, K: this launch's logical sizes (K excludes the EX trailing columns)
    const float *in_scale, *in_shift;     // MODE >= 1
    const uint32_t *seed; uint32_t thr; float dscale;   // MODE == 3 (hashed dropout, same stream as gemm.hip OpActIn<3>)
    int Kfull;                            // width used in the dropout element index (row * Kfull + col)
    double *partials;                     // [P2C_STAT_SLOTS][2][N] or NULL
    const float *w0, *b0;                 // MODE == 4: x is the folded first layer's INPUT [M,4]; w0 [K,4], b0 [K]; in_scale/in_shift = its BN
};


extern "C" int p2c_linear_fwd_pp_supported(int M, int N, int K, int in_mode);
int p2c_fwd_pp_launch(const FwdPPArgs &a, int in_mode, hipStream_t s);
