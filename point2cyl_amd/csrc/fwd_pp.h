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
    // POOL: per 32-row half of every 64-row tile (= one neighbourhood of 64) and column, the largest and the smallest pre-BN value
    // and their rows: pool_max / pool_min [2*M/64][N] fp32, pool_idx [2*M/64][N] int32 = row_of_max | row_of_min << 16 (rows 0..63)
    float *pool_max, *pool_min;
    int32_t *pool_idx;
    int lockstep;                         // fwd_pp3.hip: both halves in the same phase (set by its launcher; 0 elsewhere)
};


extern "C" int p2c_linear_fwd_pp_supported(int M, int N, int K, int in_mode);
int p2c_fwd_pp_launch(const FwdPPArgs &a, int in_mode, hipStream_t s);
extern "C" int p2c_linear_fwd_pool_supported(int M, int N, int K, int in_mode, int ns);

// bf16x3-split twin of the persistent forward (fwd_pp3.hip): same arguments, same results at fp32 accuracy, 2.7x fewer matrix-pipe cycles.
int p2c_fwd_pp3_launch(const FwdPPArgs &a, int in_mode, hipStream_t s);
// 1 (default): the persistent kernels run on the bf16 matrix pipe with three-way split operands; 0: v_mfma_f32_32x32x2_f32.
// Initialised from the environment (P2C_MFMA=f32 selects 0), changed by p2c_set_mfma_mode (A/B runs, tests).
int p2c_mfma_split();
