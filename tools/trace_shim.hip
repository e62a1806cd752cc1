// Trace build (-DP2C_TRACE) of the persistent kernels in ONE throw-away library: both forward kernels, both fused backward kernels and
// the mode switch they share, plus the plain C entry the forward trace uses (the product entry lives in gemm.hip).
#include "../point2cyl_amd/csrc/fwd_pp.hip"
#include "../point2cyl_amd/csrc/fwd_pp3.hip"
#include "../point2cyl_amd/csrc/bwd_fused.hip"
#include "../point2cyl_amd/csrc/bwd_fused3.hip"
extern "C" int p2c_trace_fwd(const float *X, int ldx, const float *W, int ldw, const float *bias, float *Y, int ldy, int M, int N, int K,
                             const float *sc, const float *sh, double *partials, void *stream)
{
    // K == 132: the grouped layer [128 features | xyz | pad], raw input (mode 0), extra columns in the epilogue
    FwdPPArgs a{X, ldx, W, ldw, bias, Y, ldy, M, N, K == 132 ? 128 : K, sc, sh, nullptr, 0u, 1.f, K, partials, nullptr, nullptr};
    return p2c_fwd_pp_launch(a, K == 132 ? 0 : 1, (hipStream_t)stream);
}
