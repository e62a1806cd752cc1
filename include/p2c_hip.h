/*
 * p2c_hip.h -- C ABI of libp2c_hip.so: the MI355X (gfx950) kernels behind Point2Cyl's hot path.
 *
 * The upstream reference (mikacuy/point2cyl) is pure Python on PyTorch: it has no FFI, no custom op and
 * no plugin ABI.  Its "interface" for this path is a set of Python functions / nn.Modules; each entry
 * point below replaces the stock-torch op sequence inside one of them (file:line cited per function).
 * INTEGRATION.md shows the ctypes stub a maintainer of the reference would add to call these.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HIP, gfx950) unless the name ends in _host;
 *   - all float tensors are fp32, row-major, POINT-MAJOR: a feature map is [rows, channels] with an
 *     explicit leading dimension (ld*, in elements) so that column slices of a wider matrix can be passed;
 *     the GEMM entry points require 16-byte aligned bases, ld % 4 == 0 and channel counts % 4 == 0
 *     (P2C_EALIGN otherwise; the host zero-pads, e.g. 131 -> 132 input channels);
 *   - index tensors are int32 on the device (the Python boundary converts to int64 where the reference
 *     returns int64);
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it, nothing synchronises, nothing
 *     allocates; buffers are caller-owned; workspace sizes are given by the *_ws_bytes helpers;
 *   - return value: 0 on success, a negative P2C_E* code on bad arguments, or the positive hipError_t of
 *     a failed launch.
 */
#ifndef P2C_HIP_H
#define P2C_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P2C_OK 0
#define P2C_EINVAL (-1)   /* bad shape / null pointer / unsupported size */
#define P2C_EALIGN (-2)   /* leading dimension or pointer not aligned as required */

int p2c_abi_version(void);            /* bumps when a signature changes */
const char *p2c_build_arch(void);     /* "gfx950" */

/* ---------------------------------------------------------------------------------------------
 * Geometry (integer outputs: bit-exact with the reference's CPU path)
 * ------------------------------------------------------------------------------------------- */

/* farthest_point_sample, models/pointnet_util.py:63-84.
 * xyz [B,N,3]; start [B] int64 = the reference's torch.randint(0,N,(B,)) draw (:75), made by the caller on
 * the CPU generator exactly like the reference; idx_out [B,npoint]; new_xyz_out [B,npoint,3] (optional,
 * = index_points(xyz, idx), :127).  N <= 65536 (above 16384 the coordinates are re-read from memory every
 * iteration; same indices). */
int p2c_fps_f32(const float *xyz, int B, int N, const int64_t *start, int npoint, int32_t *idx_out,
                float *new_xyz_out, void *stream);

/* query_ball_point (+ square_distance), models/pointnet_util.py:87-107, :19-40.
 * radius2 = (float)(radius**2).  idx_out [B,S,nsample]: first nsample in-ball indices ascending, padded
 * with the first.  nsample <= 64. */
int p2c_ball_query_f32(const float *xyz, const float *new_xyz, int B, int N, int S, float radius2, int nsample,
                       int32_t *idx_out, void *stream);

/* 3-NN + inverse-distance weights, models/pointnet_util.py:301-307.
 * xyz1 [B,N,3] dense, xyz2 [B,S,3] sparse (S>=3) -> idx_out [B,N,3], weight_out [B,N,3]
 * (w = 1/(d+1e-8), normalised); dist_out [B,N,3] optional (the 3 smallest squared distances, ascending;
 * ties broken towards the lower index). */
int p2c_three_nn_f32(const float *xyz1, const float *xyz2, int B, int N, int S, int32_t *idx_out, float *weight_out,
                     float *dist_out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Grouping / interpolation (gathers; backward = scatter-add)
 * ------------------------------------------------------------------------------------------- */

/* sample_and_group's gather + centring + concat, models/pointnet_util.py:128-139.
 * out row (b,s,j) = [xyz[b,idx]-new_xyz[b,s] (3) | feats[b,idx,:D] | 0-pad up to ldo], ldo % 4 == 0  (the reference's
 * channel order, :137), or with xyz_last != 0  [feats | xyz_rel | 0-pad]: the 16-byte aligned feature block first, so the
 * backward GEMMs can skip the 3 coordinate columns whose gradient nobody needs (the host permutes the conv weight's input
 * channels to match).  feats may be NULL (D=0).  ldf = leading dimension of feats rows. */
int p2c_group_gather_f32(const float *xyz, const float *feats, int ldf, const float *new_xyz, const int32_t *idx, int B,
                         int N, int S, int nsample, int D, float *out, int ldo, int xyz_last, void *stream);
/* backward of the feature part: dfeats[b,idx[b,s,j],:] += dout[(b,s,j), c0:c0+D], c0 = 0 if xyz_last else 3 (dfeats pre-zeroed) */
int p2c_group_gather_bwd_f32(const float *dout, int ldo, const int32_t *idx, int B, int N, int S, int nsample, int D,
                             float *dfeats, int ldf, int xyz_last, void *stream);

/* weighted 3-NN interpolation, models/pointnet_util.py:308.
 * out[b,n,:C] = sum_j w[b,n,j] * feats[b, idx[b,n,j], :C] */
int p2c_three_interp_f32(const float *feats, int ldf, const int32_t *idx, const float *weight, int B, int N, int S, int C,
                         float *out, int ldo, void *stream);
int p2c_three_interp_bwd_f32(const float *dout, int ldo, const int32_t *idx, const float *weight, int B, int N, int S,
                             int C, float *dfeats, int ldf, void *stream);
/* input of a feature-propagation level in one launch (pointnet_util.py:308-312: cat([points1, interpolated])):
 * out[b*N+n, :] = [skip[b*N+n, :Cskip] | interpolated (C) | zeros up to width], width <= ldo */
int p2c_three_interp_skip_f32(const float *feats, int ldf, const int32_t *idx, const float *weight, int B, int N, int S, int C,
                              const float *skip, int ldskip, int Cskip, float *out, int ldo, int width, void *stream);

/* Atomic-free backward of both gathers: the inverse map (target row -> entries reading it) depends only on the
 * geometry, so it is built once per batch and the backward becomes a gather.
 * p2c_build_csr_i32: idx [B,E] with values in [0,T) (w [B,E] or NULL) -> offsets [B,T+1]; rows [B,E] = source row
 *   (entry / ediv) of every entry, grouped by target; wsorted [B,E] = w in the same order.
 * p2c_csr_gather_f32: out[b,t,:C] = sum_{k in bucket(b,t)} wsorted[b,k] * src[b*rows_b + rows[b,k], coff:coff+C].
 *   three_interp backward: E = 3N, ediv = 3, rows_b = N, T = S, src = dout;  group_gather backward: E = S*nsample,
 *   ediv = 1, rows_b = E, T = N, src = dout (coff = column of the feature block), wsorted = NULL. */
int p2c_build_csr_i32(const int32_t *idx, const float *w, int B, int E, int ediv, int T, int32_t *offsets, int32_t *rows,
                      float *wsorted, void *stream);
int p2c_csr_gather_f32(const float *src, int ld_src, int coff, const int32_t *offsets, const int32_t *rows, const float *wsorted,
                       int B, int E, int rows_b, int T, int C, float *out, int ldo, void *stream);

/* "Linear before the gather": a 1x1-conv applied to gathered rows commutes with the gather, so the first layer of FP1 (and
 * of SA2) runs on the sparse set and the dense pre-BN tensor is produced by the gather itself.
 * p2c_three_interp_bias_stats_f32: out = interp(feats) + bias, BatchNorm sums of the bias-free value into stat_slots
 *   (p2c_stat_slots_bytes(C), zeroed by the caller; NULL = no sums).  C <= 256.
 * p2c_csr_gather_bn_f32: p2c_csr_gather_f32 of dY = gs*(dZ*[scale*Y+shift>0]) + q*Y + p (coef [5][C] as produced by the
 *   *_bwd_stats / bwd_finalize entry points), rebuilt per element from the stored dZ and Y. */
int p2c_three_interp_bias_stats_f32(const float *feats, int ldf, const int32_t *idx, const float *weight, int B, int N, int S, int C,
                                    const float *bias, float *out, int ldo, double *stat_slots, void *stream);
int p2c_csr_gather_bn_f32(const float *dz, int lddz, const float *y, int ldy, const float *coef, const int32_t *offsets,
                          const int32_t *rows, const float *wsorted, int B, int E, int rows_b, int T, int C, float *out, int ldo,
                          void *stream);
/* Grouped layer with point features (SA2; pointnet_util.py:128-139 + the first conv of :201): with W = [Wx | Wf] in the
 * reference's [xyz | features] order and G = F . Wf^T computed on the N points,
 *   out[(b,s,j), :] = G[b, idx[b,s,j], :] + Wx . (xyz[b, idx] - new_xyz[b, s]) + bias      (Wx [C,4] row-major, 4th column unused)
 * and the BatchNorm sums of the bias-free value go to stat_slots (NULL: none).  The grouped input tensor is never built.
 * p2c_group_linear_bwd_f32: dG[b,p,:] = sum over the grouped rows reading point p of dY (dY rebuilt from dz, y, coef as in
 *   p2c_csr_gather_bn_f32; offsets/rows = p2c_build_csr_i32 of idx with T = N; dG is ACCUMULATED with atomics: zero it first),
 *   and dWx accumulated into
 *   dwx_slots [P2C_STAT_SLOTS][3][C] (fp64, zeroed by the caller; the caller sums the slots). */
int p2c_group_linear_bias_stats_f32(const float *G, int ldg, const float *xyz, const float *new_xyz, const int32_t *idx, const float *Wx,
                                    const float *bias, int B, int N, int S, int nsample, int C, float *out, int ldo,
                                    double *stat_slots, void *stream);
int p2c_group_linear_bwd_f32(const float *dz, int lddz, const float *y, int ldy, const float *coef, const int32_t *offsets,
                             const int32_t *rows, const float *xyz, const float *new_xyz, int B, int N, int S, int nsample, int C,
                             float *dG, int ldo, double *dwx_slots, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Shared per-point MLP: 1x1 conv (+ train-mode BatchNorm + ReLU) as fp32 MFMA GEMMs
 * (models/pointnet_util.py:201-205, :317-319; models/pointnet_extrusion.py:58-65)
 *
 * A layer is  Y = act_in(X) . W^T + bias,  with the PREVIOUS layer's BatchNorm+ReLU folded into the load
 * of X (act_in) so post-activation tensors are never written to HBM:
 *   in_mode 0: act_in(x) = x
 *   in_mode 1: act_in(x) = relu(in_scale[k]*x + in_shift[k])
 *   in_mode 2: act_in(x) = relu(in_scale[k]*x + in_shift[k]) * drop_mask[m,k] * drop_scale   (F.dropout, :60)
 *   in_mode 3: as 2, but the keep-mask is regenerated from a counter hash: drop_mask points to a device
 *              uint32_t seed[2]; keep(m,k) = hash(seed, m*K+k) >= (1 - 1/drop_scale) * 2^32; ldmask is ignored.
 *              (backward-data: pass the same seed pointer as out_mask with ldmask = -1)
 * ------------------------------------------------------------------------------------------- */

/* Per-channel reductions (BatchNorm sums forward and backward) are accumulated by the producing kernels with fp64
 * atomics into P2C_STAT_SLOTS rows:  slots[s][0][c], slots[s][1][c]  (workgroup b adds into row b % 64); the
 * caller zero-initialises p2c_stat_slots_bytes(C) bytes and hands the same buffer to the matching finalize call. */
#define P2C_STAT_SLOTS 64
size_t p2c_stat_slots_bytes(int C);

/* Forward.  X [M,K] (ldx), W [N,K] row-major (ldw) = the conv weight (Co,Ci,1[,1]) as stored in the
 * reference's state_dict, bias [N] (may be NULL), Y [M,N] (ldy).  If stat_slots != NULL the kernel also
 * accumulates sum_m (y-bias) and sum_m (y-bias)^2 per output channel into it (see above). */
int p2c_linear_tile_m(void);         /* row-tile height (64, or 128 with P2C_TILE_M=128 in the environment) */
int p2c_linear_stat_tiles(int M);    /* ceil(M / p2c_linear_tile_m()) */
int p2c_linear_fwd_f32(const float *X, int ldx, const float *W, int ldw, const float *bias, float *Y, int ldy, int M, int N,
                       int K, int in_mode, const float *in_scale, const float *in_shift, const uint8_t *drop_mask,
                       int ldmask, float drop_scale, double *stat_slots, void *stream);
/* ---- Folded first layer.  A stack whose first 1x1-conv layer has <= 4 input channels (the grouped relative coordinates of
 * SA1, rows [dx,dy,dz,0]: models/pointnet_util.py:136 with no point features) never writes that layer's output: its
 * train-mode BatchNorm statistics follow from the moments of the input, and the consumers rebuild
 * Y0 = X0 W0^T + b0 from the 16-byte input row while they stage their operand tiles (268 MB less written and three
 * times less read at config 1).
 * p2c_input_moments_f32: moments[0:4] += sum_m x, moments[4:14] += upper triangle of sum_m x x^T (fp64; zero it first).
 * p2c_bn_finalize_affine_f32: stat = [scale|shift|mean|invstd] x C of that layer (and its running statistics, momentum
 *   as in p2c_bn_finalize_f32) from the moments; W0 [C,4] row-major (4th column multiplies the zero pad), b0 [C] or NULL.
 * p2c_linear_fwd_fold0_f32: Y = relu(scale0*(X0 W0^T + b0) + shift0) . W^T + bias for the NEXT layer (C0 == 64, N <= 256,
 *   M >= 8192), stat_partials as in p2c_linear_fwd_f32. */
int p2c_input_moments_f32(const float *X0, int ldx0, long long M, double *moments, void *stream);
int p2c_bn_finalize_affine_f32(const double *moments, long long M, const float *W0, const float *b0, const float *gamma,
                               const float *beta, float eps, float momentum, float *running_mean, float *running_var, int C,
                               float *stat, void *stream);
int p2c_linear_fwd_fold0_f32(const float *X0, int ldx0, const float *W0, const float *b0, const float *scale0, const float *shift0,
                             int C0, const float *W, int ldw, const float *bias, float *Y, int ldy, int M, int N,
                             double *stat_partials, void *stream);

/* Backward of the pair (folded layer, next layer): p2c_linear_bwd_fused_fold0_f32 is the fused backward of the NEXT layer
 * (grad_mode 1 arguments of p2c_linear_bwd_fused_f32; Co in {64,128}, C0 == 64) with its X operand rebuilt from X0; it
 * stores no dX and leaves 5 sums per column of the folded layer in partials5 [P2C_STAT_SLOTS][5][C0] (fp64, zeroed by the
 * caller).  p2c_fold0_bwd_finalize_f32 turns them and the moments into dgamma0, dbeta0 and dW0 [C0,4]. */
int p2c_linear_bwd_fused_fold0_f32(const float *dZ, int lddz, const float *Yfwd, int ldy, const float *coef, const float *X0, int ldx0,
                                   const float *W0, const float *b0, const float *stat0, const float *W, int ldw, float *dW, int lddw,
                                   long long dw_slot_stride, double *partials5, int M, int Co, int C0, void *stream);
int p2c_fold0_bwd_finalize_f32(const double *partials5, const double *moments, long long M, const float *W0, const float *b0,
                               const float *stat0, const float *gamma0, int C0, float *dgamma0, float *dbeta0, float *dW0, void *stream);
/* the same with dW0 written [C0, lddw0] (lddw0 in 1..4; 3 = the (C0,3,1,1) parameter's own layout) and, by extra workgroups of the
 * launch, the per-XCD copies of the consumer layer's dW summed: out[i] = sum_c src[c*stride + i], i < n (as p2c_sum_copies_f32) */
int p2c_fold0_bwd_finalize_sum_f32(const double *partials5, const double *moments, long long M, const float *W0, const float *b0,
                                   const float *stat0, const float *gamma0, int C0, float *dgamma0, float *dbeta0, float *dW0, int lddw0,
                                   const float *src, long long stride, int copies, float *out, long long n, void *stream);
/* weight gradient of a grouped first layer in the parameter's layout out [Co, 3 + Cf] = [coordinate part | feature part]
 * (pointnet_util.py:137 column order): columns 0..2 = sum over the P2C_STAT_SLOTS fp64 rows of dwx_slots [slots][3][Cs]
 * (p2c_group_linear_bwd_f32), columns 3.. = dW[:, :Cf] of the feature GEMM (dW [Co, lddw]). */
int p2c_group_weight_grad_f32(const double *dwx_slots, int Cs, const float *dW, int lddw, int Co, int Cf, float *out, void *stream);

/* Layer fed by [X | one vector per row group repeated over the group's rows] (FP3: the global feature repeated over the 128
 * points of a cloud, pointnet_util.py:298-299, :312).  The repeated part's product G = V . Wb^T is computed once per group by the
 * caller (a tiny p2c_linear_fwd_f32) and enters as a per-group additive term:
 *   p2c_linear_fwd_gbias_f32: Y[m,:] = act_in(X)[m,:] . Wa^T + gbias[m / rows_per_group, :] + bias   (rows_per_group % 64 == 0,
 *     M % rows_per_group == 0, in_mode 0/1; the BatchNorm sums include the per-group term);
 *   p2c_group_colsum_bn_f32: out[g,:] = sum over the group's rows of dY (rebuilt from dz, y, coef): the gradient of gbias. */
int p2c_linear_fwd_gbias_f32(const float *X, int ldx, const float *W, int ldw, const float *bias, const float *gbias, int ldgb,
                             int rows_per_group, float *Y, int ldy, int M, int N, int K, int in_mode, const float *in_scale,
                             const float *in_shift, double *stat_partials, void *stream);
int p2c_group_colsum_bn_f32(const float *dz, int lddz, const float *y, int ldy, const float *coef, int G, int rows_per_group, int C,
                            float *out, int ldo, void *stream);

/* 1 if p2c_linear_fwd_f32 runs this shape on the persistent weight-stationary kernel (fwd_pp.hip: M >= 8192, N <= 256,
 * K <= 128 or the grouped K == 132, no byte mask); otherwise the tiled kernel is used.  Same results either way. */
int p2c_linear_fwd_pp_supported(int M, int N, int K, int in_mode);

/* Which matrix pipe the persistent kernels (p2c_linear_fwd_f32 / _pool / _fold0 on the fwd_pp path, p2c_linear_bwd_fused_f32) use.
 *   1 (default): bf16x3 split - every fp32 operand is split into three bf16 pieces while its tile is staged and a 32x32x16 block is
 *      accumulated from six v_mfma_f32_32x32x16_bf16 products in the fp32 accumulator (fwd_pp3.hip, bwd_fused3.hip): fp32-class
 *      accuracy (the same parity tests pass in both modes) at 2.7x fewer matrix-pipe cycles.
 *   0: v_mfma_f32_32x32x2_f32 (fwd_pp.hip, bwd_fused.hip).  Initial value from the environment: P2C_MFMA=f32 selects 0.
 * p2c_set_mfma_mode returns the previous mode.  Not thread-safe against concurrent launches (a process-wide A/B switch). */
int p2c_set_mfma_mode(int split);
int p2c_get_mfma_mode(void);

/* BatchNorm batch statistics -> affine.  training != 0: mean/var from the slots (biased var for the
 * normalisation, unbiased for running_var, torch semantics), running stats updated in place with
 * `momentum`; training == 0: affine from the running stats (stat_slots unused).  The sums are taken BEFORE the bias
 * add (better conditioned); `bias` (may be NULL) is added back to the mean here.  Outputs scale[c] = gamma*invstd,
 * shift[c] = beta - mean*scale, and mean / invstd (saved for backward). */
int p2c_bn_finalize_f32(const double *stat_slots, int C, long long count, const float *bias, const float *gamma,
                        const float *beta, float eps, float momentum, int training, float *running_mean, float *running_var,
                        float *scale, float *shift, float *mean, float *invstd, void *stream);

/* Eval-mode affine of EVERY BatchNorm of a forward in one launch (the training == 0 branch of p2c_bn_finalize_f32, same arithmetic; replaces
 * torch.nn.functional.batch_norm(training=False)'s per-layer rsqrt / mul chain behind models/pointnet_util.py:187,266 and
 * models/pointnet_extrusion.py:59).  rows: DEVICE array of n_rows 48-byte records
 *   { const float *gamma, *beta, *running_mean, *running_var; float *st; int32 C; float eps; }
 * st is (4, C) row-major: scale, shift, mean, invstd.  The live parameters are read at run time (HIP-graph safe under in-place updates). */
int p2c_bn_eval_affine_batch_f32(const void *rows, int n_rows, void *stream);

/* Z = relu(scale*Y + shift), materialised (only where a consumer needs the post-activation tensor) */
int p2c_bn_relu_apply_f32(const float *Y, int ldy, const float *scale, const float *shift, int M, int C, float *Z, int ldz,
                          void *stream);

/* max over the nsample axis of relu(bn(Y)) (models/pointnet_util.py:205): Y [G*ns, C] -> out [G,C],
 * arg [G,C] (row offset j of the winner, first on ties), ywin [G,C] (optional: Y at the winner, for backward). */
int p2c_maxpool_bnrelu_f32(const float *Y, int ldy, const float *scale, const float *shift, int G, int ns, int C, float *out,
                           int ldo, int32_t *arg, float *ywin, void *stream);
/* The last layer of a set-abstraction stack with the pooling folded into the GEMM epilogue (replaces the
 * p2c_linear_fwd_f32 + p2c_maxpool_bnrelu_f32 pair, i.e. the pass that re-reads the 268-537 MB pre-BN tensor to pool it):
 *   p2c_linear_fwd_pool_f32: Y and stat slots exactly as p2c_linear_fwd_f32 with in_mode 1, plus - per 32-row half of every
 *     64-row neighbourhood and column - the largest / smallest pre-BN value and their rows: pool_max, pool_min [2*M/64, N] fp32,
 *     pool_idx [2*M/64, N] int32 (row_of_max | row_of_min << 16).  Requires p2c_linear_fwd_pool_supported(M, N, K, 1, 64)
 *     (groups of exactly 64 rows, K in {64,128}, N in {128,256}, M >= 8192, M % 64 == 0); Y may be NULL (not stored: the layer's
 *     backward through p2c_linear_bwd_pool_alg_f32 needs no Y);
 *   p2c_pool_select_f32: once the layer's BatchNorm affine exists, out [G,C] = max over the group of relu(scale*y + shift)
 *     (the largest pre-BN value for scale >= 0, the smallest for scale < 0), arg / ywin as p2c_maxpool_bnrelu_f32.
 *     Ties: lowest row among equal PRE-BN values (models/pointnet_util.py:205 leaves ties to torch.max). */
int p2c_linear_fwd_pool_supported(int M, int N, int K, int in_mode, int ns);
int p2c_linear_fwd_pool_f32(const float *X, int ldx, const float *W, int ldw, const float *bias, float *Y, int ldy, int M, int N,
                            int K, const float *in_scale, const float *in_shift, double *stat_slots, float *pool_max,
                            float *pool_min, int32_t *pool_idx, void *stream);
int p2c_pool_select_f32(const float *pool_max, const float *pool_min, const int32_t *pool_idx, const float *scale,
                        const float *shift, int G, int C, float *out, int ldo, int32_t *arg, float *ywin, void *stream);
/* Adam (torch.optim.Adam defaults: no weight decay, no amsgrad; the optimiser of train_Point2Cyl_without_sketch.py:204) over a list
 * of fp32 tensors in ONE launch with 1024-element work items.  table [n][4] int64 = (param, grad, exp_avg, exp_avg_sq) device
 * pointers, numel [n] int64, chunks [n_chunks][2] int32 = (tensor, first element / 1024), all in device memory; `step` = the
 * 1-based step count (bias corrections are formed on the host in double). */
int p2c_adam_multi_f32(const long long *table, const long long *numel, const int32_t *chunks, int n_chunks, float lr, float beta1,
                       float beta2, float eps, long long step, void *stream);
/* dZ [G*ns, C] (dense, zero except the winners) from dOut [G,C] */
int p2c_maxpool_bwd_f32(const float *dout, int ldo, const int32_t *arg, int G, int ns, int C, float *dZ, int ldz, void *stream);

/* Backward of relu(bn(Y)) in training mode, step 1: per-channel sums over rows of
 *   g = dZ * [scale*Y+shift > 0],   s1 = sum g,   s2 = sum g * (Y-mean)*invstd
 * then step 2 (finalize): dgamma = s2, dbeta = s1 and the coefficients that let a GEMM rebuild
 *   dY = gs*g + q*Y + p      (gs = gamma*invstd, q = -gs*invstd*s2/M, p = -gs*s1/M - q*mean)
 * coef_out [5,C] = {scale, shift, gs, q, p}.  slots: zeroed p2c_stat_slots_bytes(C) bytes. */
int p2c_bn_relu_bwd_stats_f32(const float *dZ, int lddz, const float *Y, int ldy, const float *scale, const float *shift,
                              const float *mean, const float *invstd, const float *gamma, int M, int C, float *dgamma,
                              float *dbeta, float *coef_out, double *slots, void *stream);

/* The same reduction for the layer that feeds the max-pool, WITHOUT materialising dZ: the sums run over the
 * pooled gradient dout [G,C] and the winners' pre-BN values ywin [G,C] (from p2c_maxpool_bnrelu_f32) only.
 * stat = [scale|shift|mean|invstd] x C of that layer.  slots: zeroed p2c_stat_slots_bytes(C) bytes. */
int p2c_maxpool_bn_bwd_stats_f32(const float *dout, int ldo, const float *ywin, const float *stat, const float *gamma, int G,
                                 int ns, int C, float *dgamma, float *dbeta, float *coef_out, double *slots, void *stream);

/* grad_mode 0: dY = G (the tensor passed as dZ is already dY; Yfwd/coef unused)
 * grad_mode 1: dY = gs*(dZ*[scale*Yfwd+shift>0]) + q*Yfwd + p   with coef [5,Co] from the stats call above
 * grad_mode 2: as 1 with dZ[m,c] = (pool_arg[m/pool_ns, c] == m%pool_ns) ? dZ_pooled[m/pool_ns, c] : 0, i.e. the
 *              `dZ` argument is the max-pool's upstream gradient [G,Co] (lddz its leading dim) */

/* dX[m,ci] = sum_co dY[m,co] * W[co,ci]   (optionally multiplied by out_mask[m,ci]*out_mask_scale: the
 * dropout in front of the layer).  dX [M,K] (lddx).
 * Fused reduction for the layer BELOW (optional, bwd_partials != NULL): dX is that layer's dZ; with its saved
 * pre-BN output Yprev [M,K] and prev_stat = [scale|shift|mean|invstd] x K the epilogue also accumulates
 * s1 = sum g, s2 = sum g*xhat (g = dX*[scale*Yprev+shift > 0]) into the slots bwd_partials (zeroed,
 * p2c_stat_slots_bytes(K)), to be finished by p2c_bn_bwd_finalize_f32. */
int p2c_linear_bwd_data_f32(const float *dZ, int lddz, const float *Yfwd, int ldy, int grad_mode, const float *coef,
                            const float *W, int ldw, float *dX, int lddx, int M, int N, int K, const uint8_t *out_mask,
                            int ldmask, float out_mask_scale, const float *Yprev, int ldyp, const float *prev_stat,
                            double *bwd_partials, const int32_t *pool_arg, int pool_ns, void *stream);
/* finish the fused reduction: dgamma, dbeta and coef [5,C] (see p2c_bn_relu_bwd_stats_f32).  stat = [scale|shift|mean|invstd] x C */
int p2c_bn_bwd_finalize_f32(const double *slots, int C, long long M, const float *stat, const float *gamma, float *dgamma,
                            float *dbeta, float *coef_out, void *stream);

/* Backward of a NARROW linear layer (Co <= 32 outputs, 128 inputs, >= 4096 rows: the per-point heads, models/pointnet_extrusion.py:56-61)
 * whose input is dropout(relu(bn(Y))), in one pass over dZ and Y (csrc/heads.hip): dW into 8 per-XCD copies (accumulated, zero them),
 * dbias (accumulated), dX, and the BatchNorm-backward sums of the layer below into `partials` (fp64 slot rows, accumulated).
 * stat = [scale|shift|mean|invstd] x 128 of that BatchNorm; seed NULL: no dropout, else the int64 counter of the forward's hashed mask
 * with keep-scale dscale.  Finish with p2c_bn_bwd_finalize_sum_f32.  p2c_linear_bwd_narrow_supported(M, Co, Ci, in_mode) -> 1/0. */
int p2c_linear_bwd_narrow_supported(int M, int Co, int Ci, int in_mode);
int p2c_linear_bwd_narrow_f32(const float *dZ, int lddz, const float *Y, int ldy, const float *stat, const void *seed, float dscale,
                              const float *W, int ldw, float *dX, int lddx, float *dW8, int lddw, long long dw_slot_stride, float *dbias,
                              double *partials, int M, int Co, int Ci, void *stream);

/* out[0..n) = sum over `copies` matrices src + c*stride (the per-XCD copies of a weight gradient the fused backward accumulates into),
 * alone or in the same launch as p2c_bn_bwd_finalize_f32 (same arguments first). */
int p2c_sum_copies_f32(const float *src, long long stride, int copies, float *out, long long n, void *stream);
int p2c_bn_bwd_finalize_sum_f32(const double *slots, int C, long long M, const float *stat, const float *gamma, float *dgamma, float *dbeta,
                                float *coef_out, const float *src, long long stride, int copies, float *out, long long n, void *stream);

/* dW[co,ci] += sum_m dY[m,co] * act_in(X)[m,ci];  dbias[co] += sum_m dY[m,co] (dbias may be NULL).
 * dW / dbias must be zero-initialised by the caller (the kernel splits the row range over workgroups and
 * accumulates with fp32 atomics).  With dw_slot_stride != 0 the atomics are spread over 8 copies of dW
 * (copy s at dW + s*dw_slot_stride elements) that the caller sums: 8x less contention per address. */
int p2c_linear_bwd_weight_f32(const float *dZ, int lddz, const float *Yfwd, int ldy, int grad_mode, const float *coef,
                              const float *X, int ldx, int in_mode, const float *in_scale, const float *in_shift,
                              const uint8_t *drop_mask, int ldmask, float drop_scale, float *dW, int lddw,
                              long long dw_slot_stride, float *dbias, int M, int N, int K, const int32_t *pool_arg, int pool_ns,
                              void *stream);

/* Both of the above in ONE launch, for layers of a few thousand rows (SA3 / FP3: pointnet_util.py:201-205, :317-319 under
 * autograd) where neither GEMM fills the chip: the workgroups of the two run side by side.  grad_mode 1 or 2, in_mode 0 or 1, no
 * dropout masks, N and K > 64; dW zeroed by the caller (atomic accumulation, one copy). */
int p2c_linear_bwd_both_f32(const float *dZ, int lddz, const float *Yfwd, int ldy, int grad_mode, const float *coef,
                            const int32_t *pool_arg, int pool_ns, const float *X, int ldx, int in_mode, const float *in_scale,
                            const float *in_shift, const float *W, int ldw, float *dX, int lddx, const float *Yprev, int ldyp,
                            const float *prev_stat, double *bwd_partials, float *dW, int lddw, int M, int N, int K, void *stream);

/* One-pass backward of a narrow layer (Co, Ci in {64,128}; in_mode 0/1): dX, dW, dbias and the fused reduction
 * for the layer below from a single stream over (dZ, Yfwd, X) -- see csrc/bwd_fused.hip.  Same argument meaning as
 * the two entry points above; dX may be NULL (then prev_stat/bwd_partials must be NULL too).
 * bwd_partials: the zeroed fp64 slots of the layer below (p2c_stat_slots_bytes(Ci)).
 * dW is accumulated into 8 copies (one per XCD, copy s at dW + s*dw_slot_stride elements, all zero-initialised by the
 * caller, who sums them); dw_slot_stride = 0 selects a single copy.  grad_mode 2 needs pool_ns >= 32, pool_ns % 16 == 0.
 * p2c_linear_bwd_fused_supported: 0 = use the two generic entry points, 1 = supported, 2 = supported for the grouped
 * layer [128 features | 4 trailing columns] (Ci = 132, in_mode 0): dW covers all 132 columns but dX[:, 128:132] is NOT
 * written (the trailing columns are the relative coordinates, which receive no gradient). */
int p2c_linear_bwd_fused_supported(int Co, int Ci, int in_mode);
int p2c_linear_bwd_fused_parts(int M, int Ci);
int p2c_linear_bwd_fused_f32(const float *dZ, int lddz, const float *Yfwd, int ldy, int grad_mode, const float *coef,
                             const int32_t *pool_arg, int pool_ns, const float *X, int ldx, int in_mode, const float *in_scale,
                             const float *in_shift, const float *W, int ldw, float *dX, int lddx, float *dW, int lddw,
                             long long dw_slot_stride, float *dbias, const float *prev_stat, double *bwd_partials, int M, int Co,
                             int Ci, void *stream);

/* n independent 2-D fp32 copies in one launch.  table (device memory): n entries of
 *   struct { const float *src; float *dst; int rows, cols, ld_src, ld_dst; }   (32 bytes each)
 * The host mirror stages every padded / re-ordered / column-sliced weight operand of a step with it (point2cyl_amd/ops.py WeightStage). */
int p2c_copy2d_batch_f32(const void *table, int n, void *stream);
/* the same launch also advances device-resident 64-bit counters (BatchNorm num_batches_tracked += 1, the dropout hash seed += stride):
 * counters (device memory): n_counters entries of struct { long long *ptr; long long inc; } (16 bytes each); n or n_counters may be 0. */
int p2c_copy2d_batch_inc_f32(const void *table, int n, const void *counters, int n_counters, void *stream);

/* n flat copies of contiguous buffers in one launch: srcs / dsts / nbytes are HOST arrays (device pointers 4-byte aligned, byte counts
 * multiples of 4); the descriptors travel in the kernel arguments, so a captured graph node needs no device table.  The hand-over of the
 * prefetched geometry between two replays of the step (point2cyl_amd/graph.py). */
int p2c_copy_flat_batch(const void *const *srcs, void *const *dsts, const long long *nbytes, int n, void *stream);

/* Backward of the LAST layer of a set-abstraction stack (conv -> train-mode BatchNorm -> ReLU -> max over ns neighbours,
 * pointnet_util.py:201-205) without its pre-BatchNorm output: Y = A W^T + b is linear in the layer's input A = relu(in_scale * X + in_shift),
 * so dX = dY W = A (W^T diag(q) W) + (q*b + p) W + (gs*G) W and dW = dY^T A = (gs*G)^T A + diag(q) (W A^T A + b 1^T A) + p 1^T A, with G
 * the pooled gradient dout [M/ns, Co] at the winner rows (pool_arg, masked where coef_scale * ywin + coef_shift <= 0).  Same results as
 * p2c_linear_bwd_fused_f32 with grad_mode 2 (dX [M, Ci], the ReLU + BatchNorm-backward sums of the layer below in bwd_partials, dW
 * [Co, Ci]) from half the HBM traffic; the forward need not store Y for this layer.  coef [5][Co] as p2c_maxpool_bn_bwd_stats_f32 writes
 * it; ws: 16-byte aligned scratch of p2c_linear_bwd_pool_alg_ws_bytes(Co, Ci) bytes (no initialisation needed).  Four launches: Q and r,
 * the main kernel, the fp64 sum of the workgroups' partial sums, the assembly of dW.  Shapes: see p2c_linear_bwd_pool_alg_supported. */
int p2c_linear_bwd_pool_alg_supported(int M, int Co, int Ci, int ns);
size_t p2c_linear_bwd_pool_alg_ws_bytes(int Co, int Ci);
int p2c_linear_bwd_pool_alg_f32(const float *dout, int lddo, const float *ywin, const int32_t *pool_arg, const float *coef, const float *X, int ldx,
                                const float *in_scale, const float *in_shift, const float *W, int ldw, const float *bias, float *dX, int lddx,
                                const float *prev_stat, double *bwd_partials, void *ws, float *dW, int lddw, int M, int Co, int Ci, int ns,
                                void *stream);

/* ---------------------------------------------------------------------------------------------
 * Extrusion-cylinder fitting  (data_utils.py:99-177, :253-266, :1650-1730; eval.py:409-436)
 * ------------------------------------------------------------------------------------------- */

/* estimate_extrusion_axis: per (b,k) the 3x3 matrices  BtB = sum_n wb^2 x x^T, CtC = sum_n wc^2 x x^T,
 * M = BtB/sb^2 - CtC/sc^2 (sb = sc = 1 unless normalize: sqrt(#gt barrel/base points of segment k)+1,
 * data_utils.py:133-160), then the eigenvector of the smallest (signed) eigenvalue of M (:170-171).
 * X [B,N,3] normals, Wb/Wc [B,N,K]; bb_gt / inst_gt [B,N] int64 only when normalize != 0.
 * axis_out [B,K,3] (sign convention: largest-magnitude component positive);
 * eig_out [B,K,12] = {lambda0..2 ascending, v1 (3), v2 (3), 1/sb^2, 1/sc^2, sign} saved for backward (may be NULL).
 * axis64_out [B,K,3] double (may be NULL): the same unit vector before its rounding to fp32 - the scatter sums and the Jacobi
 * sweeps run in fp64, and eval.py's axis-angle metric (:398-405, acos next to its clamp) is evaluated on this copy.
 * K <= 16. */
int p2c_extrusion_axis_f32(const float *X, const float *Wb, const float *Wc, const int64_t *bb_gt, const int64_t *inst_gt,
                           int normalize, int B, int N, int K, float *axis_out, float *eig_out, double *axis64_out, void *stream);
/* backward: d axis [B,K,3] -> dX [B,N,3], dWb, dWc [B,N,K] */
int p2c_extrusion_axis_bwd_f32(const float *daxis, const float *axis, const float *eig, const float *X, const float *Wb,
                               const float *Wc, int B, int N, int K, float *dX, float *dWb, float *dWc, void *stream);

/* estimate_extrusion_centers: c[b,k,:] = (1/N) sum_n W[b,n,k] * P[b,n,:]    (data_utils.py:253-266) */
int p2c_extrusion_centers_f32(const float *W, const float *P, int B, int N, int K, float *centers_out, void *stream);
int p2c_extrusion_centers_bwd_f32(const float *dcenters, const float *P, int B, int N, int K, float *dW, void *stream);

/* hard per-segment centroids (eval.py:409-436): mean of points whose label == k; found[b,k] = count > 1 */
int p2c_segment_centroids_f32(const float *P, const int64_t *label, int B, int N, int K, float *centroids_out,
                              float *found_out, void *stream);

/* get_extrusion_extents (data_utils.py:1650-1730): min/max of (p - c).a over the sampled barrel points.
 * rand_idx [B,K,S] int64 = the reference's torch.randint(0, n_barrel(b,k), (S,)) draws (:1696) made by the
 * caller (ignored where the segment is not found).  extents_out [K,B,2], found_out [B,K].
 * ws: p2c_extents_ws_bytes(B,K) bytes. */
size_t p2c_extents_ws_bytes(int B, int K);
int p2c_extrusion_extents_f32(const float *P, const int64_t *seg, const int64_t *bb, const float *axes, const float *centers,
                              const int64_t *rand_idx, int B, int N, int K, int S, float *extents_out, float *found_out,
                              void *ws, void *stream);

/* The fitting-only chain of eval.py on pre-segmented clouds in one pass over each cloud (BASELINE configs[3]):
 * estimate_extrusion_axis (eval.py:397 -> data_utils.py:99-177) -> hard per-segment centroids (eval.py:409-436) ->
 * get_extrusion_extents (data_utils.py:1650-1730) on the axes and centroids just fitted.  Same outputs as the three entry points above
 * called in that order: axis_out [B,K,3], centroids_out [B,K,3] + cfound_out [B,K], extents_out [K,B,2] + found_out [B,K];
 * axis64_out [B,K,3] double (may be NULL) as for p2c_extrusion_axis_f32 (the extents are taken along the fp32 axis_out).
 * bb_gt / inst_gt are the per-point base-barrel and segment labels (both always read: the barrel lists and the centroids need them).
 * p2c_fit_fused_supported(N, K, S) says whether the shape fits (K in {1,2,4,8}, the cloud within the LDS); ws as for the extents. */
int p2c_fit_fused_supported(int N, int K, int S);
int p2c_fit_fused_f32(const float *X, const float *Wb, const float *Wc, const int64_t *bb_gt, const int64_t *inst_gt, int normalize,
                      const float *P, const int64_t *rand_idx, int B, int N, int K, int S, float *axis_out, float *centroids_out,
                      float *cfound_out, float *extents_out, float *found_out, double *axis64_out, void *ws, void *stream);

/* sketch_implicit_projection / sketch_implicit_projection2 (data_utils.py:1014-1146, :1149-1282) and, with all_points = 1
 * (S == N, seg / bb / rand_idx unused), sketch_implicit_projection3 (:1284-1417): the S sampled barrel points and normals
 * of every segment, turned by the matrix that takes the segment's axis onto z (the reference's construction through
 * torchgeometry's angle_axis_to_rotation_matrix, restated), x and y kept, points centred on the turned centre.
 * rand_idx [B,K,S] int64 = the reference's torch.randint(0, n_barrel(b,k), (S,)) draws (:1064), made by the caller.
 * P_proj, X_proj [K,B,S,2] (8-byte aligned), scales_out [K,B] (largest radius; 1 where not found), found_out [B,K].
 * ws: p2c_extents_ws_bytes(B,K) bytes. */
int p2c_sketch_projection_f32(const float *P, const float *X, const int64_t *seg, const int64_t *bb, const float *axes,
                              const float *centers, const int64_t *rand_idx, int B, int N, int K, int S, int all_points,
                              float *P_proj, float *X_proj, float *scales_out, float *found_out, void *ws, void *stream);

/* Head post-processing for the fitting losses (train_Point2Cyl_without_sketch.py:247-265, :319-325, :342-344): from heads [B*N, ld]
 * (3 normal components at column xoff, 2K segmentation logits at woff) the unit normals X [B,N,3] (F.normalize, eps 1e-12) and the
 * softmaxed barrel / base probabilities in matched order Wb, Wc [B,N,K] (= gather(W_2K[:, :, 0::2] / [1::2], 2, matching_indices)),
 * and the backward of that map: dheads [B*N, ldd] (every column written, zero outside the two blocks) from dX, dWb, dWc (NULL = 0). */
int p2c_head_post_f32(const float *heads, int ld, int xoff, int woff, const int64_t *match, int B, int N, int K, float *X, float *Wb,
                      float *Wc, void *stream);
int p2c_head_post_bwd_f32(const float *heads, int ld, int xoff, int woff, const int64_t *match, int B, int N, int K, const float *dX,
                          const float *dWb, const float *dWc, float *dheads, int ldd, void *stream);

/* scipy.optimize.linear_sum_assignment (minimising; the reference's call site is losses.py:43) for a batch of small dense problems:
 * cost [n_problems, nr, nc] fp64 (device), nr <= nc <= 15 -> col4row_out [n_problems, nr] int32.  solver 1: one wave per problem
 * (the solver inside the matching kernels); solver 0: the single-lane restatement of the same algorithm (cross-check). */
int p2c_linear_sum_assignment_f64(const double *cost, int n_problems, int nr, int nc, int32_t *col4row_out, int solver, void *stream);

/* nn.Softplus(beta) of the sketch branch's implicit decoder (IGR/network.py:58-59, :80-82) and the derivatives its double backward
 * needs (train_Point2Cyl.py:619-646 differentiate the decoder w.r.t. its input with create_graph), one pass each over n elements
 * (16-byte aligned).  s = sigmoid(beta z); beta z > threshold is the linear region as in torch (threshold 20).
 *   fwd:      h  = softplus(z)
 *   bwd:      out = u * s(z)
 *   bwd_bwd:  du = g * s(z),  dz = g * u * beta * s(z) (1 - s(z)) */
int p2c_softplus_fwd_f32(const float *z, float *h, long long n, float beta, float threshold, void *stream);
int p2c_softplus_bwd_f32(const float *u, const float *z, float *out, long long n, float beta, float threshold, void *stream);
int p2c_softplus_bwd_bwd_f32(const float *g, const float *u, const float *z, float *du, float *dz, long long n, float beta, float threshold,
                             void *stream);
/* The same derivative fused into the product it follows: dX[M,K] = (dZ[M,N] . W[N,K]) * s(Z[M,K]) (the data gradient of a linear
 * layer fed by softplus(Z); IGR/network.py:80-82 under autograd), and the backward of THAT from its own output a:
 * t = g * s(z), dz = g * a * beta * (1 - s(z)). */
int p2c_linear_bwd_data_sig_f32(const float *dZ, int lddz, const float *W, int ldw, const float *Z, int ldz, float beta, float threshold,
                                float *dX, int lddx, int M, int N, int K, void *stream);
int p2c_softplus_sig_bwd_f32(const float *g, const float *a, const float *z, float *t, float *dz, long long n, float beta, float threshold,
                             void *stream);
/* Row-structured variants for the two ends of the decoder (one output unit; an input gradient used in two columns): z, q, e, Ein and the
 * matrix outputs are [M, K] contiguous, K % 4 == 0, w / wa / wb are [K] rows of the layer's weight.
 *   p2c_softplus_dot_f32:           out[m] = softplus(z[m,:]) . w + bias[0]                     (IGR/network.py:84-86, last layer forward)
 *   p2c_softplus_row_bwd_f32:       o[m,c] = (g ? g[m*ldg] : 1) * w[c] * sigmoid(beta z[m,c]) + (q ? q[m,c] : 0)
 *   p2c_softplus_sig_bwd_rank2_f32: E = (Ein ? Ein : 0) + ga[m*ldg] wa[c] + ga[m*ldg+1] wb[c];  t = E s(z),  dz = E e beta (1 - s(z)) */
int p2c_softplus_dot_f32(const float *z, const float *w, const float *bias, float *out, long long M, int K, float beta, float threshold, void *stream);
int p2c_softplus_row_bwd_f32(const float *g, int ldg, const float *w, const float *z, const float *q, float *o, long long M, int K, float beta,
                             float threshold, void *stream);
int p2c_softplus_sig_bwd_rank2_f32(const float *ga, int ldg, const float *wa, const float *wb, const float *Ein, const float *e, const float *z,
                                   float *t, float *dz, long long M, int K, float beta, float threshold, void *stream);

/* The same two products for the decoder's LARGE shapes (p2c_linear_big_supported: M >= 16384 rows, N, K >= 128, multiples of 4) on 128 x 256
 * tiles with W split into its three bf16 planes ONCE per call (csrc/gemm_big.hip) instead of once per workgroup: same operands, same results
 * to the fp32 contract of the split products (section "bf16 x 3" above), plus a caller-provided workspace `ws` of
 * p2c_linear_big_ws_bytes(N, K) bytes (16-byte aligned) that holds the split image.  p2c_linear_bwd_data_big_f32 with Z == NULL is the plain
 * data gradient dX = dZ . W.  IGR/network.py:20-92 (the eight 512-wide layers), train_Point2Cyl.py:608-648 (evaluated forward, backward
 * and backward-of-backward by the with-sketch step). */
int p2c_linear_big_supported(int M, int N, int K);
size_t p2c_linear_big_ws_bytes(int N, int K);
int p2c_linear_fwd_big_f32(const float *X, int ldx, const float *W, int ldw, const float *bias, float *Y, int ldy, int M, int N, int K, void *ws,
                           void *stream);
int p2c_linear_bwd_data_big_f32(const float *dZ, int lddz, const float *W, int ldw, const float *Z, int ldz, float beta, float threshold,
                                float *dX, int lddx, int M, int N, int K, void *ws, void *stream);
/* the same two products with an addend in the epilogue (ldadd % 4 == 0, 16-byte aligned, may alias the output):
 *   p2c_linear_fwd_big_add_f32:      Y  = X . W^T + bias + add          (the decoder's skip layer as two products, IGR/network.py:75-76)
 *   p2c_linear_bwd_data_big_add_f32: dX = (dZ . W) * sigmoid(beta Z) + add   (the two gradients a pre-activation receives in the double
 *   backward of train_Point2Cyl.py:608-648, summed in the product's epilogue instead of by a separate pass over 0.5 GB) */
int p2c_linear_fwd_big_add_f32(const float *X, int ldx, const float *W, int ldw, const float *bias, const float *add, int ldadd, float *Y,
                               int ldy, int M, int N, int K, void *ws, void *stream);
/* Y = softplus(X . W^T + bias [+ add], beta, threshold) (IGR/network.py:58-59, :83-84): the layer's activation in the product's epilogue, for
 * inference - the pre-activation a backward would need is not kept.  add may be NULL. */
int p2c_linear_fwd_big_sp_f32(const float *X, int ldx, const float *W, int ldw, const float *bias, const float *add, int ldadd, float beta,
                              float threshold, float *Y, int ldy, int M, int N, int K, void *ws, void *stream);
int p2c_linear_bwd_data_big_add_f32(const float *dZ, int lddz, const float *W, int ldw, const float *Z, int ldz, float beta, float threshold,
                                    const float *add, int ldadd, float *dX, int lddx, int M, int N, int K, void *ws, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Hungarian matching on the device (losses.py:22-52; scipy.optimize.linear_sum_assignment restated)
 * W [B,N,K] soft or hard segmentation, I_gt [B,N] int64 (may contain -1).  match_out [B,K] int64,
 * mask_out [B,K] uint8.  K <= 15. */
int p2c_hungarian_f32(const float *W, const int64_t *I_gt, int B, int N, int K, int64_t *match_out, uint8_t *mask_out,
                      void *stream);
/* same, from the raw head output [B*N, ld] whose 2K segmentation logits start at column woff: W = pairwise sums of
 * softmax(logits) (train_Point2Cyl_without_sketch.py:254-265) formed on the fly.
 * ws: optional scratch of p2c_hungarian_ws_bytes(B) bytes (no initialisation needed): with it (and K == 8) a cloud is
 * reduced by 8 workgroups and a second small kernel solves the assignments.  NULL = one workgroup per cloud. */
size_t p2c_hungarian_ws_bytes(int B);
int p2c_hungarian_logits_f32(const float *heads, int ld, int woff, const int64_t *I_gt, int B, int N, int K, int64_t *match_out,
                             uint8_t *mask_out, void *ws, void *stream);

/* ---------------------------------------------------------------------------------------------
 * The training losses fused (losses.py:90-143, :317-351 with collapse=True; train…:247-307):
 * normal loss mean(1-|X.n_gt|), Hungarian-matched mIoU loss, base/barrel weighted cross-entropy, forward AND gradient
 * w.r.t. the head output in two passes.  heads [B*N, ld]: normals at columns [xoff,xoff+3), 2K logits at [woff,woff+2K).
 * out[4] = {total, normal, miou, bb}; dheads [B*N, ld] = d total / d heads.  ws: zeroed p2c_seg_losses_ws_bytes(B,K).  K = 1 ... 8. */
size_t p2c_seg_losses_ws_bytes(int B, int K);
int p2c_seg_losses_f32(const float *heads, int ld, int xoff, int woff, const float *normals_gt, const int64_t *I_gt,
                       const int64_t *bb_gt, const int64_t *match, const uint8_t *mask, int B, int N, int K, float w_seg,
                       float w_normal, float w_bb, float *out, float *dheads, void *ws, void *stream);
/* dheads may be NULL in p2c_seg_losses_f32 (forward launches only).  The gradient pass on its own, scaled by the upstream gradient
 * *gscale (device scalar d / d total, NULL = 1), from what the forward call left in the same ws: */
int p2c_seg_losses_grad_f32(const float *heads, int ld, int xoff, int woff, const float *normals_gt, const int64_t *I_gt,
                            const int64_t *bb_gt, const int64_t *match, const uint8_t *mask, int B, int N, int K, float w_seg,
                            float w_normal, float w_bb, const float *gscale, float *dheads, void *ws, void *stream);

/* The two fitting terms of the full loss set on their [B,K,3] operands, forward and gradient in one launch (one workgroup):
 * out2[0] = w_ext * mean_b masked-mean_k (1 - |E_AX . gt_axes|)   (losses.py:127-143 angle_diff=False, :83-88; train_Point2Cyl_without_sketch.py:326-332),
 * out2[1] = w_center * mean_b masked-mean_k |centers - gt_centers|^2   (:342-353).  mask [B,K] bytes = k < instances of cloud b (p2c_hungarian_f32's
 * mask; a cloud without instances contributes 0).  dE / dC [B,K,3] = d out2[0] / d E_AX, d out2[1] / d centers (NULL: not wanted).  E_AX or centers
 * NULL: that term is off (0).  K <= 256. */
int p2c_fit_terms_f32(const float *E_AX, const float *gt_axes, const float *centers, const float *gt_centers, const uint8_t *mask,
                      int B, int K, float w_ext, float w_center, float *out2, float *dE, float *dC, void *stream);

/* compute_all_losses on its own inputs (losses.py:317-351, collapse=True): W [B,N,K] softmaxed membership and X [B,N,3] unit normals as the
 * reference's trainer forms them in torch (train_Point2Cyl_without_sketch.py:246-271) - what the drop-in of that function is handed.
 * match / mask [B,K] from p2c_hungarian_f32.  out2 = {mean normal loss, mean mIoU loss}; dW [B,N,K] = d out2[1] / d W and
 * dX [B,N,3] = d out2[0] / d X (unweighted: the caller applies its multipliers).  ws: zeroed p2c_all_losses_ws_bytes(B,K).  K = 1 ... 8. */
size_t p2c_all_losses_ws_bytes(int B, int K);
int p2c_all_losses_f32(const float *W, const float *X, const float *normals_gt, const int64_t *I_gt, const int64_t *match,
                       const uint8_t *mask, int B, int N, int K, float *out2, float *dW, float *dX, void *ws, void *stream);

/* ---------------------------------------------------------------------------------------------
 * The evaluation metrics of one batch in two launches (eval.py:270-446 with its default operands: predicted normals, predicted
 * segmentation and base/barrel split; replaces F.normalize + softmax + losses.hard_W_encoding(to_null_mask=True) + hungarian_matching +
 * compute_segmentation_iou + compute_normal_difference + the base/barrel accuracy + estimate_extrusion_axis on the matched memberships +
 * the hard-centroid loop, ~90 torch kernels).  heads [B*N, ld]: raw normals at columns [xoff, xoff+3), 2K logits at [woff, woff+2K);
 * pcs / gt_normals [B,N,3]; gt_inst [B,N] int64 in [-1, K) (checked by the caller); gt_bb [B,N] FLOAT 0/1 (eval.py:257 casts it);
 * gt_axes / gt_centers [B,K,3]; normalize = --norm_eig (data_utils.py:133-160); null_thr = float(N) * 0.005 as fp32 (losses.py:62);
 * pi = the reference's TORCH_PI (float32 acos(0) * 2 as a double, losses.py:17).
 * out5 [5,B] fp64 = per cloud {mIoU, normal angle error deg, base/barrel accuracy, extrusion angle error deg, centroid difference} - the
 * rows eval.py:690-715 averages.  Optional (NULL = not wanted): match [B,K] i64, mask [B,K] u8, axis64 [B,K,3] f64 (fitted unit axes of the
 * matched segments, zeros elsewhere), cen [B,K,3] f32 + found [B,K] f32 (hard centroids; <= 1 point = not found).
 * ws: p2c_eval_metrics_ws_bytes(B, K) bytes, no initialisation needed.  K in {2, 4, 8} (p2c_eval_metrics_supported). */
size_t p2c_eval_metrics_ws_bytes(int B, int K);
int p2c_eval_metrics_supported(int K);
int p2c_eval_metrics_f32(const float *heads, int ld, int xoff, int woff, const float *pcs, const float *gt_normals, const int64_t *gt_inst,
                         const float *gt_bb, const float *gt_axes, const float *gt_centers, int normalize, float null_thr, double pi,
                         int B, int N, int K, double *out5, int64_t *match_out, uint8_t *mask_out, double *axis64_out, float *cen_out,
                         float *found_out, void *ws, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* P2C_HIP_H */
