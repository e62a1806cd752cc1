"""CPU: the C-ABI library builds, loads and exports every symbol include/p2c_hip.h declares; the product
refuses to run without a HIP device (no CPU fallback)."""
import ctypes
import os

import pytest
import torch

from point2cyl_amd import _lib, build


def test_library_builds_and_exports_every_declared_symbol():
    so = build.build()
    assert os.path.exists(so)
    L = ctypes.CDLL(so)
    declared = _lib.declared_symbols()
    assert len(declared) >= 25
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert set(_lib._SIGS) <= set(declared)
    L.p2c_abi_version.restype = ctypes.c_int
    assert L.p2c_abi_version() >= 1
    L.p2c_build_arch.restype = ctypes.c_char_p
    assert L.p2c_build_arch() == b"gfx950"


def test_stat_tile_helper_matches_header():
    L = _lib.lib()
    tm = L.p2c_linear_tile_m()
    assert tm in (64, 128)
    assert L.p2c_linear_stat_tiles(1) == 1 and L.p2c_linear_stat_tiles(tm + 1) == 2
    assert L.p2c_stat_slots_bytes(64) == 64 * 2 * 64 * 8


def test_bad_arguments_are_rejected_without_touching_the_device():
    L = _lib.lib()
    assert L.p2c_fps_f32(None, 1, 16, None, 4, None, None, None) == -1
    assert L.p2c_ball_query_f32(None, None, 1, 1, 1, 0.04, 64, None, None) == -1
    assert L.p2c_hungarian_f32(None, None, 1, 1, 8, None, None, None) == -1
    assert L.p2c_build_csr_i32(None, None, 1, 8, 1, 4, None, None, None, None) == -1
    assert L.p2c_input_moments_f32(None, 4, 10, None, None) == -1
    assert L.p2c_linear_fwd_fold0_f32(None, 4, None, None, None, None, 64, None, 64, None, None, 64, 9000, 64, None, None) == -1
    assert L.p2c_fold0_bwd_finalize_f32(None, None, 10, None, None, None, None, 64, None, None, None, None) == -1
    # sketch branch entries
    assert L.p2c_sketch_projection_f32(None, None, None, None, None, None, None, 1, 16, 8, 4, 0, None, None, None, None, None, None) == -1
    assert L.p2c_softplus_fwd_f32(None, None, 8, 100.0, 20.0, None) == -1
    assert L.p2c_softplus_bwd_bwd_f32(None, None, None, None, None, 8, 100.0, 20.0, None) == -1
    assert L.p2c_linear_bwd_data_sig_f32(None, 4, None, 4, None, 4, 100.0, 20.0, None, 4, 8, 4, 4, None) == -1
    assert L.p2c_linear_sum_assignment_f64(None, 1, 8, 8, None, 1, None) == -1
    one = ctypes.c_void_p(16)          # a non-null, 16-byte aligned dummy: shapes are validated before anything is dereferenced
    assert L.p2c_linear_sum_assignment_f64(one, 1, 9, 8, one, 1, None) == -1      # more rows than columns
    assert L.p2c_linear_sum_assignment_f64(one, 1, 8, 16, one, 1, None) == -1     # more than 15 columns
    assert L.p2c_sketch_projection_f32(one, one, None, None, one, one, None, 1, 16, 8, 4, 1, one, one, one, one, one, None) == -1   # all_points needs S == N
    assert L.p2c_linear_bwd_data_sig_f32(one, 4, one, 4, one, 4, 100.0, 20.0, one, 4, 8, 6, 4, None) == -2                          # N not a multiple of 4
    # round 3 entries
    assert L.p2c_fit_fused_f32(None, None, None, None, None, 0, None, None, 1, 8192, 8, 2048, None, None, None, None, None, None, None, None) == -1
    assert L.p2c_fit_fused_f32(one, one, one, one, one, 0, one, one, 1, 8192, 3, 2048, one, one, one, one, one, None, one, None) == -1   # K not a power of two
    assert L.p2c_fit_fused_f32(one, one, one, one, one, 0, one, one, 1, 65536, 8, 2048, one, one, one, one, one, None, one, None) == -1  # cloud larger than the LDS
    assert L.p2c_linear_bwd_narrow_f32(None, 20, None, 128, None, None, 2.0, None, 128, None, 128, None, 128, 20 * 128, None, None, 262144, 20, 128, None) == -1
    assert L.p2c_linear_bwd_narrow_f32(one, 20, one, 128, one, None, 1.0, one, 128, one, 128, one, 128, 40 * 128, None, one, 262144, 40, 128, None) == -1   # more than 32 outputs
    assert L.p2c_linear_bwd_narrow_f32(one, 20, one, 130, one, None, 1.0, one, 128, one, 128, one, 128, 20 * 128, None, one, 262144, 20, 128, None) == -2   # row stride not 16-byte aligned
    assert L.p2c_sum_copies_f32(None, 16, 8, None, 16, None) == -1
    assert L.p2c_bn_bwd_finalize_sum_f32(None, 64, 10, None, None, None, None, None, None, 16, 8, None, 16, None) == -1
    # round 4 entries
    assert L.p2c_all_losses_f32(None, None, None, None, None, None, 1, 8, 8, None, None, None, None, None) == -1
    assert L.p2c_all_losses_f32(one, one, one, one, one, one, 1, 8, 9, one, one, one, one, None) == -1                              # K outside 1 ... 8
    # round 6 entries
    assert L.p2c_eval_metrics_f32(None, 20, 0, 3, None, None, None, None, None, None, 0, 40.96, 3.1415927, 1, 8192, 8, None, None, None, None, None, None, None, None) == -1
    assert L.p2c_eval_metrics_f32(one, 20, 0, 3, one, one, one, one, one, one, 0, 40.96, 3.1415927, 1, 8192, 3, one, None, None, None, None, None, one, None) == -1   # K not 2, 4, 8
    assert L.p2c_eval_metrics_f32(one, 16, 0, 3, one, one, one, one, one, one, 0, 40.96, 3.1415927, 1, 8192, 8, one, None, None, None, None, None, one, None) == -1   # row too short for 3 + 2K
    assert L.p2c_eval_metrics_supported(8) == 1 and L.p2c_eval_metrics_supported(5) == 0 and L.p2c_eval_metrics_ws_bytes(32, 8) == 32 * 8 * (28 * 8 + 3) * 8
    assert L.p2c_seg_losses_grad_f32(None, 20, 0, 4, None, None, None, None, None, 1, 8, 8, 1.0, 1.0, 1.0, None, None, None, None) == -1
    assert L.p2c_copy2d_batch_f32(None, 4, None) == -1 and L.p2c_copy2d_batch_f32(one, 0, None) == -1
    assert L.p2c_linear_bwd_pool_alg_f32(None, 128, None, None, None, None, 64, None, None, None, 64, None, None, 64, None, None, None, None, 64, 1048576, 128, 64, 64, None) == -1
    assert L.p2c_linear_bwd_pool_alg_supported(1048576, 128, 64, 64) == 1 and L.p2c_linear_bwd_pool_alg_supported(1048576, 128, 128, 64) == 0
    assert L.p2c_linear_big_supported(262144, 512, 512) == 1 and L.p2c_linear_big_supported(2048, 512, 512) == 0 and L.p2c_linear_big_supported(262144, 510, 512) == 0
    assert L.p2c_linear_big_ws_bytes(512, 512) == 3 * 2 * 512 * 512 and L.p2c_linear_big_ws_bytes(254 + 2, 260) == 3 * 2 * 512 * 256
    assert L.p2c_linear_fwd_big_f32(None, 512, None, 512, None, None, 512, 262144, 512, 512, None, None) == -1
    assert L.p2c_linear_fwd_big_f32(one, 512, one, 512, None, one, 512, 1024, 512, 512, one, None) == -1                            # too few rows for this route
    assert L.p2c_linear_fwd_big_f32(one, 514, one, 512, None, one, 512, 262144, 512, 512, one, None) == -2                          # row stride not 16-byte aligned
    assert L.p2c_linear_bwd_data_big_f32(one, 512, one, 512, one, 512, 0.0, 20.0, one, 512, 262144, 512, 512, one, None) == -1      # softplus derivative needs beta > 0
    # round 5 entries
    assert L.p2c_fit_fused_f32(one, one, None, one, one, 0, one, one, 1, 8192, 8, 2048, one, one, one, one, one, None, one, None) == -1   # Wb without Wc (both NULL = labels-implied)
    assert L.p2c_linear_fwd_big_add_f32(one, 512, one, 512, None, None, 512, one, 512, 262144, 512, 512, one, None) == -1               # addend missing
    assert L.p2c_linear_fwd_big_add_f32(one, 512, one, 512, None, one, 514, one, 512, 262144, 512, 512, one, None) == -2               # addend stride not 16-byte aligned
    assert L.p2c_linear_bwd_data_big_add_f32(one, 512, one, 512, None, 0, 0.0, 20.0, None, 512, one, 512, 262144, 512, 512, one, None) == -1
    assert L.p2c_softplus_dot_f32(None, None, None, None, 8, 512, 100.0, 20.0, None) == -1
    assert L.p2c_softplus_dot_f32(one, one, None, one, 8, 510, 100.0, 20.0, None) == -1                                             # K not a multiple of 4
    assert L.p2c_softplus_row_bwd_f32(None, 1, one, one, None, None, 8, 512, 100.0, 20.0, None) == -1
    assert L.p2c_softplus_sig_bwd_rank2_f32(None, 2, one, one, None, one, one, one, one, 8, 512, 100.0, 20.0, None) == -1
    assert L.p2c_copy2d_batch_inc_f32(None, 0, None, 0, None) == -1 and L.p2c_copy2d_batch_inc_f32(None, 2, one, 1, None) == -1
    assert L.p2c_linear_fwd_big_sp_f32(one, 512, one, 512, None, None, 0, 0.0, 20.0, one, 512, 262144, 512, 512, one, None) == -1      # beta must be positive
    assert L.p2c_linear_fwd_big_sp_f32(one, 512, one, 512, None, one, 514, 100.0, 20.0, one, 512, 262144, 512, 512, one, None) == -2   # addend stride not 16-byte aligned
    assert L.p2c_fit_terms_f32(one, one, one, one, None, 4, 8, 1.0, 1.0, one, one, one, None) == -1                                   # no mask
    assert L.p2c_fit_terms_f32(one, one, one, one, one, 4, 300, 1.0, 1.0, one, one, one, None) == -1                                  # K above 256
    assert L.p2c_fit_terms_f32(one, None, one, one, one, 4, 8, 1.0, 1.0, one, one, one, None) == -1                                   # axes without their ground truth
    assert L.p2c_three_interp_skip_f32(one, 128, one, one, 1, 64, 16, 128, None, 128, 128, one, 256, 256, None) == -1               # skip block missing
    assert L.p2c_three_interp_skip_f32(one, 128, one, one, 1, 64, 16, 128, one, 128, 128, one, 256, 200, None) == -1                # width smaller than skip + interpolated
    assert L.p2c_fold0_bwd_finalize_sum_f32(one, one, 10, one, None, one, one, 64, one, one, one, 5, one, 16, 8, one, 16, None) == -1    # dW0 leading dimension outside 1..4
    assert L.p2c_group_weight_grad_f32(None, 128, None, 128, 128, 128, None, None) == -1


def test_shape_queries_describe_the_kernel_coverage():
    L = _lib.lib()
    # persistent forward: long narrow layers only; the grouped 128+4 layer included, byte masks and deep layers excluded
    assert L.p2c_linear_fwd_pp_supported(262144, 128, 128, 1) == 1
    assert L.p2c_linear_fwd_pp_supported(262144, 128, 132, 0) == 1
    assert L.p2c_linear_fwd_pp_supported(262144, 128, 128, 2) == 0
    assert L.p2c_linear_fwd_pp_supported(4096, 1024, 512, 1) == 0
    # fused backward: Co, Ci in {64,128}; 2 = the grouped layer (no dX for the 4 trailing columns)
    assert L.p2c_linear_bwd_fused_supported(128, 128, 1) == 1
    assert L.p2c_linear_bwd_fused_supported(128, 132, 0) == 2
    assert L.p2c_linear_bwd_fused_supported(256, 128, 1) == 3        # two passes over 128 output channels each
    assert L.p2c_linear_bwd_fused_supported(256, 256, 1) == 0
    # one-pass heads backward: up to 32 outputs on 128 inputs, from 4096 rows, input = relu(bn) with or without the hashed dropout
    assert L.p2c_linear_bwd_narrow_supported(262144, 20, 128, 3) == 1 and L.p2c_linear_bwd_narrow_supported(262144, 20, 128, 1) == 1
    assert L.p2c_linear_bwd_narrow_supported(262144, 20, 128, 2) == 0 and L.p2c_linear_bwd_narrow_supported(262144, 20, 256, 1) == 0
    assert L.p2c_linear_bwd_narrow_supported(1024, 20, 128, 1) == 0
    # one-pass fitting: K a power of two <= 8, the cloud (points, keys, lists) within the LDS
    assert L.p2c_fit_fused_supported(8192, 8, 2048) == 1 and L.p2c_fit_fused_supported(1024, 4, 100) == 1
    assert L.p2c_fit_fused_supported(8192, 16, 2048) == 0 and L.p2c_fit_fused_supported(16384, 8, 2048) == 0 and L.p2c_fit_fused_supported(8192, 6, 64) == 0
    # matrix-pipe mode switch (no device needed: it only selects kernels)
    assert L.p2c_get_mfma_mode() in (0, 1)
    old = L.p2c_set_mfma_mode(0)
    assert L.p2c_get_mfma_mode() == 0
    L.p2c_set_mfma_mode(old)
    assert L.p2c_get_mfma_mode() == old


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_fails_loudly_on_cpu_tensors():
    from point2cyl_amd import ops
    from point2cyl_amd.backbone import backbone
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.fps(torch.zeros(1, 16, 3), 4, torch.zeros(1, dtype=torch.long))
    with pytest.raises(RuntimeError, match="no CPU path"):
        backbone(output_sizes=[3, 16])(torch.zeros(1, 64, 3))
    from point2cyl_amd import fitting
    from point2cyl_amd.implicit import ImplicitNet
    from point2cyl_amd.sketch import PointNetEncoder
    with pytest.raises(RuntimeError, match="no CPU path"):
        PointNetEncoder(8, 2, with_normals=True)(torch.zeros(2, 16, 4))
    with pytest.raises(RuntimeError, match="no CPU path"):
        ImplicitNet(d_in=6, dims=[8] * 8, skip_in=[4])(torch.zeros(4, 6))
    with pytest.raises(RuntimeError, match="no CPU path"):
        fitting.sketch_implicit_projection3(torch.zeros(1, 8, 3), torch.zeros(1, 8, 3), None, None, torch.zeros(1, 2, 3), torch.zeros(1, 2, 3), 8)
