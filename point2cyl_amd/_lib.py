"""ctypes binding of libp2c_hip.so (the C ABI declared in include/p2c_hip.h).

There is NO fallback: if the shared object is missing or a call fails, a RuntimeError is raised.
PyTorch only provides device memory (tensor.data_ptr()) and the HIP stream.
"""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("P2C_LIB") or os.path.join(_HERE, "libp2c_hip.so")      # (P2C_LIB: another build of the same sources, kernel A/B runs)
HEADER = os.path.join(_HERE, "..", "include", "p2c_hip.h")

_lib = None

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_f = ctypes.c_float
c_ll = ctypes.c_longlong

_SIGS = {
    "p2c_fps_f32": [c_p, c_i, c_i, c_p, c_i, c_p, c_p, c_p],
    "p2c_ball_query_f32": [c_p, c_p, c_i, c_i, c_i, c_f, c_i, c_p, c_p],
    "p2c_three_nn_f32": [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p],
    "p2c_group_gather_f32": [c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p],
    "p2c_group_gather_bwd_f32": [c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p],
    "p2c_three_interp_f32": [c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p],
    "p2c_three_interp_bwd_f32": [c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p],
    "p2c_build_csr_i32": [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p],
    "p2c_csr_gather_f32": [c_p, c_i, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_p],
    "p2c_input_moments_f32": [c_p, c_i, ctypes.c_longlong, c_p, c_p],
    "p2c_bn_finalize_affine_f32": [c_p, ctypes.c_longlong, c_p, c_p, c_p, c_p, c_f, c_f, c_p, c_p, c_i, c_p, c_p],
    "p2c_linear_fwd_fold0_f32": [c_p, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_p, c_p],
    "p2c_linear_bwd_fused_fold0_f32": [c_p, c_i, c_p, c_i, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_i, ctypes.c_longlong, c_p, c_i, c_i, c_i,
                                       c_p],
    "p2c_fold0_bwd_finalize_f32": [c_p, c_p, ctypes.c_longlong, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p],
    "p2c_three_interp_bias_stats_f32": [c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_p, c_p],
    "p2c_csr_gather_bn_f32": [c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_p],
    "p2c_group_linear_bias_stats_f32": [c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_p, c_p],
    "p2c_group_linear_bwd_f32": [c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_p, c_p],
    "p2c_linear_fwd_gbias_f32": [c_p, c_i, c_p, c_i, c_p, c_p, c_i, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p],
    "p2c_group_colsum_bn_f32": [c_p, c_i, c_p, c_i, c_p, c_i, c_i, c_i, c_p, c_i, c_p],
    "p2c_linear_fwd_f32": [c_p, c_i, c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_i, c_f, c_p, c_p],
    "p2c_bn_finalize_f32": [c_p, c_i, c_ll, c_p, c_p, c_p, c_f, c_f, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "p2c_bn_eval_affine_batch_f32": [c_p, c_i, c_p],
    "p2c_bn_bwd_finalize_f32": [c_p, c_i, c_ll, c_p, c_p, c_p, c_p, c_p, c_p],
    "p2c_sum_copies_f32": [c_p, c_ll, c_i, c_p, c_ll, c_p],
    "p2c_linear_bwd_narrow_f32": [c_p, c_i, c_p, c_i, c_p, c_p, c_f, c_p, c_i, c_p, c_i, c_p, c_i, c_ll, c_p, c_p, c_i, c_i, c_i, c_p],
    "p2c_bn_bwd_finalize_sum_f32": [c_p, c_i, c_ll, c_p, c_p, c_p, c_p, c_p, c_p, c_ll, c_i, c_p, c_ll, c_p],
    "p2c_bn_relu_apply_f32": [c_p, c_i, c_p, c_p, c_i, c_i, c_p, c_i, c_p],
    "p2c_maxpool_bnrelu_f32": [c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_p, c_p],
    "p2c_maxpool_bwd_f32": [c_p, c_i, c_p, c_i, c_i, c_i, c_p, c_i, c_p],
    "p2c_adam_multi_f32": [c_p, c_p, c_p, c_i, c_f, c_f, c_f, c_f, c_ll, c_p],
    "p2c_linear_fwd_pool_supported": [c_i, c_i, c_i, c_i, c_i],
    "p2c_linear_fwd_pool_f32": [c_p, c_i, c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "p2c_pool_select_f32": [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_i, c_p, c_p, c_p],
    "p2c_bn_relu_bwd_stats_f32": [c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p],
    "p2c_linear_bwd_data_f32": [c_p, c_i, c_p, c_i, c_i, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_f, c_p, c_i, c_p,
                                c_p, c_p, c_i, c_p],
    "p2c_maxpool_bn_bwd_stats_f32": [c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p],
    "p2c_linear_bwd_weight_f32": [c_p, c_i, c_p, c_i, c_i, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_i, c_f, c_p, c_i, c_ll, c_p,
                                  c_i, c_i, c_i, c_p, c_i, c_p],
    "p2c_linear_bwd_fused_f32": [c_p, c_i, c_p, c_i, c_i, c_p, c_p, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_i, c_p, c_i, c_p, c_i, c_ll,
                                 c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    "p2c_linear_bwd_pool_alg_f32": [c_p, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i,
                                    c_p],
    "p2c_copy2d_batch_f32": [c_p, c_i, c_p],
    "p2c_copy2d_batch_inc_f32": [c_p, c_i, c_p, c_i, c_p],
    "p2c_three_interp_skip_f32": [c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_i, c_i, c_p],
    "p2c_fold0_bwd_finalize_sum_f32": [c_p, c_p, ctypes.c_longlong, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_p, ctypes.c_longlong, c_i, c_p,
                                       ctypes.c_longlong, c_p],
    "p2c_group_weight_grad_f32": [c_p, c_i, c_p, c_i, c_i, c_i, c_p, c_p],
    "p2c_copy_flat_batch": [c_p, c_p, c_p, c_i, c_p],
    "p2c_extrusion_axis_f32": [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p],
    "p2c_extrusion_axis_bwd_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p],
    "p2c_extrusion_centers_f32": [c_p, c_p, c_i, c_i, c_i, c_p, c_p],
    "p2c_extrusion_centers_bwd_f32": [c_p, c_p, c_i, c_i, c_i, c_p, c_p],
    "p2c_segment_centroids_f32": [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p],
    "p2c_extrusion_extents_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p],
    "p2c_fit_fused_f32": [c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "p2c_sketch_projection_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p],
    "p2c_softplus_fwd_f32": [c_p, c_p, c_ll, c_f, c_f, c_p],
    "p2c_softplus_bwd_f32": [c_p, c_p, c_p, c_ll, c_f, c_f, c_p],
    "p2c_softplus_bwd_bwd_f32": [c_p, c_p, c_p, c_p, c_p, c_ll, c_f, c_f, c_p],
    "p2c_linear_bwd_data_sig_f32": [c_p, c_i, c_p, c_i, c_p, c_i, c_f, c_f, c_p, c_i, c_i, c_i, c_i, c_p],
    "p2c_softplus_sig_bwd_f32": [c_p, c_p, c_p, c_p, c_p, c_ll, c_f, c_f, c_p],
    "p2c_linear_fwd_big_f32": [c_p, c_i, c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p],
    "p2c_linear_bwd_data_big_f32": [c_p, c_i, c_p, c_i, c_p, c_i, c_f, c_f, c_p, c_i, c_i, c_i, c_i, c_p, c_p],
    "p2c_softplus_dot_f32": [c_p, c_p, c_p, c_p, c_ll, c_i, c_f, c_f, c_p],
    "p2c_softplus_row_bwd_f32": [c_p, c_i, c_p, c_p, c_p, c_p, c_ll, c_i, c_f, c_f, c_p],
    "p2c_softplus_sig_bwd_rank2_f32": [c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_ll, c_i, c_f, c_f, c_p],
    "p2c_linear_fwd_big_add_f32": [c_p, c_i, c_p, c_i, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p, c_p],
    "p2c_linear_fwd_big_sp_f32": [c_p, c_i, c_p, c_i, c_p, c_p, c_i, c_f, c_f, c_p, c_i, c_i, c_i, c_i, c_p, c_p],
    "p2c_linear_bwd_data_big_add_f32": [c_p, c_i, c_p, c_i, c_p, c_i, c_f, c_f, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p, c_p],
    "p2c_linear_sum_assignment_f64": [c_p, c_i, c_i, c_i, c_p, c_i, c_p],
    "p2c_linear_bwd_both_f32": [c_p, c_i, c_p, c_i, c_i, c_p, c_p, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "p2c_head_post_f32": [c_p, c_i, c_i, c_i, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p],
    "p2c_head_post_bwd_f32": [c_p, c_i, c_i, c_i, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_p],
    "p2c_hungarian_f32": [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p],
    "p2c_hungarian_logits_f32": [c_p, c_i, c_i, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p],
    "p2c_all_losses_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p],
    "p2c_seg_losses_grad_f32": [c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_f, c_f, c_p, c_p, c_p, c_p],
    "p2c_fit_terms_f32": [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_f, c_f, c_p, c_p, c_p, c_p],
    "p2c_seg_losses_f32": [c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_f, c_f, c_p, c_p, c_p, c_p],
    "p2c_eval_metrics_f32": [c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_f, ctypes.c_double, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
}


def declared_symbols():
    """Every function name include/p2c_hip.h declares."""
    with open(HEADER) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(p2c_[a-z0-9_]+)\s*\(", text)))


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "point2cyl_amd: %s is missing. Build it with `python -m point2cyl_amd.build` "
            "(hipcc, gfx950). There is no CPU / eager fallback." % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    for name, args in _SIGS.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = c_i
    L.p2c_abi_version.restype = c_i
    L.p2c_build_arch.restype = ctypes.c_char_p
    L.p2c_linear_stat_tiles.argtypes = [c_i]
    L.p2c_linear_stat_tiles.restype = c_i
    L.p2c_seg_losses_ws_bytes.argtypes = [c_i, c_i]
    L.p2c_seg_losses_ws_bytes.restype = ctypes.c_size_t
    L.p2c_all_losses_ws_bytes.argtypes = [c_i, c_i]
    L.p2c_all_losses_ws_bytes.restype = ctypes.c_size_t
    L.p2c_stat_slots_bytes.argtypes = [c_i]
    L.p2c_stat_slots_bytes.restype = ctypes.c_size_t
    L.p2c_linear_bwd_fused_supported.argtypes = [c_i, c_i, c_i]
    L.p2c_linear_bwd_fused_supported.restype = c_i
    L.p2c_linear_fwd_pp_supported.argtypes = [c_i, c_i, c_i, c_i]
    L.p2c_linear_fwd_pp_supported.restype = c_i
    L.p2c_set_mfma_mode.argtypes = [c_i]
    L.p2c_set_mfma_mode.restype = c_i
    L.p2c_get_mfma_mode.argtypes = []
    L.p2c_get_mfma_mode.restype = c_i
    L.p2c_linear_bwd_fused_parts.argtypes = [c_i, c_i]
    L.p2c_linear_bwd_fused_parts.restype = c_i
    L.p2c_linear_tile_m.restype = c_i
    L.p2c_eval_metrics_ws_bytes.argtypes = [c_i, c_i]
    L.p2c_eval_metrics_ws_bytes.restype = ctypes.c_size_t
    L.p2c_eval_metrics_supported.argtypes = [c_i]
    L.p2c_eval_metrics_supported.restype = c_i
    L.p2c_hungarian_ws_bytes.argtypes = [c_i]
    L.p2c_hungarian_ws_bytes.restype = ctypes.c_size_t
    L.p2c_linear_bwd_narrow_supported.argtypes = [c_i, c_i, c_i, c_i]
    L.p2c_linear_bwd_narrow_supported.restype = c_i
    L.p2c_linear_bwd_pool_alg_supported.argtypes = [c_i, c_i, c_i, c_i]
    L.p2c_linear_bwd_pool_alg_supported.restype = c_i
    L.p2c_linear_bwd_pool_alg_ws_bytes.argtypes = [c_i, c_i]
    L.p2c_linear_bwd_pool_alg_ws_bytes.restype = ctypes.c_size_t
    L.p2c_fit_fused_supported.argtypes = [c_i, c_i, c_i]
    L.p2c_fit_fused_supported.restype = c_i
    L.p2c_extents_ws_bytes.argtypes = [c_i, c_i]
    L.p2c_extents_ws_bytes.restype = ctypes.c_size_t
    L.p2c_linear_big_supported.argtypes = [c_i, c_i, c_i]
    L.p2c_linear_big_supported.restype = c_i
    L.p2c_linear_big_ws_bytes.argtypes = [c_i, c_i]
    L.p2c_linear_big_ws_bytes.restype = ctypes.c_size_t
    _lib = L
    return L


def ptr(t):
    if t is None:
        return None
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


class _Profile:
    """Optional per-launch timing with HIP events recorded on the stream the kernels are launched on
    (torch's current stream).  Used by bench.py for the roofline figures; off by default."""

    def __init__(self):
        self.enabled = False
        self.records = []

    def reset(self, enabled=False):
        self.records = []
        self.enabled = enabled

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, e0, e1, flops, nbytes in self.records:
            d = out.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += flops
            d["bytes"] += nbytes
        return out


PROFILE = _Profile()


def call(name, *args, flops=0.0, nbytes=0.0):
    fn = getattr(lib(), name)
    if PROFILE.enabled:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*args)
        e1.record()
        PROFILE.records.append((name, e0, e1, flops, nbytes))
    else:
        rc = fn(*args)
    if rc != 0:
        raise RuntimeError("%s failed with code %d" % (name, rc))


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("point2cyl_amd ops need HIP device tensors (got a %s tensor); there is no CPU path"
                               % t.device.type)
