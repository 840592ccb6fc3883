"""ctypes loader for librubiks_hip.so (the C ABI declared in include/rubiks_hip.h).

There is deliberately NO fallback: if the HIP library has not been built, or a tensor is
not on a GPU, the operators raise.  Build with `python -c "import __graft_entry__ as g;
g.build()"` or `rubiksnet_amd/csrc/build.sh`.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# RK_HIP_LIB: another build of the same library (kernel A/B runs of tools/; never a different implementation)
LIB_PATH = os.environ.get("RK_HIP_LIB") or os.path.join(_HERE, "csrc", "librubiks_hip.so")

_lock = threading.Lock()
_lib = None

_i = ctypes.c_int
_p = ctypes.c_void_p
_sz = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/rubiks_hip.h one to one
_DIMS3 = [_i] * 5 + [_i] * 6          # N,T,C,H,W, sT,sH,sW, pT,pH,pW
_DIMS2 = [_i] * 4 + [_i] * 4          # N,C,H,W, sH,sW,pH,pW
SIGNATURES = {
    "rk_version": (_i, []),
    "rk_error_string": (ctypes.c_char_p, [_i]),
    "rk_out_len": (_i, [_i, _i, _i]),
    "rk_device_count": (_i, []),
    "rk_debug_peek_launch_tag": (ctypes.c_uint, []),
    "rk_debug_set_finalize_spins": (_i, [_i]),
    "rk3d_debug_finalize_only_f32": (_i, [_p, _sz, _i, _i, _p, _i, ctypes.c_float, _p]),
    "rk3d_forward_f32": (_i, [_p, _p, _p] + _DIMS3 + [_i, _p]),
    "rk3d_forward_f64": (_i, [_p, _p, _p] + _DIMS3 + [_i, _p]),
    "rk3d_backward_workspace_bytes": (_sz, _DIMS3 + [_i]),
    "rk3d_backward_f32": (_i, [_p] * 5 + _DIMS3 + [_i, ctypes.c_float, _i, _p, _sz, _p]),
    "rk3d_backward_f64": (_i, [_p] * 5 + _DIMS3 + [_i, ctypes.c_double, _i, _p, _sz, _p]),
    "rk3d_backward_partials_f32": (_i, [_p] * 4 + _DIMS3 + [_i, _p, _sz, ctypes.POINTER(ctypes.c_int), _p]),
    "rk3d_backward_finalize_f32": (_i, [_p, _i, _i, _p, _i, ctypes.c_float, _p]),
    "rk2d_backward_workspace_bytes": (_sz, _DIMS2 + [_i]),
    "rk2d_backward_bn_workspace_bytes": (_sz, _DIMS2),
    "rk2d_bn_fused_shape": (_i, [_i] * 9),
    "rk2d_forward_bn_f32": (_i, [_p] * 4 + _DIMS2 + [_i, _p]),
    "rk2d_forward_bn_bf16_sf32": (_i, [_p] * 4 + _DIMS2 + [_i, _p]),
    "rk2d_backward_bn_f32": (_i, [_p] * 9 + _DIMS2 + [_i, _i, _p, _sz, _p]),
    "rk2d_backward_bn_bf16_sf32": (_i, [_p] * 9 + _DIMS2 + [_i, _i, _p, _sz, _p]),
    "rk_tshift3_backward_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "rk_tshift3_bn_forward_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "rk_tshift3_bn_forward_bf16": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "rk_tshift3_bn_backward_f32": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _sz, _p]),
    "rk_tshift3_bn_backward_bf16": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _sz, _p]),
    "rk_bn_relu_gather2_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "rk_bn_relu_gather2_bf16": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "rk_tshift3_bn_backward_fork_f32": (_i, [_p] * 12 + [_i] * 5 + [_p, _sz, _p]),
    "rk_tshift3_bn_backward_fork_bf16": (_i, [_p] * 12 + [_i] * 5 + [_p, _sz, _p]),
    "rk_tshift3_bn_backward_fin_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "rk_tshift3_bn_backward_fin_f32": (_i, [_p] * 11 + [_i, _i, _i, _i, _p, _sz, _p]),
    "rk_tshift3_bn_backward_fin_bf16": (_i, [_p] * 11 + [_i, _i, _i, _i, _p, _sz, _p]),
    "rk_soft_taps_forward_f32": (_i, [_p, _p, _p, _i, _p]),
    "rk_soft_taps_backward_f32": (_i, [_p, _p, _p, _p, _p, _i, _p]),
    "rk_soft_taps_many_forward_f32": (_i, [_p, _i, _p, _i, _p]),
    "rk_soft_taps_many_backward_f32": (_i, [_p, _i, _p, _p, _p, _i, _p]),
    "rk_bn_workspace_bytes": (_sz, [_i, _i, _i]),
    "rk_pw_gemm_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "rk_pw_gemm_bf16": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "rk_pw_packed_bytes": (_sz, [_i, _i]),
    "rk_pw_pack_bf16": (_i, [_p, _i, _i, _p, _p, _p]),
    "rk_pw_pack_many_bf16": (_i, [_p, _i, _p, _i, _p]),
    "rk_scatter2x2_add_bf16": (_i, [_p, _p, _p, ctypes.c_longlong, _i, _i, _p]),
    "rk_stem16_supported": (_i, [_i] * 5),
    "rk_stem_conv3x3s2_bf16out": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "rk_stem_wgrad16_workspace_bytes": (_sz, [_i] * 5),
    "rk_stem_wgrad3x3s2_bf16": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "rk_pw_odd16_supported": (_i, [_i, _i, _i, _i]),
    "rk_pw_gemm_packed_odd_bf16": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "rk_pw_wgrad_odd16_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "rk_pw_wgrad_odd16_bf16": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _sz, _p]),
    "rk_pw_gemm_packed_bf16": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "rk_pw16_stat_tiles": (_i, [_i, _i]),
    "rk_pw_gemm_packed_stats_bf16": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _i, _p]),
    "rk_pw_wgrad16_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "rk_pw_wgrad16_bf16": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _sz, _p]),
    "rk_stem_conv3x3s2_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "rk_stem_wgrad3x3s2_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "rk_pw_s2_forward_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "rk_pw_s2_forward_fused_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _i, _p]),
    "rk_pw_s2_dgrad_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "rk_pw_s2_wgrad_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "rk_bn_fold_f32": (_i, [_p, _p, _p, _p, ctypes.c_float, _p, _p, _i, _p]),
    "rk_bn_fold_many_f32": (_i, [_p, _i, _p, ctypes.c_longlong, _i, _p]),
    "rk_pw_gemm_fused_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _i, _p, _p, _i, _p]),
    "rk_pw_wgrad_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    # second-generation fp32 kernels (rk_pw2.hip): tuning / test hooks with an explicit kernel configuration
    "rk_pw2_gemm_cfg_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _i, _i, _i, _i, _p]),
    "rk_pw4_gemm_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _i, _i, _p, _p, _p, _p, _i, _p]),
    "rk_pw2_wgrad_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "rk_pw2_wgrad_cfg_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _sz, _p, _p, _i, _i, _i, _i, _p]),
    "rk_pw_wgrad_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _sz, _p]),
    "rk_pw_wgrad_bf16": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _sz, _p]),
    # planes with H * W % 4 != 0 (7x7)
    "rk_pw_gemm_odd_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "rk_pw_wgrad_odd_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "rk_pw_wgrad_odd_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _sz, _p]),
    "rk_pw_s2_forward_odd_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "rk_pw_s2_dgrad_odd_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "rk_pw_s2_wgrad_odd_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _sz, _p]),
    # training-mode fusion of the block's BatchNorms into the GEMMs (include/rubiks_hip.h)
    "rk3d_forward_bn_f32": (_i, [_p, _p, _p, _p] + _DIMS3 + [_i, _p]),
    "rk3d_backward_bn_workspace_bytes": (_sz, _DIMS3),
    "rk3d_backward_bn_f32": (_i, [_p] * 9 + _DIMS3 + [_i, ctypes.c_float, _i, _p, _sz, _p]),
    "rk_pw_tiles": (_i, [_i, _i]),
    "rk_pw_gemm_tiles": (_i, [_p, _i, _i, _i, _i, _i]),
    "rk_pw_gemm_stats_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _i, _p, _i, _p]),
    "rk_stem_conv3x3s2_stats_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _i, _p]),
    "rk_pw_gemm_bnbwd_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _i, _p]),
    "rk_pw_wgrad_pro_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _p, _i, _p, _sz, _p]),
    "rk_pw_s2_wgrad_pro_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _i, _p, _sz, _p]),
    "rk_bn_finish_tiles_f32": (_i, [_p, _i, ctypes.c_longlong, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, ctypes.c_float,
                                    ctypes.c_float, _p, _p]),
    "rk_bn_tile_stats_f32": (_i, [_p, _p, _i, _i, _i, _p]),
    "rk_bn_apply_affine_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "rk_bn_apply_affine_bf16": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "rk_bn_bwd_finish_tiles_f32": (_i, [_p, _i, ctypes.c_longlong, _p, _p, _p, _i, _p]),
    "rk_bn_bwd_dx_pre_f32": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "rk_bn_bwd_dx_pre_bf16": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "rk_bn_stats_finish_f32": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, ctypes.c_float, ctypes.c_float, _p, _p, _sz, _p]),
    "rk_bn_stats_finish_bf16": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, ctypes.c_float, ctypes.c_float, _p, _p, _sz, _p]),
    "rk_bn_stats_finish_abmi_f32": (_i, [_p] * 9 + [_i, _i, _i, ctypes.c_float, ctypes.c_float, _p, _p, _sz, _p]),
    "rk_bn_stats_finish_abmi_bf16": (_i, [_p] * 9 + [_i, _i, _i, ctypes.c_float, ctypes.c_float, _p, _p, _sz, _p]),
}
SIGNATURES["rk_se_mlp_forward_f32"] = (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p])
SIGNATURES["rk_se_mlp_backward_f32"] = (_i, [_p] * 11 + [_i, _i, _i, _p])
SIGNATURES["rk_se_dgate_f32"] = (_i, [_p, _p, _p, _i, _i, _i, _p])
SIGNATURES["rk_se_scale_add_f32"] = (_i, [_p, _p, _p, ctypes.c_float, _p, _i, _i, _i, _p])
for _sfx in ("f32", "bf16"):
    SIGNATURES["rk_clip_u8_to_chw_" + _sfx] = (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p])
    SIGNATURES["rk_se_squeeze_" + _sfx] = (_i, [_p, _p, _i, _i, _i, _p])
    SIGNATURES["rk_se_scale_" + _sfx] = (_i, [_p, _p, _p, _i, _i, _i, _p])
    SIGNATURES["rk_se_scale_backward_" + _sfx] = (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p])
    SIGNATURES["rk_bn_relu_forward_" + _sfx] = (
        _i, [_p] * 8 + [_i, _i, _i, ctypes.c_float, ctypes.c_float, _i, _i, _p, _sz, _p])
    SIGNATURES["rk_bn_relu_forward_counted_" + _sfx] = (
        _i, [_p] * 8 + [_i, _i, _i, ctypes.c_float, ctypes.c_float, _i, _p, _p, _sz, _p])
    SIGNATURES["rk_bn_relu_backward_" + _sfx] = (_i, [_p] * 10 + [_i, _i, _i, _i, _p, _sz, _p])
for _sfx in ("f32", "f64", "f16", "bf16"):
    SIGNATURES["rk2d_forward_" + _sfx] = (_i, [_p, _p, _p] + _DIMS2 + [_i, _p])
    SIGNATURES["rk2d_backward_" + _sfx] = (_i, [_p] * 5 + _DIMS2 + [_i, _i, _i, _p, _sz, _p])
    if _sfx in ("f16", "bf16"):          # 16-bit activations, fp32 shift table / d(shift)
        SIGNATURES["rk2d_forward_%s_sf32" % _sfx] = SIGNATURES["rk2d_forward_" + _sfx]
        SIGNATURES["rk2d_backward_%s_sf32" % _sfx] = SIGNATURES["rk2d_backward_" + _sfx]
    SIGNATURES["rk_tshift3_forward_" + _sfx] = (_i, [_p, _p, _p, _i, _i, _i, _i, _p])
    SIGNATURES["rk_tshift3_backward_" + _sfx] = (_i, [_p] * 5 + [_i, _i, _i, _i, _p, _sz, _p])


class RubiksHipError(RuntimeError):
    """A librubiks_hip entry point returned a negative RK_ERR_* code."""


def lib():
    """Load (once) and return the ctypes handle; raises if the library is not built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        "librubiks_hip.so not found at %s -- the HIP extension is not built and there is "
                        "no CPU fallback. Run `python -c \"import __graft_entry__ as g; g.build()\"`." % LIB_PATH
                    )
                handle = ctypes.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(handle, name)   # AttributeError here = ABI mismatch, loudly
                    fn.restype = res
                    fn.argtypes = args
                _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().rk_error_string(int(rc)).decode()
        raise RubiksHipError("%s failed: %s (code %d)" % (what, msg, rc))


def dtype_suffix(dtype):
    import torch

    return {
        torch.float32: "f32",
        torch.float64: "f64",
        torch.float16: "f16",
        torch.bfloat16: "bf16",
    }.get(dtype)
