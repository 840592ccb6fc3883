"""The C-ABI library loads (no GPU needed) and exports every symbol include/rubiks_hip.h declares;
argument validation returns error codes before anything touches a device."""
import ctypes
import os
import re

import pytest

from rubiksnet_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rubiks_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    names = set(re.findall(r"\b(rk[0-9a-z_]*?)\s*\(", text))
    names = {n for n in names if not n.endswith("_")}
    for macro, templ in (("RK_DECL_2D", ["rk2d_forward_%s", "rk2d_backward_%s"]),
                         ("RK_DECL_2D_SF32", ["rk2d_forward_%s_sf32", "rk2d_backward_%s_sf32"]),
                         ("RK_DECL_TAP", ["rk_tshift3_forward_%s", "rk_tshift3_backward_%s"]),
                         ("RK_DECL_BN", ["rk_bn_relu_forward_%s", "rk_bn_relu_forward_counted_%s", "rk_bn_relu_backward_%s"]),
                         ("RK_DECL_SE", ["rk_se_squeeze_%s", "rk_se_scale_%s", "rk_se_scale_backward_%s"])):
        for sfx in re.findall(macro + r"\((\w+)[,)]", text):
            if sfx != "SFX":
                names.update(t % sfx for t in templ)
    return sorted(n for n in names if "##" not in n)


def test_header_symbols_are_exported():
    lib = ctypes.CDLL(_native.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 25, syms
    for name in syms:
        assert hasattr(lib, name), "librubiks_hip.so does not export %s" % name
    assert set(_native.SIGNATURES) == set(syms), set(_native.SIGNATURES) ^ set(syms)


def test_version_shape_helper_and_error_strings():
    L = _native.lib()
    assert L.rk_version() >= 1
    assert L.rk_out_len(56, 1, 0) == 56 and L.rk_out_len(56, 2, 0) == 28 and L.rk_out_len(7, 2, 1) == 5
    assert L.rk_out_len(8, 0, 0) == -3
    for code in (0, -1, -2, -3, -4, -5, -6, -99):
        assert len(L.rk_error_string(code)) > 0
    assert L.rk3d_backward_workspace_bytes(32, 8, 64, 56, 56, 1, 1, 1, 0, 0, 0, 4) == 64 * 3 * 32 * 56 * 16  # per clip: max(To, H, ceil(H*W/256)) partials, 16-byte granule pairs in fp32
    assert L.rk3d_backward_workspace_bytes(32, 8, 64, 56, 56, 1, 1, 1, 0, 0, 0, 8) == 64 * 3 * 32 * 56 * 8
    assert L.rk2d_backward_workspace_bytes(4, 10, 7, 7, 1, 1, 0, 0, 4) == 10 * 2 * 4 * 16     # 16-byte granule pairs
    assert L.rk_tshift3_backward_workspace_bytes(16, 8, 5, 49) == 5 * 3 * 2 * 16
    assert L.rk_bn_workspace_bytes(256, 54, 56 * 56) == 54 * 86 * 2 * 16     # 3 frames per workgroup -> 86 groups; 16-byte granule pairs (fused statistics)
    assert L.rk_bn_workspace_bytes(0, 54, 49) == 0


def test_argument_validation_without_a_device():
    """Every entry point validates before launching, so these calls are safe on a GPU-less box."""
    L = _native.lib()
    one = ctypes.c_void_p(16)     # non-NULL dummy; never dereferenced on these paths
    dims = (2, 8, 4, 6, 6)
    assert L.rk3d_forward_f32(None, one, one, *dims, 1, 1, 1, 0, 0, 0, 0, None) == -1
    assert L.rk_bn_relu_forward_f32(one, one, one, one, one, None, None, one, 4, 3, 16, 1e-5, 0.1, 1, 1, one, 1 << 20, None) == -1
    assert L.rk_bn_relu_forward_f32(one, one, one, one, one, one, one, one, 4, 3, 16, 1e-5, 0.1, 1, 1, None, 0, None) == -4
    assert L.rk_bn_relu_backward_bf16(one, one, one, one, one, one, None, one, one, one, 4, 0, 16, 1, one, 1 << 20, None) == -2
    assert L.rk_pw_gemm_f32(one, one, None, None, 4, 8, 8, 16, 1, None) == -1
    assert L.rk_pw_gemm_f32(one, one, None, one, 4, 8, 8, 49, 1, None) == -2    # P % 4 != 0
    assert L.rk_pw_gemm_f32(one, one, None, one, 4, 7, 8, 16, 1, None) == -2    # odd K
    assert L.rk_pw_wgrad_f32(one, one, one, 4, 8, 8, 16, None, 0, None) == -4
    assert L.rk_pw_wgrad_workspace_bytes(256, 54, 54, 3136) > 0
    assert L.rk3d_forward_f32(one, one, one, 0, 8, 4, 6, 6, 1, 1, 1, 0, 0, 0, 0, None) == -2
    assert L.rk3d_forward_f64(one, one, one, *dims, 1, 0, 1, 0, 0, 0, 0, None) == -3
    assert L.rk3d_forward_f32(one, one, one, *dims, 1, 1, 1, 0, -1, 0, 0, None) == -3
    assert L.rk3d_forward_f32(one, one, one, 1 << 15, 1 << 10, 64, 56, 56, 1, 1, 1, 0, 0, 0, 0, None) == -2
    assert L.rk3d_backward_f32(one, one, one, one, one, *dims, 1, 1, 1, 0, 0, 0, 1, 1.0, 0, None, 0, None) == -4
    assert L.rk3d_backward_f32(one, one, one, None, None, *dims, 1, 1, 1, 0, 0, 0, 1, 1.0, 0, one, 1 << 20, None) == -1
    assert L.rk2d_forward_f16(one, None, one, 2, 4, 6, 6, 1, 1, 0, 0, 0, None) == -1
    assert L.rk2d_backward_bf16(one, one, one, one, one, 2, 4, 6, 6, 1, 1, 0, 0, 1, 1, 0, None, 0, None) == -4
    assert L.rk_tshift3_forward_f32(one, one, one, 15, 8, 4, 36, None) == -2        # NT % n_segment != 0
    assert L.rk_tshift3_backward_f64(one, one, one, one, one, 16, 8, 4, 36, None, 0, None) == -4
    with pytest.raises(_native.RubiksHipError, match="stride"):
        _native.check(-3, "demo")
