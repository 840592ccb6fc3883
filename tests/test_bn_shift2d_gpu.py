"""bn2 + ReLU folded into the 2-D shift (rk2d_*_bn_*, fused_bn.bn_relu_shift2d; round 5): the fused pair against
`RubiksShift2D(relu(BatchNorm2d(z)))` evaluated by PyTorch in fp64 + the oracle's 2-D shift (forward), and against the unfused
HIP path (bn_relu, then the shift module) for every gradient -- fp32 at fp32 bars, bf16 at bf16 bars; running statistics and
num_batches_tracked as nn.BatchNorm2d keeps them; integer shift components (the extra walks of rk2d_tile.hpp) included."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _setup(Fr, C, dtype, kind, seed, H=14, stride=1, padding=0):
    from rubiksnet_amd.shiftlib import RubiksShift2D
    from rubiksnet_amd.shiftlib.rubiks2d.primitive import compute_output_shape

    g = torch.Generator(device="cpu").manual_seed(seed)
    z = (torch.randn(Fr, C, H, H, generator=g) * 1.7 + 0.3).to(DEV).to(dtype)
    bn = torch.nn.BatchNorm2d(C).to(DEV).train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
    as3 = RubiksShift2D(C, stride=stride, padding=padding).to(DEV)
    with torch.no_grad():
        s = torch.rand(2, C, generator=g) * 2 - 1
        if kind == "integer":
            s[0, ::3] = torch.round(s[0, ::3] * 1.4)
            s[1, 1::4] = torch.round(s[1, 1::4] * 1.4)
            s[:, 0] = 0.0
        as3.shift.copy_(s)
    gy = torch.randn(*compute_output_shape(z, as3.stride, as3.padding), generator=g).to(DEV).to(dtype)
    return z, bn, as3, gy


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind", ["generic", "integer"])
@pytest.mark.parametrize("Fr,C,H,stride,padding", [
    (8, 6, 14, 1, 0), (256, 288, 14, 1, 0), (13, 34, 14, 1, 0), (10, 6, 56, 1, 0), (5, 3, 112, 1, 0), (9, 5, 16, 1, 0),
    (12, 7, 28, 1, 0), (3, 4, 12, 1, 0),
    # the column kernels (rk2d_column.hpp): the stride-2 layers of the -aq networks, 7 x 7, odd planes, padding
    (5, 3, 112, 2, 0), (10, 6, 56, 2, 0), (12, 7, 28, 2, 0), (13, 34, 14, 2, 0), (64, 40, 7, 1, 0), (9, 5, 7, 1, 0),
    (5, 3, 9, 2, 1), (4, 5, 13, (1, 2), (1, 0)), (6, 4, 30, 1, 1), (7, 3, 15, (2, 1), 0), (33, 2, 5, 3, 2)])
def test_fused_pair_matches_unfused_pair(Fr, C, H, stride, padding, dtype, kind):
    from rubiksnet_amd import fused_bn

    if H in (56, 112, 16, 28, 12) and (stride, padding) == (1, 0) and dtype == torch.float32:
        pytest.skip("fp32 planes the LDS-DMA kernels stream keep normalise + shift (test_other_planes_fall_back)")
    z, bn, as3, gy = _setup(Fr, C, dtype, kind, Fr * 7 + C, H, stride, padding)
    bn_u, as3_u = copy.deepcopy(bn), copy.deepcopy(as3)
    zf = z.clone().requires_grad_(True)
    zu = z.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
        yf = fused_bn.bn_relu_shift2d(bn, as3, zf)
        assert yf is not None, "this configuration must take a fused kernel"
        yu = as3_u(fused_bn.bn_relu(bn_u, zu))
    yf.backward(gy)
    yu.backward(gy)
    torch.cuda.synchronize()
    b16 = dtype == torch.bfloat16

    def close(a, b, tol, what):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        scale = max(1.0, float(b.abs().max()))
        err = float((a - b).abs().max())
        assert err <= tol * scale, "%s: max err %.3e (scale %.3g)" % (what, err, scale)

    # forward: the same activation (a z + b in one fma, rounded to the storage type) through the same interpolation
    close(yf, yu, 2e-2 if b16 else 2e-5, "y")
    close(zf.grad, zu.grad, 3e-2 if b16 else 5e-5, "d(z)")
    close(bn.weight.grad, bn_u.weight.grad, 2e-2 if b16 else 1e-4, "d(gamma)")
    close(bn.bias.grad, bn_u.bias.grad, 2e-2 if b16 else 1e-4, "d(beta)")
    close(as3.shift.grad, as3_u.shift.grad, 2e-2 if b16 else 1e-4, "d(shift)")      # unit vectors after K9
    close(bn.running_mean, bn_u.running_mean, 1e-5, "running_mean")
    close(bn.running_var, bn_u.running_var, 1e-5, "running_var")
    assert int(bn.num_batches_tracked) == int(bn_u.num_batches_tracked) == 1


@pytest.mark.parametrize("kind", ["generic", "integer"])
def test_fused_forward_against_fp64_and_the_oracle(oracle, kind):
    """fp32: y == oracle.rk2d_forward(relu(bn(z)) in fp64 rounded to fp32) to fp32 round-off of the normalisation."""
    from rubiksnet_amd import fused_bn

    z, bn, as3, _ = _setup(16, 10, torch.float32, kind, 5)
    y = fused_bn.bn_relu_shift2d(bn, as3, z.clone().requires_grad_(True))
    zd = z.double()
    mean = zd.mean(dim=(0, 2, 3), keepdim=True)
    var = zd.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
    act = torch.relu((zd - mean) / torch.sqrt(var + bn.eps) * bn.weight.double().view(1, -1, 1, 1) + bn.bias.double().view(1, -1, 1, 1))
    y_ref = oracle.rk2d_forward(act.detach().float().cpu().numpy(), as3.shift.detach().cpu().numpy(), 1, 0)
    np.testing.assert_allclose(y.detach().cpu().numpy(), y_ref, rtol=0, atol=2e-5 * max(1.0, float(np.abs(y_ref).max())))


def test_other_planes_fall_back():
    from rubiksnet_amd import fused_bn
    from rubiksnet_amd.shiftlib import RubiksShift2D

    bn = torch.nn.BatchNorm2d(8).to(DEV).train()
    as3 = RubiksShift2D(8).to(DEV)
    assert fused_bn.bn_relu_shift2d(bn, as3, torch.randn(4, 8, 28, 28, device=DEV, requires_grad=True)) is None     # fp32
    assert fused_bn.bn_relu_shift2d(bn, as3, torch.randn(4, 8, 56, 56, device=DEV, requires_grad=True)) is None     # fp32, streamed
    asq = RubiksShift2D(8, quantize=True).to(DEV)
    assert fused_bn.bn_relu_shift2d(bn, asq, torch.randn(4, 8, 14, 14, device=DEV, requires_grad=True)) is None     # quantize
    assert fused_bn.bn_relu_shift2d(bn, as3, torch.randn(4, 8, 14, 14, device=DEV)) is not None
    assert fused_bn.bn_relu_shift2d(bn.eval(), as3, torch.randn(4, 8, 14, 14, device=DEV, requires_grad=True)) is None


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_backward_takes_an_upstream_gradient_at_an_odd_storage_offset(dtype):
    """The fused backward kernels want 16-byte aligned planes; the upstream gradient autograd hands over can be a contiguous
    VIEW into a bigger buffer at any element offset (round-5 advisor finding: RK_ERR_UNSUPPORTED mid-step, after bn2's
    running statistics were already updated).  The backward copies such a gradient to a fresh buffer: same gradients as
    with an aligned one, bit for bit."""
    from rubiksnet_amd import fused_bn

    z, bn, as3, gy = _setup(16, 10, dtype, "generic", 5)
    big = torch.zeros(gy.numel() + 8, dtype=dtype, device=DEV)
    off = 1 if dtype == torch.float32 else 3                          # 4 / 6 bytes past a 16-byte boundary
    gy_odd = big[off:off + gy.numel()].view_as(gy)
    gy_odd.copy_(gy)
    assert gy_odd.is_contiguous() and gy_odd.data_ptr() % 16 != 0
    grads = []
    for g in (gy, gy_odd):
        bn_c, as3_c = copy.deepcopy(bn), copy.deepcopy(as3)
        zc = z.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
            y = fused_bn.bn_relu_shift2d(bn_c, as3_c, zc)
        assert y is not None
        y.backward(g)
        torch.cuda.synchronize()
        grads.append((zc.grad, bn_c.weight.grad, bn_c.bias.grad, as3_c.shift.grad))
    for a, b in zip(*grads):
        assert torch.equal(a, b)
