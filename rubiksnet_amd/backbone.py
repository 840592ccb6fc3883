"""2D pre-activation residual backbone built from RubiksShift blocks.

Counterpart of rubiksnet/backbone.py:14-235.  Convolutions and Linear are stock PyTorch (MIOpen /
hipBLASLt under PyTorch-ROCm); the BatchNorm2d modules are stock too, but each `relu(bn(x))` pair is
evaluated by one fused HIP operator on GPU tensors (fused_bn.py; `RK_FUSED_BN=0` restores the stock pair).  Module and attribute
names (`conv1`, `layer0..4`, `bn1`, `conv2`, `bn2`, `as3`, `se`, `conv3`, `shortcut`,
`bn_last`, `avgpool`, `fc`) are the reference's, so state dicts are interchangeable.
"""
import math

import torch
import torch.nn as nn

from . import _native, config
from .fused_bn import bn_relu, bn_relu_shift2d, bn_relu_skip, bn_relu_tshift_fork, bn_relu_tshift_skip
from .pointwise import _gathered_kind, all_frozen, conv1x1, conv_on_gathered, fork_shortcut, fused_eval_block, stem_conv
from .shiftlib import RubiksShift2D, RubiksShiftBase
from .train_block import bn_relu_from_stats, fused_train_block

__all__ = ["RubiksNetBackbone", "RubiksShiftBlock", "SELayer"]


def _pointwise(c_in, c_out, stride=1):
    return nn.Conv2d(c_in, c_out, kernel_size=1, stride=stride, bias=False)


def _unit_bn(channels):
    return nn.BatchNorm2d(channels)          # weight 1 / bias 0 is nn.BatchNorm2d's own initial state


def _he_fan_out_(conv):
    """N(0, sqrt(2 / (k*k*C_out))) -- the reference's conv init (backbone.py:14-19)."""
    kh, kw = conv.kernel_size
    nn.init.normal_(conv.weight, mean=0.0, std=math.sqrt(2.0 / (kh * kw * conv.out_channels)))


# what `RubiksNetBackbone._reset_parameters` does per module type; a module opts out with `skip_global_init = True`
_INITIALISERS = (
    (nn.Conv2d, _he_fan_out_),
    ((nn.BatchNorm2d, nn.GroupNorm), lambda m: (nn.init.ones_(m.weight), nn.init.zeros_(m.bias))),
)


class _SEGate(torch.autograd.Function):
    """y = x * gate[f, c] on [F, C, H, W] with gate [F, C]: the scale half of the SE layer as one HIP pass; its
    backward produces d(x) = dy * gate and d(gate) = sum_hw(dy * x) in one pass over (dy, x)."""

    @staticmethod
    def forward(ctx, x, gate):
        Fr, C, H, W = x.shape
        y = torch.empty_like(x)
        g32 = gate.detach().float().contiguous()
        _se_call("rk_se_scale_", x, x.data_ptr(), g32.data_ptr(), y.data_ptr(), Fr, C, H * W)
        ctx.save_for_backward(x, g32)
        ctx.gate_dtype = gate.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g32 = ctx.saved_tensors
        Fr, C, H, W = x.shape
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        dx = torch.empty_like(x)
        dgate = torch.empty_like(g32)
        _se_call("rk_se_scale_backward_", x, dy.data_ptr(), x.data_ptr(), g32.data_ptr(), dx.data_ptr(), dgate.data_ptr(),
                 Fr, C, H * W)
        return dx, dgate.to(ctx.gate_dtype)


class _SESqueeze(torch.autograd.Function):
    """mean over H*W of every (frame, channel) plane -> [F, C] fp32 (AdaptiveAvgPool2d(1) + view)."""

    @staticmethod
    def forward(ctx, x):
        Fr, C, H, W = x.shape
        mean = torch.empty(Fr, C, dtype=torch.float32, device=x.device)
        _se_call("rk_se_squeeze_", x, x.data_ptr(), mean.data_ptr(), Fr, C, H * W)
        ctx.shape, ctx.dtype = x.shape, x.dtype
        return mean

    @staticmethod
    def backward(ctx, dmean):
        Fr, C, H, W = ctx.shape
        return (dmean / float(H * W)).to(ctx.dtype).view(Fr, C, 1, 1).expand(Fr, C, H, W)


_SE_SFX = {torch.float32: "f32", torch.bfloat16: "bf16"}


def _se_call(name, x, *args):
    dev = x.device
    with torch.cuda.device(dev):
        rc = getattr(_native.lib(), name + _SE_SFX[x.dtype])(*args, torch.cuda.current_stream(dev).cuda_stream)
    _native.check(rc, name)


class SELayer(nn.Module):
    """Squeeze-and-excitation gate (backbone.py:56-71); used by the Small tier.  Same sub-modules and state-dict keys
    (`fc.0.weight`, `fc.2.weight`).  On GPU fp32 / bf16 activations the squeeze and the scale are one HIP pass each
    (rk_se_*); the two tiny Linear layers stay in PyTorch."""

    def __init__(self, channel, reduction):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Sequential(
            nn.Linear(channel, channel // reduction, bias=False),
            nn.ReLU(inplace=True),
            nn.Linear(channel // reduction, channel, bias=False),
            nn.Sigmoid(),
        )

    def forward(self, x):
        b, c = x.shape[:2]
        if x.is_cuda and x.dim() == 4 and x.dtype in _SE_SFX and x.numel() > 0 and config.switches().fused_bn:
            x = x.contiguous()
            gate = self.fc(_SESqueeze.apply(x).to(self.fc[0].weight.dtype))
            return _SEGate.apply(x, gate)
        gate = self.fc(self.avg_pool(x).view(b, c)).view(b, c, 1, 1)
        return x * gate.expand_as(x)


class RubiksShiftBlock(nn.Module):
    """BN-ReLU-1x1 -> BN-ReLU-SHIFT -> (SE) -> 1x1, plus shortcut (backbone.py:74-135)."""

    def __init__(self, in_planes, out_planes, *, stride=1, parent):
        super().__init__()
        mid_planes = int(out_planes * parent.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.bn1 = _unit_bn(in_planes)
        self.conv2 = _pointwise(in_planes, mid_planes)
        self.bn2 = _unit_bn(mid_planes)
        self.as3 = RubiksShift2D(mid_planes, stride=stride, normalize_grad=parent.normalize_grad,
                                 quantize=parent.quantize, init_shift=parent.init_shift)
        self.se = self._make_se(mid_planes, parent.use_se)
        self.conv3 = _pointwise(mid_planes, out_planes)
        projects = stride != 1 or in_planes != out_planes
        self.shortcut = _pointwise(in_planes, out_planes, stride=stride) if projects else nn.Identity()

    @staticmethod
    def _make_se(channels, use_se):
        """`use_se`: falsy -> no gate; True -> the default reduction 12; an int -> that reduction (must exceed 2)."""
        if not use_se:
            return None
        reduction = 12 if use_se is True else int(use_se)
        assert reduction > 2, ("SE reduction must > 2", use_se)
        return SELayer(channels, reduction=reduction)

    def forward(self, x):
        if not self.training:               # inference: BNs and the residual add ride on the two 1x1 GEMMs
            y = fused_eval_block(self, x)
            if y is not None:
                return y
        else:                               # training: one autograd node, BatchNorms folded into the GEMMs (train_block.py)
            y = fused_train_block(self, x)
            if y is not None:
                return y
        if isinstance(self.shortcut, nn.Identity):
            # -aq blocks in training: bn1 + ReLU folded into the AttentionShift in front of conv2 (fused_bn.py)
            if self.training and isinstance(self.conv2, nn.Sequential) and len(self.conv2) == 2:
                r = bn_relu_tshift_skip(self.bn1, self.conv2[0], x)
                if r is not None:
                    out, shortcut = r
                    z2 = conv1x1(self.conv2[1], out)
                    out = bn_relu_shift2d(self.bn2, self.as3, z2)     # bn2 + ReLU inside the 2-D shift kernels (14 x 14 planes)
                    if out is None:
                        out = self.as3(bn_relu(self.bn2, z2))
                    if self.se:
                        out = self.se(out)
                    return conv1x1(self.conv3, out, residual=shortcut)
            # relu(bn(.)) as one operator on GPU tensors (fused_bn.py); the shortcut's gradient joins inside its backward
            out, shortcut = bn_relu_skip(self.bn1, x)
        else:
            if (self.training and isinstance(self.conv2, nn.Sequential) and len(self.conv2) == 2 and x.dim() == 4
                    and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0):
                # downsampling -aq block: bn1 + ReLU inside the AttentionShift, the shortcut's operand gathered from x, the
                # shortcut's gradient joined inside the filter's backward (fused_bn.bn_relu_tshift_fork)
                kind = _gathered_kind(self.shortcut, x.shape[0], (x.shape[2] // 2) * (x.shape[3] // 2), x.dtype)
                r = bn_relu_tshift_fork(self.bn1, self.conv2[0], x) if kind is not None else None
                if r is not None:
                    out, xs = r
                    shortcut = conv_on_gathered(self.shortcut, xs, kind)
                    z2 = conv1x1(self.conv2[1], out)
                    out = bn_relu_shift2d(self.bn2, self.as3, z2)
                    if out is None:
                        out = self.as3(bn_relu(self.bn2, z2))
                    if self.se:
                        out = self.se(out)
                    return conv1x1(self.conv3, out, residual=shortcut)
            out = bn_relu(self.bn1, x)
            # (stride-2 projecting shortcut on bf16 activations: one autograd node for the two consumers of `out`, their
            # gradients joined in one pass)
            out, shortcut = fork_shortcut(self.shortcut, out)
        z2 = conv1x1(self.conv2, out)
        out = bn_relu_shift2d(self.bn2, self.as3, z2) if self.training else None
        if out is None:
            out = self.as3(bn_relu(self.bn2, z2))
        if self.se:
            out = self.se(out)
        return conv1x1(self.conv3, out, residual=shortcut)      # conv3(out) + shortcut, the add in the GEMM's epilogue


class RubiksNetBackbone(nn.Module):
    def __init__(self, width, repeats, expansion=1, num_classes=1000, use_se=False, quantize=False,
                 normalize_grad=True, init_shift="uniform"):
        super().__init__()
        self.init_shift = init_shift
        self.width = width
        self.inplanes = width
        self.expansion = expansion
        self.use_se = use_se
        self.quantize = quantize
        self.normalize_grad = normalize_grad
        self.conv1 = nn.Conv2d(3, self.inplanes, kernel_size=3, stride=2, padding=1, bias=False)

        # (planes multiplier, #blocks, stride of the first block) per stage -- backbone.py:158-165
        self.layer0 = self._make_layer(RubiksShiftBlock, width, 1, stride=1)
        self.layer1 = self._make_layer(RubiksShiftBlock, width, repeats[0], stride=2)
        self.layer2 = self._make_layer(RubiksShiftBlock, 2 * width, repeats[1], stride=2)
        self.layer3 = self._make_layer(RubiksShiftBlock, 4 * width, repeats[2], stride=2)
        self.layer4 = self._make_layer(RubiksShiftBlock, 8 * width, repeats[3], stride=2)

        self.relu = nn.ReLU(inplace=True)
        self.bn_last = _unit_bn(8 * width)
        self.avgpool = nn.AvgPool2d(7, stride=1)
        self.fc = nn.Linear(8 * width, num_classes)
        self._reset_parameters()

    def _reset_parameters(self):
        for module in self.modules():
            if getattr(module, "skip_global_init", False):
                continue
            for types, init in _INITIALISERS:
                if isinstance(module, types):
                    init(module)
        nn.init.normal_(self.fc.weight, 0.0, 0.01)

    def _make_layer(self, block, planes, repeat, stride):
        blocks = [block(self.inplanes, planes, stride=stride, parent=self)]
        self.inplanes = planes
        blocks += [block(planes, planes, stride=1, parent=self) for _ in range(repeat - 1)]
        return nn.Sequential(*blocks)

    def forward(self, x):
        x = stem_conv(self.conv1, x)
        if self.training or not torch.is_grad_enabled():
            for stage in (self.layer0, self.layer1, self.layer2, self.layer3, self.layer4):
                x = stage(x)
        else:
            with all_frozen(self):          # eval with grad mode on: one parameter walk per forward, not one per block
                for stage in (self.layer0, self.layer1, self.layer2, self.layer3, self.layer4):
                    x = stage(x)
        # statistics from the last conv3's epilogue when the last block ran fused (inputs whose last planes have H * W % 4 == 0;
        # at 224 x 224 they are 7 x 7, the block runs layer by layer and bn_last keeps its own statistics pass)
        y = bn_relu_from_stats(self.bn_last, x) if self.training else None
        x = y if y is not None else bn_relu(self.bn_last, x)
        x = self.avgpool(x)
        return self.fc(x.view(x.size(0), -1))

    # module family -> {group name: which of the module's own parameters}; the first matching family wins
    _FAMILIES = (
        ((nn.Conv2d, nn.Conv3d, nn.Linear), {"weight": "weight", "bias": "bias"}),
        ((nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d), {"bn": None}),
        ((RubiksShift2D, RubiksShiftBase), {"shift": None}),
    )
    _DECAY = {"weight": 1, "bias": 0, "bn": 0, "shift": 0}

    def get_optim_policy(self, shift_lr_mult=0.01):
        """Optimizer parameter groups `weight` / `bias` / `bn` / `shift` with `lr_mult` and `decay_mult` entries
        (same groups, order and multipliers as the reference, backbone.py:202-235): only weights decay, shifts
        train at `shift_lr_mult` of the base rate.  A module belongs to the FIRST family it matches (so no
        parameter can land in two groups); a parameter-owning LEAF module of an unknown type is an error, a
        container of an unknown type is walked through, exactly as the reference does."""
        groups = {name: [] for name in self._DECAY}
        for module in self.modules():
            family = next((rule for types, rule in self._FAMILIES if isinstance(module, types)), None)
            if family is None:
                if not module._modules and next(module.parameters(), None) is not None:
                    raise ValueError("New atomic module type: {}. Need to give it a learning policy".format(type(module)))
                continue
            own = dict(module.named_parameters(recurse=False))
            for group, attr in family.items():
                groups[group] += [p for k, p in own.items() if attr is None or k == attr]
        return [{"params": groups[name], "lr_mult": shift_lr_mult if name == "shift" else 1, "decay_mult": decay,
                 "name": name} for name, decay in self._DECAY.items()]
