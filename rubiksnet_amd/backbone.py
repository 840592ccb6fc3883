"""2D pre-activation residual backbone built from RubiksShift blocks.

Counterpart of rubiksnet/backbone.py:14-235.  Convolutions and Linear are stock PyTorch (MIOpen /
hipBLASLt under PyTorch-ROCm); the BatchNorm2d modules are stock too, but each `relu(bn(x))` pair is
evaluated by one fused HIP operator on GPU tensors (fused_bn.py; `RK_FUSED_BN=0` restores the stock pair).  Module and attribute
names (`conv1`, `layer0..4`, `bn1`, `conv2`, `bn2`, `as3`, `se`, `conv3`, `shortcut`,
`bn_last`, `avgpool`, `fc`) are the reference's, so state dicts are interchangeable.
"""
import math

import torch.nn as nn

from .fused_bn import bn_relu, bn_relu_skip
from .pointwise import conv1x1, fused_eval_block, stem_conv
from .shiftlib import RubiksShift2D, RubiksShiftBase

__all__ = ["RubiksNetBackbone", "RubiksShiftBlock", "SELayer"]


def _skip_global_init(m):
    return getattr(m, "skip_global_init", False)


def conv2d_init(m):
    """He-normal on fan-out (backbone.py:14-19)."""
    assert isinstance(m, nn.Conv2d)
    if _skip_global_init(m):
        return
    fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
    nn.init.normal_(m.weight, 0, math.sqrt(2.0 / fan_out))


def norm_layer_init(m, weight_init=1.0):
    assert isinstance(weight_init, (int, float))
    assert isinstance(m, (nn.BatchNorm2d, nn.GroupNorm))
    nn.init.constant_(m.weight, weight_init)
    nn.init.constant_(m.bias, 0)


def conv_bn_init_module(net):
    assert isinstance(net, nn.Module)
    for m in net.modules():
        if isinstance(m, nn.Conv2d):
            conv2d_init(m)
        elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
            norm_layer_init(m, weight_init=1.0)


def Conv3x3(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


def Conv1x1(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, bias=False)


def BN2d(planes, weight_init=1.0):
    bn = nn.BatchNorm2d(planes)
    norm_layer_init(bn, weight_init)
    return bn


class SELayer(nn.Module):
    """Squeeze-and-excitation gate (backbone.py:56-71); used by the Small tier."""

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
        gate = self.fc(self.avg_pool(x).view(b, c)).view(b, c, 1, 1)
        return x * gate.expand_as(x)


class RubiksShiftBlock(nn.Module):
    """BN-ReLU-1x1 -> BN-ReLU-SHIFT -> (SE) -> 1x1, plus shortcut (backbone.py:74-135)."""

    def __init__(self, in_planes, out_planes, *, stride=1, parent):
        super().__init__()
        mid_planes = int(out_planes * parent.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.bn1 = BN2d(in_planes)
        self.conv2 = Conv1x1(in_planes, mid_planes)
        self.bn2 = BN2d(mid_planes)
        self.as3 = RubiksShift2D(
            mid_planes,
            stride=stride,
            normalize_grad=parent.normalize_grad,
            quantize=parent.quantize,
            init_shift=parent.init_shift,
        )
        use_se = parent.use_se
        if use_se:
            if isinstance(use_se, bool):
                reduction = 12
            else:
                assert use_se > 2, ("SE reduction must > 2", use_se)
                reduction = use_se
            self.se = SELayer(mid_planes, reduction=reduction)
        else:
            self.se = None
        self.conv3 = Conv1x1(mid_planes, out_planes)
        if stride != 1 or in_planes != out_planes:
            self.shortcut = Conv1x1(in_planes, out_planes, stride=stride)
        else:
            self.shortcut = nn.Identity()

    def forward(self, x):
        if not self.training:               # inference: BNs and the residual add ride on the two 1x1 GEMMs
            y = fused_eval_block(self, x)
            if y is not None:
                return y
        if isinstance(self.shortcut, nn.Identity):
            # relu(bn(.)) as one operator on GPU tensors (fused_bn.py); the shortcut's gradient joins inside its backward
            out, shortcut = bn_relu_skip(self.bn1, x)
        else:
            out = bn_relu(self.bn1, x)
            shortcut = conv1x1(self.shortcut, out)
        out = bn_relu(self.bn2, conv1x1(self.conv2, out))
        out = self.as3(out)
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
        self.conv1 = Conv3x3(3, self.inplanes, stride=2)

        # (planes multiplier, #blocks, stride of the first block) per stage -- backbone.py:158-165
        self.layer0 = self._make_layer(RubiksShiftBlock, width, 1, stride=1)
        self.layer1 = self._make_layer(RubiksShiftBlock, width, repeats[0], stride=2)
        self.layer2 = self._make_layer(RubiksShiftBlock, 2 * width, repeats[1], stride=2)
        self.layer3 = self._make_layer(RubiksShiftBlock, 4 * width, repeats[2], stride=2)
        self.layer4 = self._make_layer(RubiksShiftBlock, 8 * width, repeats[3], stride=2)

        self.relu = nn.ReLU(inplace=True)
        self.bn_last = BN2d(8 * width)
        self.avgpool = nn.AvgPool2d(7, stride=1)
        self.fc = nn.Linear(8 * width, num_classes)

        conv_bn_init_module(self)
        self.fc.weight.data.normal_(0, 0.01)

    def _make_layer(self, block, planes, repeat, stride):
        blocks = [block(self.inplanes, planes, stride=stride, parent=self)]
        self.inplanes = planes
        blocks += [block(planes, planes, stride=1, parent=self) for _ in range(repeat - 1)]
        return nn.Sequential(*blocks)

    def forward(self, x):
        x = stem_conv(self.conv1, x)
        for stage in (self.layer0, self.layer1, self.layer2, self.layer3, self.layer4):
            x = stage(x)
        x = bn_relu(self.bn_last, x)
        x = self.avgpool(x)
        return self.fc(x.view(x.size(0), -1))

    def get_optim_policy(self, shift_lr_mult=0.01):
        """Parameter groups with per-group lr / weight-decay multipliers (backbone.py:202-235)."""
        weight, bias, bn, shift = [], [], [], []
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Conv3d, nn.Linear)):
                ps = list(m.parameters())
                weight.append(ps[0])
                if len(ps) == 2:
                    bias.append(ps[1])
            elif isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)):
                bn.extend(m.parameters())
            elif isinstance(m, (RubiksShift2D, RubiksShiftBase)):
                shift.extend(m.parameters())
            elif len(m._modules) == 0 and len(list(m.parameters())) > 0:
                raise ValueError("New atomic module type: {}. Need to give it a learning policy".format(type(m)))
        return [
            {"params": weight, "lr_mult": 1, "decay_mult": 1, "name": "weight"},
            {"params": bias, "lr_mult": 1, "decay_mult": 0, "name": "bias"},
            {"params": bn, "lr_mult": 1, "decay_mult": 0, "name": "bn"},
            {"params": shift, "lr_mult": shift_lr_mult, "decay_mult": 0, "name": "shift"},
        ]
