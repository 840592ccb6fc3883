"""RubiksNet video model: [N, T, 3, H, W] clips -> class logits.

Counterpart of rubiksnet/models.py:13-145 (same constructor, `load_pretrained`,
`replace_new_fc`, attribute names and state-dict keys).
"""
import os

import torch
import torch.nn as nn

from . import pointwise
from .attention_shift import AttentionShift
from .backbone import RubiksNetBackbone
from .shiftlib import RubiksShift2D, RubiksShift3D
from .utils import make_tuple

__all__ = ["RubiksNet", "TIERS"]

# models.py:28-43
TIERS = {
    "tiny": dict(width=54, repeats=[3, 4, 6, 3], use_se=False),
    "small": dict(width=72, repeats=[3, 4, 6, 3], use_se=True),
    "medium": dict(width=72, repeats=[3, 4, 23, 3], use_se=False),
    "large": dict(width=72, repeats=[3, 8, 36, 3], use_se=False),
}
_STAGES = ("layer0", "layer1", "layer2", "layer3", "layer4")


class RubiksNet(nn.Module):
    def __init__(self, tier, num_classes, num_frames=8, variant="rubiks3d", verbose=True):
        super().__init__()
        assert tier in TIERS
        assert variant in ["rubiks3d", "rubiks3d-aq"]
        self.num_frames = num_frames
        self.tier = tier
        self.variant = variant
        if verbose:
            print(f'Initializing RubiksNet-{tier.capitalize()} variant "{variant}". num_frames={num_frames}')
        self.backbone = RubiksNetBackbone(num_classes=num_classes, **TIERS[tier])
        self._prepare_backbone()
        self.feature_dim = getattr(self.backbone, self.backbone.last_layer_name).in_features
        setattr(self.backbone, self.backbone.last_layer_name, nn.Identity())
        self.new_fc = nn.Linear(self.feature_dim, num_classes)

    @classmethod
    def load_pretrained(cls, ckpt_path):
        """Checkpoint format of the reference (models.py:52-62): dict with tier / num_classes /
        num_frames / variant / model(state_dict)."""
        ckpt = torch.load(os.path.expanduser(ckpt_path), map_location="cpu")
        net = cls(tier=ckpt["tier"], num_classes=ckpt["num_classes"], num_frames=ckpt["num_frames"],
                  variant=ckpt["variant"])
        net.load_state_dict(ckpt["model"])
        return net

    def replace_new_fc(self, num_classes):
        self.new_fc = nn.Linear(self.feature_dim, num_classes)

    def _prepare_backbone(self):
        """Make every block temporal (models.py:67-110): `rubiks3d` swaps as3 for a 3D shift with
        fresh U(-1,1) shifts; `rubiks3d-aq` keeps the 2D shift and prepends AttentionShift to conv2."""
        net = self.backbone
        T = self.num_frames
        for name in _STAGES:
            if not hasattr(net, name):
                continue
            blocks = list(getattr(net, name).children())
            for b in blocks:
                if self.variant == "rubiks3d":
                    b.as3 = _Rubiks3DWrap(b.as3, n_segment=T)
                else:
                    # weights created eagerly from conv2's fan-in: no dummy device forward needed
                    b.conv2 = nn.Sequential(AttentionShift(T, num_channels=b.conv2.in_channels), b.conv2)
            setattr(net, name, nn.Sequential(*blocks))
        net.last_layer_name = "fc"
        self.input_size = 224
        self.input_mean = [0.485, 0.456, 0.406]
        self.input_std = [0.229, 0.224, 0.225]
        net.avgpool = nn.AdaptiveAvgPool2d(1)

    def forward(self, input):
        frames = input.view((-1, 3) + input.size()[-2:])
        if not self.training and frames.is_cuda:
            # (inference: every BatchNorm of the backbone folded to its affine map in one launch, pointwise.prefolded)
            with pointwise.prefolded(self.backbone):
                feats = self.backbone(frames)
        else:
            feats = self.backbone(frames)
        logits = self.new_fc(feats)
        logits = logits.view((-1, self.num_frames) + logits.size()[1:])
        return logits.mean(dim=1, keepdim=True).squeeze(1)

    @property
    def crop_size(self):
        return self.input_size

    @property
    def scale_size(self):
        return self.input_size * 256 // 224


def _temporal_config(spatial, fill):
    """(H, W) setting of a 2-D shift -> (T, H, W) with `fill` on the time axis."""
    return (fill, *make_tuple(spatial, 2))


class _Rubiks3DWrap(nn.Module):
    """Frame-batched adapter: the backbone works on [N*T, C, H, W]; the 3-D shift wants clips
    [N, T, C, H, W].  Both are views of the same memory (models.py:128-145).  Takes the place of a
    block's 2-D shift `as3`, with that layer's channel count and spatial stride / padding, never
    striding or padding time; the 3-D shifts start from their own U(-1, 1) draw.  State-dict key:
    `as3.rubiks3d.shift`."""

    def __init__(self, rubiks2d, n_segment=8):
        super().__init__()
        if not isinstance(rubiks2d, RubiksShift2D):
            raise AssertionError("expected the block's RubiksShift2D, got %s" % type(rubiks2d).__name__)
        self.n_segment = n_segment
        self.rubiks3d = RubiksShift3D(rubiks2d.num_channels,
                                      stride=_temporal_config(rubiks2d.stride, 1),
                                      padding=_temporal_config(rubiks2d.padding, 0))

    def forward(self, x):
        clips = x.unflatten(0, (-1, self.n_segment))          # [N*T, C, H, W] -> [N, T, C, H, W], no copy
        return self.rubiks3d(clips).flatten(0, 1)
