"""A bound for the model-level bench legs: what would a RubiksNet step cost if every kernel ran at its roofline?

The operator bench reports GB/s against the HBM peak; "RubiksNet-Tiny 1 400 clips/s" needs the same kind of
denominator.  `model_bound(net, batch, train)` walks the backbone (rubiksnet/backbone.py:137-200: stem, 5 stages of
RubiksShiftBlocks, BN + pool + fc) and charges every operator that MUST touch HBM -- the convolutions and the shifts;
BatchNorm / ReLU / residual adds are elementwise and count as fused into their neighbours -- with

    bytes  = one read of each operand tensor + one write of the result, at the activations' storage size
    flops  = 2 * Cin * Cout * k*k * output pixels for a convolution (x3 in training: forward, d(input), d(weight))
    t_op   = max(bytes / HBM peak, flops / dense MFMA peak of the compute type)

and sums t_op.  Training backward per operator: d(input) reads d(out) and writes d(in); d(weight) reads d(out) and the
saved input; a shift's backward reads d(out) and its saved input and writes d(in) (12 B/elem); the forward
is charged once (saved activations are the forward's own outputs).  This is a LOWER bound on time (an upper bound on
clips/s): no kernel can beat its roofline, and real steps also launch ~10^2-10^3 kernels.  Peaks: MI355X_MICROARCH.md
(HBM3E 8 TB/s; dense MFMA 157.3 TFLOP/s fp32, 2 516 TFLOP/s bf16).
"""

__all__ = ["model_bound", "HBM_PEAK", "MFMA_PEAK"]

HBM_PEAK = 8.0e12                                  # B/s
MFMA_PEAK = {"f32": 157.3e12, "bf16": 2516.0e12}   # dense FLOP/s (no sparsity)


def _conv(cin, cout, k, pix_in, pix_out, frames, es, train, first=False):
    """(bytes, flops) of one convolution: forward, + d(input) (unless `first`: the clip needs no gradient) + d(weight)."""
    x, y = frames * cin * pix_in * es, frames * cout * pix_out * es
    f = 2.0 * cin * cout * k * k * pix_out * frames
    by, fl = x + y, f
    if train:
        by += (y + x) + (0 if first else (y + x))      # d(weight): read dY, X;  d(input): read dY, write dX
        fl += f + (0 if first else f)
    return by, fl


def _shift(c, pix_in, pix_out, frames, es, train):
    x, y = frames * c * pix_in * es, frames * c * pix_out * es
    by = x + y
    if train:
        by += y + 2 * x                                # read dY, read x, write dX
    return by, 0.0


def model_bound(net, batch, train=True, compute="f32", storage_bytes=4, size=224):
    """Roofline-bound time of one step of `net` (a RubiksNet) on `batch` clips.  Returns a dict with the algorithmic
    HBM bytes, the MFMA flops, `bound_ms` = sum over operators of max(bytes / HBM peak, flops / MFMA peak) and the two
    single-resource totals."""
    bb = net.backbone
    frames = batch * net.num_frames
    es = storage_bytes
    peak = MFMA_PEAK[compute]
    ops = []
    h = size
    ops.append(_conv(bb.conv1.in_channels, bb.conv1.out_channels, 3, h * h, (h // 2) ** 2, frames, es, train, first=True))
    h //= 2
    for stage in (bb.layer0, bb.layer1, bb.layer2, bb.layer3, bb.layer4):
        for blk in stage:
            conv2 = blk.conv2[-1] if hasattr(blk.conv2, "__len__") else blk.conv2
            cin, cmid, cout = conv2.in_channels, conv2.out_channels, blk.conv3.out_channels
            sc = blk.shortcut
            projects = hasattr(sc, "weight")
            stride = int(sc.stride[0]) if projects else 1
            ho = (h - 1) // stride + 1
            if hasattr(blk.conv2, "__len__"):           # -aq variant: the temporal 3-tap filter in front of conv2
                ops.append(_shift(cin, h * h, h * h, frames, es, train))
            ops.append(_conv(cin, cmid, 1, h * h, h * h, frames, es, train))
            ops.append(_shift(cmid, h * h, ho * ho, frames, es, train))
            by, fl = _conv(cmid, cout, 1, ho * ho, ho * ho, frames, es, train)
            by += frames * cout * ho * ho * es * (2 if train else 1)      # the residual operand (and its gradient)
            ops.append((by, fl))
            if projects:
                ops.append(_conv(cin, cout, 1, h * h, ho * ho, frames, es, train))
            h = ho
    feat = net.feature_dim
    ops.append((frames * feat * h * h * es * (3 if train else 1), 0.0))   # bn_last + relu + pooling (one pass each way)
    ops.append(_conv(feat, net.new_fc.out_features, 1, 1, 1, frames, 4, train))
    tot_b = sum(b for b, _ in ops)
    tot_f = sum(f for _, f in ops)
    bound = sum(max(b / HBM_PEAK, f / peak) for b, f in ops)
    return {
        "algorithmic_bytes": tot_b, "mfma_flops": tot_f, "bound_ms": 1e3 * bound,
        "hbm_only_ms": 1e3 * tot_b / HBM_PEAK, "mfma_only_ms": 1e3 * tot_f / peak,
        "peaks": {"hbm_GBps": HBM_PEAK / 1e9, "mfma_TFLOPs": peak / 1e12, "compute": compute},
        "formula": "sum over convolutions and shifts of max(bytes / HBM peak, flops / MFMA peak); one read per operand and "
                   "one write per result, BatchNorm / ReLU / residual adds fused away; training = forward + d(input) + "
                   "d(weight) (rubiksnet_amd/roofline.py)",
    }
