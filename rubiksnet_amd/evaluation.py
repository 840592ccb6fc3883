"""Multi-view evaluation of a RubiksNet: the arithmetic of the reference's eval script
(scripts/test_models.py:150-176, :30-41) without its dataset / CLI.

A video is scored on `views` = crops x clips views (1 for "1-clip", 3 crops x 2 clips = 6 for "2-clip"); the loader
delivers them stacked along the frame axis, [B, views * T * 3, H, W]; they are folded into the batch, the logits
are averaged over the views of a video, and top-1 / top-5 precision is accumulated over the dataset.
"""
import torch

__all__ = ["accuracy", "fold_views", "video_logits", "AverageMeter", "evaluate"]


def accuracy(output, target, topk=(1,)):
    """Precision@k in percent for each k (scripts/test_models.py:30-41): a sample counts when its label is among the
    k largest logits (ties broken by torch.topk's order, as in the reference)."""
    maxk = max(topk)
    n = target.size(0)
    pred = output.topk(maxk, dim=1, largest=True, sorted=True).indices           # [n, maxk]
    hit = pred.eq(target.view(-1, 1))
    return [hit[:, :k].any(dim=1).float().sum() * (100.0 / n) for k in topk]


def fold_views(data, n_frames, views):
    """[B, views * T * 3, H, W] (loader layout) -> [B * views, T, 3, H, W] (test_models.py:159-162)."""
    B = data.size(0)
    H, W = data.size(-2), data.size(-1)
    frames = data.reshape(-1, 3, H, W)
    assert frames.size(0) == B * views * n_frames, "expected %d frames per video" % (views * n_frames)
    return frames.view(B * views, n_frames, 3, H, W)


def video_logits(model, data, n_frames, views=1):
    """Mean of the per-view logits of each video: [B, num_classes] (test_models.py:163-164)."""
    B = data.size(0)
    out = model(fold_views(data, n_frames, views))
    return out.reshape(B, views, -1).mean(1)


class AverageMeter:
    """Running average weighted by batch size (the reference's meter)."""

    def __init__(self):
        self.sum, self.count = 0.0, 0

    def update(self, val, n=1):
        self.sum += float(val) * n
        self.count += n

    @property
    def avg(self):
        return self.sum / max(self.count, 1)


@torch.no_grad()
def evaluate(model, batches, n_frames=8, views=1):
    """batches: iterable of (data [B, views*T*3, H, W], label [B]).  Returns dict(top1, top5, videos, logits, labels);
    the model is put in eval mode.  With a model wrapped for several GPUs the caller shards `batches`."""
    model.eval()
    top1, top5 = AverageMeter(), AverageMeter()
    all_logits, all_labels = [], []
    for data, label in batches:
        dev = next(model.parameters()).device
        logits = video_logits(model, data.to(dev, non_blocking=True), n_frames, views).float()
        label = label.to(logits.device)
        p1, p5 = accuracy(logits, label, topk=(1, min(5, logits.size(1))))
        top1.update(p1.item(), label.numel())
        top5.update(p5.item(), label.numel())
        all_logits.append(logits.cpu())
        all_labels.append(label.cpu())
    return {"top1": top1.avg, "top5": top5.avg, "videos": top1.count,
            "logits": torch.cat(all_logits) if all_logits else torch.empty(0),
            "labels": torch.cat(all_labels) if all_labels else torch.empty(0, dtype=torch.long)}
