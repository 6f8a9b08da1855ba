"""On-device counterparts of the reference's metrices/ package (AverageMeter.py:16-27, mIoU.py:15-41, Accuracy.py:13-30)."""
import torch

from .. import functional as HF
from .._lib import call


class AverageMeter:
    """metrices/AverageMeter.py: running mean weighted by the batch size."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.sum, self.count = 0.0, 0

    def update(self, value, n=1):
        self.sum += float(value) * n
        self.count += n

    def __call__(self):
        return self.sum / self.count if self.count else 0.0


class _Counts:
    """Per-batch [area_pred | area_inter | area_target | correct, valid] counters produced by dsrl_seg_metrics, kept on the device
    until the metric is read (one readback per epoch instead of one host-side numpy pass per batch)."""

    def __init__(self, num_classes, ignore_index=255):
        self.num_classes, self.ignore_index = num_classes, ignore_index
        self.batches = []

    def add_logits(self, logits, target):
        logits, ld = HF.pm(logits)
        N, C, H, W = logits.shape
        if target.dtype != torch.uint8:
            target = target.to(torch.uint8)
        target = target.contiguous()
        counts = torch.zeros(3 * C + 2, dtype=torch.int64, device=logits.device)
        call('dsrl_seg_metrics', logits.data_ptr(), ld, target.data_ptr(), N * H * W, C, self.ignore_index, counts.data_ptr(), HF._stream())
        self.batches.append(counts)

    def add_pred(self, pred, target, valid_labels_mask):
        """reference call shape: class maps + boolean mask (device tensors); routed through the same kernel via one-hot scores"""
        tgt = torch.where(valid_labels_mask.bool(), target.long(), torch.full_like(target.long(), self.ignore_index)).to(torch.uint8)
        scores = torch.nn.functional.one_hot(pred.long(), self.num_classes).permute(0, 3, 1, 2).float().contiguous(memory_format=torch.channels_last)
        self.add_logits(scores, tgt)

    def table(self):
        return torch.stack(self.batches).cpu().double() if self.batches else torch.zeros((0, 3 * self.num_classes + 2), dtype=torch.float64)


class mIoU:
    """metrices/mIoU.py: per batch the nan-mean over classes of intersection/union, then the nan-mean over batches, in percent."""

    def __init__(self, num_classes, ignore_index=255):
        self.num_classes = num_classes
        self._c = _Counts(num_classes, ignore_index)

    def reset(self):
        self._c.batches = []

    def update(self, pred, target, valid_labels_mask):
        assert pred.shape == target.shape, "BUG CHECK: 'pred' and 'target' must be of the same shape of (B, H, W)."
        assert len(pred.shape) == 3, "BUG CHECK: 'target' and 'pred' must be (B, H, W) channel-order dimensions."
        self._c.add_pred(pred, target, valid_labels_mask)

    def update_from_logits(self, logits, target):
        self._c.add_logits(logits, target)

    def __call__(self):
        t, C = self._c.table(), self.num_classes
        if t.shape[0] == 0:
            return 0.0
        inter, union = t[:, C:2 * C], t[:, :C] + t[:, 2 * C:3 * C] - t[:, C:2 * C]
        iou = inter / union                                     # 0/0 -> nan: classes absent from a batch are skipped (np.nanmean)
        return float(torch.nanmean(torch.nanmean(iou, dim=1)) * 100.)


class Accuracy:
    """metrices/Accuracy.py: mean over batches of correct / valid pixels, in percent."""

    def __init__(self, num_classes=19, ignore_index=255):
        self._c = _Counts(num_classes, ignore_index)

    def reset(self):
        self._c.batches = []

    def update(self, pred, target, valid_labels_mask):
        assert pred.shape == target.shape, "BUG CHECK: 'pred' and 'target' must be of the same shape of (B, H, W)."
        assert len(pred.shape) == 3, "BUG CHECK: 'target' and 'pred' must be (B, H, W) channel-order dimensions."
        self._c.add_pred(pred, target, valid_labels_mask)

    def update_from_logits(self, logits, target):
        self._c.add_logits(logits, target)

    def __call__(self):
        t, C = self._c.table(), self._c.num_classes
        if t.shape[0] == 0:
            return 0.0
        return float((t[:, 3 * C] / t[:, 3 * C + 1]).mean() * 100.)
