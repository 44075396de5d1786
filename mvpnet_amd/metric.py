"""Segmentation meters and the whole-scene evaluator matrices.

  SegAccuracy, SegIoU           mvpnet/models/metric.py:5-73   (same names, `update_dict` contract, strings)
  Evaluator, CLASS_NAMES, ...   mvpnet/evaluate_3d.py:4-92     (confusion matrix + accuracy / IoU numbers only)

MI355X side: for fp32 logits on the GPU both meters get `argmax -> mask by ignore_index -> bincount` from ONE pass of
`mvp_seg_confusion_f32` over the logits where the network left them ((B,C,N) or channels-last rows); the reference runs
argmax, two boolean-mask compactions, `eq`, `bincount` and a reshape as separate ATen kernels, twice.  The running
confusion matrix of SegIoU stays on the device; nothing is synchronised until a value is read.  Host tensors take the
torch path (the reference's own arithmetic).

The generic host utilities of the reference (`common/utils/metric_logger.py`: AverageMeter, MetricLogger; the evaluator's
table printing) are OUT OF SCOPE and not re-implemented: the reference's own `MetricLogger.add_meter(s)` accepts the two
meters below as they are (they expose `name`, `update_dict`, `__str__`, `summary_str`, `reset`, `avg`, `global_avg`).
Pinned against the imported reference classes by tests/golden/metrics.npz."""
import numpy as np
import torch

from . import _lib as L


def confusion_matrix(seg_logit, seg_label, num_classes=None, ignore_index=-100, out=None):
    """(C,C) int64 matrix [label][argmax logit] over the points whose label is not ignore_index, added to `out` if given.
    seg_logit (B,C,N) in any strides that keep (b,c,n) addressable (e.g. the transposed view of channels-last rows)."""
    B, C, N = seg_logit.shape
    n = C if num_classes is None else num_classes
    mat = out if out is not None else torch.zeros((n, n), dtype=torch.int64, device=seg_logit.device)
    if seg_logit.is_cuda and seg_logit.dtype == torch.float32 and n == C:
        label = seg_label.contiguous()
        sb, sc, sn = seg_logit.stride()
        L.call('mvp_seg_confusion_f32', seg_logit, L.ptr(seg_logit), B, C, N, sb, sc, sn, L.ptr(label), int(ignore_index), L.ptr(mat))
        return mat
    pred = seg_logit.argmax(1)
    keep = seg_label != ignore_index
    mat += torch.bincount(n * seg_label[keep] + pred[keep], minlength=n * n).reshape(n, n)
    return mat


class SegAccuracy(object):
    """Fraction of non-ignored points whose argmax equals the label (metric.py:5-24): printed as
    `<mean over the last 20 updates> (<mean over all updates>)`, both weighted by the number of valid points."""
    name = 'seg_acc'
    WINDOW = 20

    def __init__(self, ignore_index=-100):
        self.ignore_index = ignore_index
        self.reset()

    def reset(self):
        self.recent = []          # (correct, valid) of the last WINDOW updates
        self.sum, self.count = 0.0, 0

    def update_dict(self, preds, labels):
        with torch.no_grad():
            mat = confusion_matrix(preds['seg_logit'], labels['seg_label'], ignore_index=self.ignore_index)
            correct, valid = torch.stack([mat.diagonal().sum(), mat.sum()]).tolist()  # ONE device->host copy
        self.recent = (self.recent + [(correct, valid)])[-self.WINDOW:]
        self.sum += correct
        self.count += valid

    @property
    def avg(self):
        hit = sum(c for c, _ in self.recent)
        tot = sum(v for _, v in self.recent)
        return hit / tot if tot else float('nan')

    @property
    def global_avg(self):
        return self.sum / self.count if self.count else float('nan')

    def __str__(self):
        return '{:.4f} ({:.4f})'.format(self.avg, self.global_avg)

    @property
    def summary_str(self):
        return '{:.4f}'.format(self.global_avg)


class SegIoU(object):
    """Running confusion matrix and per-class IoU = tp / (gt + pred - tp) (metric.py:26-73)."""
    name = 'seg_iou'

    def __init__(self, num_classes, ignore_index=-100):
        self.num_classes = num_classes
        self.ignore_index = ignore_index
        self.mat = None

    def update_dict(self, preds, labels):
        with torch.no_grad():
            if self.mat is None:
                self.mat = torch.zeros((self.num_classes, self.num_classes), dtype=torch.int64, device=preds['seg_logit'].device)
            confusion_matrix(preds['seg_logit'], labels['seg_label'], self.num_classes, self.ignore_index, out=self.mat)

    def reset(self):
        self.mat = None

    @property
    def iou(self):
        h = self.mat.float()
        tp = torch.diag(h)
        return tp / (h.sum(1) + h.sum(0) - tp)

    @property
    def global_avg(self):
        return self.iou.mean().item()

    def __str__(self):
        return '{iou:.4f}'.format(iou=self.iou.mean().item())

    @property
    def summary_str(self):
        return str(self)


# ScanNet v2 benchmark classes (evaluate_3d.py:4-9)
CLASS_NAMES = ['wall', 'floor', 'cabinet', 'bed', 'chair', 'sofa', 'table', 'door',
               'window', 'bookshelf', 'picture', 'counter', 'desk', 'curtain',
               'refridgerator', 'showercurtain', 'toilet', 'sink', 'bathtub', 'otherfurniture']
EVAL_CLASS_IDS = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39]


class Evaluator(object):
    """Whole-scene confusion matrix on host label arrays and the numbers derived from it (evaluate_3d.py:11-92).
    A (gt, pred) pair counts when BOTH sides are one of `labels` -- what sklearn's confusion_matrix(gt, pred, labels=...)
    keeps: ignored ground truth and the "unlabelled" prediction (num_classes) drop out.  The caller's arrays are not
    modified.  The reference's table printing (tabulate) is host formatting and is not part of this build."""

    def __init__(self, class_names, labels=None):
        self.class_names = tuple(class_names)
        self.num_classes = n = len(self.class_names)
        self.labels = np.arange(n) if labels is None else np.asarray(labels)
        if self.labels.shape != (n,):
            raise ValueError('one label id per class expected')
        self._order = np.argsort(self.labels, kind='stable')
        self._sorted = self.labels[self._order]
        self.confusion_matrix = np.zeros((n, n))

    def _position(self, values):
        """index into `labels` of every value, -1 where the value is not a label"""
        v = np.asarray(values).reshape(-1)
        at = np.clip(np.searchsorted(self._sorted, v), 0, self.num_classes - 1)
        return np.where(self._sorted[at] == v, self._order[at], -1)

    def update(self, pred_label, gt_label):
        gt = np.asarray(gt_label).reshape(-1)
        if not (gt >= 0).any():
            print('Invalid label.')
            return
        row, col = self._position(gt), self._position(pred_label)
        both = (row >= 0) & (col >= 0)
        n = self.num_classes
        self.confusion_matrix += np.bincount(row[both] * n + col[both], minlength=n * n).reshape(n, n)

    def batch_update(self, pred_labels, gt_labels):
        if len(pred_labels) != len(gt_labels):
            raise ValueError('one prediction array per ground-truth array expected')
        for p, g in zip(pred_labels, gt_labels):
            self.update(p, g)

    @property
    def overall_acc(self):
        cm = self.confusion_matrix
        return np.trace(cm) / cm.sum()

    @property
    def class_seg_acc(self):
        cm = self.confusion_matrix
        with np.errstate(invalid='ignore', divide='ignore'):
            return list(np.diag(cm) / cm.sum(1))

    @property
    def class_iou(self):
        cm = self.confusion_matrix
        tp = np.diag(cm)
        union = cm.sum(0) + cm.sum(1) - tp
        with np.errstate(invalid='ignore', divide='ignore'):
            return list(np.where(union == 0, np.nan, tp / union))

    @property
    def overall_iou(self):
        return np.nanmean(self.class_iou)
