"""Meters and evaluation: same names, constructor arguments, `update_dict` contract and
string formats as the reference, so its training / test loops log the same lines:

  AverageMeter, MetricLogger    common/utils/metric_logger.py:11-107
  SegAccuracy, SegIoU           mvpnet/models/metric.py:5-73
  Evaluator, CLASS_NAMES, ...   mvpnet/evaluate_3d.py:4-92

MI355X side: for fp32 logits on the GPU both segmentation meters get `argmax -> mask by ignore_index -> bincount`
from ONE pass of `mvp_seg_confusion_f32` over the logits where the network left them ((B,C,N) or channels-last rows);
the reference runs argmax, two boolean-mask compactions, `eq`, `bincount` and a reshape as separate ATen kernels, twice.
The running confusion matrix of SegIoU stays on the device; nothing is synchronised until a value is read.
Host tensors take the torch path (the reference's own arithmetic) -- these classes are host glue, not the hot path.
Pinned against the imported reference classes by tests/golden/metrics.npz."""
import collections

import numpy as np
import torch

from . import _lib as L


class AverageMeter(object):
    """Windowed and global average of a stream of (value, count) pairs (metric_logger.py:11-50)."""
    default_fmt = '{avg:.4f} ({global_avg:.4f})'
    default_summary_fmt = '{global_avg:.4f}'

    def __init__(self, window_size=20, fmt=None, summary_fmt=None):
        self.values = collections.deque(maxlen=window_size)
        self.counts = collections.deque(maxlen=window_size)
        self.sum = 0.0
        self.count = 0
        self.fmt = fmt or self.default_fmt
        self.summary_fmt = summary_fmt or self.default_summary_fmt

    def update(self, value, count=1):
        self.values.append(value)
        self.counts.append(count)
        self.sum += value
        self.count += count

    @property
    def avg(self):
        return np.sum(self.values) / np.sum(self.counts)

    @property
    def global_avg(self):
        return self.sum / self.count if self.count != 0 else float('nan')

    def reset(self):
        self.values.clear()
        self.counts.clear()
        self.sum = 0.0
        self.count = 0

    def __str__(self):
        return self.fmt.format(avg=self.avg, global_avg=self.global_avg)

    @property
    def summary_str(self):
        return self.summary_fmt.format(global_avg=self.global_avg)


class MetricLogger(object):
    """name -> meter; every meter implements __str__, summary_str, reset (metric_logger.py:53-107)."""

    def __init__(self, delimiter='\t'):
        self.meters = collections.defaultdict(AverageMeter)
        self.delimiter = delimiter

    def update(self, **kwargs):
        for name, v in kwargs.items():
            if isinstance(v, (torch.Tensor, np.ndarray)):
                count = v.numel() if isinstance(v, torch.Tensor) else v.size
                value = v.item() if count == 1 else v.sum().item()
            else:
                assert isinstance(v, (float, int))
                value, count = v, 1
            self.meters[name].update(value, count)

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def add_meters(self, meters):
        for meter in (meters if isinstance(meters, (list, tuple)) else [meters]):
            self.add_meter(meter.name, meter)

    def __getattr__(self, attr):
        meters = self.__dict__.get('meters', {})
        if attr in meters:
            return meters[attr]
        raise AttributeError(attr)

    def __str__(self):
        return self.delimiter.join('{}: {}'.format(name, str(meter)) for name, meter in self.meters.items())

    @property
    def summary_str(self):
        return self.delimiter.join('{}: {}'.format(name, meter.summary_str) for name, meter in self.meters.items())

    def reset(self):
        for meter in self.meters.values():
            meter.reset()


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


class SegAccuracy(AverageMeter):
    """Fraction of non-ignored points whose argmax equals the label (metric.py:5-24)."""
    name = 'seg_acc'

    def __init__(self, ignore_index=-100):
        super(SegAccuracy, self).__init__()
        self.ignore_index = ignore_index

    def update_dict(self, preds, labels):
        with torch.no_grad():
            mat = confusion_matrix(preds['seg_logit'], labels['seg_label'], ignore_index=self.ignore_index)
            both = torch.stack([mat.diagonal().sum(), mat.sum()]).tolist()  # ONE device->host copy
        self.update(both[0], both[1])


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
    """Whole-scene evaluation on host label arrays (evaluate_3d.py:11-92).  The confusion matrix is what
    sklearn.metrics.confusion_matrix(gt, pred, labels=self.labels) returns: pairs with either side outside
    `labels` (ignored ground truth, the "unlabelled" prediction num_classes) are dropped.  Unlike the reference,
    `update` does not overwrite the -100 entries of the caller's gt array."""

    def __init__(self, class_names, labels=None):
        self.class_names = tuple(class_names)
        self.num_classes = len(class_names)
        self.labels = np.arange(self.num_classes) if labels is None else np.array(labels)
        assert self.labels.shape[0] == self.num_classes
        self.confusion_matrix = np.zeros((self.num_classes, self.num_classes))

    def update(self, pred_label, gt_label):
        gt = np.asarray(gt_label).reshape(-1)
        pred = np.asarray(pred_label).reshape(-1)
        if np.all(gt < 0):
            print('Invalid label.')
            return
        n = self.num_classes
        lut = {int(v): i for i, v in enumerate(self.labels)}
        to_pos = np.vectorize(lambda v: lut.get(int(v), -1), otypes=[np.int64])
        gi, pi = to_pos(gt), to_pos(pred)
        keep = (gi >= 0) & (pi >= 0)
        self.confusion_matrix += np.bincount(gi[keep] * n + pi[keep], minlength=n * n).reshape(n, n)

    def batch_update(self, pred_labels, gt_labels):
        assert len(pred_labels) == len(gt_labels)
        for pred_label, gt_label in zip(pred_labels, gt_labels):
            self.update(pred_label, gt_label)

    @property
    def overall_acc(self):
        return np.sum(np.diag(self.confusion_matrix)) / np.sum(self.confusion_matrix)

    @property
    def overall_iou(self):
        return np.nanmean(self.class_iou)

    @property
    def class_seg_acc(self):
        return [self.confusion_matrix[i, i] / np.sum(self.confusion_matrix[i]) for i in range(self.num_classes)]

    @property
    def class_iou(self):
        cm = self.confusion_matrix
        out = []
        for i in range(self.num_classes):
            union = cm[:, i].sum() + cm[i, :].sum() - cm[i, i]
            out.append(float('nan') if union == 0 else cm[i, i] / union)
        return out

    def print_table(self):
        from tabulate import tabulate
        acc, iou = self.class_seg_acc, self.class_iou
        rows = [[name, acc[i] * 100, iou[i] * 100, int(self.confusion_matrix[i].sum())] for i, name in enumerate(self.class_names)]
        return tabulate(rows, headers=['Class', 'Accuracy', 'IOU', 'Total'], tablefmt='psql', floatfmt='.2f')

    def save_table(self, filename):
        from tabulate import tabulate
        header = ('overall acc', 'overall iou') + self.class_names
        with open(filename, 'w') as f:  # no alignment, to keep one format across runs
            f.write(tabulate([[self.overall_acc, self.overall_iou] + self.class_iou], headers=header, tablefmt='tsv', floatfmt='.5f',
                             numalign=None, stralign=None))
