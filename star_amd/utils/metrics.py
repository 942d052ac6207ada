"""Three-scale occupancy IoU evaluator with the reference's interface (star/utils/metrics.py:128-218:
`Metrics(nbr_classes, num_iterations_epoch)`, `add_batch`, `update_IoU`, `get_average_IoU`,
`get_occupancy_IoU`).  The confusion counts (threshold, two MaxPool3d stages, 2x2 histogram per scale) come
from one HIP kernel pair (star_iou_confusion) instead of numpy on the host; the IoU arithmetic on the
3 x 2 x 2 integers is the reference's, in float64."""
import numpy as np
import torch

from .. import ops


class Metrics:
    def __init__(self, nbr_classes, num_iterations_epoch=0):
        assert nbr_classes == 2, "occupancy evaluation is binary"
        self.nbr_classes = nbr_classes
        self.scales = ['1', '2', '4']
        self.conf = {s: np.zeros((2, 2), dtype=np.int64) for s in self.scales}
        self.every_batch_IoU = []
        self.best_metric_record = {'mIoU': 0, 'IoU': 0, 'epoch': 0, 'loss': 99999999}

    def add_batch(self, prediction, target):
        """prediction, target: float [N, Z, H, W] (any device; moved to the GPU for counting)."""
        dev = prediction.device if prediction.is_cuda else (target.device if target.is_cuda else torch.device("cuda"))
        conf = ops.iou_confusion(prediction.detach().to(dev).float(), target.detach().to(dev).float()).cpu().numpy()
        for i, s in enumerate(self.scales):
            self.conf[s] += conf[i]

    def get_occupancy_IoU(self, scale):
        c = self.conf[scale]
        tp, fp, fn = np.sum(c[1:, 1:]), np.sum(c[1:, 0]), np.sum(c[0, 1:])
        return tp / (tp + fp + fn + 1e-15)

    def update_IoU(self):
        self.every_batch_IoU.append(np.asarray([self.get_occupancy_IoU(s) for s in self.scales]))
        self.reset_evaluator()

    def get_average_IoU(self):
        return np.mean(np.asarray(self.every_batch_IoU), axis=0)

    def reset_evaluator(self):
        for s in self.scales:
            self.conf[s][:] = 0

    def update_best_metric_record(self, mIoU, IoU, loss, epoch):
        self.best_metric_record.update(mIoU=mIoU, IoU=IoU, loss=loss, epoch=epoch)
