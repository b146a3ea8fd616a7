"""Anchor <-> ground-truth assignment (mirror of the reference's ``utils/roi.py``)."""

import torch

from . import box


class RoI:
    """``RoI(iou_threshold)(anchors[A,4], labels[B,N,5]) -> (bbox_offset, bbox_mask, class_labels)``.

    ``utils/roi.py:18-109``: an anchor takes the ground-truth box of highest IoU when that IoU reaches
    the threshold; afterwards every ground-truth ROW (padding rows of -1 included - a reference quirk
    kept for parity) greedily claims the globally best remaining anchor.  Class 0 is background.
    """

    def __init__(self, iou_threshold=0.5) -> None:
        self.iou_threshold = iou_threshold

    def __call__(self, anchors: torch.Tensor, labels: torch.Tensor):
        num_anchors = anchors.shape[0]
        offsets, masks, classes = [], [], []
        for label in labels:
            amap = self._assign_anchor_to_box(label[:, 1:], anchors)
            bbox_mask = (amap >= 0).float().unsqueeze(-1).repeat(1, 4)
            class_labels = torch.zeros(num_anchors, dtype=torch.long, device=anchors.device)
            assigned_bb = torch.zeros((num_anchors, 4), dtype=torch.float32, device=anchors.device)
            positive = torch.nonzero(amap >= 0)
            gt_row = amap[positive]
            class_labels[positive] = label[gt_row, 0].long() + 1
            assigned_bb[positive] = label[gt_row, 1:]
            offsets.append(box.offset_boxes(anchors, assigned_bb) * bbox_mask)
            masks.append(bbox_mask)
            classes.append(class_labels)
        return torch.stack(offsets), torch.stack(masks), torch.stack(classes)

    def _assign_anchor_to_box(self, ground_truth: torch.Tensor, anchors: torch.Tensor) -> torch.Tensor:
        num_anchors, num_gt = anchors.shape[0], ground_truth.shape[0]
        iou = box.box_iou(anchors, ground_truth)
        amap = torch.full((num_anchors,), -1, dtype=torch.long, device=anchors.device)
        best_iou, best_gt = torch.max(iou, dim=1)
        above = best_iou >= self.iou_threshold
        amap[torch.nonzero(above).reshape(-1)] = best_gt[above]
        for _ in range(num_gt):
            flat = torch.argmax(iou)
            gt_idx = (flat % num_gt).long()
            anc_idx = (flat / num_gt).long()  # float division + truncation, as the reference does
            amap[anc_idx] = gt_idx
            iou[:, gt_idx] = -1
            iou[anc_idx, :] = -1
        return amap
