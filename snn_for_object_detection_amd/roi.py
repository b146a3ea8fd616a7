"""Anchor <-> ground-truth assignment (mirror of the reference's ``utils/roi.py``).

Same arithmetic and the same results as ``utils/roi.py:18-109`` (pinned bit-exactly by
``tests/golden/detect_roi.npz``).  Device tensors take ONE HIP kernel launch for the whole batch
(``snn_roi_assign``, ``csrc/targets.hip``: one block per sample, the greedy per-ground-truth phase included); the
tensor form below is the host path the fixtures pin, written without ``nonzero`` / boolean indexing.
"""

import torch

from . import box


class RoI:
    """``RoI(iou_threshold)(anchors[A,4], labels[B,N,5]) -> (bbox_offset, bbox_mask, class_labels)``.

    An anchor takes the ground-truth box of highest IoU when that IoU reaches the threshold; afterwards
    every ground-truth ROW (padding rows of -1 included - a reference quirk kept for parity) greedily
    claims the globally best remaining anchor.  Class 0 is background.
    """

    def __init__(self, iou_threshold=0.5) -> None:
        self.iou_threshold = iou_threshold

    def __call__(self, anchors: torch.Tensor, labels: torch.Tensor):
        if anchors.is_cuda:
            return self._device(anchors, labels)
        offsets, masks, classes = [], [], []
        for label in labels:
            amap = self._assign_anchor_to_box(label[:, 1:], anchors)
            assigned = amap >= 0
            bbox_mask = assigned.float().unsqueeze(-1).repeat(1, 4)
            row = amap.clamp(min=0)
            class_labels = torch.where(assigned, label[row, 0].long() + 1, torch.zeros_like(amap))
            assigned_bb = torch.where(assigned.unsqueeze(-1), label[row, 1:], torch.zeros_like(anchors))
            offsets.append(box.offset_boxes(anchors, assigned_bb) * bbox_mask)
            masks.append(bbox_mask)
            classes.append(class_labels)
        return torch.stack(offsets), torch.stack(masks), torch.stack(classes)

    def _device(self, anchors: torch.Tensor, labels: torch.Tensor):
        from . import _hip
        anchors = anchors.detach().contiguous().float()
        labels = labels.detach().to(anchors.device).contiguous().float()
        B, N, _ = labels.shape
        A = anchors.shape[0]
        dev = anchors.device
        ws = torch.empty(_hip.query("snn_roi_workspace_size", B, A, N), device=dev, dtype=torch.uint8)
        offset = torch.empty((B, A, 4), device=dev, dtype=torch.float32)
        mask = torch.empty((B, A, 4), device=dev, dtype=torch.float32)
        cls = torch.empty((B, A), device=dev, dtype=torch.int64)
        _hip.call("snn_roi_assign", anchors.data_ptr(), labels.data_ptr(), B, A, N, float(self.iou_threshold),
                  ws.data_ptr(), offset.data_ptr(), mask.data_ptr(), cls.data_ptr(),
                  torch.cuda.current_stream().cuda_stream)
        return offset, mask, cls

    def _assign_anchor_to_box(self, ground_truth: torch.Tensor, anchors: torch.Tensor) -> torch.Tensor:
        num_gt = ground_truth.shape[0]
        iou = box.box_iou(anchors, ground_truth)
        best_iou, best_gt = torch.max(iou, dim=1)
        amap = torch.where(best_iou >= self.iou_threshold, best_gt, torch.full_like(best_gt, -1))
        for _ in range(num_gt):
            flat = torch.argmax(iou)
            gt_idx = (flat % num_gt).long()
            anc_idx = (flat / num_gt).long()  # float division + truncation, as the reference does
            amap.index_fill_(0, anc_idx.reshape(1), 0)
            amap.index_add_(0, anc_idx.reshape(1), gt_idx.reshape(1))
            iou.index_fill_(1, gt_idx.reshape(1), -1.0)
            iou.index_fill_(0, anc_idx.reshape(1), -1.0)
        return amap
