"""Box utilities (mirror of the reference's ``utils/box.py``): IoU, offsets, NMS, detection decode.

Tail of the step (tiny tensors: ``[B, A, C+1]`` / ``[B, A, 4]``).  Target-side helpers are torch tensor ops;
``multibox_detection`` on device tensors runs the HIP decode + greedy-NMS kernels of ``csrc/detect.hip`` (SURVEY
section 8f rank 2), the loop below is the host form the fixtures pin.
"""

import torch


def box_corner_to_center(boxes: torch.Tensor) -> torch.Tensor:
    """(x1, y1, x2, y2) -> (cx, cy, w, h)."""
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    return torch.stack(((x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1), dim=-1)


def box_center_to_corner(boxes: torch.Tensor) -> torch.Tensor:
    """(cx, cy, w, h) -> (x1, y1, x2, y2)."""
    cx, cy, w, h = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    return torch.stack((cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h), dim=-1)


def box_iou(boxes1: torch.Tensor, boxes2: torch.Tensor) -> torch.Tensor:
    """Pairwise IoU ``[len(boxes1), len(boxes2)]`` of corner boxes (utils/box.py:31-59)."""
    assert boxes1.shape == (boxes1.shape[0], 4), "Wrong box shape"
    assert boxes2.shape == (boxes2.shape[0], 4), "Wrong box shape"
    area1 = torch.prod(boxes1[:, 2:] - boxes1[:, :2], dim=1)
    area2 = torch.prod(boxes2[:, 2:] - boxes2[:, :2], dim=1)
    top_left = torch.max(boxes1[:, None, :2], boxes2[:, :2])
    bottom_right = torch.min(boxes1[:, None, 2:], boxes2[:, 2:])
    overlap = torch.prod(torch.clamp(bottom_right - top_left, min=0), dim=2)
    return overlap / (area1[:, None] + area2 - overlap)


def offset_boxes(anchors: torch.Tensor, assigned_bb: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """Regression targets: ``10*dxy/wh_a`` and ``5*log(eps + wh/wh_a)`` (utils/box.py:62-69)."""
    anc = box_corner_to_center(anchors)
    tgt = box_corner_to_center(assigned_bb)
    d_xy = 10 * (tgt[:, :2] - anc[:, :2]) / anc[:, 2:]
    d_wh = 5 * torch.log(eps + tgt[:, 2:] / anc[:, 2:])
    return torch.cat([d_xy, d_wh], dim=1)


def offset_inverse(anchors: torch.Tensor, offset_preds: torch.Tensor) -> torch.Tensor:
    """Decode predicted offsets back to corner boxes (utils/box.py:72-79)."""
    anc = box_corner_to_center(anchors)
    xy = (offset_preds[:, :2] * anc[:, 2:] / 10) + anc[:, :2]
    wh = torch.exp(offset_preds[:, 2:] / 5) * anc[:, 2:]
    return box_center_to_corner(torch.cat((xy, wh), dim=1))


def nms(boxes, scores, class_id, num_classes, iou_threshold) -> torch.Tensor:
    """Per-class greedy NMS; returns kept indices in class-then-score order (utils/box.py:82-99)."""
    kept = []
    for cls in range(num_classes - 1):
        members = torch.nonzero(class_id == cls).squeeze(dim=1)
        cls_boxes = boxes[members]
        order = torch.argsort(scores[members], descending=True)
        while order.numel() > 0:
            best = order[0]
            kept.append(members[best])
            if order.numel() == 1:
                break
            iou = box_iou(cls_boxes[best, :].reshape(-1, 4), cls_boxes[order[1:], :].reshape(-1, 4)).reshape(-1)
            order = order[torch.nonzero(iou <= iou_threshold).reshape(-1) + 1]
    return torch.tensor(kept, device=boxes.device, dtype=torch.long)


def _multibox_detection_device(cls_probs: torch.Tensor, offset_preds: torch.Tensor, anchors: torch.Tensor,
                               nms_threshold: float, pos_threshold: float) -> torch.Tensor:
    """Same result as the loop below, on the device and without host synchronisation: decode and greedy NMS are
    HIP kernels (``csrc/detect.hip``), ordering uses torch's device sort / prefix sums as plumbing."""
    from . import _hip
    B, A, K = cls_probs.shape
    dev = cls_probs.device
    st = torch.cuda.current_stream().cuda_stream
    probs = cls_probs.contiguous().float()
    offs = offset_preds.contiguous().float()
    anc = anchors.contiguous().float()
    classes = torch.arange(K, device=dev, dtype=torch.int32)
    out = []
    for b in range(B):
        conf = torch.empty(A, device=dev, dtype=torch.float32)
        cls = torch.empty(A, device=dev, dtype=torch.int32)
        boxes = torch.empty(A, 4, device=dev, dtype=torch.float32)
        _hip.call("snn_detect_decode", probs[b].data_ptr(), offs[b].data_ptr(), anc.data_ptr(), A, K, conf.data_ptr(),
                  cls.data_ptr(), boxes.data_ptr(), st)
        # (class ascending, confidence descending); background (class -1) sorts first and is not a candidate
        key = (cls + 1).double() * 2.0 - conf.double()
        order = torch.argsort(key, stable=True).to(torch.int32)
        counts = (cls.unsqueeze(1) + 1 == classes.unsqueeze(0)).sum(0)          # [K]: background, class 0, ...
        seg = torch.cumsum(counts, 0).to(torch.int32).contiguous()              # seg[c] = first member of class c
        kept = torch.empty(A, device=dev, dtype=torch.int32)
        nkept = torch.zeros(K - 1, device=dev, dtype=torch.int32)
        flag = torch.zeros(A, device=dev, dtype=torch.uint8)
        rank = torch.zeros(A, device=dev, dtype=torch.int32)
        _hip.call("snn_nms_sorted", boxes.data_ptr(), order.data_ptr(), seg.data_ptr(), K - 1, float(nms_threshold),
                  kept.data_ptr(), nkept.data_ptr(), flag.data_ptr(), rank.data_ptr(), st)
        # output order (utils/box.py:134-141): kept rows in class-then-score order, then the rest by anchor index
        is_kept = flag.bool()
        class_off = torch.cumsum(nkept, 0) - nkept
        pos_kept = class_off[cls.clamp(min=0).long()].long() + rank.long()
        not_kept = (~is_kept).long()
        pos_rest = nkept.sum().long() + torch.cumsum(not_kept, 0) - not_kept
        pos = torch.where(is_kept, pos_kept, pos_rest)
        weak = conf < pos_threshold
        class_out = torch.where(is_kept & ~weak, cls, torch.full_like(cls, -1)).float()
        conf_out = torch.where(weak, 1 - conf, conf)
        rows = torch.cat((class_out.unsqueeze(1), conf_out.unsqueeze(1), boxes), dim=1)
        res = torch.empty_like(rows)
        res.index_copy_(0, pos, rows)
        out.append(res)
    return torch.stack(out)


def multibox_detection(cls_probs: torch.Tensor, offset_preds: torch.Tensor, anchors: torch.Tensor,
                       nms_threshold: float = 0.1, pos_threshold: float = 0.009999999) -> torch.Tensor:
    """``[B, A, 6]`` rows ``(class, conf, x1, y1, x2, y2)``; suppressed / background rows get class -1
    (utils/box.py:102-153)."""
    if cls_probs.is_cuda:
        return _multibox_detection_device(cls_probs, offset_preds, anchors, nms_threshold, pos_threshold)
    device = cls_probs.device
    _, num_anchors, num_classes = cls_probs.shape
    out = []
    for cls_prob, offset_pred in zip(cls_probs, offset_preds):
        conf, class_id = torch.max(cls_prob, 1)
        predicted_bb = offset_inverse(anchors, offset_pred)
        class_id -= 1
        keep = nms(predicted_bb, conf, class_id, num_classes, nms_threshold)
        every = torch.arange(num_anchors, dtype=torch.long, device=device)
        uniques, counts = torch.cat((keep, every)).unique(return_counts=True)
        non_keep = uniques[counts == 1]
        order = torch.cat((keep, non_keep))
        class_id[non_keep] = -1
        class_id = class_id[order]
        conf, predicted_bb = conf[order], predicted_bb[order]
        weak = conf < pos_threshold
        class_id[weak] = -1
        conf[weak] = 1 - conf[weak]
        out.append(torch.cat((class_id.unsqueeze(1), conf.unsqueeze(1), predicted_bb), dim=1))
    return torch.stack(out)
