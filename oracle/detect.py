"""Oracle restatement of the reference's detection utilities (TEST INFRASTRUCTURE ONLY).

PINNED: every function here is checked bit-for-bit against outputs of the
reference's own files executed in the build container (``tests/golden/
make_golden.py`` -> ``tests/golden/detect_*.npz``; ``tests/test_oracle_detect.py``).

Follows (reference file:line):
* ``anchor_boxes``        - ``utils/anchors.py:46-85`` (``AnchorGenerator._cal_anchors``)
* ``head_anchor_sizes``   - ``models/generator.py:389-401`` (sizes / ratios tables)
* ``box_iou`` ...         - ``utils/box.py:9-79``
* ``assign_anchors``      - ``utils/roi.py:65-109`` (threshold pass + greedy per-GT pass,
  padded ``-1`` label rows are NOT skipped - reference quirk kept on purpose)
* ``roi_targets``         - ``utils/roi.py:18-63``
* ``nms`` / ``multibox_detection`` - ``utils/box.py:82-153``
"""

from typing import List, Tuple

import torch


# --------------------------------------------------------------------------- anchors
def head_anchor_sizes(num_maps: int, per_pixel: int = 3) -> Tuple[torch.Tensor, torch.Tensor]:
    lo, hi = 0.08, 0.75
    sizes = torch.arange(lo, hi, (hi - lo) / (num_maps * per_pixel), dtype=torch.float32)
    return sizes.reshape((-1, per_pixel)), torch.tensor((0.5, 1.0, 2), dtype=torch.float32)


def anchor_boxes(height: int, width: int, sizes: torch.Tensor, ratios: torch.Tensor) -> torch.Tensor:
    """Corner-format normalised anchors ``[height*width*len(sizes)*len(ratios), 4]``."""
    per_pixel = len(sizes) * len(ratios)
    cy = (torch.arange(height) + 0.5) * (1.0 / height)
    cx = (torch.arange(width) + 0.5) * (1.0 / width)
    gy, gx = torch.meshgrid(cy, cx, indexing="ij")
    gy, gx = gy.reshape(-1), gx.reshape(-1)
    half_w = torch.cat([sizes * r for r in ratios]) * height / width
    half_h = torch.cat([sizes / r for r in ratios]) * width / height
    deltas = torch.stack((-half_w, -half_h, half_w, half_h)).T.repeat(height * width, 1) / 2
    centres = torch.stack([gx, gy, gx, gy], dim=1).repeat_interleave(per_pixel, dim=0)
    return centres + deltas


# --------------------------------------------------------------------------- boxes
def corner_to_center(b: torch.Tensor) -> torch.Tensor:
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    return torch.stack(((x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1), dim=-1)


def center_to_corner(b: torch.Tensor) -> torch.Tensor:
    cx, cy, w, h = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    return torch.stack((cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h), dim=-1)


def box_iou(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    assert a.shape == (a.shape[0], 4) and b.shape == (b.shape[0], 4), "Wrong box shape"
    area_a = torch.prod(a[:, 2:] - a[:, :2], dim=1)
    area_b = torch.prod(b[:, 2:] - b[:, :2], dim=1)
    lo = torch.max(a[:, None, :2], b[:, :2])
    hi = torch.min(a[:, None, 2:], b[:, 2:])
    inter = torch.prod(torch.clamp(hi - lo, min=0), dim=2)
    return inter / (area_a[:, None] + area_b - inter)


def offset_boxes(anchors: torch.Tensor, assigned: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    ca, cb = corner_to_center(anchors), corner_to_center(assigned)
    d_xy = 10 * (cb[:, :2] - ca[:, :2]) / ca[:, 2:]
    d_wh = 5 * torch.log(eps + cb[:, 2:] / ca[:, 2:])
    return torch.cat([d_xy, d_wh], dim=1)


def offset_inverse(anchors: torch.Tensor, offsets: torch.Tensor) -> torch.Tensor:
    ca = corner_to_center(anchors)
    xy = (offsets[:, :2] * ca[:, 2:] / 10) + ca[:, :2]
    wh = torch.exp(offsets[:, 2:] / 5) * ca[:, 2:]
    return center_to_corner(torch.cat((xy, wh), dim=1))


# --------------------------------------------------------------------------- targets
def assign_anchors(gt: torch.Tensor, anchors: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    """``[A]`` int64 map anchor -> gt row (or -1)."""
    num_a, num_g = anchors.shape[0], gt.shape[0]
    iou = box_iou(anchors, gt)
    amap = torch.full((num_a,), -1, dtype=torch.long, device=anchors.device)
    best, best_j = torch.max(iou, dim=1)
    keep = best >= iou_threshold
    amap[torch.nonzero(keep).reshape(-1)] = best_j[keep]
    for _ in range(num_g):
        flat = torch.argmax(iou)
        j = (flat % num_g).long()
        a = (flat / num_g).long()  # float division then truncation, as upstream
        amap[a] = j
        iou[:, j] = -1
        iou[a, :] = -1
    return amap


def roi_targets(anchors: torch.Tensor, labels: torch.Tensor, iou_threshold: float):
    """-> ``(bbox_offset[B,A,4], bbox_mask[B,A,4], class_labels[B,A] int64)``."""
    num_a = anchors.shape[0]
    offs: List[torch.Tensor] = []
    masks: List[torch.Tensor] = []
    clss: List[torch.Tensor] = []
    for lab in labels:
        amap = assign_anchors(lab[:, 1:], anchors, iou_threshold)
        mask = (amap >= 0).float().unsqueeze(-1).repeat(1, 4)
        cls = torch.zeros(num_a, dtype=torch.long, device=anchors.device)
        boxes = torch.zeros((num_a, 4), dtype=torch.float32, device=anchors.device)
        pos = torch.nonzero(amap >= 0)
        who = amap[pos]
        cls[pos] = lab[who, 0].long() + 1
        boxes[pos] = lab[who, 1:]
        offs.append(offset_boxes(anchors, boxes) * mask)
        masks.append(mask)
        clss.append(cls)
    return torch.stack(offs), torch.stack(masks), torch.stack(clss)


# --------------------------------------------------------------------------- decode
def nms(boxes, scores, class_id, num_classes, iou_threshold) -> torch.Tensor:
    keep = []
    for c in range(num_classes - 1):
        members = torch.nonzero(class_id == c).squeeze(dim=1)
        mboxes = boxes[members]
        order = torch.argsort(scores[members], descending=True)
        while order.numel() > 0:
            top = order[0]
            keep.append(members[top])
            if order.numel() == 1:
                break
            iou = box_iou(mboxes[top, :].reshape(-1, 4), mboxes[order[1:], :].reshape(-1, 4)).reshape(-1)
            order = order[torch.nonzero(iou <= iou_threshold).reshape(-1) + 1]
    return torch.tensor(keep, device=boxes.device, dtype=torch.long)


def multibox_detection(cls_probs, offset_preds, anchors, nms_threshold: float = 0.1,
                       pos_threshold: float = 0.009999999) -> torch.Tensor:
    """-> ``[B, A, 6]`` rows ``(class, conf, x1, y1, x2, y2)``, kept rows first."""
    _, num_a, num_c = cls_probs.shape
    out = []
    for prob, off in zip(cls_probs, offset_preds):
        conf, cid = torch.max(prob, 1)
        boxes = offset_inverse(anchors, off)
        cid -= 1
        keep = nms(boxes, conf, cid, num_c, nms_threshold)
        everything = torch.arange(num_a, dtype=torch.long, device=prob.device)
        uniq, cnt = torch.cat((keep, everything)).unique(return_counts=True)
        dropped = uniq[cnt == 1]
        order = torch.cat((keep, dropped))
        cid[dropped] = -1
        cid = cid[order]
        conf, boxes = conf[order], boxes[order]
        low = conf < pos_threshold
        cid[low] = -1
        conf[low] = 1 - conf[low]
        out.append(torch.cat((cid.unsqueeze(1), conf.unsqueeze(1), boxes), dim=1))
    return torch.stack(out)
