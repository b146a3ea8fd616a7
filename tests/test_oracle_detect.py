"""oracle/detect.py and the product's anchors/box/roi against vectors produced by the REFERENCE's
own utils/{box,anchors,roi}.py (tests/golden/make_golden.py).  Bit-exact: same torch ops, same order."""
import os

import numpy as np
import pytest
import torch

from oracle import detect
from snn_for_object_detection_amd import box as pbox
from snn_for_object_detection_amd.anchors import AnchorGenerator
from snn_for_object_detection_amd.roi import RoI


def _npz(golden_dir, name):
    return {k: v for k, v in np.load(os.path.join(golden_dir, name)).items()}


def _t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("tag", ["gen1", "tiny"])
def test_anchors_match_reference(golden_dir, tag):
    g = _npz(golden_dir, "detect_anchors.npz")
    sizes, ratios = detect.head_anchor_sizes(3)
    assert torch.equal(sizes, _t(g["sizes"])) and torch.equal(ratios, _t(g["ratios"]))
    oracle_all, product_all = [], []
    for idx, (h, w) in enumerate(g[f"shapes_{tag}"]):
        oracle_all.append(detect.anchor_boxes(int(h), int(w), sizes[idx], ratios))
        product_all.append(AnchorGenerator(sizes[idx].clone(), ratios.clone())(torch.zeros(1, 1, int(h), int(w))))
    want = _t(g[f"anchors_{tag}"])
    assert torch.equal(torch.cat(oracle_all), want)
    assert torch.equal(torch.cat(product_all), want)
    if tag == "gen1":
        assert want.shape == (13545, 4)  # 9 * (30*38 + 15*19 + 8*10), SURVEY section 8a-10


@pytest.mark.parametrize("mod", [detect, pbox], ids=["oracle", "product"])
def test_box_primitives_match_reference(golden_dir, mod):
    g = _npz(golden_dir, "detect_box.npz")
    a, b, offs = _t(g["boxes_a"]), _t(g["boxes_b"]), _t(g["offs"])
    c2c = getattr(mod, "corner_to_center", None) or mod.box_corner_to_center
    c2c_inv = getattr(mod, "center_to_corner", None) or mod.box_center_to_corner
    assert torch.equal(mod.box_iou(a, b), _t(g["iou"]))
    assert torch.equal(c2c(a), _t(g["c2c"]))
    assert torch.equal(c2c_inv(c2c(a)), _t(g["c2c_inv"]))
    assert torch.equal(mod.offset_boxes(a, a.flip(0)), _t(g["offset_boxes"]))
    assert torch.equal(mod.offset_inverse(a, offs), _t(g["offset_inverse"]))


@pytest.mark.parametrize("tag", ["plain", "padded"])
def test_roi_targets_match_reference(golden_dir, tag):
    g = _npz(golden_dir, "detect_roi.npz")
    anchors = _t(_npz(golden_dir, "detect_anchors.npz")["anchors_gen1"])
    labels = _t(g[f"labels_{tag}"])
    thr = float(g["iou_threshold"])
    for name, fn in (("oracle", lambda: detect.roi_targets(anchors, labels.clone(), thr)),
                     ("product", lambda: RoI(thr)(anchors, labels.clone()))):
        off, mask, cls = fn()
        assert torch.equal(off, _t(g[f"offset_{tag}"])), name
        assert torch.equal(mask, _t(g[f"mask_{tag}"])), name
        assert torch.equal(cls, _t(g[f"cls_{tag}"])), name
    if tag == "padded":
        # the reference quirk: padding rows still claim an anchor -> mask 1 with class 0
        cls, mask = _t(g["cls_padded"]), _t(g["mask_padded"])
        assert ((cls == 0) & (mask[..., 0] == 1)).any()


@pytest.mark.parametrize("mod", [detect, pbox], ids=["oracle", "product"])
def test_multibox_detection_matches_reference(golden_dir, mod):
    g = _npz(golden_dir, "detect_nms.npz")
    anchors = _t(_npz(golden_dir, "detect_anchors.npz")["anchors_tiny"])
    det = mod.multibox_detection(_t(g["probs"]).clone(), _t(g["offsets"]).clone(), anchors)
    assert torch.equal(det, _t(g["detections"]))
