"""Device detection decode: HIP decode + greedy NMS kernels behind ``box.multibox_detection`` and
``SODa.predict``, against the reference-generated goldens and the host path."""
import os

import pytest
import torch

from tests.util import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg(hip_lib):
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import snn_for_object_detection_amd as p
    return p


# ------------------------------------------------------------------------------------------- detection decode
def _rows_equal_up_to_ties(det, ref, atol=1e-5):
    """Class and confidence columns must match row by row (that IS the output order); rows with identical
    (class, confidence) may be permuted - the reference orders ties with an unstable sort (utils/box.py:88)."""
    assert det.shape == ref.shape
    assert torch.equal(det[..., 0], ref[..., 0])
    assert torch.allclose(det[..., 1], ref[..., 1], rtol=0, atol=atol)
    for d, r in zip(det, ref):
        def canon(t):
            t = t.double()
            order = list(range(t.shape[0]))
            order.sort(key=lambda i: (round(float(t[i, 0])), round(float(t[i, 1]), 5), round(float(t[i, 2]), 4),
                                      round(float(t[i, 3]), 4)))
            return t[order]
        assert torch.allclose(canon(d)[:, 2:], canon(r)[:, 2:], rtol=1e-5, atol=atol)


def _npz(name):
    import os

    import numpy as np
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name))


@pytest.mark.parametrize("fixture,anchor_key", [("detect_nms.npz", None), ("detect_nms_mid.npz", "anchors")])
def test_device_nms_decode_matches_reference_goldens(pkg, fixture, anchor_key):
    """multibox_detection on device tensors = HIP decode + greedy NMS kernels (csrc/detect.hip); the goldens are
    outputs of the reference's own utils/box.py (tests/golden/make_golden.py)."""
    from snn_for_object_detection_amd import box
    g = _npz(fixture)
    anchors = torch.from_numpy(g[anchor_key] if anchor_key else _npz("detect_anchors.npz")["anchors_tiny"])
    probs, offs, ref = (torch.from_numpy(g[k]) for k in ("probs", "offsets", "detections"))
    det = box.multibox_detection(probs.cuda(), offs.cuda(), anchors.cuda()).cpu()
    _rows_equal_up_to_ties(det, ref)


def test_device_nms_decode_gen1_size_equals_host_path(pkg):
    """GEN1 anchor count (13 545), 2 classes + background, clustered boxes: several hundred kept boxes per class, so
    the kernel's kept list spans several 256-box tiles; compared with the host (torch, CPU) path of the same module."""
    from snn_for_object_detection_amd import box
    torch.manual_seed(21)
    A, K = 13545, 3
    centers = torch.rand(A, 2)
    wh = 0.02 + 0.1 * torch.rand(A, 2)
    anchors = torch.cat([centers - wh / 2, centers + wh / 2], dim=1)
    probs = torch.softmax(4 * torch.randn(2, A, K), dim=2)
    offs = 0.3 * torch.randn(2, A, 4)
    ref = box.multibox_detection(probs.clone(), offs.clone(), anchors)
    det = box.multibox_detection(probs.cuda(), offs.cuda(), anchors.cuda()).cpu()
    n_kept = int((ref[0, :, 0] >= 0).sum())
    assert n_kept > 600, n_kept
    _rows_equal_up_to_ties(det, ref)


def test_predict_streams_without_host_sync_in_decode(pkg):
    """SODa.predict on the device path returns the same detections as the host decode of the same predictions."""
    from snn_for_object_detection_amd import box
    import torch.nn.functional as F
    torch.manual_seed(3)
    m = pkg.TinyYolo(num_classes=2, time_window=0).cuda().eval()
    frame = (torch.rand(2, 64, 96) < 0.1).float().cuda()
    with torch.no_grad():
        det, state = m.predict(frame, None)
        preds, _ = m._forward_impl(frame.unsqueeze(0), None)
        anchors, cls, bbox = preds
        host = box.multibox_detection(F.softmax(cls, dim=2).cpu(), bbox.cpu(), anchors.cpu()).squeeze(0)
    host = host[host[:, 0] >= 0]
    host[:, 2:] = torch.clamp(host[:, 2:], min=0.0, max=1.0)
    assert det.shape == host.shape
    assert torch.equal(det[:, 0].cpu(), host[:, 0]) and torch.allclose(det[:, 1:].cpu(), host[:, 1:], atol=1e-5)


# ------------------------------------------------------------------------------------------- event feed (8f rank 1)
def test_event_batcher_matches_oracle_and_feeds_the_model(pkg):
    """Raw events -> device batch (pinned copy on a side stream + HIP scatter) equals the oracle voxelisation / collate
    bit for bit, and the model consumes the channels-last batch without a layout pass."""
    import numpy as np
    from oracle import events as OE
    T, B, H, W, step = 6, 3, 32, 48, 1000
    rng = np.random.default_rng(5)
    samples_np, samples_t, labels = [], [], []
    for b in range(B):
        n = 4000 + 500 * b
        t0 = 10_000 * (b + 1)
        t = rng.integers(t0 - 500, t0 + (T + 1) * step, n)      # some before t0, some past the window
        x = rng.integers(0, W + 4, n)                            # some past the frame: clipped
        y = rng.integers(0, H, n)
        p = rng.integers(0, 2, n)
        lab = rng.random((b + 1, 5)).astype(np.float32)
        samples_np.append((OE.voxelize(t, x, y, p, t0, step, T, H, W), lab))
        samples_t.append(tuple(torch.from_numpy(v).pin_memory() for v in (t, x, y, p)) + (t0,))
        labels.append(torch.from_numpy(lab))
    X_ref, lab_ref = OE.stack_batch(samples_np)
    batcher = pkg.EventBatcher(T, H, W, step)
    X, lab = batcher(samples_t, labels)
    assert X.shape == (T, B, 2, H, W)
    assert torch.equal(X.cpu(), torch.from_numpy(X_ref))
    assert torch.equal(lab.cpu(), torch.from_numpy(lab_ref))
    from snn_for_object_detection_amd import functional as HF
    assert HF.is_channels_last(X)                               # no NCHW -> NHWC pass in front of the first conv
    torch.manual_seed(1)
    m = pkg.TinyYolo(num_classes=2, time_window=0).cuda().eval()
    with torch.no_grad():
        a = m(X)
        b = m(torch.from_numpy(X_ref).cuda())
    assert all(torch.equal(u, v) for u, v in zip(a, b))


def test_event_batcher_matches_reference_generated_fixture(pkg, golden_dir):
    """``k_events`` / ``EventBatcher`` against the REFERENCE's own voxelisation (``tests/golden/events.npz``: outputs of
    ``utils/datasets.py`` ``parse_data`` / ``_stack_data`` executed unmodified by ``make_golden.py``) - no oracle in
    between: the batcher is handed the raw event stream and the window start the reference derived, and must set exactly
    the cells the reference set (GEN1 and 1 Mpx with x past the frame; the multi-target window), labels padded with -1."""
    import numpy as np
    z = np.load(os.path.join(golden_dir, "events.npz"))

    def case(tag):
        T, shift, step_us, clock, H, W, _ = (int(v) for v in z[f"{tag}_params"])
        ev = tuple(torch.from_numpy(z[f"{tag}_events_{k}"].astype(np.int64)).pin_memory() for k in "txyp")
        return T, shift, step_us, clock, H, W, ev

    for tag in ("st_gen1", "st_1mpx"):
        T, shift, step_us, clock, H, W, ev = case(tag)
        # the window the reference's parse_data chose (datasets.py:408-411): first labelled step at or after
        # start_step + T, moved back by T - time_shift steps
        gt = z[f"{tag}_gt"]
        first = gt[gt[:, 0] >= clock // step_us + T][0, 0]
        t0 = int(first) * step_us - step_us * (T - shift)
        X, lab = pkg.EventBatcher(T, H, W, step_us)([ev + (t0,)], [torch.from_numpy(z[f"{tag}_labels"])])
        got = np.flatnonzero(X[:, 0].contiguous().cpu().numpy().reshape(-1))
        assert np.array_equal(got, z[f"{tag}_nonzero"]), tag
        assert float(X.sum()) == len(z[f"{tag}_nonzero"])            # flags, not counts
        assert torch.equal(lab[0].cpu(), torch.from_numpy(z[f"{tag}_labels"]))
    T, _, step_us, clock, H, W, ev = case("mt")                       # multi-target: bins count from the window's first step
    X = pkg.EventBatcher(T, H, W, step_us)([ev + ((clock // step_us) * step_us,)])
    assert np.array_equal(np.flatnonzero(X[:, 0].contiguous().cpu().numpy().reshape(-1)), z["mt_nonzero"])
    # collate of ragged label lists (one sample without a box): the reference's padded targets
    labs = [torch.from_numpy(z[f"stack_labels_{b}"]) for b in range(3)]
    empty = tuple(torch.zeros(0, dtype=torch.int64) for _ in range(4)) + (0,)
    _, lab = pkg.EventBatcher(3, 4, 6, 1000)([empty] * 3, labs)
    assert torch.equal(lab.cpu(), torch.from_numpy(z["stack_out_targets"]))


def test_roi_assign_kernel_matches_reference_vectors(hip_lib, golden_dir):
    """snn_roi_assign (one block per sample) against vectors produced by the reference's own utils/roi.py: class labels
    and masks bit for bit (plain labels and labels with -1 padding rows: the reference gives every padding ROW an
    anchor too), offsets to the last bit except for the device logf (<= 2 ulp)."""
    import numpy as np
    from snn_for_object_detection_amd.roi import RoI
    z = np.load(os.path.join(golden_dir, "detect_roi.npz"))
    anchors = torch.from_numpy(np.load(os.path.join(golden_dir, "detect_anchors.npz"))["anchors_gen1"]).cuda()
    for tag in ("plain", "padded"):
        off, mask, cls = RoI(float(z["iou_threshold"]))(anchors, torch.from_numpy(z[f"labels_{tag}"]).cuda())
        assert cls.dtype == torch.int64 and torch.equal(cls.cpu(), torch.from_numpy(z[f"cls_{tag}"])), tag
        assert torch.equal(mask.cpu(), torch.from_numpy(z[f"mask_{tag}"])), tag
        want = torch.from_numpy(z[f"offset_{tag}"])
        assert torch.equal(off.cpu()[..., :2], want[..., :2]), tag                      # 10 * dxy / wh: exact
        assert torch.allclose(off.cpu(), want, rtol=3e-7, atol=1e-6), tag               # 5 * log(...)


def test_detection_loss_kernels_match_torch(hip_lib):
    """snn_det_loss_fwd / _bwd against the reference's torch expression (models/soda.py:259-281): loss within 1e-6
    relative, gradients within 1e-5; a sample set without positives gives NaN on both sides."""
    from snn_for_object_detection_amd import functional as HF
    torch.manual_seed(21)
    for B, A, K in ((3, 1357, 3), (2, 4001, 8)):
        cls_p = torch.randn(B, A, K, device="cuda").requires_grad_()
        box_p = torch.randn(B, A, 4, device="cuda").requires_grad_()
        labels = (torch.rand(B, A, device="cuda") < 0.02).long() * torch.randint(1, K, (B, A), device="cuda")
        mask = (labels > 0).float().unsqueeze(-1).repeat(1, 1, 4)
        mask[0, 5] = 1.0                                       # the padded-label quirk: class 0 with a box mask
        offset = torch.randn(B, A, 4, device="cuda") * mask
        ratio = 0.04
        loss = HF.detection_loss(cls_p, box_p, offset, mask, labels, ratio)
        loss.backward()
        cr, br = cls_p.detach().clone().requires_grad_(), box_p.detach().clone().requires_grad_()
        ce = torch.nn.functional.cross_entropy(cr.reshape(-1, K), labels.reshape(-1), reduction="none")
        pos = labels.reshape(-1) > 0
        ref = ce[pos].mean() * ratio + ce[~pos].mean() * (1 - ratio) + torch.nn.functional.l1_loss(br * mask, offset * mask)
        ref.backward()
        assert abs(loss.item() - ref.item()) <= 1e-6 * abs(ref.item())
        assert rel_err(cls_p.grad, cr.grad) < 1e-5 and rel_err(box_p.grad, br.grad) < 1e-5
    none = HF.detection_loss(cls_p.detach(), box_p.detach(), offset * 0, mask * 0, labels * 0, ratio)
    assert torch.isnan(none)


def test_small_gemm_matches_torch(hip_lib):
    from snn_for_object_detection_amd import functional as HF
    torch.manual_seed(22)
    for M, N, K in ((64, 32, 64), (128, 256, 64), (37, 19, 250)):
        a, b = torch.randn(M, K, device="cuda"), torch.randn(K, N, device="cuda")
        want = a.double() @ b.double()
        for ta, tb in ((False, False), (True, False), (False, True), (True, True)):
            am = a.t().contiguous() if ta else a
            bm = b.t().contiguous() if tb else b
            c = torch.full((M, N), float("nan"), device="cuda")
            ct = torch.full((N, M), float("nan"), device="cuda")
            HF._small_gemm(am, ta, bm, tb, c, 0, ct=ct)
            assert rel_err(c, want) < 1e-6, (M, N, K, ta, tb)
            assert torch.equal(ct, c.t())   # the transposed copy of the same launch
            # ... which is, bit for bit, what the transposed product computes (same fmaf chains: the composed 1x1
            # convolution takes the operand of its data gradient from the forward launch)
            ct2 = torch.empty_like(ct)
            HF._small_gemm(bm, not tb, am, not ta, ct2, 0)
            assert torch.equal(ct2, ct)
            HF._small_gemm(am, ta, bm, tb, c, 1)
            assert rel_err(c, 2 * want) < 1e-6
