"""Sibling 1x1 convolutions of one input as ONE convolution (``functional._SiblingConv1x1``,
``generator.BlockGen._plan_siblings``): the two ``Conv(c/2, 1)`` that open the branches of a C2f block
(reference ``models/tiny_yolo.py:84-85``), composed with the ``Conv(c, 1)`` in front (``:76-82``).

The fused form must compute what the separate convolutions compute: forward values bit for bit (same k-ordered
products per output element), gradients to the re-association tolerance of a different summation order; against the CPU
oracle (the reference's per-layer ``nn.Conv2d``) to the model tolerances.  The backward pass must take the zero-copy
route: the gradient of the first part is accumulated INTO the concat-gradient slice next to the second part's.
"""
import pytest
import torch

from tests.util import rel_err, synthetic_events

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S(hip_lib):
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import snn_for_object_detection_amd as pkg
    return pkg


def _c2f(S, c, n):
    half = c // 2

    def rec(k):
        if k == 0:
            return []
        inner = [S.Residual([[S.Conv(), S.Norm(), S.LIF()], [S.Pass()]]), *rec(k - 1)]
        return [S.Dense([inner, [S.Pass()]])]
    return [S.Conv(c, 3, 2), S.Norm(), S.LIF(), S.Conv(c, 1), S.Dense([[S.Conv(half, 1), *rec(n)], [S.Conv(half, 1)]]),
            S.Conv(c, 1)]


def _build(S, cfg, fused, seed=3):
    HF = S.functional
    was = HF.USE_SIBLING_FUSION
    HF.USE_SIBLING_FUSION = fused
    try:
        torch.manual_seed(seed)
        blk = S.BlockGen(2, cfg)
    finally:
        HF.USE_SIBLING_FUSION = was
    return blk.cuda().train()


def _run(S, blk, x, fused, probe):
    HF = S.functional
    was = HF.USE_SIBLING_FUSION
    HF.USE_SIBLING_FUSION = fused
    try:
        for p in blk.parameters():
            p.grad = None
        x = x.clone().requires_grad_()
        out, _ = blk(x)
        (out * probe).sum().backward()
        torch.cuda.synchronize()
        HF.wgrad_stream_sync()
        torch.cuda.synchronize()
    finally:
        HF.USE_SIBLING_FUSION = was
    return out.detach(), x.grad.detach(), {n: p.grad.detach().clone() for n, p in blk.named_parameters()}


@pytest.mark.parametrize("c,n,H,W", [(64, 2, 24, 40), (128, 3, 16, 24), (32, 1, 9, 13)], ids=["c64-n2", "c128-n3", "c32-n1-odd"])
def test_fused_siblings_equal_the_separate_convolutions(S, c, n, H, W):
    T, B = 3, 2
    fused, plain = _build(S, _c2f(S, c, n), True), _build(S, _c2f(S, c, n), False)
    plain.load_state_dict(fused.state_dict())
    dense = [m for m in fused.modules() if isinstance(m, S.BlockGen) and m._siblings]
    assert len(dense) == 1 and not [m for m in plain.modules() if isinstance(m, S.BlockGen) and m._siblings]
    # the two outputs sit side by side: (pass slot at the end of the nested block's slice, second branch right behind)
    assert dense[0]._siblings == [(n * c // 2, c // 2, 1), ((n + 1) * c // 2, c // 2, None)]
    x = synthetic_events(T, B, H, W, p=0.3, seed=1).cuda()
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    probe = torch.randn(T, B, c, Ho, Wo, generator=torch.Generator().manual_seed(4)).cuda()
    calls = []
    from snn_for_object_detection_amd import _hip

    class Spy:
        def before(self, name, args):
            calls.append((name, args))
            return None

        def after(self, tok):
            pass
    _hip.PROFILER = Spy()
    try:
        y1, gx1, g1 = _run(S, fused, x, True, probe)
    finally:
        _hip.PROFILER = None
    y0, gx0, g0 = _run(S, plain, x, False, probe)
    assert torch.equal(y1, y0)                        # forward: the same products in the same order
    assert rel_err(gx1, gx0) < 2e-5
    for k in g0:
        assert rel_err(g1[k], g0[k]) < 5e-5, k
    # ONE forward convolution, ONE data gradient and ONE pixel reduction for the pair - and no gather copy: the joined
    # gradient is a channel slice of the concat gradient (pixel stride = concat width)
    width = (n + 2) * c // 2
    fwd = [a for nm, a in calls if nm == "snn_conv2d_fwd" and a[14] == 1 and a[15] == 1 and a[12] == c and a[9] == c]
    assert len(fwd) == 1 and fwd[0][5] == width      # writes c channels into the concat buffer
    dg = [a for nm, a in calls if nm == "snn_conv2d_dgrad" and a[13] == 1 and a[12] == c and a[9] == c]
    assert len(dg) == 1 and dg[0][1] == width, [a[1] for a in dg]
    assert not [nm for nm, _ in calls if nm in ("snn_copy_channels", "snn_add")]


def test_fused_siblings_with_the_trainer_kept_stacked_weight(S):
    """FlatTrainer composes [w2a; w2b] w1 (and its transpose) in its batched launch: same bits as the per-call products,
    stale products never used."""
    from snn_for_object_detection_amd.trainer import FlatTrainer
    T, B, H, W, c, n = 2, 2, 16, 24, 64, 2
    a, b = _build(S, _c2f(S, c, n), True), _build(S, _c2f(S, c, n), True)
    b.load_state_dict(a.state_dict())
    tr = FlatTrainer(b, lr=1e-2)
    x = synthetic_events(T, B, H, W, p=0.3, seed=2).cuda()
    probe = torch.randn(T, B, c, H // 2, W // 2, generator=torch.Generator().manual_seed(5)).cuda()
    for it in range(3):
        ya, _, ga = _run(S, a, x, True, probe)
        tr.zero_grad()
        yb, _ = b(x)
        (yb * probe).sum().backward()
        tr.synchronize()
        gb = {k: v.clone() for k, v in tr.grads_by_name(b).items()}
        assert torch.equal(ya, yb.detach()), it
        for k in ga:
            assert torch.equal(ga[k], gb[k]), (it, k)
        w2a = [m for m in b.modules() if isinstance(m, S.BlockGen) and m._siblings][0].net[0][0].weight
        if it >= 1:
            assert getattr(w2a, "_snn_sibling_weight", None) is not None     # served from the batched launch by now
        tr.step()
        with torch.no_grad():                      # the untrained twin follows by hand
            for (_, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
                pa.copy_(pb)


def test_siblings_without_a_composed_front_convolution(S):
    """Dense([[Conv(c,1), ...], [Conv(c,1)]]) directly behind a LIF (no 1x1 in front to compose with): the stacked weight
    is the two weights themselves."""
    cfg = [S.Conv(32, 3), S.Norm(), S.LIF(), S.Dense([[S.Conv(16, 1)], [S.Conv(48, 1)]]), S.Conv(8, 1)]
    fused, plain = _build(S, cfg, True), _build(S, cfg, False)
    plain.load_state_dict(fused.state_dict())
    assert [m._siblings for m in fused.modules() if isinstance(m, S.BlockGen) and m._siblings] == [[(0, 16, None), (16, 48, None)]]
    x = synthetic_events(3, 2, 12, 20, p=0.3, seed=3).cuda()
    probe = torch.randn(3, 2, 8, 12, 20, generator=torch.Generator().manual_seed(6)).cuda()
    y1, gx1, g1 = _run(S, fused, x, True, probe)
    y0, gx0, g0 = _run(S, plain, x, False, probe)
    assert torch.equal(y1, y0) and rel_err(gx1, gx0) < 2e-5
    for k in g0:
        assert rel_err(g1[k], g0[k]) < 5e-5, k


def test_data_gradient_accumulates_in_place_over_its_second_addend(hip_lib):
    """What the joined gradient relies on (include/snn_hip.h: "passing the destination itself accumulates in place"): dx
    written over addend2, both a channel slice of a wider buffer - implicit GEMM (1x1), halo-resident 3x3 at 32 / 64 / 128
    channels."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import snn_for_object_detection_amd as pkg
    from snn_for_object_detection_amd import _hip
    HF = pkg.functional
    g = torch.Generator().manual_seed(9)
    for C, H, W, k in ((32, 20, 24, 3), (64, 12, 20, 3), (128, 8, 10, 3), (64, 9, 11, 1)):
        N, wide, off = 6, C + 96, 32
        dy = torch.randn(N, H, W, C, generator=g).cuda()
        wt = (0.1 * torch.randn(C, k, k, C, generator=g)).cuda()        # [Cin][KH][KW][Cout] of the data gradient
        a1 = torch.randn(N, H, W, C, generator=g).cuda()
        buf = torch.randn(N, H, W, wide, generator=g).cuda()
        want_buf = buf.clone()
        ref = torch.empty(N, H, W, C, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        slice_ptr = buf.data_ptr() + 4 * off
        if k == 3 and _hip.query("snn_conv3x3_halo_supported", N, H, W, C, C):
            img = HF._frag_image(wt, C, C, 1, _hip.PREC_BF16X3)
            for dst, ld, add2, ld2 in ((ref.data_ptr(), C, slice_ptr, wide), (slice_ptr, wide, slice_ptr, wide)):
                _hip.call("snn_conv3x3_halo", dy.data_ptr(), C, img.data_ptr(), dst, ld, N, H, W, C, C, a1.data_ptr(), C,
                          add2, ld2, None, 0, None, _hip.PREC_BF16X3, st)
        else:
            for dst, ld, add2, ld2 in ((ref.data_ptr(), C, slice_ptr, wide), (slice_ptr, wide, slice_ptr, wide)):
                _hip.call("snn_conv2d_dgrad", dy.data_ptr(), C, wt.data_ptr(), None, dst, ld, N, H, W, C, H, W, C, k, k, 1,
                          k // 2, a1.data_ptr(), C, add2, ld2, _hip.PREC_BF16X3, st)
        torch.cuda.synchronize()
        want_buf[..., off:off + C] = ref
        assert torch.equal(buf, want_buf), (C, k)      # the slice holds the sum, the rest of the buffer is untouched
