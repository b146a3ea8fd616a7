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
    # (behind a stage-entry LIF the fused convolution reads saved potentials: snn_conv1x1_spikes_fwd, same position of ldy)
    fwd = [a for nm, a in calls if (nm == "snn_conv2d_fwd" and a[14] == 1 and a[15] == 1 and a[12] == c and a[9] == c)
           or (nm == "snn_conv1x1_spikes_fwd" and a[9] == c and a[10] == c)]
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


# ------------------------------------------------------------------------------------------- spikes that are never stored
@pytest.mark.parametrize("Cin,Cout,N,H,W,ld", [(64, 64, 6, 24, 40, 64), (128, 128, 4, 15, 19, 128), (256, 256, 3, 8, 10, 256),
                                               (32, 96, 5, 9, 13, 32), (64, 32, 2, 30, 38, 96)],
                         ids=["64-64", "128-128", "256-256-small", "32-96-odd", "64-32-sliced"])
def test_spike_kernels_equal_the_plain_kernels_on_stored_spikes(hip_lib, Cin, Cout, N, H, W, ld):
    """snn_conv1x1_spikes_fwd / _wgrad read saved potentials and threshold on load (two MFMA products: the low piece of a
    spike is zero): bit for bit what snn_conv2d_fwd / snn_conv2d_wgrad give on the stored spike tensor."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from snn_for_object_detection_amd import _hip
    g = torch.Generator().manual_seed(11)
    v_th = 1.0
    vdec_wide = (1.0 + 0.8 * torch.randn(N, H, W, ld, generator=g)).cuda()
    vdec_wide[0, 0, 0, :4] = torch.tensor([1.0, 1.0 + 2 ** -23, 1.0 - 2 ** -24, 0.0])   # exactly at / next to the threshold
    z = (vdec_wide > v_th).float()
    w = (0.2 * torch.randn(Cout, Cin, generator=g)).cuda()
    dy = torch.randn(N, H, W, Cout, generator=g).cuda()
    st = torch.cuda.current_stream().cuda_stream
    assert _hip.query("snn_conv1x1_spikes_supported", N, H, W, Cin, Cout, ld, _hip.PREC_FP16X3, _hip.PREC_BF16X3)
    y_ref, y_new = torch.empty(N, H, W, Cout, device="cuda"), torch.empty(N, H, W, Cout, device="cuda")
    _hip.call("snn_conv2d_fwd", z.data_ptr(), ld, w.data_ptr(), None, y_ref.data_ptr(), Cout, N, H, W, Cin, H, W, Cout, 1, 1,
              1, 0, None, 0, None, 0, None, _hip.PREC_FP16X3, st)
    _hip.call("snn_conv1x1_spikes_fwd", vdec_wide.data_ptr(), ld, v_th, w.data_ptr(), y_new.data_ptr(), Cout, N, H, W, Cin,
              Cout, st)
    assert torch.equal(y_new, y_ref)
    rel = float((y_ref.double() - torch.einsum("nhwc,oc->nhwo", z[..., :Cin].double(), w.double())).norm()
                / y_ref.double().norm())
    assert rel < 1e-6, rel
    splitk = _hip.query("snn_conv2d_wgrad_splitk", N, H, W, Cin, H, W, Cout, 1, 1, 1, 0, _hip.PREC_BF16X3)
    ws = torch.empty(splitk, Cout * Cin, device="cuda")
    g_ref, g_new = torch.empty(Cout, Cin, device="cuda"), torch.full((Cout, Cin), 0.5, device="cuda")
    _hip.call("snn_conv2d_wgrad", z.data_ptr(), ld, dy.data_ptr(), Cout, g_ref.data_ptr(), N, H, W, Cin, H, W, Cout, 1, 1, 1,
              0, 0, ws.data_ptr(), splitk, _hip.PREC_BF16X3, st)
    _hip.call("snn_conv1x1_spikes_wgrad", vdec_wide.data_ptr(), ld, v_th, dy.data_ptr(), Cout, g_new.data_ptr(), N, H, W,
              Cin, Cout, 1, ws.data_ptr(), splitk, st)                     # accumulate onto 0.5
    assert torch.equal(g_new, g_ref + 0.5)
    # refusals: arithmetic / shapes the thresholding kernels do not cover
    assert not _hip.query("snn_conv1x1_spikes_supported", N, H, W, Cin + 8, Cout, ld + 8, _hip.PREC_FP16X3, _hip.PREC_BF16X3)
    assert not _hip.query("snn_conv1x1_spikes_supported", N, H, W, Cin, Cout, ld, _hip.PREC_FP32, _hip.PREC_FP32)
    with pytest.raises(RuntimeError, match="negative threshold"):
        _hip.call("snn_conv1x1_spikes_fwd", vdec_wide.data_ptr(), ld, -0.5, w.data_ptr(), y_new.data_ptr(), Cout, N, H, W, Cin,
                  Cout, st)


def test_stage_entry_lif_writes_no_spike_tensor_and_nothing_changes(S):
    """Conv -> Norm -> LIF in front of a C2f split: with USE_SPIKES_FROM_VDEC the scan writes only the potentials it saves
    anyway and the fused sibling convolution thresholds them on load - outputs and every gradient bit for bit those of the
    path that stores the spikes; in exact-fp32 arithmetic (not covered by the thresholding kernels) the spikes are stored."""
    from snn_for_object_detection_amd import _hip
    HF = S.functional
    T, B, H, W, c, n = 4, 2, 24, 40, 64, 2
    blk = _build(S, _c2f(S, c, n), True)
    x = synthetic_events(T, B, H, W, p=0.3, seed=1).cuda()
    probe = torch.randn(T, B, c, H // 2, W // 2, generator=torch.Generator().manual_seed(4)).cuda()

    def run(on):
        was = HF.USE_SPIKES_FROM_VDEC
        HF.USE_SPIKES_FROM_VDEC = on
        calls = []

        class Spy:
            def before(self, name, args):
                calls.append((name, args))

            def after(self, tok):
                pass
        _hip.PROFILER = Spy()
        try:
            return _run(S, blk, x, True, probe), calls
        finally:
            _hip.PROFILER = None
            HF.USE_SPIKES_FROM_VDEC = was
    (y1, gx1, g1), calls1 = run(True)
    (y0, gx0, g0), calls0 = run(False)
    assert torch.equal(y1, y0) and torch.equal(gx1, gx0)
    for k in g0:
        assert torch.equal(g1[k], g0[k]), k
    names1, names0 = [nm for nm, _ in calls1], [nm for nm, _ in calls0]
    assert names1.count("snn_conv1x1_spikes_fwd") == 1 and names1.count("snn_conv1x1_spikes_wgrad") == 1
    assert "snn_conv1x1_spikes_fwd" not in names0
    entry = [a for nm, a in calls1 if nm == "snn_affine_neuron_fwd" and a[7] is None]
    assert len(entry) == 1 and entry[0][18] & _hip.SCAN_SPIKES_FROM_VDEC       # ONE scan without an output tensor
    assert not [a for nm, a in calls0 if nm == "snn_affine_neuron_fwd" and a[7] is None]
    HF.set_forward_precision("fp32")
    HF.set_backward_precision("fp32")
    try:
        (y2, _, _), calls2 = run(True)
    finally:
        HF.set_forward_precision(HF.DEFAULT_FORWARD_PRECISION)
        HF.set_backward_precision(HF.DEFAULT_BACKWARD_PRECISION)
    assert "snn_conv1x1_spikes_fwd" not in [nm for nm, _ in calls2]
    assert float((y2 - y0).abs().max()) < 1e-4


def test_three_sibling_branches_and_bf16_storage(S):
    """Three branch-opening 1x1 convolutions (one of them followed by more layers inside a nested pass-through block is the
    TinyYolo case; here: three plain ones of different widths) fuse as well; the same block in the bf16-storage mode runs the
    fused path on bf16 tensors (own tolerances: one rounding per stored value)."""
    HF = S.functional
    cfg = [S.Conv(32, 3), S.Norm(), S.LIF(), S.Conv(64, 1), S.Dense([[S.Conv(32, 1)], [S.Conv(64, 1)], [S.Conv(32, 1)]]),
           S.Conv(32, 1)]
    fused, plain = _build(S, cfg, True), _build(S, cfg, False)
    plain.load_state_dict(fused.state_dict())
    plans = [m._siblings for m in fused.modules() if isinstance(m, S.BlockGen) and m._siblings]
    assert plans == [[(0, 32, None), (32, 64, None), (96, 32, None)]]
    x = synthetic_events(3, 2, 12, 20, p=0.3, seed=3).cuda()
    probe = torch.randn(3, 2, 32, 12, 20, generator=torch.Generator().manual_seed(6)).cuda()
    y1, gx1, g1 = _run(S, fused, x, True, probe)
    y0, gx0, g0 = _run(S, plain, x, False, probe)
    assert torch.equal(y1, y0) and rel_err(gx1, gx0) < 2e-5
    for k in g0:
        assert rel_err(g1[k], g0[k]) < 5e-5, k
    HF.set_activation_storage("bf16")
    try:
        yb, gxb, gb = _run(S, fused, x, True, probe)
        yp, gxp, gp = _run(S, plain, x, False, probe)
    finally:
        HF.set_activation_storage("fp32")
    assert rel_err(yb.float(), yp.float()) < 1e-2 and rel_err(yb.float(), y0) < 5e-2
    for k in g0:
        assert rel_err(gb[k], gp[k]) < 5e-2, k


SPIKE_CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride       what runs (forward / weight gradient)
    (6, 30, 38, 128, 128, 3, 1),          # halo-resident strip tiles, 128-wide / implicit GEMM (below the halo wgrad's size)
    (140, 30, 38, 64, 64, 3, 1),          # ... 64-wide / halo-resident weight gradient (two products)
    (3, 24, 100, 64, 64, 3, 1),           # rectangles (rows longer than 78 pixels)
    (90, 40, 44, 32, 32, 3, 1),           # the 32-channel tile / halo-resident weight gradient, K-steps split over the waves
    (330, 37, 52, 64, 128, 3, 2),         # stride 2: implicit GEMM forward, halo-resident weight gradient (de-interleaved halo)
    (4, 17, 23, 32, 48, 5, 1),            # 5x5: implicit GEMM both ways
    (5, 12, 19, 96, 36, 3, 1),            # 36 output channels: no halo tile, implicit GEMM
]


@pytest.mark.parametrize("N,H,W,Cin,Cout,k,s", SPIKE_CONV_CASES)
def test_spike_convolutions_of_any_kernel_size_equal_the_plain_kernels_on_stored_spikes(hip_lib, N, H, W, Cin, Cout, k, s):
    """snn_conv2d_spikes_fwd / snn_conv3x3_halo_spikes / snn_conv2d_spikes_wgrad (a LIF layer in front of a PLAIN convolution
    writes no spike tensor either): bit for bit the plain kernels on the stored spikes, statistics partials included."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import ctypes
    from snn_for_object_detection_amd import _hip
    g = torch.Generator().manual_seed(N + H + Cin)
    v_th, pad = 1.0, k // 2
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    vdec = (1.0 + 0.8 * torch.randn(N, H, W, Cin, generator=g)).cuda()
    vdec[0, 0, 0, :4] = torch.tensor([1.0, 1.0 + 2 ** -23, 1.0 - 2 ** -24, 0.0])
    z = (vdec > v_th).float()
    w = (torch.randn(Cout, k, k, Cin, generator=g) / (k * k * Cin) ** 0.5).cuda()
    dy = torch.randn(N, Ho, Wo, Cout, generator=g).cuda()
    st = torch.cuda.current_stream().cuda_stream
    assert _hip.query("snn_conv2d_spikes_supported", N, H, W, Cin, Ho, Wo, Cout, k, k, s, pad, Cin, _hip.PREC_FP16X3,
                      _hip.PREC_BF16X3)
    B = N // 2 if N % 2 == 0 else N          # frames per "timestep" of the statistics
    n_part = _hip.query("snn_conv2d_fwd_bn_partial_size", N, B, Ho, Wo, Cout)
    outs = []
    halo = (k, s) == (3, 1) and _hip.query("snn_conv3x3_halo_supported", N, H, W, Cin, Cout)
    for spikes in (False, True):
        y = torch.full((N, Ho, Wo, Cout), float("nan"), device="cuda")
        part = torch.zeros(max(n_part, 1), device="cuda", dtype=torch.float64)
        lay = (ctypes.c_int * 2)()
        pp = part.data_ptr() if n_part else None
        if halo:
            img = torch.empty(9 * Cout * Cin, device="cuda")
            table = torch.tensor([[0, 0, Cout, Cin]], dtype=torch.int64, device="cuda")
            _hip.call("snn_weight_frag_image_batched", w.data_ptr(), img.data_ptr(), table.data_ptr(), 1,
                      9 * (Cin // 32) * (Cout // 32) * 128, 0, _hip.PREC_FP16X3, st)
            if spikes:
                _hip.call("snn_conv3x3_halo_spikes", vdec.data_ptr(), Cin, v_th, img.data_ptr(), y.data_ptr(), Cout, N, H, W,
                          Cin, Cout, pp, B, lay, st)
            else:
                _hip.call("snn_conv3x3_halo", z.data_ptr(), Cin, img.data_ptr(), y.data_ptr(), Cout, N, H, W, Cin, Cout, None,
                          0, None, 0, pp, B, lay, _hip.PREC_FP16X3, st)
        elif spikes:
            _hip.call("snn_conv2d_spikes_fwd", vdec.data_ptr(), Cin, v_th, w.data_ptr(), y.data_ptr(), Cout, N, H, W, Cin, Ho,
                      Wo, Cout, k, k, s, pad, pp, B, lay, st)
        else:
            _hip.call("snn_conv2d_fwd", z.data_ptr(), Cin, w.data_ptr(), None, y.data_ptr(), Cout, N, H, W, Cin, Ho, Wo, Cout,
                      k, k, s, pad, None, 0, pp, B, lay, _hip.PREC_FP16X3, st)
        outs.append((y, part, (lay[0], lay[1])))
    assert torch.isfinite(outs[0][0]).all() and torch.equal(outs[0][0], outs[1][0])
    assert outs[0][2] == outs[1][2] and torch.equal(outs[0][1], outs[1][1])
    ref = torch.nn.functional.conv2d(z.permute(0, 3, 1, 2).double().cpu(), w.permute(0, 3, 1, 2).double().cpu(), stride=s,
                                     padding=pad).permute(0, 2, 3, 1)
    assert rel_err(outs[1][0], ref) < 2e-6
    splitk = _hip.query("snn_conv2d_wgrad_splitk", N, H, W, Cin, Ho, Wo, Cout, k, k, s, pad, _hip.PREC_BF16X3)
    ws = torch.empty(splitk, Cout * k * k * Cin, device="cuda")
    g_ref, g_new = torch.empty(Cout, k, k, Cin, device="cuda"), torch.full((Cout, k, k, Cin), 0.25, device="cuda")
    _hip.call("snn_conv2d_wgrad", z.data_ptr(), Cin, dy.data_ptr(), Cout, g_ref.data_ptr(), N, H, W, Cin, Ho, Wo, Cout, k, k, s,
              pad, 0, ws.data_ptr(), splitk, _hip.PREC_BF16X3, st)
    _hip.call("snn_conv2d_spikes_wgrad", vdec.data_ptr(), Cin, v_th, dy.data_ptr(), Cout, g_new.data_ptr(), N, H, W, Cin, Ho,
              Wo, Cout, k, k, s, pad, 1, ws.data_ptr(), splitk, st)
    assert torch.equal(g_new, g_ref + 0.25)


def test_plain_convolution_stack_writes_no_spike_tensor_between_its_layers(S):
    """Conv -> Norm -> LIF -> Conv -> Norm -> LIF -> Conv (the deep backbones of BASELINE configs[4]): with
    USE_SPIKES_FROM_VDEC the hidden LIF layers write potentials only and the next convolution thresholds on load - outputs and
    every gradient bit for bit those of the path that stores the spikes."""
    from snn_for_object_detection_amd import _hip
    HF = S.functional
    T, B, H, W, c = 4, 2, 36, 44, 64
    torch.manual_seed(7)
    cfg = [S.Conv(c, 3, 2), S.Norm(), S.LIF(), S.Conv(c, 3), S.Norm(), S.LIF(), S.Conv(c, 5), S.Norm(), S.LIF(),
           S.Conv(2 * c, 3, 2), S.Norm(), S.LIF()]
    blk = S.BlockGen(2, [cfg]).cuda().train()
    x = synthetic_events(T, B, H, W, p=0.3, seed=1).cuda()
    probe = torch.randn(T, B, 2 * c, H // 4, W // 4, generator=torch.Generator().manual_seed(4)).cuda()

    def run(on):
        was = HF.USE_SPIKES_FROM_VDEC
        HF.USE_SPIKES_FROM_VDEC = on
        calls = []

        class Spy:
            def before(self, name, args):
                calls.append((name, args))

            def after(self, tok):
                pass
        _hip.PROFILER = Spy()
        try:
            blk.zero_grad(set_to_none=True)
            y, _ = blk(x)
            (y * probe).sum().backward()
            torch.cuda.synchronize()
            return y.detach().clone(), {k: p.grad.detach().clone() for k, p in blk.named_parameters()}, calls
        finally:
            _hip.PROFILER = None
            HF.USE_SPIKES_FROM_VDEC = was
    y1, g1, calls1 = run(True)
    y0, g0, calls0 = run(False)
    assert float(y0.abs().sum()) > 0 and torch.equal(y1, y0)
    for k in g0:
        assert torch.equal(g1[k], g0[k]), k
    names1, names0 = [nm for nm, _ in calls1], [nm for nm, _ in calls0]
    assert names1.count("snn_conv3x3_halo_spikes") + names1.count("snn_conv2d_spikes_fwd") == 3
    assert names1.count("snn_conv2d_spikes_wgrad") == 3
    assert not any("spikes" in nm for nm in names0)
    no_out = [a for nm, a in calls1 if nm == "snn_affine_neuron_fwd" and a[7] is None]
    assert len(no_out) == 3 and all(a[18] & _hip.SCAN_SPIKES_FROM_VDEC for a in no_out)   # the last LIF feeds nobody here
