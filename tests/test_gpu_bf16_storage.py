"""The labelled bf16-STORAGE throughput mode (``functional.set_activation_storage("bf16")``; C ABI: ``SNN_PREC_BF16S`` for the
convolutions, ``SNN_SCAN_BF16_STORAGE`` for the Norm + neuron scans, ``snn_bn_stats_bf16`` / ``snn_bn_bwd_apply_bf16``): the
wide activation tensors of a step - convolution outputs, spikes, saved decayed potentials, gradients - are bf16 in HBM,
neuron state and every accumulation stay fp32.  This is the dtype BASELINE configs[1] names.

Opt-in, NOT a parity mode and never a default.  What is asserted here (tolerances measured, then rounded up):
  * every kernel computes the fp32 result OF THE STORED (bf16) OPERANDS and rounds it once to bf16 on the way out:
      - convolutions (forward / data gradient; halo-resident, implicit GEMM, stride-2 one-pass) against torch's CPU
        convolution in fp64 on the bf16-rounded operands: |y - ref| <= 2^-8 |ref| + 2e-3 max|ref| elementwise would be one
        output rounding; asserted as relative L2 error < 3e-3 (RNE to 8 bits: 2^-9 * ~0.6 average);
      - weight gradients (fp32 out) against fp64 on the rounded operands: relative L2 error < 2e-5 (bf16 x bf16
        products are exact in fp32; only the fp32 accumulation order differs);
      - the scans, the statistics and the BatchNorm-backward apply against the fp32 kernels run on the up-converted
        operands: outputs equal the bf16 rounding of the fp32 kernel's outputs BIT FOR BIT (same arithmetic, one rounding),
        the BatchNorm-backward sums agree to 1e-5 (fp32 pre-sums per thread, grouped per launch shape);
      - the event-frame layer likewise (fp32 frames in, bf16 out: bit-equal to the rounded fp32 kernel output);
  * a TinyYolo training step at 32x48, T=4 against the CPU oracle: loss within 6 %, everything finite, and the weight
    gradient of the first layer - the one furthest from the loss, behind 22 spiking layers whose near-threshold neurons
    flip under 8-bit storage - still points the oracle's way: cosine similarity > 0.6 (measured 0.82 and 0.97 on two
    seeds, relative L2 error 0.74 and 0.25; the fp32-storage bf16 mode measures 0.99 / 0.11); the activations between
    the layers really are bf16, the predictions fp32;
  * the default (fp32) mode is untouched afterwards.
"""
import pytest
import torch
import torch.nn.functional as F

from tests.util import make_pair, rel_err, synthetic_events, synthetic_labels

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def H_(hip_lib):
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from snn_for_object_detection_amd import _hip
    return _hip


def _st():
    return torch.cuda.current_stream().cuda_stream


def _image(_hip, src, O, I, flip):
    img = torch.empty(9 * O * I, device="cuda")
    table = torch.tensor([[0, 0, O, I]], dtype=torch.int64, device="cuda")
    _hip.call("snn_weight_frag_image_batched", src.data_ptr(), img.data_ptr(), table.data_ptr(), 1,
              9 * (I // 32) * (O // 32) * 128, flip, _hip.PREC_BF16X3, _st())
    return img


def _ref_conv(x_bf, w, gy_bf, stride, pad):
    """fp64 forward / data gradient / weight gradient of the STORED operands: bf16 activations, bf16-rounded weights (OHWI)."""
    xr = x_bf.double().cpu().permute(0, 3, 1, 2).requires_grad_()
    wr = w.to(BF).double().cpu().permute(0, 3, 1, 2).requires_grad_()
    yr = F.conv2d(xr, wr, stride=stride, padding=pad)
    yr.backward(gy_bf.double().cpu().permute(0, 3, 1, 2))
    return (yr.detach().permute(0, 2, 3, 1), xr.grad.permute(0, 2, 3, 1), wr.grad.permute(0, 2, 3, 1))


CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride   (what the case exercises)
    (6, 30, 38, 128, 128, 3, 1),     # halo-resident, 128-wide tile, strip tiles
    (3, 60, 76, 64, 64, 3, 1),       # halo-resident, 64-wide tile
    (2, 24, 100, 64, 128, 3, 1),     # halo-resident, rectangles (W > 78)
    (5, 33, 41, 32, 32, 3, 1),       # 32 channels: the implicit GEMM (no bf16 form of the direct 3x3 kernel)
    (4, 24, 20, 128, 64, 1, 1),      # 1x1: implicit GEMM, pipelined weight gradient
    (7, 37, 52, 64, 128, 3, 2),      # stride 2: implicit GEMM forward, one-pass data gradient, halo weight gradient
    (3, 15, 19, 256, 128, 3, 1),     # small deep map: pipelined (not halo-resident) 3x3 weight gradient
    (2, 9, 11, 96, 40, 1, 1),        # 40 output channels: partial channel tile forward; its data gradient (K = 40) is refused
]


@pytest.mark.parametrize("N,H,W,Cin,Cout,k,s", CONV_CASES)
def test_bf16_storage_convolutions_against_fp64_of_the_stored_operands(H_, N, H, W, Cin, Cout, k, s):
    _hip = H_
    torch.manual_seed(N + H + W + Cin + Cout + k)
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    x = torch.randn(N, H, W, Cin, device="cuda").to(BF)
    w = torch.randn(Cout, k, k, Cin, device="cuda") / (k * k * Cin) ** 0.5
    gy = torch.randn(N, Ho, Wo, Cout, device="cuda").to(BF)
    y_ref, dx_ref, dw_ref = _ref_conv(x, w, gy, s, pad)
    st = _st()
    # ---- forward
    y = torch.full((N, Ho, Wo, Cout), float("nan"), device="cuda", dtype=BF)
    halo = k == 3 and s == 1 and _hip.query("snn_conv3x3_halo_supported", N, H, W, Cin, Cout) == 1
    if halo:
        _hip.call("snn_conv3x3_halo", x.data_ptr(), Cin, _image(_hip, w, Cout, Cin, 0).data_ptr(), y.data_ptr(), Cout, N, H, W,
                  Cin, Cout, None, 0, None, 0, None, 0, None, _hip.PREC_BF16S, st)
    else:
        _hip.call("snn_conv2d_fwd", x.data_ptr(), Cin, w.data_ptr(), None, y.data_ptr(), Cout, N, H, W, Cin, Ho, Wo, Cout, k, k,
                  s, pad, None, 0, None, 0, None, _hip.PREC_BF16S, st)
    assert torch.isfinite(y.float()).all()
    assert rel_err(y, y_ref) < 3e-3
    # one rounding: nearly every element is the bf16 neighbour of the exact value (fp32 accumulation can flip a tie)
    exact = y_ref.to(BF).float()
    assert (y.float().cpu() != exact).float().mean().item() < 0.02
    # ---- data gradient
    wt = torch.empty(Cin, k, k, Cout, device="cuda")
    _hip.call("snn_weight_transpose", w.data_ptr(), wt.data_ptr(), Cout, k, k, Cin, st)
    dx = torch.full((N, H, W, Cin), float("nan"), device="cuda", dtype=BF)
    if k == 3 and s == 1 and _hip.query("snn_conv3x3_halo_supported", N, H, W, Cout, Cin) == 1:
        _hip.call("snn_conv3x3_halo", gy.data_ptr(), Cout, _image(_hip, wt, Cin, Cout, 1).data_ptr(), dx.data_ptr(), Cin, N, H, W,
                  Cout, Cin, None, 0, None, 0, None, 0, None, _hip.PREC_BF16S, st)
    elif k == 3 and s == 2 and _hip.query("snn_conv3x3_s2_dgrad_supported", N, H, W, Cin, Ho, Wo, Cout) == 1:
        _hip.call("snn_conv3x3_s2_dgrad", gy.data_ptr(), Cout, _image(_hip, wt, Cin, Cout, 1).data_ptr(), dx.data_ptr(), Cin, N,
                  H, W, Cin, Ho, Wo, Cout, None, 0, None, 0, _hip.PREC_BF16S, st)
    elif Cout % 32:
        with pytest.raises(RuntimeError, match="bf16 storage covers"):   # the gathered tensor needs whole 32-channel k-steps
            _hip.call("snn_conv2d_dgrad", gy.data_ptr(), Cout, wt.data_ptr(), None, dx.data_ptr(), Cin, N, H, W, Cin, Ho, Wo,
                      Cout, k, k, s, pad, None, 0, None, 0, _hip.PREC_BF16S, st)
        dx = None
    else:
        _hip.call("snn_conv2d_dgrad", gy.data_ptr(), Cout, wt.data_ptr(), None, dx.data_ptr(), Cin, N, H, W, Cin, Ho, Wo, Cout,
                  k, k, s, pad, None, 0, None, 0, _hip.PREC_BF16S, st)
    if dx is not None:
        assert torch.isfinite(dx.float()).all()
        assert rel_err(dx, dx_ref) < 3e-3
    # ---- weight gradient (fp32 result)
    splitk = _hip.query("snn_conv2d_wgrad_splitk", N, H, W, Cin, Ho, Wo, Cout, k, k, s, pad, _hip.PREC_BF16S)
    ws = torch.empty(splitk * Cout * k * k * Cin, device="cuda")
    dw = torch.full((Cout, k, k, Cin), float("nan"), device="cuda")
    _hip.call("snn_conv2d_wgrad", x.data_ptr(), Cin, gy.data_ptr(), Cout, dw.data_ptr(), N, H, W, Cin, Ho, Wo, Cout, k, k, s,
              pad, 0, ws.data_ptr(), splitk, _hip.PREC_BF16S, st)
    # the stored activations are exact operands; the weight gradient does not involve the weights
    assert rel_err(dw, dw_ref) < 2e-5
    dw2 = torch.empty_like(dw)
    _hip.call("snn_conv2d_wgrad", x.data_ptr(), Cin, gy.data_ptr(), Cout, dw2.data_ptr(), N, H, W, Cin, Ho, Wo, Cout, k, k, s,
              pad, 0, ws.data_ptr(), splitk, _hip.PREC_BF16S, st)
    assert torch.equal(dw, dw2)       # ordered slab reduction: reproducible


def test_bf16_storage_channel_slices_addends_and_statistics(H_):
    """Channel-sliced bf16 operands (the zero-copy Dense merge), the fused addends of the data gradient, and the
    BatchNorm statistics partials out of the forward epilogues against snn_bn_stats_bf16 over the stored output."""
    _hip = H_
    torch.manual_seed(3)
    st = _st()
    for (N, H, W, Cin, Cout, k) in [(4, 15, 19, 64, 128, 3), (4, 15, 19, 64, 96, 1)]:
        pad = k // 2
        xbuf = torch.randn(N, H, W, Cin + 32, device="cuda").to(BF)
        x = xbuf[..., 16:16 + Cin]
        w = torch.randn(Cout, k, k, Cin, device="cuda") / (k * k * Cin) ** 0.5
        ybuf = torch.full((N, H, W, Cout + 64), 7.0, device="cuda", dtype=BF)
        y = ybuf[..., 32:32 + Cout]
        T, fps = 2, N // 2
        n_part = _hip.query("snn_conv2d_fwd_bn_partial_size", N, fps, H, W, Cout)
        partial = torch.zeros(n_part, device="cuda", dtype=torch.float64)
        import ctypes
        layout = (ctypes.c_int * 2)()
        if k == 3:
            _hip.call("snn_conv3x3_halo", x.data_ptr(), Cin + 32, _image(_hip, w, Cout, Cin, 0).data_ptr(), y.data_ptr(),
                      Cout + 64, N, H, W, Cin, Cout, None, 0, None, 0, partial.data_ptr(), fps, layout, _hip.PREC_BF16S, st)
        else:
            _hip.call("snn_conv2d_fwd", x.data_ptr(), Cin + 32, w.data_ptr(), None, y.data_ptr(), Cout + 64, N, H, W, Cin, H, W,
                      Cout, k, k, 1, pad, None, 0, partial.data_ptr(), fps, layout, _hip.PREC_BF16S, st)
        y_ref, _, _ = _ref_conv(x, w, torch.zeros(N, H, W, Cout, device="cuda").to(BF), 1, pad)
        assert rel_err(y, y_ref) < 3e-3
        assert bool((ybuf[..., :32] == 7.0).all()) and bool((ybuf[..., 32 + Cout:] == 7.0).all())
        assert layout[0] > 0
        # the statistics are of the fp32 accumulators (what the epilogue has), the stored tensor is their rounding: the
        # per-(t, c) mean / mean of squares agree with the pass over the stored bf16 tensor to bf16 resolution
        sums = torch.empty(T, Cout, 2, device="cuda", dtype=torch.float64)
        _hip.call("snn_bn_stats_reduce", partial.data_ptr(), int(layout[0]), int(layout[1]), T, fps * H * W, Cout,
                  sums.data_ptr(), st)
        n2 = _hip.query("snn_bn_stats_partial_size", T, fps * H * W, Cout)
        p2 = torch.zeros(n2, device="cuda", dtype=torch.float64)
        _hip.call("snn_bn_stats_bf16", y.data_ptr(), Cout + 64, T, fps * H * W, Cout, p2.data_ptr(), st)
        sums2 = torch.empty_like(sums)
        _hip.call("snn_bn_stats_reduce", p2.data_ptr(), 0, 0, T, fps * H * W, Cout, sums2.data_ptr(), st)
        ref = y.float().double().view(T, fps * H * W, Cout)
        assert torch.allclose(sums2[..., 0], ref.sum(1), rtol=1e-12, atol=1e-9)
        assert torch.allclose(sums2[..., 1], (ref * ref).sum(1), rtol=1e-12, atol=1e-9)
        assert rel_err(sums[..., 1], sums2[..., 1]) < 2e-3
        # data gradient with two addends (their own pixel strides)
        gy = torch.randn(N, H, W, Cout, device="cuda").to(BF)
        a1 = torch.randn(N, H, W, Cin + 8, device="cuda").to(BF)
        a2 = torch.randn(N, H, W, Cin, device="cuda").to(BF)
        wt = torch.empty(Cin, k, k, Cout, device="cuda")
        _hip.call("snn_weight_transpose", w.data_ptr(), wt.data_ptr(), Cout, k, k, Cin, st)
        dx = torch.empty(N, H, W, Cin, device="cuda", dtype=BF)
        if k == 3:
            _hip.call("snn_conv3x3_halo", gy.data_ptr(), Cout, _image(_hip, wt, Cin, Cout, 1).data_ptr(), dx.data_ptr(), Cin, N,
                      H, W, Cout, Cin, a1[..., 4:].data_ptr(), Cin + 8, a2.data_ptr(), Cin, None, 0, None, _hip.PREC_BF16S, st)
        else:
            _hip.call("snn_conv2d_dgrad", gy.data_ptr(), Cout, wt.data_ptr(), None, dx.data_ptr(), Cin, N, H, W, Cin, H, W,
                      Cout, k, k, 1, pad, a1[..., 4:].data_ptr(), Cin + 8, a2.data_ptr(), Cin, _hip.PREC_BF16S, st)
        _, dx_ref, _ = _ref_conv(x.contiguous(), w, gy, 1, pad)
        dx_ref = dx_ref + a1[..., 4:4 + Cin].double().cpu() + a2.double().cpu()
        assert rel_err(dx, dx_ref) < 3e-3


def test_bf16_storage_refusals(H_):
    _hip = H_
    st = _st()
    x = torch.zeros(2, 8, 8, 24, device="cuda", dtype=BF)
    w = torch.zeros(32, 1, 1, 24, device="cuda")
    y = torch.zeros(2, 8, 8, 32, device="cuda", dtype=BF)
    with pytest.raises(RuntimeError, match="multiple of 32"):   # 24 input channels: no bf16-storage kernel
        _hip.call("snn_conv2d_fwd", x.data_ptr(), 24, w.data_ptr(), None, y.data_ptr(), 32, 2, 8, 8, 24, 8, 8, 32, 1, 1, 1, 0,
                  None, 0, None, 0, None, _hip.PREC_BF16S, st)
    with pytest.raises(RuntimeError, match="SNN_PREC_BF16X3 or SNN_PREC_BF16S"):
        _hip.call("snn_conv3x3_s2_dgrad", x.data_ptr(), 64, x.data_ptr(), y.data_ptr(), 64, 2, 16, 16, 64, 8, 8, 64, None, 0,
                  None, 0, _hip.PREC_FP16X3, st)


def test_bf16_storage_event_frame_layer_is_the_rounded_fp32_kernel(H_):
    _hip = H_
    torch.manual_seed(11)
    st = _st()
    N, H, W, Cout = 6, 40, 52, 32
    x = (torch.rand(N, H, W, 2, device="cuda") < 0.1).float()
    w = torch.randn(Cout, 3, 3, 2, device="cuda") / 18 ** 0.5
    y32 = torch.empty(N, H, W, Cout, device="cuda")
    y16 = torch.empty(N, H, W, Cout, device="cuda", dtype=BF)
    _hip.call("snn_conv2d_fwd", x.data_ptr(), 2, w.data_ptr(), None, y32.data_ptr(), Cout, N, H, W, 2, H, W, Cout, 3, 3, 1, 1,
              None, 0, None, 0, None, _hip.PREC_FP16X3, st)
    _hip.call("snn_conv2d_fwd", x.data_ptr(), 2, w.data_ptr(), None, y16.data_ptr(), Cout, N, H, W, 2, H, W, Cout, 3, 3, 1, 1,
              None, 0, None, 0, None, _hip.PREC_BF16S, st)
    assert torch.equal(y16, y32.to(BF))
    gy = torch.randn(N, H, W, Cout, device="cuda").to(BF)
    gy32 = gy.float()
    dws = []
    for g, prec in ((gy32, _hip.PREC_BF16X3), (gy, _hip.PREC_BF16S)):
        splitk = _hip.query("snn_conv2d_wgrad_splitk", N, H, W, 2, H, W, Cout, 3, 3, 1, 1, prec)
        ws = torch.empty(splitk * Cout * 18, device="cuda")
        dw = torch.empty(Cout, 3, 3, 2, device="cuda")
        _hip.call("snn_conv2d_wgrad", x.data_ptr(), 2, g.data_ptr(), Cout, dw.data_ptr(), N, H, W, 2, H, W, Cout, 3, 3, 1, 1, 0,
                  ws.data_ptr(), splitk, prec, st)
        dws.append(dw)
    assert torch.equal(dws[0], dws[1])     # same fmaf chains over the same values


@pytest.mark.parametrize("neuron_name,last_only", [("NONE", False), ("LIF", False), ("LI", False), ("LI_TANH", False),
                                                   ("LIF", True), ("LI", True), ("LI_TANH", True)])
def test_bf16_storage_scans_are_the_rounded_fp32_kernels(H_, neuron_name, last_only):
    _hip = H_
    from snn_for_object_detection_amd import functional as HF
    neuron = getattr(_hip, "NEURON_" + neuron_name)
    torch.manual_seed(17)
    st = _st()
    T, M, C, ldy = 6, 5 * 9 * 7, 48, 64
    params = HF.neuron_params()
    ybuf = (torch.randn(T, M, ldy, device="cuda") * 2).to(BF)
    y = ybuf[..., 8:8 + C]
    alpha = torch.rand(T, C, device="cuda") + 0.5
    beta = torch.randn(T, C, device="cuda") * 0.3
    has_state = neuron != _hip.NEURON_NONE
    saves = neuron == _hip.NEURON_LIF
    flags = _hip.SCAN_LAST_STEP_ONLY if last_only else 0
    res = {}
    for tag, yt, dt, fl in (("f32", ybuf.float()[..., 8:8 + C], torch.float32, flags),
                            ("b16", y, BF, flags | _hip.SCAN_BF16_STORAGE)):
        out = torch.full((M, C) if last_only else (T, M, C), float("nan"), device="cuda", dtype=dt)
        vT = torch.empty(M, C, device="cuda")
        iT = torch.empty(M, C, device="cuda")
        vdec = torch.full((T, M, C), float("nan"), device="cuda", dtype=dt) if saves else None
        _hip.call("snn_affine_neuron_fwd", neuron, yt.data_ptr(), ldy, alpha.data_ptr(), beta.data_ptr(), None, None,
                  out.data_ptr(), C, None, 0, vT.data_ptr() if has_state else None, iT.data_ptr() if has_state else None,
                  None if vdec is None else vdec.data_ptr(), T, M, C, params, fl, st)
        res[tag] = (out, vT, iT, vdec)
    o32, v32, i32, d32 = res["f32"]
    o16, v16, i16, d16 = res["b16"]
    assert torch.equal(o16, o32.to(BF))
    if has_state:
        assert torch.equal(v16, v32) and torch.equal(i16, i32)     # the state never leaves fp32
    if saves:
        assert torch.equal(d16, d32.to(BF))
    # ---- reverse scan on the STORED tensors (both kernels get the same values)
    g_out = (torch.randn(o16.shape, device="cuda")).to(BF)
    state16 = d16 if saves else (o16 if neuron == _hip.NEURON_LI_TANH else None)
    gxs = {}
    for tag, cast, fl in (("f32", lambda t: None if t is None else t.float(), flags),
                          ("b16", lambda t: t, flags | _hip.SCAN_BF16_STORAGE)):
        yt = ybuf.float()[..., 8:8 + C] if tag == "f32" else y
        go, stt = cast(g_out), cast(state16)
        gx = torch.full((T, M, C), float("nan"), device="cuda", dtype=torch.float32 if tag == "f32" else BF)
        n_sums = _hip.query("snn_affine_neuron_bwd_sums_size", T, M, C)
        sums = torch.zeros(n_sums, device="cuda", dtype=torch.float64)
        _hip.call("snn_affine_neuron_bwd", neuron, go.data_ptr(), C, None if stt is None else stt.data_ptr(), yt.data_ptr(), ldy,
                  None, None, alpha.data_ptr(), beta.data_ptr(), 0, gx.data_ptr(), None, None, sums.data_ptr(), T, M, C, params,
                  fl, st)
        raw = torch.empty(T, C, 2, device="cuda", dtype=torch.float64)
        _hip.call("snn_bn_bwd_reduce", sums.data_ptr(), T, M, C, raw.data_ptr(), st)
        gxs[tag] = (gx, raw)
    assert torch.equal(gxs["b16"][0], gxs["f32"][0].to(BF))
    # (the threads pre-add their rows in fp32 before the fp64 tree; the two launches may group rows differently)
    assert torch.allclose(gxs["b16"][1], gxs["f32"][1], rtol=1e-5, atol=1e-5)
    # ---- the BatchNorm-backward apply, in place, accumulate off / on
    coef = torch.randn(3, T, C, device="cuda")
    gx16 = gxs["b16"][0]
    gx32 = gx16.float()
    y32 = ybuf.float()[..., 8:8 + C]
    for accumulate in (0, 1):
        d32_ = torch.randn(T, M, C, device="cuda").to(BF).float()
        d16_ = d32_.to(BF)
        _hip.call("snn_bn_bwd_apply", gx32.data_ptr(), y32.data_ptr(), ldy, coef[0].data_ptr(),
                  coef[1].data_ptr(), coef[2].data_ptr(), d32_.data_ptr(), C, T, M, C, accumulate, st)
        _hip.call("snn_bn_bwd_apply_bf16", gx16.data_ptr(), y.data_ptr(), ldy, coef[0].data_ptr(), coef[1].data_ptr(),
                  coef[2].data_ptr(), d16_.data_ptr(), C, T, M, C, accumulate, st)
        assert torch.equal(d16_, d32_.to(BF))


def test_bf16_storage_training_step_tolerance_and_dtypes(H_):
    import snn_for_object_detection_amd as S
    HF = S.functional
    T, B, H, W = 4, 2, 32, 48
    product, oracle = make_pair(S.TinyYolo, num_classes=2, time_window=0, state_storage=True)
    X, labels = synthetic_events(T, B, H, W, p=0.08), synthetic_labels(B)
    product.train()
    oracle.train()
    loss_ref = oracle.training_step((X, labels))
    loss_ref.backward()
    seen = []
    first = product.base_net.net.net[0]
    hook = first[0].register_forward_hook(lambda m, i, o: seen.append(o.dtype))
    HF.set_activation_storage("bf16")
    try:
        assert HF.get_activation_storage() == "bf16"
        loss = product.training_step((X.cuda(), labels.cuda()))
        loss.backward()
    finally:
        HF.set_activation_storage("fp32")
        hook.remove()
    assert seen and seen[0] == BF                     # the event-frame layer wrote bf16: the step ran in bf16 storage
    assert loss.dtype == torch.float32 and torch.isfinite(loss)
    assert abs(loss.item() - loss_ref.item()) <= 0.06 * abs(loss_ref.item()), (loss.item(), loss_ref.item())
    g, g_ref = first[0].weight.grad, oracle.base_net.net.net[0][0].weight.grad
    assert g.dtype == torch.float32 and torch.isfinite(g).all()
    a, b = g.double().cpu().flatten(), g_ref.double().flatten()
    cos = float((a @ b) / (a.norm() * b.norm()))
    assert cos > 0.6 and rel_err(g, g_ref) < 1.0, (cos, rel_err(g, g_ref))
    for p in product.parameters():
        assert p.grad is None or (p.grad.dtype == torch.float32 and torch.isfinite(p.grad).all())
    # the default mode is untouched: the same model, same batch, fp32 storage again -> the parity-grade loss
    product.zero_grad()
    loss32 = product.training_step((X.cuda(), labels.cuda()))
    assert abs(loss32.item() - loss_ref.item()) <= 1e-4 * abs(loss_ref.item())
    with pytest.raises(ValueError):
        HF.set_activation_storage("fp16")


def test_bf16_storage_with_the_trainer_and_the_other_entry_points(H_):
    """The mode through the rest of the API: FlatTrainer steps (its per-step forward weight image switches to bf16 pieces),
    streaming ``predict`` with carried fp32 state, the literal time-outer loop and the layer-major pass in eval mode (the two
    orders agree to the mode's resolution: they round the same values at the same places, only sums differ in order)."""
    import snn_for_object_detection_amd as S
    from snn_for_object_detection_amd.trainer import FlatTrainer
    HF = S.functional
    _hip = H_
    T, B, H, W = 4, 2, 32, 48
    torch.manual_seed(3)
    model = S.TinyYolo(num_classes=2, time_window=0).cuda().train()
    X, labels = synthetic_events(T, B, H, W, p=0.1).cuda(), synthetic_labels(B).cuda()
    HF.set_activation_storage("bf16")
    try:
        tr = FlatTrainer(model, lr=2e-3)
        losses = []
        for _ in range(6):
            tr.zero_grad()
            loss = model.training_step((X, labels))
            loss.backward()
            tr.step()
            losses.append(loss.item())
        assert all(torch.isfinite(torch.tensor(losses))) and min(losses[1:]) < losses[0]
        assert {getattr(p, "_snn_wfrag_prec", None) for p in model.parameters()} == {_hip.PREC_BF16X3, None}
        model.eval()
        with torch.no_grad():
            state = None
            for t in range(T):
                det, state = model.predict(X[t, 0], state)
            assert det.dtype == torch.float32 and det.dim() == 2 and det.shape[1] == 6
            outer = model(X, time_outer=True)
            major = model(X)
        assert all(t.dtype == torch.float32 for t in outer) and all(t.dtype == torch.float32 for t in major)
        assert float((outer[1] - major[1]).abs().max()) < 5e-3 and float((outer[2] - major[2]).abs().max()) < 5e-3
    finally:
        HF.set_activation_storage("fp32")


def test_bf16_storage_merges_and_the_domain_boundary(H_):
    """Copy / add of (channel-sliced) bf16 tensors and the fp32 <-> bf16 conversion at the boundary of the bf16 domain:
    the sum is formed in fp32 and rounded once (= torch's bf16 add), conversions round to nearest even (= torch's cast)."""
    _hip = H_
    from snn_for_object_detection_amd import functional as HF
    torch.manual_seed(23)
    st = _st()
    M, C = 777, 40
    for C, lda, ldb, ldd, oa, ob, od in ((40, 64, 48, 56, 8, 4, 12), (6, 7, 9, 6, 1, 2, 0)):   # 8-byte accesses / scalar path
        abuf = torch.randn(M, lda, device="cuda").to(BF)
        bbuf = torch.randn(M, ldb, device="cuda").to(BF)
        dbuf = torch.full((M, ldd), 7.0, device="cuda", dtype=BF)
        a, b, d = abuf[:, oa:oa + C], bbuf[:, ob:ob + C], dbuf[:, od:od + C]
        _hip.call("snn_add_bf16", a.data_ptr(), lda, b.data_ptr(), ldb, d.data_ptr(), ldd, M, C, st)
        assert torch.equal(d, a + b)
        assert bool((dbuf[:, :od] == 7.0).all()) and bool((dbuf[:, od + C:] == 7.0).all())
        _hip.call("snn_copy_channels_bf16", b.data_ptr(), ldb, d.data_ptr(), ldd, M, C, st)
        assert torch.equal(d, b)
    x = torch.randn(5, 33, 17, device="cuda") * 3
    x[0, 0, 0], x[0, 0, 1] = float("inf"), 1e-40
    y = torch.empty(x.shape, device="cuda", dtype=BF)
    _hip.call("snn_convert_bf16", x.data_ptr(), y.data_ptr(), x.numel(), 1, st)
    assert torch.equal(y, x.to(BF))
    z = torch.empty_like(x)
    _hip.call("snn_convert_bf16", y.data_ptr(), z.data_ptr(), x.numel(), 0, st)
    assert torch.equal(z, y.float())
    # autograd form: the gradient crosses the boundary the other way
    t = (torch.randn(2, 3, 8, 5, 6, device="cuda")).to(BF).requires_grad_()
    f = HF.to_float32(t)
    assert f.dtype == torch.float32 and torch.equal(f, t.detach().float())
    g = torch.randn_like(f)
    f.backward(g)
    assert t.grad.dtype == BF and torch.equal(t.grad, g.to(BF))
