"""Operator-level parity on a real MI355X: every HIP entry point, called through the C ABI (ctypes),
against torch's CPU kernels / the oracle on identical seeded inputs.

Tolerances: conv = fp32 MFMA fmaf chain vs oneDNN's accumulation order -> rel 1e-5 (L2);
pointwise / scans on identical inputs -> 1e-6; spikes -> exact except neurons within 1e-5 of threshold.
"""
import pytest
import torch
import torch.nn.functional as F

from oracle import neurons as ON
from tests.util import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def HF(hip_lib):
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from snn_for_object_detection_amd import functional
    return functional


CONV_CASES = [
    # T, B, Cin, H, W, Cout, k, s
    (2, 2, 2, 17, 23, 64, 3, 2),      # first layer (direct row kernels k_conv_first): Cin=2, K=18, stride 2
    (1, 3, 2, 9, 11, 16, 3, 1),       # the same kernels at stride 1, 16 channels (64 pixel lanes > row length)
    (2, 1, 2, 10, 13, 36, 3, 2),      # Cin=2 with a channel count the row kernels do not take: generic scalar gather
    (2, 1, 32, 12, 19, 32, 3, 1),     # BN=32 tile
    (1, 2, 64, 9, 11, 128, 3, 2),     # stride 2, odd sizes, BN=128 tile
    (2, 1, 128, 8, 10, 64, 1, 1),     # 1x1
    (1, 1, 96, 7, 5, 36, 1, 1),       # head box conv: Cout=36
    (1, 2, 256, 4, 5, 27, 1, 1),      # head cls conv: Cout=27 (not a multiple of 4)
    (1, 1, 8, 13, 9, 12, 5, 1),       # k=5 (README example nets)
    (1, 1, 4, 16, 15, 8, 7, 2),       # k=7 stride 2
    (3, 2, 64, 30, 38, 64, 3, 1),     # several 128-pixel tiles, K loop of 18 stages
    (1, 1, 8, 9, 7, 16, 1, 2),        # k=1 stride 2: three of the four dgrad stride-phase classes have no tap
    (1, 2, 4, 11, 10, 8, 3, 3),       # stride 3
    (2, 1, 128, 15, 19, 256, 3, 2),   # neck downsampling conv shape (odd sizes)
    # halo-resident direct 3x3 kernel (<= 32 output channels, Cin % 32 == 0), 8x16 patches:
    (2, 2, 64, 13, 21, 16, 3, 1),     # two 32-channel chunks, partial patches on both edges
    (1, 3, 32, 8, 16, 8, 3, 1),       # exactly one patch per image, 8 output channels
    (1, 1, 96, 5, 7, 32, 3, 1),       # image smaller than a patch, three chunks
    (4, 3, 32, 24, 40, 32, 3, 1),     # 108 patches: the persistent blocks loop over several patches each
]


@pytest.mark.parametrize("T,B,Cin,H,W,Cout,k,s", CONV_CASES)
def test_conv2d_fwd_bwd(HF, T, B, Cin, H, W, Cout, k, s):
    torch.manual_seed(T * 1000 + Cin + Cout + k)
    x = torch.randn(T, B, Cin, H, W)
    w = torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    pad = int(k / 2)
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    yr = F.conv2d(xr.flatten(0, 1), wr, stride=s, padding=pad)
    gy = torch.randn_like(yr)
    yr.backward(gy)

    xd, wd = x.cuda().requires_grad_(), w.cuda().requires_grad_()
    yd = HF.conv2d(xd, wd, stride=s, padding=pad)
    assert yd.shape == (T, B) + yr.shape[1:]
    yd.backward(gy.view(yd.shape).cuda())
    assert rel_err(yd.flatten(0, 1), yr) < 1e-5
    assert rel_err(xd.grad, xr.grad) < 1e-5
    assert rel_err(wd.grad, wr.grad) < 1e-5


BN_EPILOGUE_CASES = [
    # T, B, Cin, H, W, Cout, k, s, kernel that takes the shape
    (3, 2, 2, 34, 46, 64, 3, 2, "first"),       # row kernel: one group of blocks per timestep
    (8, 1, 2, 10, 12, 16, 3, 1, "first"),       # fewer rows per timestep than blocks
    (3, 2, 32, 21, 27, 32, 3, 1, "direct3"),    # patches never straddle a frame
    (2, 3, 64, 13, 21, 16, 3, 1, "direct3"),
    (3, 2, 64, 30, 38, 64, 3, 1, "gather"),     # 2280 rows per step: 128-row tiles straddle the timesteps
    (5, 1, 64, 19, 23, 128, 3, 2, "gather"),    # 120 output pixels per step < one tile: no partials, plain pass
    (4, 3, 64, 18, 22, 128, 3, 2, "gather"),    # two channel tiles... 297 rows per step
    (2, 2, 128, 16, 16, 64, 1, 1, "gather"),    # 512 rows per step = 4 whole tiles: nothing straddles
    (3, 1, 96, 13, 11, 36, 1, 1, "gather"),     # 36 channels in a 64-wide tile, 143 rows per step
    (2, 2, 8, 13, 9, 27, 5, 1, "gather"),       # generic loader, Cout not a multiple of 4 (scalar stores)
]


@pytest.mark.parametrize("T,B,Cin,H,W,Cout,k,s,kernel", BN_EPILOGUE_CASES)
def test_bn_statistics_from_the_conv_epilogue(HF, hip_lib, T, B, Cin, H, W, Cout, k, s, kernel):
    """snn_conv2d_fwd bn_partial: the partials a convolution leaves for the BatchNorm behind it give the same
    mean / invstd / running statistics as the separate pass over y.  The MFMA kernels sum in fp64 throughout (only the
    order differs from snn_bn_stats); the first-layer row kernel sums a thread's <= ceil(Wo / pixel lanes) pixels of
    one row in fp32 before going to fp64 (stated tolerance 1e-6 on the sums, far inside the convolution's own 5e-7
    per-element error)."""
    import ctypes
    from snn_for_object_detection_amd import _hip
    torch.manual_seed(Cin * 7 + Cout + T)
    dev, st, pad = torch.device("cuda"), torch.cuda.current_stream().cuda_stream, k // 2
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    N, M = T * B, B * Ho * Wo
    x = torch.randn(N, H, W, Cin, device=dev) + 0.5
    w = torch.randn(Cout, k, k, Cin, device=dev) / (Cin * k * k) ** 0.5
    y = torch.empty(N, Ho, Wo, Cout, device=dev)
    n_part = _hip.query("snn_conv2d_fwd_bn_partial_size", N, B, Ho, Wo, Cout)
    partial = torch.full((n_part,), float("nan"), device=dev, dtype=torch.float64)   # unwritten slots must not be read
    layout = (ctypes.c_int * 2)()
    _hip.call("snn_conv2d_fwd", x.data_ptr(), Cin, w.data_ptr(), None, y.data_ptr(), Cout, N, H, W, Cin, Ho, Wo, Cout, k, k,
              s, pad, None, 0, partial.data_ptr(), B, layout, _hip.PREC_FP16X3, st)
    if M < 128 and kernel == "gather":
        assert layout[0] == 0   # a 128-row tile would meet more than two timesteps: the caller runs snn_bn_stats
        return
    assert layout[0] > 0 and (layout[1] == 128) == (kernel == "gather")
    assert T * layout[0] * Cout * 2 <= n_part
    yref = torch.empty_like(y)   # the same convolution without the statistics: identical values
    _hip.call("snn_conv2d_fwd", x.data_ptr(), Cin, w.data_ptr(), None, yref.data_ptr(), Cout, N, H, W, Cin, Ho, Wo, Cout, k,
              k, s, pad, None, 0, None, 0, None, _hip.PREC_FP16X3, st)
    assert torch.equal(y, yref)

    gamma = torch.rand(Cout, device=dev) + 0.5
    bias = torch.randn(Cout, device=dev)

    def finalize(part, chunks, rpc):
        out = [torch.empty(T, Cout, device=dev) for _ in range(4)]
        rm, rv = torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev)
        _hip.call("snn_bn_stats_finalize", part.data_ptr(), chunks, rpc, T, M, Cout, gamma.data_ptr(), bias.data_ptr(),
                  1e-5, 0.1, rm.data_ptr(), rv.data_ptr(), 0, *[o.data_ptr() for o in out], st)
        sums = torch.empty(T, Cout, 2, device=dev, dtype=torch.float64)
        _hip.call("snn_bn_stats_reduce", part.data_ptr(), chunks, rpc, T, M, Cout, sums.data_ptr(), st)
        return out + [rm, rv], sums

    got, got_sums = finalize(partial, layout[0], layout[1])
    part2 = torch.empty(_hip.query("snn_bn_stats_partial_size", T, M, Cout), device=dev, dtype=torch.float64)
    _hip.call("snn_bn_stats", y.data_ptr(), Cout, T, M, Cout, part2.data_ptr(), st)
    want, want_sums = finalize(part2, 0, 0)
    y64 = y.double().view(T, M, Cout)
    exact = torch.stack([y64.sum(1), (y64 * y64).sum(1)], dim=-1)
    assert torch.isfinite(got_sums).all()
    tol = 1e-6 if kernel == "first" else 1e-13
    assert rel_err(got_sums, exact) < tol and rel_err(want_sums, exact) < 1e-13
    for g_, w_ in zip(got, want):
        assert torch.isfinite(g_).all()
        assert rel_err(g_, w_) < (2e-6 if kernel == "first" else 2e-7)   # fp32 roundings of sums that agree to `tol`
    # deterministic: the same launch again gives the same bits
    partial_b = torch.zeros_like(partial)
    _hip.call("snn_conv2d_fwd", x.data_ptr(), Cin, w.data_ptr(), None, y.data_ptr(), Cout, N, H, W, Cin, Ho, Wo, Cout, k, k,
              s, pad, None, 0, partial_b.data_ptr(), B, layout, _hip.PREC_FP16X3, st)
    again, again_sums = finalize(partial_b, layout[0], layout[1])
    assert torch.equal(again_sums, got_sums)


PRESPLIT_CASES = [
    # N, Cin, H, W, Cout, k, s
    (6, 128, 30, 38, 128, 3, 1),   # pipelined kernel, 128-wide tile
    (4, 64, 33, 41, 64, 3, 1),     # 64-wide tile
    (3, 64, 31, 45, 128, 3, 2),    # stride 2: four phase classes in the data gradient
    (5, 128, 20, 24, 64, 1, 1),    # 1x1
    (2, 96, 9, 11, 36, 1, 1),      # 36 channels in a 64-wide tile (rows past Cout read offset -1)
    (2, 32, 12, 19, 32, 3, 1),     # halo-resident direct kernel: ignores the image
    (2, 8, 13, 9, 16, 5, 1),       # generic loader: ignores the image
]


@pytest.mark.parametrize("N,Cin,H,W,Cout,k,s", PRESPLIT_CASES)
def test_presplit_weight_images_give_the_same_bits(HF, hip_lib, N, Cin, H, W, Cout, k, s):
    """snn_weight_presplit + the w_split / wt_split arguments: forward and data gradient with ready-made weight pieces
    equal the conversion on the fly bit for bit (the image holds exactly the pieces the kernels derive themselves)."""
    from snn_for_object_detection_amd import _hip
    torch.manual_seed(N * 100 + Cin + Cout + k)
    dev, st, pad = torch.device("cuda"), torch.cuda.current_stream().cuda_stream, k // 2
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    x = torch.randn(N, H, W, Cin, device=dev)
    w = torch.randn(Cout, k, k, Cin, device=dev) / (Cin * k * k) ** 0.5
    w[0, 0, 0, :4] = torch.tensor([3e-6, -1e-7, 200.0, 0.0], device=dev)   # tiny, subnormal-piece and large weights
    wt = w.permute(3, 1, 2, 0).contiguous()
    dy = torch.randn(N, Ho, Wo, Cout, device=dev)
    w16, wt16 = torch.empty_like(w), torch.empty_like(wt)
    _hip.call("snn_weight_presplit", w.data_ptr(), w16.data_ptr(), w.numel(), _hip.PREC_FP16X3, st)
    _hip.call("snn_weight_presplit", wt.data_ptr(), wt16.data_ptr(), wt.numel(), _hip.PREC_BF16X3, st)
    ys, dxs = [], []
    for a, b in ((None, None), (w16.data_ptr(), wt16.data_ptr())):
        y, dx = torch.empty(N, Ho, Wo, Cout, device=dev), torch.empty(N, H, W, Cin, device=dev)
        _hip.call("snn_conv2d_fwd", x.data_ptr(), Cin, w.data_ptr(), a, y.data_ptr(), Cout, N, H, W, Cin, Ho, Wo, Cout, k,
                  k, s, pad, None, 0, None, 0, None, _hip.PREC_FP16X3, st)
        _hip.call("snn_conv2d_dgrad", dy.data_ptr(), Cout, wt.data_ptr(), b, dx.data_ptr(), Cin, N, H, W, Cin, Ho, Wo,
                  Cout, k, k, s, pad, None, 0, None, 0, _hip.PREC_BF16X3, st)
        ys.append(y)
        dxs.append(dx)
    assert torch.equal(ys[0], ys[1]) and torch.equal(dxs[0], dxs[1])
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), stride=s, padding=pad)
    assert rel_err(ys[1].permute(0, 3, 1, 2), ref) < 2e-6
    with pytest.raises(RuntimeError, match="pre-split"):   # an image is tied to its arithmetic
        _hip.call("snn_conv2d_fwd", x.data_ptr(), Cin, w.data_ptr(), w16.data_ptr(), ys[0].data_ptr(), Cout, N, H, W, Cin,
                  Ho, Wo, Cout, k, k, s, pad, None, 0, None, 0, None, _hip.PREC_BF16X6, st)


def test_flat_trainer_presplit_weights_leave_the_step_bit_identical(HF):
    """The training step of a generated block with FlatTrainer's pre-split weight images and without: identical loss,
    gradients and updated weights, and the images follow the weights through optimiser steps."""
    from snn_for_object_detection_amd.generator import BlockGen
    from snn_for_object_detection_amd.layer_gen import Conv, LIF, Norm
    from snn_for_object_detection_amd.trainer import FlatTrainer
    outs = []
    for use in (True, False):
        torch.manual_seed(3)
        blk = BlockGen(32, [Conv(64, 3), Norm(), LIF(), Conv(128, 3, 2), Norm(), LIF(), Conv(64, 1)]).cuda().train()
        tr = FlatTrainer(blk, lr=1e-2)
        x = (torch.rand(4, 2, 32, 24, 30, device="cuda") < 0.3).float()
        HF.USE_PRESPLIT_WEIGHTS = use
        try:
            losses = []
            for _ in range(3):
                tr.zero_grad()
                y, _ = blk(x)
                loss = (y * y).mean()
                loss.backward()
                tr.step()
                losses.append(loss.item())
            tr.synchronize()
        finally:
            HF.USE_PRESPLIT_WEIGHTS = True
        outs.append((losses, tr.flat_grad.clone(), tr.flat_param.clone()))
    assert outs[0][0] == outs[1][0]
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])


def test_conv_feeds_batchnorm_statistics_through_the_modules(HF):
    """HipConv2d -> HipBatchNorm2d -> LIF in a generated block: with and without the epilogue statistics the outputs
    and every gradient agree (the statistics agree to fp64 rounding)."""
    from snn_for_object_detection_amd.generator import BlockGen
    from snn_for_object_detection_amd.layer_gen import Conv, LIF, Norm
    torch.manual_seed(5)
    T, B, C, H, W = 3, 2, 32, 24, 30
    results = []
    for use in (True, False):
        torch.manual_seed(11)
        blk = BlockGen(C, [Conv(32, 3), Norm(), LIF(), Conv(64, 3, 2), Norm(), LIF()]).cuda()
        blk.train()
        x = (torch.rand(T, B, C, H, W, device="cuda") < 0.3).float().requires_grad_()
        HF.USE_CONV_BN_STATS = use
        try:
            y, _ = blk(x)
            (y * torch.linspace(0.5, 1.5, y.numel(), device="cuda").view_as(y)).sum().backward()
        finally:
            HF.USE_CONV_BN_STATS = True
        results.append((y.detach(), x.grad, [p.grad for p in blk.parameters()],
                        [b.clone() for b in blk.buffers()]))
    (ya, ga, pa, ba), (yb, gb, pb, bb) = results
    assert (ya != yb).float().mean() < 1e-4    # spikes: identical unless a membrane sits on the threshold
    assert rel_err(ga, gb) < 1e-4
    for u, v in zip(pa, pb):
        assert rel_err(u, v) < 1e-4
    for u, v in zip(ba, bb):
        assert rel_err(u.float(), v.float()) < 1e-6


@pytest.mark.parametrize("neuron", ["LIF", "LI"])
def test_long_backward_scan_in_segments_equals_one_launch(HF, neuron):
    """Sequences longer than SCAN_SEGMENT_T run their backward scan in segments (the per-(t, c) BatchNorm sums of a
    launch live in LDS, and their footprint limits the channels per block): the carried (g_v, g_i) make the input
    gradient identical bit for bit; the parameter gradients are summed in a different (fixed) order."""
    from snn_for_object_detection_amd import _hip
    kind = _hip.NEURON_LIF if neuron == "LIF" else _hip.NEURON_LI
    torch.manual_seed(31)
    T, B, C, H, W = 70, 2, 64, 12, 10
    y0 = 2.0 * torch.randn(T, B, C, H, W) + 0.2
    g = torch.randn(T, B, C, H, W).cuda()
    gv = torch.randn(B, C, H, W).cuda()
    results = []
    for seg in (None, 32):
        HF.SCAN_SEGMENT_T = seg
        try:
            bn = torch.nn.BatchNorm2d(C).cuda().train()
            y = y0.cuda().requires_grad_()
            out, state = HF.affine_neuron(y, kind, None, bn=bn)
            results.append(torch.autograd.grad((out, state.v), (y, bn.weight, bn.bias), (g, gv)))
        finally:
            HF.SCAN_SEGMENT_T = 32
    (gy_a, gw_a, gb_a), (gy_b, gw_b, gb_b) = results
    assert torch.isfinite(gy_a).all() and rel_err(gy_b, gy_a) < 1e-6      # coefficients use t-ordered sums per segment
    assert rel_err(gw_b, gw_a) < 1e-5 and rel_err(gb_b, gb_a) < 1e-5


HALO_WGRAD_CASES = [
    # N, Cin, H, W, Cout, stride      (3x3, pad 1: the halo-resident weight gradient, csrc/wgrad_halo.hip, takes the
    # layers with at least 150 000 output pixels; the smaller cases below cover the implicit-GEMM path on the same shapes)
    (140, 128, 30, 38, 128, 1),  # four waves over output channels, four input-channel tiles, ragged patch columns
    (90, 32, 40, 44, 32, 1),     # K-steps of a patch split over the four waves (32 output channels)
    (96, 64, 21, 76, 64, 1),     # two waves over channels x two over K; partial last patch row
    (330, 64, 37, 52, 128, 2),   # stride 2, odd height: parity-de-interleaved halo columns
    (540, 128, 30, 38, 256, 2),  # stride 2, two output-channel tiles
    (1900, 256, 15, 19, 256, 2), # stride 2, odd sizes, eight input-channel tiles
    (1900, 128, 8, 10, 128, 1),  # one patch per image
    (2100, 32, 9, 8, 64, 1),     # images narrower than most patches
    (3, 32, 90, 640, 32, 1),     # 1 Mpx-class rows: 40 column patches per row
    (11, 64, 181, 320, 128, 2),  # 1 Mpx-class stride-2 layer, odd height
    (6, 128, 30, 38, 128, 1), (3, 64, 37, 52, 128, 2),   # below the size threshold: implicit-GEMM kernel
]


@pytest.mark.parametrize("N,Cin,H,W,Cout,s", HALO_WGRAD_CASES)
def test_wgrad_halo_kernel_against_fp64(HF, hip_lib, N, Cin, H, W, Cout, s):
    """snn_conv2d_wgrad through the C ABI on the shapes that take the halo-resident kernel: against an fp64
    convolution backward, with x and dy being channel slices of wider buffers (pixel strides > channel counts),
    bitwise reproducible, and `accumulate` adding to the destination."""
    from snn_for_object_detection_amd import _hip
    torch.manual_seed(N * 100 + Cin + Cout + s)
    st = torch.cuda.current_stream().cuda_stream
    Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
    xw = torch.randn(N, H, W, Cin + 32, device="cuda")
    dyw = torch.randn(N, Ho, Wo, Cout + 64, device="cuda")
    x, dy = xw[..., 32:], dyw[..., 32:32 + Cout]             # 16-byte aligned channel slices
    prec = _hip.PREC_BF16X3
    splitk = _hip.query("snn_conv2d_wgrad_splitk", N, H, W, Cin, Ho, Wo, Cout, 3, 3, s, 1, prec)
    ws = torch.empty(splitk, Cout * 9 * Cin, device="cuda")
    outs = []
    for _ in range(2):
        ws.fill_(float("nan"))                                # every slab element must be written
        dw = torch.full((Cout, 3, 3, Cin), float("nan"), device="cuda")
        _hip.call("snn_conv2d_wgrad", x.data_ptr(), Cin + 32, dy.data_ptr(), Cout + 64, dw.data_ptr(), N, H, W, Cin, Ho,
                  Wo, Cout, 3, 3, s, 1, 0, ws.data_ptr(), splitk, prec, st)
        outs.append(dw)
    assert torch.equal(outs[0], outs[1])
    wref = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    yref = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), wref, stride=s, padding=1)
    yref.backward(dy.permute(0, 3, 1, 2).double().cpu())
    ref = wref.grad.permute(0, 2, 3, 1)                       # OHWI like dw
    assert rel_err(outs[0], ref) < 1e-5
    _hip.call("snn_conv2d_wgrad", x.data_ptr(), Cin + 32, dy.data_ptr(), Cout + 64, outs[0].data_ptr(), N, H, W, Cin, Ho,
              Wo, Cout, 3, 3, s, 1, 1, ws.data_ptr(), splitk, prec, st)
    assert rel_err(outs[0], 2.0 * ref) < 1e-5


def test_conv2d_single_step_and_determinism(HF):
    torch.manual_seed(3)
    x = torch.randn(2, 16, 10, 12).cuda().requires_grad_()
    w = (torch.randn(24, 16, 3, 3) / 12).cuda().requires_grad_()
    y1 = HF.conv2d(x, w, 1, 1)
    assert y1.shape == (2, 24, 10, 12)
    g = torch.randn_like(y1)
    (gx1, gw1) = torch.autograd.grad(y1, (x, w), g)
    y2 = HF.conv2d(x, w, 1, 1)
    (gx2, gw2) = torch.autograd.grad(y2, (x, w), g)
    assert torch.equal(y1, y2) and torch.equal(gx1, gx2) and torch.equal(gw1, gw2)  # bitwise reproducible
    ref = F.conv2d(x.detach().cpu(), w.detach().cpu(), padding=1)
    assert rel_err(y1, ref) < 1e-5


def test_dgrad_two_fused_addends(HF, hip_lib):
    """dx = conv^T(dy) + addend + addend2 in one epilogue (C ABI), addends being channel slices of wider buffers."""
    from snn_for_object_detection_amd import _hip
    torch.manual_seed(11)
    st = torch.cuda.current_stream().cuda_stream
    for (N, H, W, Cin, Cout, k, s) in [(3, 12, 19, 32, 32, 3, 1), (2, 9, 11, 64, 128, 3, 2), (2, 8, 10, 128, 64, 1, 1)]:
        pad = k // 2
        Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
        dy = torch.randn(N, Ho, Wo, Cout, device="cuda")
        wt = torch.randn(Cin, k, k, Cout, device="cuda") * 0.05
        wide1 = torch.randn(N, H, W, Cin + 8, device="cuda")    # addend = channels [4, 4+Cin) of a wider buffer
        wide2 = torch.randn(N, H, W, 2 * Cin, device="cuda")    # addend2 = channels [Cin, 2Cin)
        a1, a2 = wide1[..., 4:4 + Cin], wide2[..., Cin:]
        plain = torch.empty(N, H, W, Cin, device="cuda")
        fused = torch.empty_like(plain)
        args = (dy.data_ptr(), Cout, wt.data_ptr(), None)
        geom = (N, H, W, Cin, Ho, Wo, Cout, k, k, s, pad)
        prec = _hip.PREC_BF16X3   # the arithmetic is an argument of every call (ABI v5)
        _hip.call("snn_conv2d_dgrad", *args, plain.data_ptr(), Cin, *geom, None, 0, None, 0, prec, st)
        _hip.call("snn_conv2d_dgrad", *args, fused.data_ptr(), Cin, *geom, a1.data_ptr(), Cin + 8, a2.data_ptr(),
                  2 * Cin, prec, st)
        assert torch.equal(fused, (plain + a1) + a2)
        only2 = torch.empty_like(plain)
        _hip.call("snn_conv2d_dgrad", *args, only2.data_ptr(), Cin, *geom, None, 0, a2.data_ptr(), 2 * Cin, prec, st)
        with pytest.raises(RuntimeError, match="precision"):   # a forward-only mode is refused, loudly
            _hip.call("snn_conv2d_dgrad", *args, only2.data_ptr(), Cin, *geom, None, 0, None, 0, _hip.PREC_FP16X3, st)
        assert torch.equal(only2, plain + a2)


def test_composed_1x1_convolutions_equal_the_chain(HF):
    """composed_conv1x1(x, w1, w2) == conv(conv(x, w1), w2) (values and all three gradients), without the middle tensor."""
    torch.manual_seed(15)
    T, B, Cin, C1, C2, H, W = 3, 2, 64, 64, 32, 9, 13
    x = torch.randn(T, B, Cin, H, W)
    w1 = torch.randn(C1, Cin, 1, 1) / Cin ** 0.5
    w2 = torch.randn(C2, C1, 1, 1) / C1 ** 0.5
    xr, w1r, w2r = (t.clone().double().requires_grad_() for t in (x, w1, w2))
    yr = F.conv2d(F.conv2d(xr.flatten(0, 1), w1r), w2r)
    g = torch.randn_like(yr)
    yr.backward(g)
    xd, w1d, w2d = (t.cuda().requires_grad_() for t in (x, w1, w2))
    yd = HF.composed_conv1x1(xd, w1d, w2d)
    yd.backward(g.float().view(yd.shape).cuda())
    assert rel_err(yd.flatten(0, 1), yr) < 2e-6
    assert rel_err(xd.grad, xr.grad) < 3e-5 and rel_err(w1d.grad, w1r.grad) < 3e-5 and rel_err(w2d.grad, w2r.grad) < 3e-5
    with pytest.raises(RuntimeError):
        HF.composed_conv1x1(xd, w2d, w1d)


def test_shortcut_fused_into_lif_store(HF):
    """affine_neuron(addend=x): out = LIF(BN(y)) + x written by the scan kernel; d out / d x = identity."""
    torch.manual_seed(12)
    T, B, C, H, W = 5, 2, 32, 6, 9
    y = torch.randn(T, B, C, H, W).cuda().requires_grad_()
    x = torch.randn(T, B, C, H, W).cuda().requires_grad_()
    bn = torch.nn.BatchNorm2d(C).cuda().train()
    bn2 = torch.nn.BatchNorm2d(C).cuda().train()
    from snn_for_object_detection_amd import _hip
    plain, st_p = HF.affine_neuron(y, _hip.NEURON_LIF, None, bn=bn)
    fused, st_f = HF.affine_neuron(y, _hip.NEURON_LIF, None, bn=bn2, addend=x)
    assert torch.equal(fused, plain + x)
    assert torch.equal(st_f.v, st_p.v) and torch.equal(st_f.i, st_p.i)
    g = torch.randn_like(fused)
    gy_p, = torch.autograd.grad(plain, (y,), g, retain_graph=True)
    gy_f, gx_f = torch.autograd.grad(fused, (y, x), g)
    assert torch.equal(gy_f, gy_p)
    assert torch.equal(gx_f, g)
    with pytest.raises(RuntimeError):
        HF.affine_neuron(y, _hip.NEURON_LI_TANH, None, bn=bn, addend=x)


@pytest.mark.parametrize("C,H,W,T,B,with_bn", [(64, 9, 11, 32, 2, True), (32, 6, 7, 6, 3, True), (6, 5, 4, 7, 2, True),
                                                 (16, 8, 8, 9, 2, False), (3, 5, 4, 5, 2, False)])
def test_checkpointed_lif_is_bit_identical(HF, C, H, W, T, B, with_bn):
    """snn_lif_fwd_ckpt / snn_lif_bwd_ckpt (state saved every 4th step, recomputed in the backward scan) against the
    per-step-state kernels: outputs, final state and every gradient must agree bit for bit, including a T that is
    not a multiple of the checkpoint interval, a scalar-path channel count, a carried state and a fused shortcut."""
    from snn_for_object_detection_amd import _hip
    torch.manual_seed(C * T + B)
    y = (2.5 * torch.randn(T, B, C, H, W) + 0.3).cuda().requires_grad_()
    x = torch.randn(T, B, C, H, W).cuda().requires_grad_()
    v0 = torch.rand(B, C, H, W).cuda().requires_grad_()
    i0 = torch.randn(B, C, H, W).cuda().requires_grad_()
    g = torch.randn(T, B, C, H, W).cuda()
    g_v, g_i = torch.randn(B, C, H, W).cuda(), torch.randn(B, C, H, W).cuda()
    results = []
    for threshold in (None, 0):
        HF.LIF_CHECKPOINT_BYTES = threshold
        try:
            bn = torch.nn.BatchNorm2d(C).cuda().train() if with_bn else None
            out, st = HF.affine_neuron(y, _hip.NEURON_LIF, HF.NeuronState(v0, i0), bn=bn, addend=x)
            inputs = (y, x, v0, i0) + ((bn.weight, bn.bias) if with_bn else ())
            grads = torch.autograd.grad((out, st.v, st.i), inputs, (g, g_v, g_i))
        finally:
            HF.LIF_CHECKPOINT_BYTES = None
        results.append((out, st.v, st.i) + tuple(grads))
    assert float((results[0][0] - x).detach().abs().sum()) > 0  # spikes present
    # C = 6 takes the LDS-atomics reduction (group count not a power of two): its BatchNorm sums are not ordered
    # in either kernel, so there the comparison is to rounding instead of to the bit
    ordered = not with_bn or C % 4 == 0
    for a, b in zip(*results):
        if ordered:
            assert torch.equal(a, b)
        else:
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("C,H,W,T,B", [(64, 60, 76, 4, 5), (24, 96, 120, 3, 2), (256, 30, 38, 2, 3)])
def test_backward_scan_addressing_variants_agree(HF, C, H, W, T, B):
    """Layers with several pixel rows per block take the buffer-addressed, branch-free backward scan; the call flag
    SNN_SCAN_WIDE_ADDRESSING forces the pointer / per-pixel-branch kernel (the library picks it by itself for tensors
    whose timestep exceeds the 31-bit buffer offsets).  Same arithmetic in the same order: every gradient must agree bit
    for bit where the BatchNorm sums are ordered (power-of-two group counts), to rounding where they use LDS atomics
    (C = 24)."""
    from snn_for_object_detection_amd import _hip
    torch.manual_seed(C + H)
    y = (2.5 * torch.randn(T, B, C, H, W) + 0.3).cuda().requires_grad_()
    x = torch.randn(T, B, C, H, W).cuda().requires_grad_()
    g = torch.randn(T, B, C, H, W).cuda()
    results = []
    was = HF.USE_SUMS_FROM_STATE
    HF.USE_SUMS_FROM_STATE = False   # both forms of the scan that READS y (the other one has a buffer-addressed form only)
    for no_buf in (False, True):
        HF.SCAN_FLAGS = _hip.SCAN_WIDE_ADDRESSING if no_buf else 0
        try:
            bn = torch.nn.BatchNorm2d(C).cuda().train()
            out, st = HF.affine_neuron(y, _hip.NEURON_LIF, None, bn=bn, addend=x)
            results.append(torch.autograd.grad(out, (y, x, bn.weight, bn.bias), g))
        finally:
            HF.SCAN_FLAGS = 0
    HF.USE_SUMS_FROM_STATE = was
    ordered = (C // 4) & (C // 4 - 1) == 0
    for a, b in zip(*results):
        assert torch.isfinite(a).all()
        if ordered:
            assert torch.equal(a, b)
        else:
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("C,H,W,T,B,shortcut", [(64, 60, 76, 8, 5, True), (32, 120, 152, 3, 2, False), (128, 15, 19, 5, 2, True),
                                                   (256, 8, 10, 4, 5, False), (64, 9, 11, 1, 2, False), (32, 6, 7, 2, 3, True),
                                                   (128, 30, 38, 32, 2, True), (64, 12, 10, 70, 2, True)])
def test_reverse_scan_statistic_from_the_saved_state(HF, C, H, W, T, B, shortcut):
    """SNN_SCAN_SUMS_FROM_STATE: the reverse LIF scan of a train-mode Norm -> LIF layer does not read y; the BatchNorm
    statistic sum(gx * y) is replaced by sum(gx * x) with the neuron input x rebuilt from the saved potentials
    (snn_bn_bwd_finalize_from_state converts).  Against the scan that reads y: the recurrence is untouched, so everything
    that does not pass through the statistic is bit-identical (shortcut gradient, dbias = sum gx); dy / dgamma agree to the
    rounding of the rebuilt input (1e-6 relative measured; asserted 1e-5).  One channel has gamma == 0 exactly - x then holds
    no trace of y and the finalize kernel sums gx * y for that channel itself - and the bias is not zero.
    Cases: three / two / one pixel rows per thread, T = 1 and 2 (the two statistics owed after the loop), T = 32, and T = 70
    in three segments (SNN_SCAN_STATE_LOOKBACK: a segment behind the first one rebuilds the state it starts from out of the two
    saved potentials in front of it)."""
    from snn_for_object_detection_amd import _hip
    torch.manual_seed(C + H + T)
    y = (2.5 * torch.randn(T, B, C, H, W) + 0.3).cuda().requires_grad_()
    x = torch.randn(T, B, C, H, W).cuda().requires_grad_() if shortcut else None
    g = torch.randn(T, B, C, H, W).cuda()
    gamma = 1.0 + 0.3 * torch.randn(C)
    gamma[1] = 0.0
    gamma[2] = -0.7
    bias = 0.4 * torch.randn(C)
    results, flags = [], []
    for on in (True, False):
        was = HF.USE_SUMS_FROM_STATE
        HF.USE_SUMS_FROM_STATE = on
        calls = []

        class Spy:
            def before(self, name, args):
                calls.append((name, args))

            def after(self, tok):
                pass
        _hip.PROFILER = Spy()
        try:
            bn = torch.nn.BatchNorm2d(C).cuda().train()
            bn.weight.data.copy_(gamma)
            bn.bias.data.copy_(bias)
            out, st = HF.affine_neuron(y, _hip.NEURON_LIF, None, bn=bn, addend=x)
            inputs = (y, bn.weight, bn.bias) + ((x,) if shortcut else ())
            results.append(torch.autograd.grad(out, inputs, g))
        finally:
            _hip.PROFILER = None
            HF.USE_SUMS_FROM_STATE = was
        flags.append([a[19] for nm, a in calls if nm == "snn_affine_neuron_bwd"])
        names = [nm for nm, _ in calls]
        assert ("snn_bn_bwd_finalize_from_state" in names) == on and ("snn_bn_bwd_finalize" in names) == (not on)
    if T <= 32:
        assert flags[0] == [_hip.SCAN_SUMS_FROM_STATE] and flags[1] == [0]
    else:   # last segment first; the one that starts at step 0 looks back on nothing
        look = _hip.SCAN_SUMS_FROM_STATE | _hip.SCAN_STATE_LOOKBACK
        assert flags[0] == [look, look, _hip.SCAN_SUMS_FROM_STATE] and flags[1] == [0, 0, 0]
    (dy1, dg1, db1, *rest1), (dy0, dg0, db0, *rest0) = results
    assert torch.isfinite(dy1).all() and float(dy0.abs().sum()) > 0
    assert torch.equal(db1, db0)
    if shortcut:
        assert torch.equal(rest1[0], rest0[0])
    assert rel_err(dy1, dy0) < 1e-5 and rel_err(dg1, dg0) < 1e-5
    assert abs(float(dg1[1] - dg0[1])) <= 1e-5 * max(1.0, abs(float(dg0[1])))     # the gamma == 0 channel: summed directly
    assert float(dy1[:, :, 1].abs().max()) == 0.0 and float(dy0[:, :, 1].abs().max()) == 0.0


def _oracle_norm_neuron(y, bn, cell, tanh=False, state=None):
    outs = []
    for t in range(y.shape[0]):
        x = bn(y[t]) if bn is not None else y[t]
        if cell is not None:
            o, state = cell(x, state)
        else:
            o = x
        outs.append(torch.tanh(o) if tanh else o)
    return torch.stack(outs), state


@pytest.mark.parametrize("C,H,W,T,B", [(32, 6, 7, 5, 3), (64, 12, 19, 8, 2), (6, 5, 4, 3, 2)])
def test_bn_lif_fused_train(HF, C, H, W, T, B):
    from snn_for_object_detection_amd import _hip
    from snn_for_object_detection_amd.layer_gen import HipBatchNorm2d
    torch.manual_seed(C + T)
    y = 2.5 * torch.randn(T, B, C, H, W) + 0.3
    gamma = 1.0 + 0.2 * torch.randn(C)

    bn = torch.nn.BatchNorm2d(C)
    bn.bias = None
    bn.weight.data.copy_(gamma)
    yr = y.clone().requires_grad_()
    cell = ON.LIFCell()
    # run the oracle step by step, keeping v_dec for the near-threshold analysis
    state, zs, vdecs = None, [], []
    for t in range(T):
        xt = bn(yr[t])
        if state is None:
            state = cell.initial_state(xt)
        i_new = state.i + xt
        vdecs.append((state.v + 0.1 * ((0.0 - state.v) + i_new)).detach())
        z, state = cell(xt, state)
        zs.append(z)
    zr = torch.stack(zs)
    gz = torch.randn_like(zr)
    (zr * gz).sum().backward()

    bnd = HipBatchNorm2d(C)
    bnd.bias = None
    bnd.weight.data.copy_(gamma)
    bnd = bnd.cuda()
    yd = y.cuda().requires_grad_()
    zd, st = HF.affine_neuron(yd, _hip.NEURON_LIF, None, bn=bnd)
    (zd * gz.cuda()).sum().backward()

    zd_c = zd.detach().cpu()
    mism = zd_c != zr.detach()
    assert mism.float().mean().item() < 1e-4
    if mism.any():  # the first disagreement of a neuron must sit on the threshold
        first = mism.float().cumsum(0).eq(1) & mism
        assert (torch.stack(vdecs)[first] - 1.0).abs().max().item() < 1e-5
    if not mism.any():
        assert rel_err(st.v, state.v) < 1e-6 and rel_err(st.i, state.i) < 1e-6
        assert rel_err(yd.grad, yr.grad) < 2e-5
        assert rel_err(bnd.weight.grad, bn.weight.grad) < 2e-5
    assert rel_err(bnd.running_mean, bn.running_mean) < 1e-6
    assert rel_err(bnd.running_var, bn.running_var) < 1e-6
    assert int(bnd.num_batches_tracked) == int(bn.num_batches_tracked) == T


def test_bn_li_tanh_fused_and_eval_mode(HF):
    from snn_for_object_detection_amd import _hip
    from snn_for_object_detection_amd.layer_gen import HipBatchNorm2d
    torch.manual_seed(11)
    T, B, C, H, W = 6, 2, 16, 5, 6
    y = torch.randn(T, B, C, H, W) * 3
    for training in (True, False):
        bn = torch.nn.BatchNorm2d(C)
        bn.weight.data.uniform_(0.5, 1.5)
        bn.bias.data.uniform_(-0.2, 0.2)
        bn.running_mean.uniform_(-0.5, 0.5)
        bn.running_var.uniform_(0.5, 2.0)
        bnd = HipBatchNorm2d(C)
        bnd.load_state_dict(bn.state_dict())
        bn.train(training)
        bnd.train(training)
        bnd = bnd.cuda()
        yr = y.clone().requires_grad_()
        outr, str_ = _oracle_norm_neuron(yr, bn, ON.LICell(), tanh=True)
        g = torch.randn_like(outr)
        (outr * g).sum().backward()
        yd = y.cuda().requires_grad_()
        outd, std = HF.affine_neuron(yd, _hip.NEURON_LI_TANH, None, bn=bnd)
        (outd * g.cuda()).sum().backward()
        assert rel_err(outd, outr) < 1e-6
        assert rel_err(std.v, str_.v) < 1e-6 and rel_err(std.i, str_.i) < 1e-6
        assert rel_err(yd.grad, yr.grad) < 2e-5
        assert rel_err(bnd.weight.grad, bn.weight.grad) < 2e-5
        assert rel_err(bnd.bias.grad, bn.bias.grad) < 2e-5


def test_last_step_only_lif_long_sequence_in_segments(HF):
    """The same for LIF at T = 70 (> SCAN_SEGMENT_T): the backward scan runs in segments, only the last of which has an
    output gradient."""
    from snn_for_object_detection_amd import _hip
    from snn_for_object_detection_amd.layer_gen import HipBatchNorm2d
    torch.manual_seed(23)
    T, B, C, H, W = 70, 2, 32, 7, 9
    x = torch.randn(T, B, C, H, W, device="cuda") * 2
    g_last = torch.randn(B, C, H, W, device="cuda")
    res = []
    was = HF.USE_SUMS_FROM_STATE
    HF.USE_SUMS_FROM_STATE = False   # the full scan that reads y, as the last-step-only one does: same bits
    try:
        for last_only in (True, False):
            bn = HipBatchNorm2d(C).cuda().train()
            xin = x.clone().requires_grad_()
            out, st = HF.affine_neuron(xin, _hip.NEURON_LIF, None, bn=bn, last_only=last_only)
            (out if last_only else out[-1]).backward(g_last)
            res.append(((out if last_only else out[-1]).detach(), xin.grad, bn.weight.grad, st.v.detach(), st.i.detach()))
    finally:
        HF.USE_SUMS_FROM_STATE = was
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert res[0][0].sum() > 0 and res[0][1].abs().sum() > 0


def test_last_step_only_never_takes_the_checkpointed_scan(HF):
    """``last_only`` allocates ONE step of output; the checkpointed LIF kernels write / read all T of them, so the
    memory lever must not apply (it used to: an out-of-bounds write of (T-1)*M*C floats).  Same bits with the lever set
    as without."""
    from snn_for_object_detection_amd import _hip
    from snn_for_object_detection_amd.layer_gen import HipBatchNorm2d
    torch.manual_seed(29)
    T, B, C, H, W = 12, 2, 32, 7, 9
    x = torch.randn(T, B, C, H, W, device="cuda") * 2
    g_last = torch.randn(B, C, H, W, device="cuda")
    res = []
    for lever in (0, None):
        HF.LIF_CHECKPOINT_BYTES = lever          # 0 = every LIF layer, None = never
        try:
            bn = HipBatchNorm2d(C).cuda().train()
            xin = x.clone().requires_grad_()
            guard = torch.full((4 * T * B * C * H * W,), 7.0, device="cuda")   # neighbours of the small output buffer
            out, st = HF.affine_neuron(xin, _hip.NEURON_LIF, None, bn=bn, last_only=True)
            assert out.shape == (B, C, H, W)
            out.backward(g_last)
            assert bool((guard == 7.0).all())
            res.append((out.detach(), xin.grad, bn.weight.grad, st.v.detach(), st.i.detach()))
        finally:
            HF.LIF_CHECKPOINT_BYTES = None
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize("tanh", [True, False])
def test_last_step_only_scan_equals_the_full_scan(HF, tanh):
    """SNN_SCAN_LAST_STEP_ONLY (the detection head keeps the last timestep only): the [B,C,H,W] output equals the last
    step of the full scan, and every gradient equals the one the full scan gets from an output gradient that is zero
    before the last step - bit for bit."""
    from snn_for_object_detection_amd import _hip
    from snn_for_object_detection_amd.layer_gen import HipBatchNorm2d
    torch.manual_seed(17)
    T, B, C, H, W = 6, 3, 64, 9, 11
    neuron = _hip.NEURON_LI_TANH if tanh else _hip.NEURON_LI
    x = torch.randn(T, B, C, H, W, device="cuda")
    g_last = torch.randn(B, C, H, W, device="cuda")
    res = []
    for last_only in (True, False):
        torch.manual_seed(1)
        bn = HipBatchNorm2d(C).cuda().train()
        bn.weight.data.uniform_(0.5, 1.5)
        xin = x.clone().requires_grad_()
        out, st = HF.affine_neuron(xin, neuron, None, bn=bn, last_only=last_only)
        if last_only:
            assert out.shape == (B, C, H, W)
            out.backward(g_last)
        else:
            out[-1].backward(g_last)
            out = out[-1]
        res.append((out.detach(), xin.grad, bn.weight.grad, st.v.detach(), bn.running_mean.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_lif_state_carry_equals_sequence(HF):
    """T single-step calls with carried state (reference protocol) == one sequence call, bitwise; and
    gradients flow through the carried state (time-outer BPTT)."""
    from snn_for_object_detection_amd import _hip
    torch.manual_seed(5)
    T, B, C, H, W = 7, 2, 8, 4, 5
    x = (1.2 * torch.rand(T, B, C, H, W)).cuda()
    xs = x.clone().requires_grad_()
    z_seq, st_seq = HF.affine_neuron(xs, _hip.NEURON_LIF, None)
    g = torch.randn_like(z_seq)
    (z_seq * g).sum().backward()
    xt = x.clone().requires_grad_()
    state, zs = None, []
    for t in range(T):
        z, state = HF.affine_neuron(xt[t], _hip.NEURON_LIF, state)
        zs.append(z)
    z_loop = torch.stack(zs)
    (z_loop * g).sum().backward()
    assert torch.equal(z_seq, z_loop)
    assert torch.equal(st_seq.v, state.v) and torch.equal(st_seq.i, state.i)
    assert rel_err(xt.grad, xs.grad) < 1e-6
    # and against the oracle
    cell, st, zr = ON.LIFCell(), None, []
    xc = x.cpu().requires_grad_()
    for t in range(T):
        z, st = cell(xc[t], st)
        zr.append(z)
    zr = torch.stack(zr)
    (zr * g.cpu()).sum().backward()
    assert torch.equal(z_seq.cpu(), zr)
    assert rel_err(xs.grad, xc.grad) < 1e-6


def test_bn_backward_is_bitwise_reproducible(HF):
    """ordered block / wave reduction of the BatchNorm sums: two runs give identical bits."""
    from snn_for_object_detection_amd import _hip
    from snn_for_object_detection_amd.layer_gen import HipBatchNorm2d
    torch.manual_seed(21)
    for C in (32, 64, 24):   # 24 -> 6 channel groups: the non-power-of-two fallback is exempt
        y = (2 * torch.randn(6, 3, C, 11, 13)).cuda()
        g = torch.randn(6, 3, C, 11, 13).cuda()
        res = []
        for _ in range(2):
            bn = HipBatchNorm2d(C).cuda()
            yd = y.clone().requires_grad_()
            z, _ = HF.affine_neuron(yd, _hip.NEURON_LIF, None, bn=bn)
            (z * g).sum().backward()
            res.append((yd.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone()))
        if C != 24:
            for a, b in zip(*res):
                assert torch.equal(a, b)
        else:
            for a, b in zip(*res):
                assert rel_err(a, b) < 1e-5


def test_standalone_norm_matches_batchnorm(HF):
    from snn_for_object_detection_amd.layer_gen import HipBatchNorm2d
    torch.manual_seed(2)
    x = torch.randn(4, 10, 9, 7) * 2 + 1
    bn = torch.nn.BatchNorm2d(10)
    bnd = HipBatchNorm2d(10).cuda()
    xr, xd = x.clone().requires_grad_(), x.cuda().requires_grad_()
    yr, yd = bn(xr), bnd(xd)
    g = torch.randn_like(yr)
    yr.backward(g)
    yd.backward(g.cuda())
    assert rel_err(yd, yr) < 1e-6 and rel_err(xd.grad, xr.grad) < 2e-5
    assert rel_err(bnd.weight.grad, bn.weight.grad) < 2e-5 and rel_err(bnd.bias.grad, bn.bias.grad) < 2e-5


def test_merges_pool_upsample_activation(HF):
    from snn_for_object_detection_amd import _hip
    torch.manual_seed(9)
    a, b, c = torch.randn(2, 3, 8, 6, 5), torch.randn(2, 3, 12, 6, 5), torch.randn(2, 3, 8, 6, 5)
    ad, bd, cd = (t.cuda().requires_grad_() for t in (a, b, c))
    ar, br, cr = (t.clone().requires_grad_() for t in (a, b, c))
    outd = HF.concat_channels([HF.sum_tensors([ad, cd]), bd])
    outr = torch.cat([torch.stack([ar, cr]).sum(0), br], dim=2)
    g = torch.randn_like(outr)
    outd.backward(g.cuda())
    outr.backward(g)
    assert torch.equal(outd.cpu(), outr)
    for d, r in ((ad, ar), (bd, br), (cd, cr)):
        assert torch.equal(d.grad.cpu(), r.grad)

    x = torch.randn(3, 6, 9, 11)
    pools = ((_hip.POOL_AVG, 2, 2, lambda t: F.avg_pool2d(t, 2, 2)), (_hip.POOL_MAX, 2, 2, lambda t: F.max_pool2d(t, 2, 2)),
             (_hip.POOL_SUM, 2, 2, lambda t: F.avg_pool2d(t, 2, 2) * 2 * 2), (_hip.POOL_AVG, 3, 2, lambda t: F.avg_pool2d(t, 3, 2)),
             (_hip.POOL_MAX, 3, 1, lambda t: F.max_pool2d(t, 3, 1)))
    for kind, k, s, ref in pools:
        xr, xd = x.clone().requires_grad_(), x.cuda().requires_grad_()
        yr = ref(xr)
        yd = HF.pool2d(xd, kind, k, s)
        gg = torch.randn_like(yr)
        yr.backward(gg)
        yd.backward(gg.cuda())
        assert rel_err(yd, yr) < 1e-6 and rel_err(xd.grad, xr.grad) < 1e-6

    xr, xd = x.clone().requires_grad_(), x.cuda().requires_grad_()
    yr, yd = F.interpolate(xr, scale_factor=2, mode="nearest"), HF.upsample_nearest(xd, 2)
    gg = torch.randn_like(yr)
    yr.backward(gg)
    yd.backward(gg.cuda())
    assert torch.equal(yd.cpu(), yr) and rel_err(xd.grad, xr.grad) < 1e-6

    for act, ref in ((_hip.ACT_RELU, torch.relu), (_hip.ACT_SILU, F.silu), (_hip.ACT_TANH, torch.tanh)):
        xr, xd = x.clone().requires_grad_(), x.cuda().requires_grad_()
        yr, yd = ref(xr), HF.activation(xd, act)
        ga = torch.randn_like(yr)
        yr.backward(ga)
        yd.backward(ga.cuda())
        assert rel_err(yd, yr) < 1e-6 and rel_err(xd.grad, xr.grad) < 1e-5


def test_layout_roundtrip_and_events(HF):
    torch.manual_seed(1)
    for C in (1, 2, 3, 7, 64):
        x = torch.randn(3, C, 9, 14).cuda()
        cl = HF._raw_to_cl(x)
        assert HF.is_channels_last(cl) and torch.equal(cl, x)
        assert torch.equal(HF._raw_to_nchw(cl).contiguous(), x)
    T, H, W, n = 4, 12, 16, 500
    g = torch.Generator().manual_seed(0)
    tb = torch.randint(0, T, (n,), generator=g, dtype=torch.int32)
    xs = torch.randint(0, W + 3, (n,), generator=g, dtype=torch.int32)  # some beyond the frame: clipped
    ys = torch.randint(0, H, (n,), generator=g, dtype=torch.int32)
    ps = torch.randint(0, 2, (n,), generator=g, dtype=torch.int32)
    want = torch.zeros(T, 2, H, W)
    want[tb.long(), ps.long(), ys.long(), xs.clamp(0, W - 1).long()] = 1
    got = HF.events_to_frames(tb.cuda(), xs.cuda(), ys.cuda(), ps.cuda(), T, H, W)
    assert torch.equal(got.cpu(), want)


def test_errors_surface_as_exceptions(HF):
    from snn_for_object_detection_amd import Pool, Residual, BlockGen, Conv
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        HF.conv2d(torch.zeros(1, 2, 4, 4), torch.zeros(3, 2, 3, 3), 1, 1)
    with pytest.raises(ValueError):
        Pool("X")
    with pytest.raises(RuntimeError, match="residual"):
        BlockGen(4, Residual([[Conv(8, 1)], [Conv(6, 1)]]))
    with pytest.raises(RuntimeError, match="channels"):
        HF.conv2d(torch.zeros(1, 2, 4, 4).cuda(), torch.zeros(3, 5, 3, 3).cuda(), 1, 1)


@pytest.mark.parametrize("mode,tol", [("fp32", 2e-6), ("bf16x3", 3e-5)])
def test_backward_precision_modes_against_fp64(HF, mode, tol):
    """Gradient accuracy of the two backward arithmetics against an fp64 reference (forward is always fp32)."""
    torch.manual_seed(12)
    T, B, Cin, H, W, Cout, k, s = 2, 2, 64, 20, 24, 128, 3, 1
    x = torch.randn(T, B, Cin, H, W)
    w = torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    xr, wr = x.double().requires_grad_(), w.double().requires_grad_()
    yr = F.conv2d(xr.flatten(0, 1), wr, stride=s, padding=1)
    gy = torch.randn_like(yr)
    yr.backward(gy)
    prev = HF.get_backward_precision()
    try:
        HF.set_backward_precision(mode)
        assert HF.get_backward_precision() == mode
        xd, wd = x.cuda().requires_grad_(), w.cuda().requires_grad_()
        yd = HF.conv2d(xd, wd, stride=s, padding=1)
        yd.backward(gy.float().view(yd.shape).cuda())
        assert rel_err(yd.flatten(0, 1), yr) < 2e-6          # forward: fp32-grade in every mode
        assert rel_err(xd.grad, xr.grad) < tol and rel_err(wd.grad, wr.grad) < tol
    finally:
        HF.set_backward_precision(prev)
    with pytest.raises(ValueError):
        HF.set_backward_precision("fp8")


@pytest.mark.parametrize("mode", ["bf16x6", "fp32", "fp16x3"])
def test_forward_precision_modes_against_fp64(HF, mode):
    """Both forward arithmetics are fp32-grade: the 3-way bf16 split (6 products) carries all 24 significant bits."""
    torch.manual_seed(13)
    prev = HF.get_forward_precision()
    try:
        HF.set_forward_precision(mode)
        assert HF.get_forward_precision() == mode
        for (Cin, Cout, k, s, H, W) in [(64, 128, 3, 1, 20, 24), (96, 64, 1, 1, 17, 19), (32, 32, 3, 2, 31, 30)]:
            x = torch.randn(3, 2, Cin, H, W) * 3.0
            w = torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5
            ref = F.conv2d(x.double().flatten(0, 1), w.double(), stride=s, padding=k // 2)
            y = HF.conv2d(x.cuda(), w.cuda(), stride=s, padding=k // 2)
            assert rel_err(y.flatten(0, 1), ref) < 1.5e-6
    finally:
        HF.set_forward_precision(prev)
    with pytest.raises(ValueError):
        HF.set_forward_precision("bf16x2")


def test_fp16x3_range_contract(HF):
    """Default forward arithmetic: fp32-grade inside its range, graceful below it, loud (non-finite) above it."""
    torch.manual_seed(14)
    prev = HF.get_forward_precision()
    try:
        HF.set_forward_precision("fp16x3")
        w = torch.randn(64, 64, 3, 3) / 24.0
        for scale, tol in [(1.0, 1.5e-6), (200.0, 1.5e-6), (1e-2, 1.5e-6), (1e-4, 1e-4)]:
            x = torch.randn(2, 2, 64, 12, 16) * scale
            ref = F.conv2d(x.double().flatten(0, 1), w.double(), padding=1)
            y = HF.conv2d(x.cuda(), w.cuda(), stride=1, padding=1)
            assert rel_err(y.flatten(0, 1), ref) < tol, scale
        x = torch.randn(1, 1, 64, 12, 16)
        x[0, 0, 3, 5, 5] = 1e5   # beyond the fp16 range after the 2^4 pre-scale
        y = HF.conv2d(x.cuda(), w.cuda(), stride=1, padding=1)
        assert not torch.isfinite(y).all()
        HF.set_forward_precision("bf16x6")   # the any-range mode handles the same input
        y = HF.conv2d(x.cuda(), w.cuda(), stride=1, padding=1)
        assert torch.isfinite(y).all()
    finally:
        HF.set_forward_precision(prev)
