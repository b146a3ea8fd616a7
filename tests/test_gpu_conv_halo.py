"""Halo-resident 3x3 convolution (csrc/conv_halo.hip, ``snn_conv3x3_halo``) through the C ABI: forward and data
gradient against torch's CPU convolution in fp64 (the op the reference instantiates, models/modules/layer_gen.py:129-136),
channel-sliced operands, fused addends, BatchNorm statistics partials, determinism, and the padded-strip edge cases
(images narrower than a tile, one-pixel images, the widest supported row, tiles that end inside a group)."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from tests.util import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H_(hip_lib):
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from snn_for_object_detection_amd import _hip
    return _hip


def _image(_hip, src, O, I, flip, prec):
    img = torch.empty(9 * O * I, device="cuda")
    table = torch.tensor([[0, 0, O, I]], dtype=torch.int64, device="cuda")
    _hip.call("snn_weight_frag_image_batched", src.data_ptr(), img.data_ptr(), table.data_ptr(), 1,
              9 * (I // 32) * (O // 32) * 128, flip, prec, torch.cuda.current_stream().cuda_stream)
    return img


HALO_CASES = [
    # N, H, W, Cin, Cout
    (6, 30, 38, 128, 128),    # the neck-1 bottleneck shape: 128-wide tile, four chunks
    (3, 60, 76, 64, 64),      # backbone stage 2: 64-wide tile, two chunks
    (7, 15, 19, 128, 128),    # neck-2 maps
    (9, 8, 10, 128, 128),     # neck-3 maps: a tile spans more than one image
    (5, 7, 5, 32, 64),        # one chunk, rows far shorter than a tile (several images per tile)
    (3, 1, 1, 64, 64),        # one-pixel images: every tap but the centre reads a pad cell
    (2, 9, 78, 64, 128),      # the widest supported row (PW = 79)
    (2, 33, 41, 96, 192),     # three chunks, three 64-wide channel tiles
    (1, 5, 6, 32, 256),       # two 128-wide channel tiles
    (4, 12, 70, 64, 128),     # tiles ending inside an image row
    # rows longer than 78 pixels: 4 x 32 rectangles of one image (RECT) instead of strip tiles
    (2, 24, 100, 64, 64),     # partial rectangles on the right edge (100 = 3 * 32 + 4)
    (1, 7, 304, 64, 128),     # the deep-backbone row length, partial rectangles on the bottom edge (7 = 4 + 3)
    (3, 33, 81, 32, 64),      # one chunk, 81 = 2 * 32 + 17
    (2, 90, 160, 128, 128),   # a 1 Mpx neck map
    # 32 output channels: four waves side by side along the cells (the full-resolution stage)
    (2, 40, 52, 32, 32),      # strip tiles
    (2, 21, 152, 32, 32),     # rectangles, partial on both edges
    (3, 9, 30, 64, 32),       # two chunks into one 32-channel tile (data gradient: 32 -> 64)
]


@pytest.mark.parametrize("N,H,W,Cin,Cout", HALO_CASES)
def test_halo_forward_and_data_gradient_against_fp64(H_, N, H, W, Cin, Cout):
    _hip = H_
    assert _hip.query("snn_conv3x3_halo_supported", N, H, W, Cin, Cout) == 1
    torch.manual_seed(N * 100 + H + W + Cin)
    st = torch.cuda.current_stream().cuda_stream
    x = torch.randn(N, H, W, Cin)
    w = torch.randn(Cout, 3, 3, Cin) / (9 * Cin) ** 0.5              # OHWI
    gy = torch.randn(N, H, W, Cout)
    xr = x.permute(0, 3, 1, 2).double().requires_grad_()
    yr = F.conv2d(xr, w.permute(0, 3, 1, 2).double(), padding=1)
    yr.backward(gy.permute(0, 3, 1, 2).double())
    y_ref, dx_ref = yr.detach().permute(0, 2, 3, 1), xr.grad.permute(0, 2, 3, 1)

    xd, wd, gyd = x.cuda(), w.cuda(), gy.cuda()
    # forward: fp16 x 3 image of the OHWI weights
    img = _image(_hip, wd, Cout, Cin, 0, _hip.PREC_FP16X3)
    y = torch.full((N, H, W, Cout), float("nan"), device="cuda")
    _hip.call("snn_conv3x3_halo", xd.data_ptr(), Cin, img.data_ptr(), y.data_ptr(), Cout, N, H, W, Cin, Cout, None, 0, None,
              0, None, 0, None, _hip.PREC_FP16X3, st)
    assert torch.isfinite(y).all()
    assert rel_err(y, y_ref) < 2e-6
    # the implicit-GEMM kernel computes the same products (other summation order)
    y_g = torch.empty_like(y)
    _hip.call("snn_conv2d_fwd", xd.data_ptr(), Cin, wd.data_ptr(), None, y_g.data_ptr(), Cout, N, H, W, Cin, H, W, Cout, 3, 3,
              1, 1, None, 0, None, 0, None, _hip.PREC_FP16X3, st)
    assert rel_err(y, y_g) < 2e-6
    # data gradient: bf16 x 3 image of the transposed weights with mirrored taps; "Cin" of the call = the layer's Cout
    wt = torch.empty(Cin, 3, 3, Cout, device="cuda")
    _hip.call("snn_weight_transpose", wd.data_ptr(), wt.data_ptr(), Cout, 3, 3, Cin, st)
    if _hip.query("snn_conv3x3_halo_supported", N, H, W, Cout, Cin) == 1:
        img_t = _image(_hip, wt, Cin, Cout, 1, _hip.PREC_BF16X3)
        dx = torch.full((N, H, W, Cin), float("nan"), device="cuda")
        _hip.call("snn_conv3x3_halo", gyd.data_ptr(), Cout, img_t.data_ptr(), dx.data_ptr(), Cin, N, H, W, Cout, Cin, None, 0,
                  None, 0, None, 0, None, _hip.PREC_BF16X3, st)
        assert torch.isfinite(dx).all()
        assert rel_err(dx, dx_ref) < 3e-5                           # bf16 x 3: 16-bit products
        # bitwise reproducible
        dx2 = torch.empty_like(dx)
        _hip.call("snn_conv3x3_halo", gyd.data_ptr(), Cout, img_t.data_ptr(), dx2.data_ptr(), Cin, N, H, W, Cout, Cin, None,
                  0, None, 0, None, 0, None, _hip.PREC_BF16X3, st)
        assert torch.equal(dx, dx2)


def test_halo_channel_slices_and_fused_addends(H_):
    """Operands that are channel slices of wider channels-last buffers (pixel stride > channel count: the zero-copy Dense
    merge) and the two fused addends of the data-gradient epilogue; untouched neighbours of the output slice."""
    _hip = H_
    torch.manual_seed(5)
    st = torch.cuda.current_stream().cuda_stream
    N, H, W, Cin, Cout = 4, 15, 19, 64, 128
    xbuf = torch.randn(N, H, W, Cin + 32, device="cuda")             # x = channels 16 .. 16+Cin of a wider buffer
    x = xbuf[..., 16:16 + Cin]
    ybuf = torch.full((N, H, W, Cout + 64), 7.0, device="cuda")      # y = channels 32 .. 32+Cout
    w = torch.randn(Cout, 3, 3, Cin, device="cuda") / (9 * Cin) ** 0.5
    a1 = torch.randn(N, H, W, Cout + 8, device="cuda")               # addends with their own pixel strides
    a2 = torch.randn(N, H, W, Cout, device="cuda")
    img = _image(_hip, w, Cout, Cin, 0, _hip.PREC_FP16X3)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.permute(0, 3, 1, 2).double().cpu(), padding=1).permute(0, 2, 3, 1)
    ref = ref + a1[..., 4:4 + Cout].double().cpu() + a2.double().cpu()
    _hip.call("snn_conv3x3_halo", x.data_ptr(), Cin + 32, img.data_ptr(), ybuf[..., 32:].data_ptr(), Cout + 64, N, H, W, Cin,
              Cout, a1[..., 4:].data_ptr(), Cout + 8, a2.data_ptr(), Cout, None, 0, None, _hip.PREC_FP16X3, st)
    assert rel_err(ybuf[..., 32:32 + Cout], ref) < 2e-6
    assert bool((ybuf[..., :32] == 7.0).all()) and bool((ybuf[..., 32 + Cout:] == 7.0).all())
    # an output slice that is not 16-byte aligned takes the scalar store path
    ybuf2 = torch.full((N, H, W, Cout + 3), 7.0, device="cuda")
    _hip.call("snn_conv3x3_halo", x.data_ptr(), Cin + 32, img.data_ptr(), ybuf2[..., 1:].data_ptr(), Cout + 3, N, H, W, Cin,
              Cout, None, 0, a2.data_ptr(), Cout, None, 0, None, _hip.PREC_FP16X3, st)
    assert rel_err(ybuf2[..., 1:1 + Cout], ref - a1[..., 4:4 + Cout].double().cpu()) < 2e-6
    assert bool((ybuf2[..., 0] == 7.0).all()) and bool((ybuf2[..., 1 + Cout:] == 7.0).all())


@pytest.mark.parametrize("T,B,H,W,Cin,Cout", [(3, 2, 30, 38, 64, 64), (4, 3, 15, 19, 128, 128), (5, 1, 8, 10, 64, 128),
                                             (2, 5, 60, 76, 64, 64), (6, 2, 3, 4, 32, 64),
                                             (3, 2, 13, 100, 64, 64), (2, 1, 21, 304, 64, 64),    # these two: RECT tiles
                                             (3, 2, 20, 26, 32, 32), (2, 2, 9, 90, 32, 32)])      # the 32-channel tile
def test_halo_batchnorm_statistics_partials(H_, T, B, H, W, Cin, Cout):
    """The partials the forward leaves for the BatchNorm behind it (per timestep, tiles never straddle two steps) give
    the sums of the stored values - against fp64 sums of y and against the separate statistics pass."""
    _hip = H_
    torch.manual_seed(T + B + H)
    st = torch.cuda.current_stream().cuda_stream
    N, M = T * B, B * H * W
    x = torch.randn(N, H, W, Cin, device="cuda") + 0.5
    w = torch.randn(Cout, 3, 3, Cin, device="cuda") / (9 * Cin) ** 0.5
    img = _image(_hip, w, Cout, Cin, 0, _hip.PREC_FP16X3)
    y = torch.empty(N, H, W, Cout, device="cuda")
    n_part = _hip.query("snn_conv2d_fwd_bn_partial_size", N, B, H, W, Cout)
    partial = torch.full((n_part,), float("nan"), device="cuda", dtype=torch.float64)
    layout = (ctypes.c_int * 2)()
    _hip.call("snn_conv3x3_halo", x.data_ptr(), Cin, img.data_ptr(), y.data_ptr(), Cout, N, H, W, Cin, Cout, None, 0, None, 0,
              partial.data_ptr(), B, layout, _hip.PREC_FP16X3, st)
    assert layout[0] == _hip.query("snn_conv3x3_halo_bn_chunks", B, H, W) and layout[1] == 0
    assert T * layout[0] * Cout * 2 <= n_part
    yref = torch.empty_like(y)
    _hip.call("snn_conv3x3_halo", x.data_ptr(), Cin, img.data_ptr(), yref.data_ptr(), Cout, N, H, W, Cin, Cout, None, 0, None,
              0, None, 0, None, _hip.PREC_FP16X3, st)
    assert torch.equal(y, yref)                                        # statistics do not change the values
    sums = torch.empty(T, Cout, 2, device="cuda", dtype=torch.float64)
    _hip.call("snn_bn_stats_reduce", partial.data_ptr(), layout[0], layout[1], T, M, Cout, sums.data_ptr(), st)
    y64 = y.double().view(T, M, Cout)
    exact = torch.stack([y64.sum(1), (y64 * y64).sum(1)], dim=-1)
    assert torch.isfinite(sums).all() and rel_err(sums, exact) < 1e-13
    # and through the finalize kernel: mean / invstd equal to the separate pass's
    gamma, bias = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda")

    def finalize(part, chunks, rpc):
        out = [torch.empty(T, Cout, device="cuda") for _ in range(4)]
        rm, rv = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
        _hip.call("snn_bn_stats_finalize", part.data_ptr(), chunks, rpc, T, M, Cout, gamma.data_ptr(), bias.data_ptr(), 1e-5,
                  0.1, rm.data_ptr(), rv.data_ptr(), 0, *[o.data_ptr() for o in out], st)
        return out + [rm, rv]

    part2 = torch.empty(_hip.query("snn_bn_stats_partial_size", T, M, Cout), device="cuda", dtype=torch.float64)
    _hip.call("snn_bn_stats", y.data_ptr(), Cout, T, M, Cout, part2.data_ptr(), st)
    for g_, w_ in zip(finalize(partial, layout[0], layout[1]), finalize(part2, 0, 0)):
        assert torch.isfinite(g_).all() and rel_err(g_, w_) < 2e-7


def test_halo_refuses_uncovered_shapes(H_):
    _hip = H_
    st = torch.cuda.current_stream().cuda_stream
    x = torch.zeros(1, 4, 100, 64, device="cuda")
    img = torch.zeros(9 * 64 * 64, device="cuda")
    y = torch.zeros(1, 4, 100, 64, device="cuda")
    assert _hip.query("snn_conv3x3_halo_supported", 1, 4, 100, 64, 96) == 0          # channel tiles: 32, or multiples of 64
    assert _hip.query("snn_conv3x3_halo_supported", 1, 4, 100, 64, 32) == 1
    assert _hip.query("snn_conv3x3_halo_supported", 1, 4, 100, 48, 64) == 0          # input channels not a multiple of 32
    with pytest.raises(RuntimeError, match="shape not covered"):
        _hip.call("snn_conv3x3_halo", x.data_ptr(), 64, img.data_ptr(), y.data_ptr(), 64, 1, 4, 100, 64, 96, None, 0, None, 0,
                  None, 0, None, _hip.PREC_FP16X3, st)
    with pytest.raises(RuntimeError, match="precision"):
        _hip.call("snn_conv3x3_halo", x.data_ptr(), 64, img.data_ptr(), y.data_ptr(), 64, 1, 4, 50, 64, 64, None, 0, None, 0,
                  None, 0, None, _hip.PREC_FP32, st)


def test_halo_kernel_serves_the_model_layers_and_matches_the_implicit_gemm(H_):
    """Module level: the same BlockGen with the halo-resident kernel on and off (functional.USE_HALO_CONV) - outputs,
    input gradient and weight gradients agree to the split-product tolerances; with a FlatTrainer the per-step weight
    images give the same bits as images built per call."""
    from snn_for_object_detection_amd import BlockGen, Conv, LIF, Norm
    from snn_for_object_detection_amd import functional as HF
    from snn_for_object_detection_amd.trainer import FlatTrainer
    torch.manual_seed(11)
    T, B, H, W = 3, 2, 15, 19
    cfg = lambda: [Conv(64, 3), Norm(), LIF(), Conv(128, 3), Norm(), LIF(), Conv(128, 3)]  # noqa: E731
    blk = BlockGen(32, cfg()).cuda().train()
    x = (torch.rand(T, B, 32, H, W, device="cuda") < 0.3).float()
    res = {}
    for on in (True, False):
        HF.USE_HALO_CONV = on
        try:
            for m in blk.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.reset_running_stats()
            xin = x.clone().requires_grad_()
            out, _ = blk(xin)
            blk.zero_grad(set_to_none=True)
            out.square().mean().backward()
            res[on] = (out.detach().clone(), xin.grad.clone(), [p.grad.clone() for p in blk.parameters()])
        finally:
            HF.USE_HALO_CONV = True
    assert rel_err(res[True][0], res[False][0]) < 1e-5
    assert rel_err(res[True][1], res[False][1]) < 1e-4
    for a, b in zip(res[True][2], res[False][2]):
        assert rel_err(a, b) < 1e-4
    # trainer-kept images == per-call images (bit for bit)
    tr = FlatTrainer(blk, lr=1e-3)
    conv_w = [p for p in blk.parameters() if p.dim() == 4 and p.shape[1] >= 64][0]
    assert isinstance(conv_w._snn_wfrag, torch.Tensor) and isinstance(conv_w._snn_wtfrag, torch.Tensor)
    tr.zero_grad()
    xin = x.clone().requires_grad_()
    out_t, _ = blk(xin)
    out_t.square().mean().backward()
    tr.synchronize()
    assert torch.equal(xin.grad, res[True][1])


S2_CASES = [
    # N, H, W, Cin (dx channels), Cout (dy channels)
    (4, 120, 152, 64, 128),   # backbone 64 -> 128 (the slowest data gradient of the step)
    (3, 60, 76, 128, 256),    # neck-1 entry
    (5, 30, 38, 256, 256),    # neck-2 entry
    (6, 15, 19, 256, 256),    # neck-3 entry: odd input size (15 -> 8, 19 -> 10)
    (3, 9, 11, 64, 32),       # odd sizes, one K chunk
    (2, 1, 1, 64, 64),        # one-pixel images
    (2, 2, 2, 64, 64),        # even size 2: dx pixels (1, .) only reach dy row 0 through kh = 2
    (1, 17, 310, 64, 64),     # the widest dy row of the strip form (Wo = 155, PW = 156)
    # wider rows: 4 x 32 rectangles of dy cells of one image
    (2, 21, 330, 64, 64),     # Wo = 165 (5 rectangles + a partial one), Ho = 11 (partial bottom rectangle), odd input height
    (1, 90, 640, 64, 32),     # a 1 Mpx stage entry (180x320 dy would be the real one): even sizes, one K chunk
    (2, 7, 641, 128, 96),     # odd width 641 -> Wo = 321: the last dx column comes from kw = 1 only; three K chunks
]


@pytest.mark.parametrize("N,H,W,Cin,Cout", S2_CASES)
def test_one_pass_stride2_data_gradient_against_fp64(H_, N, H, W, Cin, Cout):
    _hip = H_
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    assert _hip.query("snn_conv3x3_s2_dgrad_supported", N, H, W, Cin, Ho, Wo, Cout) == 1
    torch.manual_seed(N + H + W)
    st = torch.cuda.current_stream().cuda_stream
    w = torch.randn(Cout, 3, 3, Cin) / (9 * Cin) ** 0.5
    gy = torch.randn(N, Ho, Wo, Cout)
    xr = torch.zeros(N, Cin, H, W, dtype=torch.float64, requires_grad=True)
    F.conv2d(xr, w.permute(0, 3, 1, 2).double(), stride=2, padding=1).backward(gy.permute(0, 3, 1, 2).double())
    dx_ref = xr.grad.permute(0, 2, 3, 1)
    wd, gyd = w.cuda(), gy.cuda()
    wt = torch.empty(Cin, 3, 3, Cout, device="cuda")
    _hip.call("snn_weight_transpose", wd.data_ptr(), wt.data_ptr(), Cout, 3, 3, Cin, st)
    img_t = _image(_hip, wt, Cin, Cout, 1, _hip.PREC_BF16X3)
    a1 = torch.randn(N, H, W, Cin, device="cuda")
    dx = torch.full((N, H, W, Cin), float("nan"), device="cuda")
    _hip.call("snn_conv3x3_s2_dgrad", gyd.data_ptr(), Cout, img_t.data_ptr(), dx.data_ptr(), Cin, N, H, W, Cin, Ho, Wo, Cout,
              None, 0, None, 0, _hip.PREC_BF16X3, st)
    assert torch.isfinite(dx).all()                                   # every dx pixel of every class was written
    assert rel_err(dx, dx_ref) < 3e-5
    # the four-launch implicit GEMM computes the same products
    dx_g = torch.empty_like(dx)
    _hip.call("snn_conv2d_dgrad", gyd.data_ptr(), Cout, wt.data_ptr(), None, dx_g.data_ptr(), Cin, N, H, W, Cin, Ho, Wo, Cout,
              3, 3, 2, 1, None, 0, None, 0, _hip.PREC_BF16X3, st)
    assert rel_err(dx, dx_g) < 3e-6
    # fused addend, determinism
    dx2 = torch.empty_like(dx)
    _hip.call("snn_conv3x3_s2_dgrad", gyd.data_ptr(), Cout, img_t.data_ptr(), dx2.data_ptr(), Cin, N, H, W, Cin, Ho, Wo, Cout,
              a1.data_ptr(), Cin, None, 0, _hip.PREC_BF16X3, st)
    assert torch.equal(dx2, dx + a1)


@pytest.mark.parametrize("T,B,H,W,C,Cdx", [(4, 5, 30, 38, 128, 128), (3, 2, 60, 76, 64, 64), (6, 5, 8, 10, 128, 128),
                                          (5, 3, 15, 19, 64, 128)])
def test_halo_data_gradient_with_fused_batchnorm_apply(H_, T, B, H, W, C, Cdx):
    """snn_conv3x3_halo_bn: dy = A*gx + B*y + C formed while gx is staged.  dy_out must equal snn_bn_bwd_apply's result
    bit for bit (same statement, same roundings) and dx the plain halo-resident data gradient of that dy, bit for bit -
    with a fused addend, timesteps changing inside a tile (small images) and cells no tile owns (pad cells)."""
    _hip = H_
    N = T * B
    assert _hip.query("snn_conv3x3_halo_bn_supported", N, H, W, C, Cdx, B) == 1
    torch.manual_seed(T * 10 + B)
    st = torch.cuda.current_stream().cuda_stream
    gx = torch.randn(N, H, W, C, device="cuda")
    y = torch.randn(N, H, W, C, device="cuda")
    coef = torch.randn(3, T, C, device="cuda")
    wt = torch.randn(Cdx, 3, 3, C, device="cuda") / (9 * C) ** 0.5            # [Cin][KH][KW][Cout] of the layer
    img = _image(_hip, wt, Cdx, C, 1, _hip.PREC_BF16X3)
    add = torch.randn(N, H, W, Cdx, device="cuda")
    # reference: the separate apply pass, then the plain kernel
    dy_ref = torch.empty_like(gx)
    _hip.call("snn_bn_bwd_apply", gx.data_ptr(), y.data_ptr(), C, coef[0].data_ptr(), coef[1].data_ptr(), coef[2].data_ptr(),
              dy_ref.data_ptr(), C, T, B * H * W, C, 0, st)
    dx_ref = torch.empty(N, H, W, Cdx, device="cuda")
    _hip.call("snn_conv3x3_halo", dy_ref.data_ptr(), C, img.data_ptr(), dx_ref.data_ptr(), Cdx, N, H, W, C, Cdx, add.data_ptr(),
              Cdx, None, 0, None, 0, None, _hip.PREC_BF16X3, st)
    dy_out = torch.full_like(gx, float("nan"))
    dx = torch.full_like(dx_ref, float("nan"))
    _hip.call("snn_conv3x3_halo_bn", gx.data_ptr(), y.data_ptr(), coef.data_ptr(), B, dy_out.data_ptr(), img.data_ptr(),
              dx.data_ptr(), Cdx, N, H, W, C, Cdx, add.data_ptr(), Cdx, None, 0, st)
    assert torch.equal(dy_out, dy_ref)        # every cell stored exactly once, by the tile that owns it
    assert torch.equal(dx, dx_ref)
    # against fp64 too (independent of the kernels above)
    dy64 = (coef[0].double().repeat_interleave(B, 0)[:, None, None, :] * gx.double()
            + coef[1].double().repeat_interleave(B, 0)[:, None, None, :] * y.double()
            + coef[2].double().repeat_interleave(B, 0)[:, None, None, :])
    xr = torch.zeros(N, Cdx, H, W, dtype=torch.float64, device="cuda", requires_grad=True)
    w_ohwi = wt.permute(3, 1, 2, 0).double()                                      # [Cout][KH][KW][Cin]
    F.conv2d(xr, w_ohwi.permute(0, 3, 1, 2), padding=1).backward(dy64.permute(0, 3, 1, 2))
    assert rel_err(dx - add, xr.grad.permute(0, 2, 3, 1)) < 5e-5


def test_deferred_batchnorm_apply_in_bottleneck_layers_is_bit_identical(H_):
    """Module level: a stack whose 3x3 convolutions sit behind train-mode BatchNorms, with the deferred apply on and off
    (functional.USE_DEFERRED_BN_APPLY): input gradient and every parameter gradient equal bit for bit; the apply launches
    of the covered layers are gone."""
    from snn_for_object_detection_amd import BlockGen, Conv, LIF, Norm, _hip
    from snn_for_object_detection_amd import functional as HF
    from snn_for_object_detection_amd.trainer import FlatTrainer
    T, B, H, W = 3, 2, 15, 19
    x = (torch.rand(T, B, 64, H, W, device="cuda") < 0.3).float()
    calls = []
    real_call = _hip.call

    def spy(name, *a):
        calls.append(name)
        return real_call(name, *a)

    res = {}
    for fused in (True, False):
        HF.USE_DEFERRED_BN_APPLY = HF.USE_DEFERRED_BN_APPLY_DGRAD = fused   # (the data-gradient form is opt-in)
        try:
            torch.manual_seed(12)
            blk = BlockGen(64, [Conv(64, 3), Norm(), LIF(), Conv(128, 3), Norm(), LIF(), Conv(128, 3), Norm(), LIF()])
            blk = blk.cuda().train()
            tr = FlatTrainer(blk, lr=1e-3)
            tr.zero_grad()
            xin = x.clone().requires_grad_()
            calls.clear()
            _hip.call = spy
            try:
                out, _ = blk(xin)
                (out * torch.linspace(0.5, 1.5, out.numel(), device="cuda").view(out.shape)).mean().backward()
            finally:
                _hip.call = real_call
            tr.synchronize()
            res[fused] = (xin.grad.clone(), tr.flat_grad.clone(), list(calls))
            assert not HF._PENDING_APPLY
        finally:
            HF.USE_DEFERRED_BN_APPLY, HF.USE_DEFERRED_BN_APPLY_DGRAD = True, False
    # layers 2 and 3 (64 -> 128, 128 -> 128: ONE channel tile of dx) fuse; layer 1's dx has 64 channels: it fuses too
    assert res[True][2].count("snn_conv3x3_halo_bn") == 3 and res[True][2].count("snn_bn_bwd_apply") == 0
    assert res[False][2].count("snn_conv3x3_halo_bn") == 0 and res[False][2].count("snn_bn_bwd_apply") == 3
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])


def test_staging_passes_without_memory_access_do_not_break_the_counted_weight_dma_wait(H_):
    """Regression (round 3): the k-step's ``s_waitcnt vmcnt(1)`` ("all but the youngest operation have landed" = the
    weight tile's LDS-DMA is complete) is only valid when the youngest operation - the staging pass of the next chunk - is
    a real memory access.  A pass whose lanes are all out of range is answered at once, AHEAD of the older DMA; the next tap
    then multiplied the previous weight tile on some tiles, differently from run to run.  It showed on the rectangle form
    of the stride-2 kernel at sizes with many tiles in flight (passes 6-8 lie outside its 165-cell halo); those steps now
    issue no load and drain the queue.  Same-size check: run-to-run equality and equality with the four-launch implicit
    GEMM (both are bf16 x 3 with the same products: the sums agree to fp32 rounding)."""
    _hip = H_
    st = torch.cuda.current_stream().cuda_stream
    N, H, W, Cin, Cout = 16, 360, 640, 64, 128
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    torch.manual_seed(0)
    w = torch.randn(Cout, 3, 3, Cin, device="cuda") / (9 * Cin) ** 0.5
    gy = torch.randn(N, Ho, Wo, Cout, device="cuda")
    wt = torch.empty(Cin, 3, 3, Cout, device="cuda")
    _hip.call("snn_weight_transpose", w.data_ptr(), wt.data_ptr(), Cout, 3, 3, Cin, st)
    img = _image(_hip, wt, Cin, Cout, 1, _hip.PREC_BF16X3)
    ref = torch.empty(N, H, W, Cin, device="cuda")
    _hip.call("snn_conv2d_dgrad", gy.data_ptr(), Cout, wt.data_ptr(), None, ref.data_ptr(), Cin, N, H, W, Cin, Ho, Wo, Cout, 3, 3,
              2, 1, None, 0, None, 0, _hip.PREC_BF16X3, st)
    outs = []
    for _ in range(3):
        dx = torch.full((N, H, W, Cin), float("nan"), device="cuda")
        _hip.call("snn_conv3x3_s2_dgrad", gy.data_ptr(), Cout, img.data_ptr(), dx.data_ptr(), Cin, N, H, W, Cin, Ho, Wo, Cout,
                  None, 0, None, 0, _hip.PREC_BF16X3, st)
        outs.append(dx)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert float((outs[0] - ref).abs().max()) < 1e-4 * float(ref.abs().max())
    # the stride-1 kernel on rectangles (its passes 7-8 lie outside the 204-cell halo) and on strips whose pad rows hold
    # whole passes (W + 1 = 77 cells): run-to-run equality at a size with many tiles in flight
    for (n, h, ww, ci, co) in ((32, 120, 152, 32, 32), (160, 60, 76, 64, 64)):
        x = torch.randn(n, h, ww, ci, device="cuda")
        wk = torch.randn(co, 3, 3, ci, device="cuda") / (9 * ci) ** 0.5
        im = _image(_hip, wk, co, ci, 0, _hip.PREC_FP16X3)
        ys = []
        for _ in range(3):
            y = torch.full((n, h, ww, co), float("nan"), device="cuda")
            _hip.call("snn_conv3x3_halo", x.data_ptr(), ci, im.data_ptr(), y.data_ptr(), co, n, h, ww, ci, co, None, 0, None, 0,
                      None, 0, None, _hip.PREC_FP16X3, st)
            ys.append(y)
        assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2])
        yg = torch.empty_like(ys[0])
        _hip.call("snn_conv2d_fwd", x.data_ptr(), ci, wk.data_ptr(), None, yg.data_ptr(), co, n, h, ww, ci, h, ww, co, 3, 3, 1, 1,
                  None, 0, None, 0, None, _hip.PREC_BF16X6, st)
        assert rel_err(ys[0], yg) < 2e-6
