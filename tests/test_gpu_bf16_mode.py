"""The labelled bf16 THROUGHPUT mode (``forward_precision = backward_precision = "bf16"``: conv operands rounded once to
bf16, one MFMA product per multiply-add, fp32 accumulation and storage) - the dtype BASELINE configs[1] names.

It is opt-in and NOT a parity mode: 8 significant bits in the products flip near-threshold neurons, and spike trains
diverge from the fp32 reference layer by layer.  Stated tolerances (measured, then rounded up):
  * single convolution (fwd / dgrad / wgrad) against fp64: relative L2 error < 1e-2 (bf16 rounding of both operands);
  * TinyYolo training step at 32x48, T=4 against the CPU oracle: loss within 5 %, first-layer weight gradient within
    25 % relative L2, spikes of the FIRST LIF layer (exact {0,1} event input, bf16-rounded weights) differ for < 2 % of
    the neurons; everything stays finite;
  * the default modes are untouched by it (a conv in default mode after a bf16 call is still fp32-grade).
"""
import pytest
import torch
import torch.nn.functional as F

from tests.util import make_pair, rel_err, synthetic_events, synthetic_labels

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S(hip_lib):
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import snn_for_object_detection_amd as pkg
    return pkg


@pytest.mark.parametrize("Cin,Cout,k,s,H,W,N", [(64, 128, 3, 1, 30, 38, 6), (128, 64, 1, 1, 24, 20, 4),
                                                (64, 128, 3, 2, 37, 52, 330), (32, 32, 3, 1, 40, 44, 90)])
def test_bf16_convolution_error_level(S, Cin, Cout, k, s, H, W, N):
    HF = S.functional
    torch.manual_seed(Cin + Cout + k)
    x = torch.randn(N, 1, Cin, H, W)
    w = torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    xr, wr = x.double().requires_grad_(), w.double().requires_grad_()
    yr = F.conv2d(xr.flatten(0, 1), wr, stride=s, padding=k // 2)
    gy = torch.randn_like(yr)
    yr.backward(gy)
    xd, wd = x.cuda().requires_grad_(), w.cuda().requires_grad_()
    yd = HF.conv2d(xd, wd, stride=s, padding=k // 2, forward_precision="bf16", backward_precision="bf16")
    yd.backward(gy.float().view(yd.shape).cuda())
    errs = (rel_err(yd.flatten(0, 1), yr), rel_err(xd.grad, xr.grad), rel_err(wd.grad, wr.grad))
    assert all(1e-4 < e < 1e-2 for e in errs), errs          # bf16-level: far from fp32-grade, far from wrong
    y32 = HF.conv2d(x.cuda(), w.cuda(), stride=s, padding=k // 2)   # session default: still fp32-grade
    assert rel_err(y32.flatten(0, 1), yr) < 2e-6


def test_bf16_training_step_tolerance(S):
    T, B, H, W = 4, 2, 32, 48
    product, oracle = make_pair(S.TinyYolo, num_classes=2, time_window=0, state_storage=True)
    X, labels = synthetic_events(T, B, H, W, p=0.08), synthetic_labels(B)
    product.train()
    oracle.train()
    loss_ref = oracle.training_step((X, labels))
    loss_ref.backward()
    HF = S.functional
    HF.set_forward_precision("bf16")
    HF.set_backward_precision("bf16")
    try:
        loss = product.training_step((X.cuda(), labels.cuda()))
        loss.backward()
        # first-layer spikes in eval-style taps: run the first block alone on the same input
        first = product.base_net.net.net[0]
        with torch.no_grad():
            y = first[0](X.cuda())
            z, _ = HF.affine_neuron(y, 1, None, bn=first[1])
    finally:
        HF.set_forward_precision(HF.DEFAULT_FORWARD_PRECISION)
        HF.set_backward_precision(HF.DEFAULT_BACKWARD_PRECISION)
    assert torch.isfinite(loss) and abs(loss.item() - loss_ref.item()) <= 0.05 * abs(loss_ref.item())
    g, g_ref = product.base_net.net.net[0][0].weight.grad, oracle.base_net.net.net[0][0].weight.grad
    assert torch.isfinite(g).all() and rel_err(g, g_ref) < 0.25, rel_err(g, g_ref)
    for p in product.parameters():
        assert p.grad is None or torch.isfinite(p.grad).all()
    ref_first = oracle.base_net.net.net[0]
    with torch.no_grad():
        state, zs = None, []
        ref_first[1].train()
        for t in range(T):
            zt, state = ref_first[2].module(ref_first[1](ref_first[0](X[t])), state)
            zs.append(zt)
    assert (z.cpu() != torch.stack(zs)).float().mean().item() < 0.02
    with pytest.raises(ValueError):
        HF.set_forward_precision("bf16x2")
