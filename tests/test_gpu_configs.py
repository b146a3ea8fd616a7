"""The other BASELINE.json configurations as parity / shape cases (not bench lines):
configs[0] TinyYolo B=1 T=8 at the full GEN1 frame (against the oracle), configs[3] the 1Mpx 1280x720 frame
with 7 classes (shapes / finiteness / determinism at reduced B,T - fp32 activations of B=8,T=32 exceed HBM),
configs[4] the deep 12 x {Conv(64,3), Norm, LIF} backbone at T=128 (against the oracle at reduced frame size)."""
import pytest
import torch

from tests.util import make_pair, rel_err, synthetic_events, synthetic_labels

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S(hip_lib):
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import snn_for_object_detection_amd as pkg
    return pkg


def test_config0_gen1_frame_b1_t8_forward_matches_oracle(S):
    T, B, H, W = 8, 1, 240, 304
    product, oracle = make_pair(S.TinyYolo, num_classes=2, time_window=0)
    product.eval()
    oracle.eval()
    X = synthetic_events(T, B, H, W, p=0.05)
    with torch.no_grad():
        anchors, cls, box = product(X.cuda())
        anchors_r, cls_r, box_r = oracle(X)
    assert torch.equal(anchors.cpu(), anchors_r) and anchors.shape == (13545, 4)
    assert rel_err(cls, cls_r) < 1e-4 and rel_err(box, box_r) < 1e-4


def test_config3_1mpx_frame_runs(S):
    T, B, H, W = 2, 1, 720, 1280
    torch.manual_seed(2)
    model = S.TinyYolo(num_classes=7, time_window=0).cuda().train()
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == 4_263_104
    X, labels = synthetic_events(T, B, H, W, p=0.02).cuda(), synthetic_labels(B, n_classes=7).cuda()
    loss = model.training_step((X, labels))
    loss.backward()
    anchors, cls, box = model(X)
    assert anchors.shape == (170280, 4) and cls.shape == (B, 170280, 8) and box.shape == (B, 170280, 4)
    assert torch.isfinite(loss) and all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)


def test_config4_deep12_t128_matches_oracle(S):
    """12 x {Conv(64,3), Norm, LIF}, T=128.  Spiking nets are chaotic: ONE neuron whose potential sits within
    rounding of the threshold flips, and the flip spreads through the following layers (measured here: layers
    0-4 bit-identical, layer 5 one flip at t=28, ~5 % of the spikes differ from layer 8 on).  So the check is
    layer by layer: the early layers must be exact, every layer's firing rate must agree, and the first
    disagreement of the stack must be a single-neuron event."""
    from oracle.net import BlockRef
    from oracle.net import StateStorage as RefTap
    from snn_for_object_detection_amd import BlockGen, Conv, LIF, Norm
    from snn_for_object_detection_amd.layer_gen import StateStorage

    def cfg():
        layers = []
        for _ in range(12):
            layers += [Conv(64, 3), Norm(), LIF(state_storage=True)]
        return layers

    T, B, H, W = 128, 2, 12, 16
    torch.manual_seed(5)
    blk = BlockGen(2, cfg())
    ref = BlockRef(2, cfg())
    for m in blk.modules():
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
    ref.load_state_dict(blk.state_dict())
    blk = blk.cuda()
    X = synthetic_events(T, B, H, W, p=0.3, seed=4)

    def run(train):
        blk.train(train)
        ref.train(train)
        with torch.no_grad():
            blk(X.cuda())
            state = None
            for t in range(T):
                _, state = ref(X[t], state)

    run(True)    # T sequential running-stat updates on both sides
    run(False)   # eval: the taps record every layer's spikes
    taps_p = [m for m in blk.modules() if isinstance(m, StateStorage)]
    taps_r = [m for m in ref.modules() if isinstance(m, RefTap)]
    assert len(taps_p) == len(taps_r) == 12
    rates = []
    for k, (tp, tr) in enumerate(zip(taps_p, taps_r)):
        zp, zr = tp.get_spikes().cpu(), tr.get_spikes()
        assert zp.shape == zr.shape == (T, B, 64, H, W)
        mism = (zp != zr).float().mean().item()
        rates.append((mism, zp.mean().item(), zr.mean().item()))
        if k < 1:  # layers before the first near-threshold flip are bit-identical; WHICH layer flips first depends
            assert mism == 0.0, (k, mism)  # on the rounding pattern of the arithmetic mode (layer 3 .. 5 measured)
        assert abs(zp.mean().item() - zr.mean().item()) < 0.1 * zr.mean().item(), (k, rates[-1])
    first_bad = next((k for k, r in enumerate(rates) if r[0] > 0), None)
    if first_bad is not None:   # the stack diverges from a handful of near-threshold neurons, not from a wrong layer
        assert rates[first_bad][0] < 1e-3, rates
