"""The other BASELINE.json configurations as parity cases:
configs[0] TinyYolo B=1 T=8 at the full GEN1 frame (against the oracle);
configs[3] the 1Mpx 1280x720 frame with 7 classes: against the oracle at B=1, T=4 (what the CPU finishes in
seconds) and at FULL size (B=8, T=32, ~250 GiB live) through size-independent properties (bitwise-deterministic
training step, chunked == whole sequence);
configs[4] the deep 12 x {Conv(64,3), Norm, LIF} backbone at T=128: every one of the 12 layers against the oracle
(teacher-forced: layer k is fed the ORACLE's layer k-1 spikes, so a near-threshold flip cannot hide a wrong later
layer), plus BPTT gradients through T=128 on a 3-layer stack."""
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


def _train_mode_layerwise(S, T, H, W, classes, p=0.05, B=1):
    """Forward pass in TRAIN mode (per-timestep batch statistics) on product and oracle with every LIF / LI output
    recorded (the taps are switched to recording although BatchNorm keeps training).  Returns the predictions and,
    per tapped layer in execution order, (name, oracle spikes, mismatching spikes per timestep | LI rel. error)."""
    from oracle.net import StateStorage as RefTap
    from snn_for_object_detection_amd.layer_gen import StateStorage
    product, oracle = make_pair(S.TinyYolo, num_classes=classes, time_window=0, state_storage=True)
    X = synthetic_events(T, B, H, W, p=p)
    product.train()
    oracle.train()
    for m in list(product.modules()) + list(oracle.modules()):
        if isinstance(m, (StateStorage, RefTap)):
            m.training = False
    with torch.no_grad():
        preds, preds_r = product(X.cuda()), oracle(X)
    taps, taps_r = product.spike_taps(), oracle.spike_taps()
    assert list(taps) == list(taps_r) and len(taps_r) == 22
    layers = []
    for name, zr in taps_r.items():
        z = taps[name].cpu()
        assert z.shape == zr.shape, name
        if "head_net" in name:
            layers.append((name, None, rel_err(z, zr)))
        else:
            layers.append((name, float(zr.sum()), [int((z[t] != zr[t]).sum()) for t in range(T)], float(z.sum())))
    return preds, preds_r, layers


def test_config0_gen1_frame_train_mode_all_layers_match_oracle(S):
    """BASELINE configs[0] shape (TinyYolo, GEN1 304x240, B=1, T=8) with batch-statistics BatchNorm, where all 19 LIF
    layers are firing by t = 8: every spike tensor and the predictions against the oracle."""
    preds, preds_r, layers = _train_mode_layerwise(S, 8, 240, 304, 2)
    spikes = mism = 0
    for row in layers:
        if row[1] is None:
            assert row[2] < 1e-4, row
        else:
            assert row[1] > 0, row          # the layer is alive
            spikes += row[1]
            mism += sum(row[2])
    assert mism <= 1e-5 * spikes, (mism, spikes)
    if mism == 0:
        assert rel_err(preds[1], preds_r[1]) < 1e-4 and rel_err(preds[2], preds_r[2]) < 1e-4


def test_config1_gen1_b5_t32_forward_matches_oracle_layer_by_layer(S):
    """BASELINE configs[1] at its FULL size (TinyYolo, GEN1 304x240, B=5, T=32: the workload of the bench line), forward
    pass in train mode against the oracle - 26 M neurons per step over 32 steps, 845 M neuron-timesteps.

    Measured (`tools/diag_config1.py`): every one of the 19 LIF layers is bit-identical to the oracle for the first four
    timesteps; the first disagreement anywhere is ONE neuron of a 60x76 layer at t = 4 (a potential within fp32 rounding
    of the threshold: the product's split-precision convolution and torch's CPU convolution differ by 5e-7), the
    120x152 entry layer differs in 6 of its 3.29 M spikes, and from there batch statistics and recurrence spread the
    difference (chaos: a third of the spikes of the smallest maps by t = 32) while every layer's firing rate stays within
    0.4 % of the oracle's.  Asserted: that structure - exact prefix, single-neuron first event, a near-exact entry
    layer, firing rates within 1 %.  (The same run with the separate statistics pass instead of the convolution
    epilogues gives the identical mismatch counts, layer for layer.)"""
    T = 32
    preds, preds_r, layers = _train_mode_layerwise(S, T, 240, 304, 2, B=5)
    assert torch.equal(preds[0].cpu(), preds_r[0]) and preds[0].shape == (13545, 4)
    lif = [row for row in layers if row[1] is not None]
    assert len(lif) == 19
    firsts = [next((t for t, n in enumerate(row[2]) if n > 0), T) for row in lif]
    t_star = min(firsts)
    assert t_star >= 3, firsts                       # >= 3 x 26 M neuron-timesteps bit-identical in every layer
    if t_star < T:
        k = firsts.index(t_star)                     # the shallowest layer that disagrees at t*
        assert lif[k][2][t_star] <= 3, (lif[k][0], lif[k][2][t_star])
    entry = lif[0]
    assert sum(entry[2]) <= 1e-5 * entry[1], (sum(entry[2]), entry[1])
    for name, n_ref, per_t, n_prod in lif:
        assert n_ref > 0 and abs(n_prod - n_ref) <= 0.01 * n_ref, (name, n_ref, n_prod)
    if t_star == T:
        assert rel_err(preds[1], preds_r[1]) < 1e-4 and rel_err(preds[2], preds_r[2]) < 1e-4
    else:
        assert rel_err(preds[1], preds_r[1]) < 0.3 and rel_err(preds[2], preds_r[2]) < 0.3


def test_config3_1mpx_frame_b1_t8_matches_oracle_layer_by_layer(S):
    """BASELINE configs[3] frame (1280x720, 7 classes) at B=1, T=8 against the oracle, train-mode BatchNorm.

    With 62 M neurons per step a neuron whose potential sits within rounding of the threshold is a certainty, and with
    batch statistics ONE flipped spike shifts every neuron of the following layers (measured: the first one to four
    layers bit-exact over 1-2.3 M spikes - which neuron flips first depends on the rounding pattern of the build -
    then one flipped spike, thousands of differing spikes in the small deep maps by t = 8: SURVEY section 7,
    "spike-flip sensitivity").  So the statement is layer by layer: the first (360x640) layer and everything else
    before the first disagreement is EXACT, the first disagreement is a single-neuron event, not a wrong layer, and
    after it firing rates and predictions agree to the level that chaos allows."""
    T = 8
    preds, preds_r, layers = _train_mode_layerwise(S, T, 720, 1280, 7)
    assert torch.equal(preds[0].cpu(), preds_r[0]) and preds[0].shape == (170280, 4)
    lif = [row for row in layers if row[1] is not None]
    first_bad = next((k for k, row in enumerate(lif) if sum(row[2]) > 0), None)
    exact_spikes = sum(row[1] for row in (lif if first_bad is None else lif[:first_bad]))
    assert first_bad is None or first_bad >= 1, lif[:2]            # the 360x640 entry layer is exact
    assert exact_spikes > 1e6
    if first_bad is not None:
        per_t = lif[first_bad][2]
        first_t = next(t for t, n in enumerate(per_t) if n > 0)
        assert per_t[first_t] <= 3, lif[first_bad]                 # a near-threshold neuron, not a wrong layer
    for name, n_ref, per_t, n_prod in lif:
        assert n_ref > 0 and abs(n_prod - n_ref) <= 0.1 * n_ref, (name, n_ref, n_prod)
    if first_bad is None:
        assert rel_err(preds[1], preds_r[1]) < 1e-4 and rel_err(preds[2], preds_r[2]) < 1e-4
    else:
        assert rel_err(preds[1], preds_r[1]) < 0.2 and rel_err(preds[2], preds_r[2]) < 0.2


def test_config3_1mpx_full_size_b8_t32(S):
    """BASELINE configs[3] at full size on one MI355X: a training step is bitwise deterministic, and the sequence run
    as two halves with the carried state equals one layer-major pass (eval mode), bit for bit."""
    import gc
    from snn_for_object_detection_amd.trainer import FlatTrainer
    T, B, H, W = 32, 8, 720, 1280
    X, labels = synthetic_events(T, B, H, W, p=0.02).cuda(), synthetic_labels(B, n_classes=7).cuda()
    results = []
    for _ in range(2):
        torch.manual_seed(2)
        model = S.TinyYolo(num_classes=7, time_window=0).cuda().train()
        tr = FlatTrainer(model)
        tr.zero_grad()
        loss = model.training_step((X, labels))
        loss.backward()
        tr.synchronize()
        results.append((loss.detach().clone(), tr.flat_grad.clone()))
        del model, tr, loss
        gc.collect()
    assert torch.equal(results[0][0], results[1][0]) and torch.equal(results[0][1], results[1][1])
    g = results[0][1]
    assert torch.isfinite(results[0][0]) and torch.isfinite(g).all() and g.abs().max() > 0
    del results, g
    gc.collect()
    torch.manual_seed(2)
    model = S.TinyYolo(num_classes=7, time_window=0).cuda().eval()
    with torch.no_grad():
        (anchors, cls_w, box_w), _ = model._forward_impl(X, None)
        (_, _, _), st_h = model._forward_impl(X[: T // 2], None)
        (_, cls_c, box_c), _ = model._forward_impl(X[T // 2:], st_h)
    assert anchors.shape == (170280, 4) and cls_w.shape == (B, 170280, 8) and box_w.shape == (B, 170280, 4)
    assert torch.equal(cls_w, cls_c) and torch.equal(box_w, box_c)
    del model, X
    gc.collect()
    torch.cuda.empty_cache()


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


def _oracle_layer(conv, bn, x):
    """One {Conv, Norm(train), LIF} layer of the oracle, time-outer, also returning the pre-reset potentials."""
    from oracle.neurons import LIFCell, LIFParameters
    cell, p, dt = LIFCell(), LIFParameters(), 0.001
    state, zs, vds = None, [], []
    for t in range(x.shape[0]):
        cur = bn(conv(x[t]))
        if state is None:
            state = cell.initial_state(cur)
        i_new = state.i + cur                                          # the statements of lif_feed_forward_step
        vds.append(state.v + dt * p.tau_mem_inv * ((p.v_leak - state.v) + i_new))
        z, state = cell(cur, state)
        zs.append(z)
    return torch.stack(zs), torch.stack(vds)


def test_config4_deep12_t128_teacher_forced_every_layer(S):
    """Each of the 12 layers is compared with the oracle on IDENTICAL inputs (the oracle's previous-layer spikes): all
    spikes equal except neurons whose FIRST disagreement happens with the oracle's pre-reset potential within 1e-5 of
    the threshold (after a flip that neuron's own later steps differ by construction)."""
    from snn_for_object_detection_amd import BlockGen, Conv, LIF, Norm
    T, B, H, W, C = 128, 2, 24, 32, 64
    torch.manual_seed(5)
    x = synthetic_events(T, B, H, W, p=0.3, seed=4)
    cin = 2
    with torch.no_grad():
        for k in range(12):
            conv = torch.nn.Conv2d(cin, C, 3, padding=1, bias=False)
            torch.nn.init.kaiming_normal_(conv.weight, mode="fan_out", nonlinearity="relu")
            bn = torch.nn.BatchNorm2d(C)
            bn.bias = None
            bn.train()
            blk = BlockGen(cin, [Conv(C, 3), Norm(), LIF()])
            blk.load_state_dict({"net.0.0.weight": conv.weight, "net.0.1.weight": bn.weight,
                                 "net.0.1.running_mean": bn.running_mean, "net.0.1.running_var": bn.running_var,
                                 "net.0.1.num_batches_tracked": bn.num_batches_tracked})
            blk = blk.cuda().train()
            z_ref, vdec_ref = _oracle_layer(conv, bn, x)
            z, _ = blk(x.cuda())
            z = z.cpu()
            assert z.shape == z_ref.shape == (T, B, C, H, W)
            diff = z != z_ref
            if diff.any():
                first = diff.float().argmax(dim=0)                      # first disagreeing timestep per neuron
                bad = diff.any(dim=0)
                margin = (vdec_ref.gather(0, first.unsqueeze(0)).squeeze(0) - 1.0).abs()[bad]
                assert margin.max().item() < 1e-5, (k, margin.max().item())
                assert bad.float().mean().item() < 1e-3, (k, bad.float().mean().item())
            assert 0.0 < z_ref.mean().item() < 0.5, k                  # the layer is alive (not all-zero / saturated)
            assert rel_err(blk.net[0][1].running_var, bn.running_var) < 1e-5
            x, cin = z_ref, C                                           # teacher forcing


def test_config4_bptt_t128_gradients_match_oracle(S):
    """BPTT through T=128 on a 3-layer {Conv(64,3), Norm, LIF} stack, train mode: input and parameter gradients
    within 1e-3 (relative L2) of the oracle when the spike trains agree exactly; a near-threshold flip perturbs
    the surrogate-gradient path of that neuron, which is then allowed 2e-2."""
    from oracle.net import BlockRef
    from snn_for_object_detection_amd import BlockGen, Conv, LIF, Norm

    def cfg():
        layers = []
        for _ in range(3):
            layers += [Conv(64, 3), Norm(), LIF()]
        return layers

    T, B, H, W = 128, 2, 12, 16
    torch.manual_seed(7)
    blk, ref = BlockGen(2, cfg()), BlockRef(2, cfg())
    for m in blk.modules():
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
    ref.load_state_dict(blk.state_dict())
    blk = blk.cuda().train()
    ref.train()
    x = synthetic_events(T, B, H, W, p=0.3, seed=9)
    xd, xr = x.cuda().requires_grad_(), x.clone().requires_grad_()
    out, _ = blk(xd)
    state, outs = None, []
    for t in range(T):
        o, state = ref(xr[t], state)
        outs.append(o)
    out_r = torch.stack(outs)
    g = torch.randn(out_r.shape, generator=torch.Generator().manual_seed(3))
    (out * g.cuda()).sum().backward()
    (out_r * g).sum().backward()
    flips = int((out.detach().cpu() != out_r.detach()).sum())
    tol = 1e-3 if flips == 0 else 2e-2
    assert flips <= 1e-5 * out_r.numel(), flips
    assert rel_err(xd.grad, xr.grad) < tol, (flips, rel_err(xd.grad, xr.grad))
    for (name, pd), pr in zip(blk.named_parameters(), ref.parameters()):
        assert rel_err(pd.grad, pr.grad) < tol, (name, flips, rel_err(pd.grad, pr.grad))


def _randomize_running_stats(model, seed):
    """Eval-mode BatchNorm with NON-trivial running statistics (fresh buffers are mean 0 / var 1: an identity)."""
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_((0.2 * torch.randn(m.running_mean.shape, generator=g)).to(m.running_mean.device))
            m.running_var.copy_((0.5 + torch.rand(m.running_var.shape, generator=g)).to(m.running_var.device))


def test_config3_1mpx_full_size_backward_is_additive_over_a_batch_split(S):
    """The FULL-size backward pass (1280x720, 7 classes, B=8, T=32: data-gradient, weight-gradient and scan kernels
    addressing > 2 GiB tensors) checked against something other than itself: with eval-mode BatchNorm the samples
    decouple, so for a loss that is a SUM over samples the parameter gradient of the batch of 8 equals the gradient of
    samples 0-3 plus the gradient of samples 4-7.  Different batch sizes mean different tile boundaries, split-K
    partitions and pixel offsets for the same samples - a wrong (but repeatable) address past 2 GiB cannot pass.
    Sums are ordered and fp32: agreement to 2e-5 per parameter tensor."""
    import gc
    from snn_for_object_detection_amd.trainer import FlatTrainer
    T, B, H, W = 32, 8, 720, 1280
    X = synthetic_events(T, B, H, W, p=0.02).cuda()
    A = 170280
    g = torch.Generator().manual_seed(11)
    P_cls = torch.randn(B, A, 8, generator=g).cuda() / A
    P_box = torch.randn(B, A, 4, generator=g).cuda() / A

    def grads(sl):
        torch.manual_seed(2)
        model = S.TinyYolo(num_classes=7, time_window=0).cuda()
        _randomize_running_stats(model, 3)
        model.eval()
        tr = FlatTrainer(model)
        tr.zero_grad()
        _, cls, box = model(X[:, sl])
        loss = (cls * P_cls[sl]).sum() + (box * P_box[sl]).sum()
        loss.backward()
        tr.synchronize()
        out = tr.flat_grad.clone()
        offsets, names = list(tr._offsets), [n for n, p in model.named_parameters() if p.requires_grad]
        del model, tr, loss, cls, box
        gc.collect()
        torch.cuda.empty_cache()
        return out, offsets, names

    g_all, offsets, names = grads(slice(0, 8))
    g_lo, _, _ = grads(slice(0, 4))
    g_hi, _, _ = grads(slice(4, 8))
    g_sum = g_lo + g_hi
    assert torch.isfinite(g_all).all() and g_all.abs().max() > 0
    assert not torch.equal(g_lo, g_hi)
    worst = ("", 0.0)
    for k, name in enumerate(names):
        a, b = g_all[offsets[k]:offsets[k + 1]], g_sum[offsets[k]:offsets[k + 1]]
        if b.norm() > 0:
            e = rel_err(a, b)
            if e > worst[1]:
                worst = (name, e)
    assert worst[1] < 2e-5, worst
    del X, g_all, g_lo, g_hi, g_sum
    gc.collect()
    torch.cuda.empty_cache()


def _deep12(S):
    from snn_for_object_detection_amd import BlockGen, Conv, LIF, Norm
    layers = []
    for _ in range(12):
        layers += [Conv(64, 3), Norm(), LIF()]
    torch.manual_seed(5)
    blk = BlockGen(2, layers)
    for m in blk.modules():
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
    return blk.cuda()


def test_config4_deep12_real_per_gpu_share_b2_t128(S):
    """BASELINE configs[4] at the size one GPU really gets (240x304, B=2 of the 16, T=128, ~170 GiB): (1) the backward
    scan in four 32-step segments (functional.SCAN_SEGMENT_T) equals the one-launch scan at that size (the per-(t, c)
    BatchNorm sums are added over another block partition and the parameter gradients over another segment order, so
    "equal" is to fp32 summation order: 1e-5 on the whole gradient; the small-size test checks gx bit for bit);
    (2) with eval-mode BatchNorm the gradient of the batch of 2 is the sum of the two single-sample gradients
    (sum-over-samples read-out) - the size-triggered paths (long pixel ranges of the weight gradient, > 2 GiB pixel
    splits, segmented scans) against an independent decomposition of the same work."""
    import gc
    from snn_for_object_detection_amd import functional as HF
    from snn_for_object_detection_amd.trainer import FlatTrainer
    T, B, H, W = 128, 2, 240, 304
    X = synthetic_events(T, B, H, W, p=0.3, seed=4).cuda()
    probe = torch.randn(B, 64, H, W, generator=torch.Generator().manual_seed(3)).cuda() / (H * W)

    def grads(sl, train, segment):
        blk = _deep12(S)
        _randomize_running_stats(blk, 7)
        blk.train(train)
        tr = FlatTrainer(blk)
        tr.zero_grad()
        old = HF.SCAN_SEGMENT_T
        HF.SCAN_SEGMENT_T = segment
        try:
            out, _ = blk(X[:, sl], last_only=True)
            (out * probe[sl]).sum().backward()
            tr.synchronize()
        finally:
            HF.SCAN_SEGMENT_T = old
        g = tr.flat_grad.clone()
        del blk, tr, out
        gc.collect()
        torch.cuda.empty_cache()
        return g

    # (1) train mode (the bench's mode): segmented == one launch
    g_seg = grads(slice(0, 2), True, 32)
    g_one = grads(slice(0, 2), True, None)
    assert torch.isfinite(g_seg).all() and g_seg.abs().max() > 0
    assert rel_err(g_seg, g_one) < 1e-5
    del g_seg, g_one
    # (2) eval mode: additive over the batch split
    g_all = grads(slice(0, 2), False, 32)
    g_sum = grads(slice(0, 1), False, 32) + grads(slice(1, 2), False, 32)
    assert g_all.abs().max() > 0 and rel_err(g_all, g_sum) < 2e-5
    del X, g_all, g_sum
    gc.collect()
    torch.cuda.empty_cache()
