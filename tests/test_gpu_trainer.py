"""Flat-buffer trainer on the device: zero-copy gradient slots == autograd gradients, fused Adamax ==
torch.optim.Adamax, and a few full steps reduce the loss."""
import pytest
import torch

from tests.util import rel_err, synthetic_events, synthetic_labels

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S(hip_lib):
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import snn_for_object_detection_amd as pkg
    return pkg


def test_grad_slots_equal_autograd_and_adamax_matches_torch(S):
    from snn_for_object_detection_amd.trainer import FlatTrainer
    T, B, H, W = 3, 2, 32, 48
    X, labels = synthetic_events(T, B, H, W, p=0.1).cuda(), synthetic_labels(B).cuda()
    torch.manual_seed(2)
    a = S.TinyYolo(num_classes=2, time_window=0).cuda().train()
    torch.manual_seed(2)
    b = S.TinyYolo(num_classes=2, time_window=0).cuda().train()
    tr = FlatTrainer(a, lr=1e-3)
    opt = torch.optim.Adamax([p for p in b.parameters() if p.requires_grad], lr=1e-3)
    for it in range(3):
        tr.zero_grad()
        la = a.training_step((X, labels))
        la.backward()
        opt.zero_grad()
        lb = b.training_step((X, labels))
        lb.backward()
        assert abs(la.item() - lb.item()) < 1e-4 * abs(lb.item())
        if it == 0:
            by_name = tr.grads_by_name(a)
            for name, p in b.named_parameters():
                if p.requires_grad:
                    assert rel_err(by_name[name], p.grad) < 1e-4, name   # same kernels; the scan's LDS atomics reorder sums
        tr.step()
        opt.step()
        for (n, pa), pb in zip(a.named_parameters(), b.parameters()):
            if pa.requires_grad:
                assert rel_err(pa, pb) < 1e-5, (it, n)


def test_training_reduces_loss(S):
    from snn_for_object_detection_amd.trainer import FlatTrainer
    T, B, H, W = 4, 2, 32, 48
    X, labels = synthetic_events(T, B, H, W, p=0.1).cuda(), synthetic_labels(B).cuda()
    torch.manual_seed(0)
    m = S.TinyYolo(num_classes=2, time_window=0).cuda().train()
    tr = FlatTrainer(m, lr=2e-3)
    losses = []
    for _ in range(12):
        tr.zero_grad()
        loss = m.training_step((X, labels))
        loss.backward()
        tr.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0], losses


def test_time_outer_training_accumulates_into_grad_slots(S):
    """The reference's literal time loop (T single-step calls, state carried) under the FlatTrainer: every conv /
    norm is used T times per step, so the gradient slots are written once and then accumulated T-1 times on the
    weight-gradient side stream.  Must equal the layer-major step."""
    from snn_for_object_detection_amd.trainer import FlatTrainer
    T, B, H, W = 3, 2, 32, 48
    X, labels = synthetic_events(T, B, H, W, p=0.1).cuda(), synthetic_labels(B).cuda()
    grads = []
    for time_outer in (False, True):
        torch.manual_seed(2)
        m = S.TinyYolo(num_classes=2, time_window=0).cuda().train()
        tr = FlatTrainer(m)
        tr.zero_grad()
        preds = m(X, time_outer=time_outer)
        loss = m._loss(preds, labels)
        loss.backward()
        tr.synchronize()
        grads.append((loss.detach().clone(), tr.flat_grad.clone()))
    assert torch.equal(grads[0][0], grads[1][0])
    assert rel_err(grads[1][1], grads[0][1]) < 1e-5


def test_odd_channel_counts_and_tiny_batches(S):
    """Channel counts that are not multiples of 4 (scalar kernel paths), B=1, T=1."""
    from oracle.net import BlockRef
    from snn_for_object_detection_amd import BlockGen, Conv, Dense, LIF, Norm, Pass, Pool

    def cfg():
        return [Conv(6, 3, 2), Norm(), LIF(), Dense([[Conv(5, 3), Norm(bias=True), LIF()], [Pass()]]), Pool("A"),
                Conv(7, 1)]

    torch.manual_seed(8)
    blk, ref = BlockGen(3, cfg()), BlockRef(3, cfg())
    ref.load_state_dict(blk.state_dict())
    blk = blk.cuda()
    for T, B in ((1, 1), (3, 2)):
        x = 3.0 * torch.rand(T, B, 3, 13, 10)
        xd, xr = x.cuda().requires_grad_(), x.clone().requires_grad_()
        out, _ = blk(xd)
        state, outs = None, []
        for t in range(T):
            o, state = ref(xr[t], state)
            outs.append(o)
        out_r = torch.stack(outs)
        assert out.shape == out_r.shape == (T, B, 7, 3, 2)
        assert rel_err(out, out_r) < 1e-4
        g = torch.randn_like(out_r)
        blk.zero_grad()
        ref.zero_grad()
        (out * g.cuda()).sum().backward()
        (out_r * g).sum().backward()
        assert rel_err(xd.grad, xr.grad) < 1e-3
        for pd, pr in zip(blk.parameters(), ref.parameters()):
            if pr.grad.norm() > 1e-7:
                assert rel_err(pd.grad, pr.grad) < 1e-3


def test_repeated_steps_and_cached_transposes_are_reproducible(S):
    """Steps with FlatTrainer's cached w^T: the same batch gives bit-identical loss and gradients step after step, new
    labels are picked up, and a torch-side in-place weight update invalidates the cached transposes (per-layer
    fallback) without changing the result."""
    from snn_for_object_detection_amd.trainer import FlatTrainer
    T, B, H, W = 3, 2, 32, 48
    X = synthetic_events(T, B, H, W, p=0.1).cuda()
    labels, labels2 = synthetic_labels(B).cuda(), synthetic_labels(B, seed=7).cuda()
    torch.manual_seed(2)
    m = S.TinyYolo(num_classes=2, time_window=0).cuda().train()
    tr = FlatTrainer(m, lr=1e-3)

    def grads(lab):
        tr.zero_grad()
        loss = m.training_step((X, lab))
        loss.backward()
        tr.synchronize()
        return loss.detach().clone(), tr.flat_grad.clone()

    l_a, g_a = grads(labels)
    l_b, g_b = grads(labels)
    assert torch.equal(l_a, l_b) and torch.equal(g_a, g_b)
    l_new, _ = grads(labels2)
    assert not torch.equal(l_new, l_a)
    # a torch-side in-place weight update invalidates the cached transposes (per-layer fallback)
    conv_w = next(p for p in m.parameters() if p.dim() == 4 and p.shape[1] > 2)
    with torch.no_grad():
        conv_w.mul_(1.5)
    assert conv_w._snn_wt_version != conv_w._version
    _, g_fallback = grads(labels)
    tr.refresh_transposed_weights()
    assert conv_w._snn_wt_version == conv_w._version
    _, g_cached = grads(labels)
    assert torch.equal(g_fallback, g_cached)


def test_optimizer_state_round_trip_and_torch_interchange(S):
    """``FlatTrainer.state_dict()`` has torch.optim.Adamax's layout: a torch optimiser loads it and continues exactly
    like the fused kernel does; a fresh FlatTrainer resumes from it bit for bit (checkpoint / resume)."""
    from snn_for_object_detection_amd.trainer import FlatTrainer
    T, B, H, W = 3, 2, 32, 48
    X, labels = synthetic_events(T, B, H, W, p=0.1).cuda(), synthetic_labels(B).cuda()

    def fresh():
        torch.manual_seed(2)
        return S.TinyYolo(num_classes=2, time_window=0).cuda().train()

    def one_step(model, trainer):
        trainer.zero_grad()
        model.training_step((X, labels)).backward()
        trainer.step()

    a = fresh()
    tr_a = FlatTrainer(a, lr=2e-3)
    for _ in range(2):
        one_step(a, tr_a)
    ckpt_model = {k: v.clone() for k, v in a.state_dict().items()}
    ckpt_opt = tr_a.state_dict()
    assert set(ckpt_opt) == {"state", "param_groups"} and len(ckpt_opt["state"]) == len(tr_a.params)
    assert ckpt_opt["state"][0]["exp_avg"].shape == tr_a.params[0].shape
    one_step(a, tr_a)                                        # the step to reproduce after a resume
    # resume into a new model + trainer
    b = fresh()
    b.load_state_dict(ckpt_model)
    tr_b = FlatTrainer(b, lr=1e-3)                           # lr comes from the checkpoint
    tr_b.load_state_dict(ckpt_opt)
    assert tr_b.step_count == 2 and tr_b.lr == 2e-3
    one_step(b, tr_b)
    for (n, pa), pb in zip(a.named_parameters(), b.parameters()):
        assert torch.equal(pa, pb), n
    # the same state drives torch.optim.Adamax
    c = fresh()
    c.load_state_dict(ckpt_model)
    opt = torch.optim.Adamax([p for p in c.parameters() if p.requires_grad], lr=1e-3)
    opt.load_state_dict(ckpt_opt)
    opt.zero_grad()
    c.training_step((X, labels)).backward()
    opt.step()
    for (n, pa), pc in zip(a.named_parameters(), c.parameters()):
        if pa.requires_grad:
            assert rel_err(pc, pa) < 1e-5, n


def test_parameters_without_gradient_are_left_alone(S):
    """torch.optim.Adamax skips parameters whose grad is None; the fused step does the same (no decay of their moments,
    no movement), and a model moved after the trainer was built is refused instead of silently training a copy."""
    from snn_for_object_detection_amd import BlockGen, Conv, LIF, Norm
    from snn_for_object_detection_amd.trainer import FlatTrainer
    torch.manual_seed(3)
    blk = BlockGen(2, [Conv(8, 3), Norm(), LIF(), Conv(8, 1)]).cuda().train()
    extra = BlockGen(8, [Conv(4, 1)]).cuda()                 # never used in the forward pass: no gradient
    model = torch.nn.ModuleList([blk, extra])
    tr = FlatTrainer(model, lr=1e-2)
    x = synthetic_events(3, 2, 12, 16, p=0.3).cuda()
    w_extra = extra.net[0][0].weight
    for it in range(2):
        before = w_extra.detach().clone()
        tr.zero_grad()
        if it == 1:                                          # give it non-zero moments first, then starve it
            w_extra.grad = torch.ones_like(w_extra)
        out, _ = blk(x)
        out.square().mean().backward()
        tr.step()
        if it == 0:
            assert torch.equal(w_extra.detach(), before)     # untouched, moments still zero
            assert float(tr.exp_inf[tr._offsets[-2]:tr._offsets[-1]].abs().max()) == 0.0
        else:
            assert not torch.equal(w_extra.detach(), before)
    moments = tr.exp_avg[tr._offsets[-2]:tr._offsets[-1]].clone()
    before = w_extra.detach().clone()
    tr.zero_grad()
    out, _ = blk(x)
    out.square().mean().backward()
    tr.step()                                                # no gradient for `extra` this time
    assert torch.equal(w_extra.detach(), before) and torch.equal(tr.exp_avg[tr._offsets[-2]:tr._offsets[-1]], moments)
    assert not torch.equal(blk.net[0][0].weight.detach(), torch.zeros_like(blk.net[0][0].weight))
    model.float().cpu()
    with pytest.raises(RuntimeError, match="flat parameter buffer"):
        tr.step()


def test_cached_weight_images_follow_the_parameter_not_the_trainer(S):
    """The pre-split / transposed weight images are tensor VIEWS kept on the parameter: when the trainer is dropped and
    the allocator recycles memory, an eval forward still reads valid images (they used to be raw addresses into buffers
    only the trainer kept alive - a use-after-free with silently wrong convolution outputs).  A second trainer clears
    what the first one left."""
    import gc
    from snn_for_object_detection_amd import functional as HF
    from snn_for_object_detection_amd.trainer import FlatTrainer
    T, B, H, W = 3, 2, 32, 48
    X, labels = synthetic_events(T, B, H, W, p=0.1).cuda(), synthetic_labels(B).cuda()
    torch.manual_seed(2)
    m = S.TinyYolo(num_classes=2, time_window=0).cuda().train()
    tr = FlatTrainer(m, lr=1e-3)
    tr.zero_grad()
    m.training_step((X, labels)).backward()
    tr.step()
    torch.cuda.synchronize()
    conv_w = next(p for p in m.parameters() if p.dim() == 4 and p.shape[1] > 2)
    assert not hasattr(conv_w, "_snn_w16_ptr") and isinstance(conv_w._snn_w16, torch.Tensor)
    del tr
    gc.collect()
    torch.cuda.empty_cache()
    junk = [torch.full((1 << 22,), float("nan"), device="cuda") for _ in range(16)]   # recycle whatever was freed
    m.eval()
    with torch.no_grad():
        _, cls_a, box_a = m(X)
        HF.USE_PRESPLIT_WEIGHTS = False          # the same forward converting the weights in every block
        try:
            _, cls_b, box_b = m(X)
        finally:
            HF.USE_PRESPLIT_WEIGHTS = True
    assert torch.isfinite(cls_a).all() and torch.equal(cls_a, cls_b) and torch.equal(box_a, box_b)
    del junk
    # a new trainer starts from a clean slate: no image of the old one survives on the parameters
    m.train()
    tr2 = FlatTrainer(m, lr=1e-3)
    assert conv_w._snn_w16.untyped_storage().data_ptr() == tr2.flat_w16.untyped_storage().data_ptr()
    assert not hasattr(conv_w, "_snn_wt16") or tr2.flat_wt16 is not None


def test_skipped_parameters_keep_their_own_step_count_like_torch(S):
    """torch.optim.Adamax keeps ``step`` per parameter and leaves it alone for a parameter without a gradient, so the
    bias correction of a parameter that skipped steps differs from the others'.  Same model, same gradients: the fused
    trainer and torch.optim.Adamax must stay together, and their state_dicts must be interchangeable."""
    from snn_for_object_detection_amd import BlockGen, Conv, LIF, Norm
    from snn_for_object_detection_amd.trainer import FlatTrainer

    def build():
        torch.manual_seed(3)
        blk = BlockGen(2, [Conv(8, 3), Norm(), LIF(), Conv(8, 1)]).cuda().train()
        extra = BlockGen(8, [Conv(4, 1)]).cuda()
        never = BlockGen(8, [Conv(2, 1)]).cuda()             # never receives a gradient
        return torch.nn.ModuleList([blk, extra, never]), blk, extra

    ma, blk_a, extra_a = build()
    mb, blk_b, extra_b = build()
    tr = FlatTrainer(ma, lr=1e-2)
    opt = torch.optim.Adamax(list(mb.parameters()), lr=1e-2)
    x = synthetic_events(3, 2, 12, 16, p=0.3).cuda()
    g_extra = torch.randn(extra_a.net[0][0].weight.shape, device="cuda")
    for it in range(5):
        tr.zero_grad()
        opt.zero_grad(set_to_none=True)
        if it in (1, 4):                                     # `extra` gets a gradient in steps 1 and 4 only
            extra_a.net[0][0].weight.grad = g_extra.clone() * (it + 1)
            extra_b.net[0][0].weight.grad = g_extra.clone() * (it + 1)
        for blk in (blk_a, blk_b):
            out, _ = blk(x)
            out.square().mean().backward()
        tr.step()
        opt.step()
        for (n, pa), pb in zip(ma.named_parameters(), mb.parameters()):
            assert rel_err(pa, pb) < 1e-5, (it, n)
    assert tr.step_count == 5
    k_extra = [i for i, p in enumerate(tr.params) if p is extra_a.net[0][0].weight][0]
    assert tr.param_steps[k_extra] == 2 and tr.param_steps[0] == 5 and tr.param_steps[-1] == 0
    sd, sd_t = tr.state_dict(), opt.state_dict()
    assert set(sd["state"]) == set(sd_t["state"])           # the never-updated parameter has no entry, as in torch
    for k in sd["state"]:
        assert float(sd["state"][k]["step"]) == float(sd_t["state"][k]["step"])
        assert rel_err(sd["state"][k]["exp_avg"], sd_t["state"][k]["exp_avg"]) < 1e-5
        assert rel_err(sd["state"][k]["exp_inf"], sd_t["state"][k]["exp_inf"]) < 1e-5
    # torch's state (differing per-parameter steps) loads into a fresh fused trainer and continues identically
    mc, blk_c, extra_c = build()
    mc.load_state_dict(mb.state_dict())
    tr_c = FlatTrainer(mc, lr=1.0)
    tr_c.load_state_dict(sd_t)
    assert tr_c.param_steps == tr.param_steps
    for model, blk, stepper in ((mb, blk_b, opt), (mc, blk_c, tr_c)):
        stepper.zero_grad()
        out, _ = blk(x)
        out.square().mean().backward()
        stepper.step()
    for (n, pb), pc in zip(mb.named_parameters(), mc.parameters()):
        assert rel_err(pc, pb) < 1e-5, n


def test_event_frame_layer_forms_dy_inside_its_weight_gradient(S):
    """The BatchNorm behind the event-frame convolution hands its producer gx instead of dy (functional.PendingBnApply):
    ``snn_conv2d_wgrad_bn`` applies dy = A*gx + B*y + C while reading - the apply pass and the dy tensor are gone.  Same
    statement, same roundings: the flat gradient is BIT-identical to the two-kernel path, and without grad slots (plain
    autograd) the classic path still runs."""
    from snn_for_object_detection_amd import BlockGen, Conv, LIF, Norm, _hip
    from snn_for_object_detection_amd import functional as HF
    from snn_for_object_detection_amd.trainer import FlatTrainer
    T, B, H, W = 5, 3, 34, 46
    x = synthetic_events(T, B, H, W, p=0.1).cuda()
    calls = []
    real_call = _hip.call

    def spy(name, *a):
        calls.append(name)
        return real_call(name, *a)

    res = {}
    for fused in (True, False):
        HF.USE_DEFERRED_BN_APPLY = fused
        try:
            torch.manual_seed(4)
            blk = BlockGen(2, [Conv(64, 3, 2), Norm(), LIF(), Conv(16, 1), Norm(), LIF()]).cuda().train()
            tr = FlatTrainer(blk, lr=1e-3)
            tr.zero_grad()
            calls.clear()
            _hip.call = spy
            try:
                out, _ = blk(x)
                (out * torch.linspace(0.5, 1.5, out.numel(), device="cuda").view(out.shape)).mean().backward()
            finally:
                _hip.call = real_call
            tr.synchronize()
            res[fused] = (tr.flat_grad.clone(), list(calls))
            assert not HF._PENDING_APPLY                       # every record was consumed
        finally:
            HF.USE_DEFERRED_BN_APPLY = True
    assert res[True][1].count("snn_conv2d_wgrad_bn") == 1 and res[False][1].count("snn_conv2d_wgrad_bn") == 0
    assert res[True][1].count("snn_bn_bwd_apply") == res[False][1].count("snn_bn_bwd_apply") - 1
    assert res[True][0].abs().max() > 0 and torch.equal(res[True][0], res[False][0])


def test_composed_1x1_weights_come_from_one_batched_launch_per_step(S):
    """The C2f entry groups (models/tiny_yolo.py:76-85: Conv(c,1) composed with the two branch-opening Conv(c/2,1), run as
    one convolution with the row-stacked weight) register themselves during the first forward pass; from the next optimiser
    step on FlatTrainer composes [w2a; w2b] w1 (and its transpose) for ALL of them in one launch (snn_small_gemm_batched)
    and the forward pass issues no composition GEMM any more - with the same bits as the per-call products, and with a
    stale cache (weights changed behind the trainer's back) never used."""
    from snn_for_object_detection_amd import _hip
    from snn_for_object_detection_amd.trainer import FlatTrainer
    T, B, H, W = 3, 2, 32, 48
    X, labels = synthetic_events(T, B, H, W, p=0.1).cuda(), synthetic_labels(B).cuda()
    torch.manual_seed(4)
    model = S.TinyYolo(num_classes=2, time_window=0).cuda().train()
    tr = FlatTrainer(model, lr=1e-3)

    class Count:
        def __init__(self):
            self.names = []

        def before(self, name, args):
            self.names.append(name)

        def after(self, token):
            pass

    def run():
        tr.zero_grad()
        loss = model.training_step((X, labels))
        loss.backward()
        S.functional.wgrad_stream_sync()
        return loss.detach().clone(), tr.flat_grad.clone()

    run()                                   # registers the pairs
    tr.step()                               # update + refresh: composes them in one launch
    pairs = [p for p in model.parameters() if getattr(p, "_snn_sibling_weight", None) is not None]
    assert len(pairs) == 5                  # one group per C2f block
    _hip.PROFILER = cnt = Count()
    try:
        loss_cached, grad_cached = run()
    finally:
        _hip.PROFILER = None
    n_gemm_cached = cnt.names.count("snn_small_gemm")
    for p in pairs:
        del p._snn_sibling_weight           # the per-call composition again
    _hip.PROFILER = cnt2 = Count()
    try:
        loss_call, grad_call = run()
    finally:
        _hip.PROFILER = None
    assert cnt2.names.count("snn_small_gemm") == n_gemm_cached + 2 * len(pairs)   # one forward GEMM per branch came back
    assert torch.equal(loss_cached, loss_call) and torch.equal(grad_cached, grad_call)
    # a weight changed behind the trainer (version counter moved): the cached product is not used
    tr.step()
    with torch.no_grad():
        pairs[0].mul_(1.5)
    _hip.PROFILER = cnt3 = Count()
    try:
        run()
    finally:
        _hip.PROFILER = None
    assert cnt3.names.count("snn_small_gemm") == n_gemm_cached + 2


@pytest.mark.gpu
def test_side_stream_is_probed_for_a_hardware_queue_of_its_own(hip_lib):
    """HIP maps streams onto a few hardware queues; two streams on one queue serialise (the weight-gradient stream lost
    its overlap when an RCCL communicator existed before the model: +1.3 ms per GEN1 step).  ``concurrent_stream`` must
    hand out a stream whose work overtakes a long job on the main stream, and the probe must call a stream that IS the
    main stream serial."""
    from snn_for_object_detection_amd import functional as HF
    main = torch.cuda.current_stream()
    assert not HF.runs_concurrently(main, main)
    st = HF.concurrent_stream(main)
    assert st.cuda_stream != main.cuda_stream
    assert HF.runs_concurrently(st, main)
    assert HF._STREAM_PROBE_LOG and HF._STREAM_PROBE_LOG[-1][2] is True
