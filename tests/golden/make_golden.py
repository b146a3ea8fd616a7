"""Generate golden vectors from the REFERENCE's own torch-only files (run in the build container only).

    python tests/golden/make_golden.py            # needs /root/reference

Loads ``utils/box.py``, ``utils/anchors.py``, ``utils/roi.py`` and ``models/modules/conv_lstm.py`` of the
reference BY FILE PATH (the package ``__init__`` files pull in Lightning / OpenCV / norse, which are not
installed), runs them on seeded inputs and stores inputs + outputs in ``tests/golden/*.npz``.  The network
DESCRIPTION ``models/tiny_yolo.py`` is executed against recording stand-ins of the layer generators (it only
builds nested lists) and its structure stored as ``tiny_yolo_desc.json``.  The step functions of ``models/modules/sli.py``
and ``models/modules/synapse.py`` run unmodified (``sli.npz`` / ``synapse.npz``).  Only data is stored - no reference
source.  The fixtures pin ``oracle/detect.py`` and the product's ``anchors/box/roi`` modules
(``tests/test_oracle_detect.py``).  ``/root/reference`` never travels to the GPU box; the fixtures do.
"""

import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("SNN_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference_utils():
    pkg = types.ModuleType("utils")
    pkg.__path__ = []  # stub package: expose only the torch-only submodules
    sys.modules["utils"] = pkg
    box = _load("utils.box", os.path.join(REF, "utils", "box.py"))
    pkg.box = box
    anchors = _load("utils.anchors", os.path.join(REF, "utils", "anchors.py"))
    roi = _load("utils.roi", os.path.join(REF, "utils", "roi.py"))
    return box, anchors, roi


def head_sizes(num_maps=3, per_pixel=3):
    # models/generator.py:389-401 (the file itself needs python >= 3.12, so the table is rebuilt here)
    lo, hi = 0.08, 0.75
    sizes = torch.arange(lo, hi, (hi - lo) / (num_maps * per_pixel), dtype=torch.float32).reshape((-1, per_pixel))
    return sizes, torch.tensor((0.5, 1.0, 2), dtype=torch.float32)


def random_labels(gen, batch, n_boxes, n_classes, pad_rows=0):
    out = torch.full((batch, n_boxes + pad_rows, 5), -1.0)
    for b in range(batch):
        for k in range(n_boxes):
            while True:
                xy = torch.rand(2, 2, generator=gen)
                lo, hi = xy.min(0).values, xy.max(0).values
                if (hi - lo).prod() > 0.01:
                    break
            out[b, k, 0] = float(torch.randint(0, n_classes, (1,), generator=gen))
            out[b, k, 1:3], out[b, k, 3:5] = lo, hi
    return out


def convlstm_golden(gen):
    """``models/modules/conv_lstm.py:51-78``: seeded input sequence -> (h, c) after each of 3 steps, plus the
    gradients of a seeded scalar loss w.r.t. the inputs and the gate weight (pins the oracle's restatement and the
    HIP ``LSTM()`` layer)."""
    mod = _load("ref_conv_lstm", os.path.join(REF, "models", "modules", "conv_lstm.py"))
    T, B, Cin, Ch, H, W = 3, 2, 5, 4, 6, 7
    cell = mod.ConvLSTM(Cin, Ch)
    w = 0.5 * torch.randn(cell.conv.weight.shape, generator=gen)
    with torch.no_grad():
        cell.conv.weight.copy_(w)
    x = torch.randn(T, B, Cin, H, W, generator=gen).requires_grad_()
    gh = torch.randn(T, B, Ch, H, W, generator=gen)
    gc = torch.randn(B, Ch, H, W, generator=gen)
    state, hs, cs = None, [], []
    for t in range(T):
        h, state = cell(x[t], state)
        hs.append(h)
        cs.append(state[1])
    loss = (torch.stack(hs) * gh).sum() + (state[1] * gc).sum()
    loss.backward()
    np.savez_compressed(os.path.join(OUT, "convlstm.npz"), weight=w.numpy(), x=x.detach().numpy(), gh=gh.numpy(),
                        gc=gc.numpy(), h=torch.stack(hs).detach().numpy(), c=torch.stack(cs).detach().numpy(),
                        gx=x.grad.numpy(), gw=cell.conv.weight.grad.numpy())


def sli_synapse_golden(gen):
    """``models/modules/sli.py:110-126`` (``sli_feed_forward_step``) and ``models/modules/synapse.py:73-103``
    (``synapse_feed_forward_step``): the reference's own step FUNCTIONS, executed over seeded sequences with autograd.
    Both files import norse only for the base classes of their Cell wrappers (``SNNCell`` / ``SNN``); empty stand-in
    classes satisfy that import, the step functions themselves are plain torch and run unmodified.  The SLI step is
    the only in-tree witness of norse's LI ordering (current jump first, then the voltage update from the jumped
    current), so it is pinned as DATA here: ``sli.npz`` / ``synapse.npz`` hold inputs, per-step outputs, final state and
    the gradients of a seeded scalar loss."""
    snn = types.ModuleType("norse.torch.module.snn")
    snn.SNNCell = type("SNNCell", (), {"__init__": lambda self, *a, **k: None})
    snn.SNN = type("SNN", (), {"__init__": lambda self, *a, **k: None})
    names = ("norse", "norse.torch", "norse.torch.module", "norse.torch.module.snn")
    saved = {k: sys.modules.get(k) for k in names}
    for k in names[:-1]:
        m = types.ModuleType(k)
        m.__path__ = []
        sys.modules[k] = m
    sys.modules[names[-1]] = snn
    try:
        sli = _load("ref_sli", os.path.join(REF, "models", "modules", "sli.py"))
        syn = _load("ref_synapse", os.path.join(REF, "models", "modules", "synapse.py"))
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    T, B, C, H, W = 6, 2, 3, 4, 5
    # ---- SLI: inputs large enough for |v| to approach the saturation potential (sigmoid(v_st - |v|) far from 1/2)
    x = (8.0 * torch.randn(T, B, C, H, W, generator=gen)).requires_grad_()
    gv = torch.randn(T, B, C, H, W, generator=gen)
    gi = torch.randn(B, C, H, W, generator=gen)
    p = sli.SLIParameters()
    v0 = p.v_leak.detach().clone().requires_grad_()          # SLICell.initial_state (sli.py:97-107)
    state = sli.SLIState(v=v0, i=torch.zeros(B, C, H, W))
    vs = []
    for t in range(T):
        v, state = sli.sli_feed_forward_step(x[t], state, p, 0.001)
        vs.append(v)
    vs = torch.stack(vs)
    ((vs * gv).sum() + (state.i * gi).sum()).backward()
    np.savez_compressed(os.path.join(OUT, "sli.npz"), x=x.detach().numpy(), gv=gv.numpy(), gi=gi.numpy(),
                        v=vs.detach().numpy(), i_final=state.i.detach().numpy(), gx=x.grad.numpy(),
                        gv0=v0.grad.numpy(), dt=0.001)
    # ---- Synapse: signed inputs (secretion for x > 0, dissociation otherwise), without and with inhibition
    out = {}
    xs = torch.randn(T, B, C, H, W, generator=gen)
    gg = torch.randn(T, B, C, H, W, generator=gen)
    out.update(x=xs.numpy(), gg=gg.numpy(), dt=0.001)
    for tag, sigma in (("s0", 0.0), ("s07", 0.7)):
        pp = syn.SynapseParameters(sigma_inhibition=torch.as_tensor(sigma))
        xin = xs.clone().requires_grad_()
        st = syn.SynapseState(p=torch.zeros(B, C, H, W))
        gs = []
        for t in range(T):
            g, st = syn.synapse_feed_forward_step(xin[t], st, pp, 0.001)
            gs.append(g)
        gs = torch.stack(gs)
        (gs * gg).sum().backward()
        out.update({f"g_{tag}": gs.detach().numpy(), f"p_final_{tag}": st.p.detach().numpy(),
                    f"gx_{tag}": xin.grad.numpy(), f"sigma_{tag}": sigma})
    np.savez_compressed(os.path.join(OUT, "synapse.npz"), **out)


def events_golden():
    """``utils/datasets.py``: ``STPropheseeDataset.parse_data`` (:378-435), ``MTPropheseeDataset.parse_data`` (:311-344) and
    ``PropheseeDataModule._stack_data`` (:127-135) - the step FEEDING the hot path - executed
    unmodified.  The module imports ``lightning`` (base class of the data module) and
    ``prophesee_toolbox.src.io.psee_loader.PSEELoader`` (the recording reader, an un-fetched submodule) at import time
    only; empty stand-in modules satisfy both imports, and the method bodies see a FAKE loader - an object with the three
    members they use (``done``, ``current_time``, ``load_delta_t``) that serves a seeded structured ``(t, x, y, p)``
    array - so what is pinned is the reference's own binning / clipping / flagging / label selection / padding, as DATA
    (``events.npz``: the event stream, the boxes, and the non-zero cells + labels that came out)."""
    lightning = types.ModuleType("lightning")
    lightning.LightningDataModule = type("LightningDataModule", (), {"__init__": lambda self, *a, **k: None})
    loader_mod = types.ModuleType("prophesee_toolbox.src.io.psee_loader")
    loader_mod.PSEELoader = type("PSEELoader", (), {})
    names = ("lightning", "prophesee_toolbox", "prophesee_toolbox.src", "prophesee_toolbox.src.io",
             "prophesee_toolbox.src.io.psee_loader")
    saved = {k: sys.modules.get(k) for k in names}
    sys.modules["lightning"] = lightning
    for k in names[1:-1]:
        m = types.ModuleType(k)
        m.__path__ = []
        sys.modules[k] = m
    sys.modules[names[-1]] = loader_mod
    try:
        ds = _load("ref_datasets", os.path.join(REF, "utils", "datasets.py"))
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v

    class FakeLoader:
        """Serves ``events`` (sorted by t) the way the methods consume a recording: ``load_delta_t(d)`` returns the events
        of ``[current_time, current_time + d)`` and advances the clock; ``done`` once the stream is exhausted."""

        def __init__(self, events, start_us, end_us):
            self.events, self.current_time, self.end = events, int(start_us), int(end_us)

        @property
        def done(self):
            return self.current_time >= self.end

        def reset(self):
            self.current_time = 0

        def load_delta_t(self, delta):
            lo, hi = self.current_time, self.current_time + int(delta)
            self.current_time = hi
            t = self.events["t"]
            return self.events[(t >= lo) & (t < hi)].copy()

    def stream(rng, n, t_lo, t_hi, width, height, x_over):
        ev = np.zeros(n, dtype=[("t", "<i8"), ("x", "<i4"), ("y", "<i4"), ("p", "<i4")])
        ev["t"] = np.sort(rng.integers(t_lo, t_hi, n))
        ev["x"] = rng.integers(0, width + x_over, n)        # x past the frame: the 1Mpx quirk parse_data clips (:425-426)
        ev["y"] = rng.integers(0, height, n)
        ev["p"] = rng.integers(0, 2, n)
        return ev

    out = {}
    rng = np.random.default_rng(77)
    # ---- single-target samples (training): GEN1 frame and a 1Mpx frame with out-of-frame x
    st_cases = (("st_gen1", "gen1", 4, 2, 16, 70_000, 0), ("st_1mpx", "1mpx", 3, 1, 16, 40_000, 9),
                ("st_sparse", "gen1", 4, 2, 16, 900, 0))
    for tag, name, T, shift, step_ms, n_events, x_over in st_cases:
        d = ds.STPropheseeDataset(num_steps=T, time_shift=shift, gt_files=["g"], data_files=["d"], time_step=step_ms,
                                  num_load_file=1, name=name)
        step_us = d.time_step_us
        start_us = 3 * step_us + 1234                          # the recording clock is not on a step boundary
        start_step = start_us // step_us
        # boxes (step, class, x1, y1, x2, y2): rows BEFORE start_step + T are skipped, the first later step's rows are
        # the sample's labels, a box below the 1 % area threshold is dropped, later steps are ignored
        s0 = start_step + T + 2
        gt = torch.tensor([[start_step + 1, 0, 0.1, 0.1, 0.5, 0.5], [s0, 1, 0.20, 0.25, 0.60, 0.70],
                           [s0, 0, 0.40, 0.40, 0.45, 0.45], [s0, 0, 0.05, 0.30, 0.55, 0.95],
                           [s0 + 5, 1, 0.3, 0.3, 0.9, 0.9]], dtype=torch.float32)
        ev = stream(rng, n_events, start_us, (s0 + shift + 2) * step_us, d._width, d._height, x_over)
        loader = FakeLoader(ev, start_us, 10 ** 9)
        sample, more = d.parse_data(gt.clone(), loader)
        out.update({f"{tag}_events_t": ev["t"], f"{tag}_events_x": ev["x"], f"{tag}_events_y": ev["y"],
                    f"{tag}_events_p": ev["p"], f"{tag}_gt": gt.numpy(),
                    f"{tag}_params": np.array([T, shift, step_us, start_us, d._height, d._width, d.events_threshold]),
                    f"{tag}_box_size_threshold": d.box_size_threshold,
                    f"{tag}_more": bool(more), f"{tag}_clock_after": loader.current_time,
                    f"{tag}_rejected": sample is None})
        if sample is not None:
            feats, labels = sample
            assert feats.shape == (T, 2, d._height, d._width) and set(feats.unique().tolist()) <= {0.0, 1.0}
            out[f"{tag}_nonzero"] = np.flatnonzero(feats.numpy().reshape(-1)).astype(np.int64)
            out[f"{tag}_labels"] = labels.numpy()
    # ---- multi-target sample (evaluation)
    T, step_ms = 5, 16
    d = ds.MTPropheseeDataset(num_steps=T, gt_files=["g"], data_files=["d"], time_step=step_ms, num_load_file=1, name="gen1")
    step_us = d.time_step_us
    # (the evaluation reader starts at 0 and advances by whole windows: its clock is always on a step boundary - off one,
    #  the events of the last partial step index frame T and the reference raises IndexError)
    start_us = 7 * step_us
    start_step = start_us // step_us
    gt = torch.tensor([[start_step - 1, 0, 0.1, 0.1, 0.5, 0.5], [start_step, 1, 0.2, 0.2, 0.6, 0.7],
                       [start_step + 2, 0, 0.3, 0.1, 0.7, 0.4], [start_step + T - 1, 1, 0.5, 0.5, 0.9, 0.9],
                       [start_step + T, 0, 0.1, 0.6, 0.3, 0.9]], dtype=torch.float32)
    ev = stream(rng, 30_000, start_us, start_us + (T + 2) * step_us, d._width, d._height, 0)
    loader = FakeLoader(ev, start_us, 10 ** 9)
    feats, labels = d.parse_data(gt.clone(), loader)
    out.update({"mt_events_t": ev["t"], "mt_events_x": ev["x"], "mt_events_y": ev["y"], "mt_events_p": ev["p"],
                "mt_gt": gt.numpy(), "mt_params": np.array([T, 0, step_us, start_us, d._height, d._width, 0]),
                "mt_clock_after": loader.current_time,
                "mt_nonzero": np.flatnonzero(feats.numpy().reshape(-1)).astype(np.int64), "mt_labels": labels.numpy()})
    # an empty window: the features stay zero and the label set is EMPTY (:328-329)
    empty = FakeLoader(ev[:0], start_us, 10 ** 9)
    feats0, labels0 = d.parse_data(gt.clone(), empty)
    out.update({"mt_empty_nonzero_count": int(feats0.count_nonzero()), "mt_empty_labels_shape": np.array(labels0.shape)})
    # ---- collate: ragged label lists padded with -1, features stacked on dim 1 (:127-135)
    g = torch.Generator().manual_seed(5)
    batch = [((torch.rand(3, 2, 4, 6, generator=g) < 0.2).float(), torch.rand(n, 5, generator=g)) for n in (2, 0, 3)]
    features, targets = ds.PropheseeDataModule._stack_data(None, batch)
    for b, (f, l) in enumerate(batch):
        out[f"stack_features_{b}"] = f.numpy()
        out[f"stack_labels_{b}"] = l.numpy()
    out.update({"stack_out_features": features.numpy(), "stack_out_targets": targets.numpy()})
    np.savez_compressed(os.path.join(OUT, "events.npz"), **out)


def executor_golden():
    """The reference's own EXECUTOR, heads and loss on a network description without spiking neurons: ``BlockGen`` /
    ``BackboneGen`` / ``NeckGen`` / ``Head`` / ``HeadGen`` (``models/generator.py:82-538``), the layer generators and modules of
    ``models/modules/{layer_gen,common,conv_lstm}.py``, ``AnchorGenerator``, and ``SODa.__init__ / forward / _forward_impl /
    _loss`` (``models/soda.py:66-96,138-144,235-244,259-281``) - a real subclass of the reference's ``SODa`` whose three
    description hooks use ``Conv, Norm, ReLU, SiLU, Tanh, Pool, Up, LSTM, Pass, Return, Residual, Dense`` only (everything but
    the norse neurons), run forward over T steps (the ConvLSTM state threads through the reference's time loop), loss,
    backward.  What stands in, and for what: ``lightning`` (``LightningModule`` = ``nn.Module`` + ``save_hyperparameters``),
    ``torchmetrics.detection`` and ``utils.plotter`` (constructed / imported, never called here), norse (imported by the layer
    files for classes this description never instantiates; ``norse.torch.utils.state._is_module_stateful`` is restated as
    norse 1.1.0 defines it - "``forward`` has a parameter named ``state``" - and is the one piece of third-party logic this
    fixture rests on).  ``models/generator.py`` uses two PEP 695 ``type`` aliases (:31-32) that python 3.10 cannot parse: the
    file is compiled IN MEMORY with those two statements rewritten as plain assignments (they only feed annotations); every
    other statement of every file runs as it is on disk.  Stored: the state_dict (keys = the reference's module-tree layout),
    inputs, labels, the predictions of every prefix length, the loss, every parameter gradient, BatchNorm buffers afterwards."""
    import inspect
    import re

    def standin(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        return m

    class _LightningModule(torch.nn.Module):
        def save_hyperparameters(self, *args, ignore=(), **kw):
            frame = inspect.currentframe().f_back
            names = [n for n in inspect.signature(type(self).__init__).parameters if n != "self"]
            # (the subclass forwards **kwargs: take the base class's named arguments from the calling frame)
            local = frame.f_locals
            hp = {k: local[k] for k in inspect.signature(frame.f_globals["SODa"].__init__).parameters
                  if k not in ("self", *ignore) and k in local}
            self.hparams = types.SimpleNamespace(**hp)

        def log(self, *a, **k):
            pass

    dummy = type("Dummy", (), {"__init__": lambda self, *a, **k: None})
    norse_torch = standin("norse.torch", LIFCell=dummy, LICell=dummy)
    mods = {
        "lightning": standin("lightning", LightningModule=_LightningModule),
        "torchmetrics": standin("torchmetrics"),
        "torchmetrics.detection": standin("torchmetrics.detection", MeanAveragePrecision=dummy),
        "norse": standin("norse", torch=norse_torch),
        "norse.torch": norse_torch,
        "norse.torch.module": standin("norse.torch.module"),
        "norse.torch.module.snn": standin("norse.torch.module.snn", SNNCell=dummy, SNN=dummy,
                                          _merge_states=lambda states: states),
        "norse.torch.utils": standin("norse.torch.utils"),
        "norse.torch.utils.state": standin(
            "norse.torch.utils.state",
            _is_module_stateful=lambda m: "state" in inspect.signature(m.forward).parameters),
        "utils.plotter": standin("utils.plotter", Plotter=dummy),
    }
    mods["torchmetrics"].detection = mods["torchmetrics.detection"]
    for pkg, sub in (("utils", "utils"), ("models", "models")):   # packages WITHOUT their __init__ (datasets / Lightning CLI)
        m = standin(pkg)
        m.__path__ = [os.path.join(REF, sub)]
        mods[pkg] = m
    for k in ("norse", "norse.torch", "norse.torch.module", "norse.torch.utils", "torchmetrics"):
        mods[k].__path__ = []
    touched = set(mods) | {"utils.anchors", "utils.box", "utils.roi", "models.modules", "models.modules.layer_gen",
                           "models.modules.common", "models.modules.conv_lstm", "models.modules.sli",
                           "models.modules.synapse", "models.generator", "models.soda"}
    saved = {k: sys.modules.get(k) for k in touched}
    sys.modules.update(mods)
    try:
        import importlib
        gen = types.ModuleType("models.generator")
        gen.__file__ = os.path.join(REF, "models", "generator.py")
        src = open(gen.__file__).read()
        src, n = re.subn(r"(?m)^type (ListGen|ListState) = ", r"\1 = ", src)
        assert n == 2                                   # exactly the two PEP 695 statements, nothing else
        src = src.replace("ListGen = List[LayerGen | ListGen]", "ListGen = List").replace(
            "ListState = List[torch.Tensor | None | ListState]", "ListState = List")
        sys.modules["models.generator"] = gen
        exec(compile(src, gen.__file__, "exec"), gen.__dict__)
        soda = importlib.import_module("models.soda")
        L = importlib.import_module("models.modules")

        class Net(soda.SODa):
            def backbone_cfgs(self):
                return [L.Conv(8, 3, 2), L.Norm(), L.ReLU(),
                        L.Dense([[L.Conv(8, 1), L.Residual([[L.Conv(kernel_size=3), L.Norm(bias=True), L.Tanh()], [L.Pass()]])],
                                 [L.Pool("S"), L.Up(), L.Conv(4, 1)]]),
                        L.Conv(16, 1)]

            def neck_cfgs(self):
                return [L.Conv(16, 3, 2), L.Norm(), L.Tanh(), L.LSTM(), L.Return(),
                        L.Conv(24, 3, 2), L.Norm(), L.SiLU(), L.Pool("M", 1, 1), L.Return()]

            def head_cfgs(self, box_out, cls_out):
                return [[L.Conv(kernel_size=1), L.Norm(), L.Tanh()], [L.Conv(box_out, 1)], [L.Conv(cls_out, 1)]]

        torch.manual_seed(11)
        net = Net(num_classes=3, time_window=0)
        net.train()
        g = torch.Generator().manual_seed(12)
        T, B, H, W = 3, 2, 24, 32
        X = (torch.rand(T, B, 2, H, W, generator=g) < 0.2).float()
        labels = random_labels(g, batch=B, n_boxes=2, n_classes=3, pad_rows=1)
        out = {"X": X.numpy(), "labels": labels.numpy(), "num_classes": 3,
               "loss_ratio": net.hparams.loss_ratio, "iou_threshold": net.hparams.iou_threshold}
        sd0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
        out["state_keys"] = np.array(list(sd0.keys()))
        for k, v in sd0.items():
            out["init/" + k] = v.numpy()
        anchors, cls_preds, bbox_preds = net(X)                       # soda.py:138-144: time loop, last step's predictions
        loss = net._loss((anchors, cls_preds, bbox_preds), labels)   # soda.py:259-281
        loss.backward()
        out.update(anchors=anchors.detach().numpy(), cls_preds=cls_preds.detach().numpy(),
                   bbox_preds=bbox_preds.detach().numpy(), loss=float(loss.detach()))
        out["no_grad"] = np.array([k for k, p in net.named_parameters() if p.grad is None])
        for k, p in net.named_parameters():
            if p.grad is not None:
                out["grad/" + k] = p.grad.numpy()
        for k, v in net.state_dict().items():
            if "running_" in k or "num_batches" in k:
                out["after/" + k] = v.detach().numpy()
        # training_step with the reference's default time_window = 16 (soda.py:146-158, 246-257): a random prefix of the
        # sequence is dropped, the draw taken from torch's global generator
        torch.manual_seed(21)
        net2 = Net(num_classes=3)                        # time_window = 16, the default
        net2.load_state_dict(sd0)
        net2.train()
        Xl = (torch.rand(18, B, 2, H, W, generator=g) < 0.2).float()
        torch.manual_seed(5)
        loss2 = net2.training_step((Xl, labels), 0)
        loss2.backward()
        out.update(ts_X=Xl.numpy(), ts_seed=5, ts_time_window=net2.hparams.time_window, ts_loss=float(loss2.detach()),
                   ts_grad_first=net2.base_net.net.net[0][0].weight.grad.numpy(),
                   ts_nbt=int(net2.base_net.net.net[0][1].num_batches_tracked))   # = frames that were actually run
        # streaming inference (soda.py:202-233): eval mode, one frame at a time, the detector state threaded by the caller
        net.eval()
        st = None
        with torch.no_grad():
            for t in range(T):
                det, st = net.predict(X[t, 0], st)
                out[f"predict_{t}"] = det.numpy()
        pflat = []

        def pwalk(s_, path):
            if isinstance(s_, (list, tuple)):
                for i_, e_ in enumerate(s_):
                    pwalk(e_, path + [i_])
            elif isinstance(s_, torch.Tensor):
                pflat.append((".".join(map(str, path)), s_))
        pwalk(st, [])
        out["predict_state_paths"] = np.array([p_ for p_, _ in pflat])
        for p_, t_ in pflat:
            out["predict_state/" + p_] = t_.numpy()
        # one block on its own with an explicit state tree: the executor's state threading (generator.py:169-198)
        torch.manual_seed(13)
        blk = gen.BlockGen(4, [L.Conv(6, 3), L.Norm(), L.Tanh(), L.Dense([[L.LSTM(5)], [L.Pass()], [L.Conv(3, 1), L.SiLU()]]),
                               L.Residual([[L.Conv(kernel_size=1)], [L.LSTM()]])])
        blk.eval()
        xs = torch.randn(4, 2, 4, 6, 7, generator=g)
        state, ys = None, []
        with torch.no_grad():
            for t in range(4):
                y, state = blk(xs[t], state)
                ys.append(y)
        out.update(blk_x=xs.numpy(), blk_y=torch.stack(ys).numpy(), blk_out_channels=blk.out_channels,
                   blk_state_mask=np.array(json_dumps(blk.branch_state)))
        for k, v in blk.state_dict().items():
            out["blk/" + k] = v.detach().numpy()
        flat = []

        def walk(s, path):
            if isinstance(s, (list, tuple)):
                for i, e in enumerate(s):
                    walk(e, path + [i])
            elif isinstance(s, torch.Tensor):
                flat.append((".".join(map(str, path)), s))
        walk(state, [])
        out["blk_state_paths"] = np.array([p for p, _ in flat])
        for p_, t_ in flat:
            out["blk_state/" + p_] = t_.numpy()
        np.savez_compressed(os.path.join(OUT, "executor.npz"), **out)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def json_dumps(obj):
    import json
    return json.dumps(obj)


def tiny_yolo_description():
    """Execute ``models/tiny_yolo.py`` with recording stand-ins for ``models.soda.SODa`` / ``models.generator`` /
    ``models.modules`` (the real ones need Lightning, norse and python >= 3.12): the file only BUILDS nested lists of
    layer generators, so the stand-ins record class name + constructor arguments.  Output: the nested structure of
    ``backbone_cfgs() / neck_cfgs() / head_cfgs(36, 27)`` as JSON - data that pins the product's transcription of
    the description independently of the oracle."""
    import json

    def recorder(kind, defaults):
        class _Gen:
            def __init__(self, *args, **kwargs):
                vals = dict(defaults)
                for k, v in zip(list(defaults), args):
                    vals[k] = v
                vals.update(kwargs)
                self.kind, self.vals = kind, vals
        _Gen.__name__ = kind
        return _Gen

    spec = {  # constructor signatures of models/modules/layer_gen.py:96-347
        "Pass": {}, "Conv": {"out_channels": None, "kernel_size": 3, "stride": 1}, "Norm": {"bias": False},
        "LIF": {"state_storage": False}, "LI": {"state_storage": False}, "ReLU": {}, "SiLU": {}, "Tanh": {},
        "LSTM": {"hidden_size": None}, "Pool": {"type": None, "kernel_size": 2, "stride": None},
        "Up": {"scale": 2, "mode": "nearest"}, "Return": {}, "Synapse": {}, "SLI": {"state_storage": False},
    }
    modules = types.ModuleType("models.modules")
    for kind, defaults in spec.items():
        setattr(modules, kind, recorder(kind, defaults))
    modules.Residual = type("Residual", (list,), {})
    modules.Dense = type("Dense", (list,), {})
    modules.__all__ = list(spec) + ["Residual", "Dense"]
    soda = types.ModuleType("models.soda")

    class SODa:  # only what the description touches: self.hparams.state_storage
        def __init__(self, state_storage=False):
            self.hparams = types.SimpleNamespace(state_storage=state_storage)
    soda.SODa = SODa
    generator = types.ModuleType("models.generator")
    generator.ListGen = list
    pkg = types.ModuleType("models")
    pkg.__path__ = []
    saved = {k: sys.modules.get(k) for k in ("models", "models.soda", "models.generator", "models.modules")}
    sys.modules.update({"models": pkg, "models.soda": soda, "models.generator": generator, "models.modules": modules})
    try:
        ty = _load("ref_tiny_yolo", os.path.join(REF, "models", "tiny_yolo.py"))
        net = ty.TinyYolo()

        def render(item):
            if isinstance(item, (list, tuple)):
                tag = type(item).__name__ if type(item).__name__ in ("Residual", "Dense") else "list"
                return {"merge": tag, "items": [render(i) for i in item]}
            return {"layer": item.kind, **item.vals}

        desc = {"backbone": render(net.backbone_cfgs()), "neck": render(net.neck_cfgs()),
                "head": render(net.head_cfgs(36, 27))}
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    with open(os.path.join(OUT, "tiny_yolo_desc.json"), "w") as f:
        json.dump(desc, f, indent=0, sort_keys=True)


def main():
    box, anchors_mod, roi_mod = load_reference_utils()
    torch.manual_seed(0)
    gen = torch.Generator().manual_seed(1234)
    sizes, ratios = head_sizes()

    # ---- anchors for the three GEN1 taps and a tiny pyramid
    fix = {}
    for tag, shapes in (("gen1", [(30, 38), (15, 19), (8, 10)]), ("tiny", [(4, 6), (2, 3), (1, 2)])):
        per_map = []
        for idx, (h, w) in enumerate(shapes):
            g = anchors_mod.AnchorGenerator(sizes=sizes[idx].clone(), ratios=ratios.clone())
            per_map.append(g(torch.zeros(1, 1, h, w)).clone())
        fix[f"anchors_{tag}"] = torch.cat(per_map).numpy()
        fix[f"shapes_{tag}"] = np.array(shapes)
    np.savez_compressed(os.path.join(OUT, "detect_anchors.npz"), sizes=sizes.numpy(), ratios=ratios.numpy(), **fix)

    # ---- box primitives
    a = torch.rand(64, 2, 2, generator=gen)
    boxes_a = torch.cat([a.min(1).values, a.max(1).values + 0.01], dim=1)
    b = torch.rand(7, 2, 2, generator=gen)
    boxes_b = torch.cat([b.min(1).values, b.max(1).values + 0.01], dim=1)
    offs = torch.randn(64, 4, generator=gen)
    np.savez_compressed(
        os.path.join(OUT, "detect_box.npz"),
        boxes_a=boxes_a.numpy(), boxes_b=boxes_b.numpy(), offs=offs.numpy(),
        iou=box.box_iou(boxes_a, boxes_b).numpy(),
        c2c=box.box_corner_to_center(boxes_a).numpy(),
        c2c_inv=box.box_center_to_corner(box.box_corner_to_center(boxes_a)).numpy(),
        offset_boxes=box.offset_boxes(boxes_a, boxes_a.flip(0)).numpy(),
        offset_inverse=box.offset_inverse(boxes_a, offs).numpy(),
    )

    # ---- RoI targets on the GEN1 anchor set: plain labels and labels with padding rows (-1)
    anc = torch.from_numpy(fix["anchors_gen1"])
    cases = {}
    for tag, pad in (("plain", 0), ("padded", 2)):
        labels = random_labels(gen, batch=3, n_boxes=2, n_classes=2, pad_rows=pad)
        off, mask, cls = roi_mod.RoI(0.4)(anc, labels.clone())
        cases.update({f"labels_{tag}": labels.numpy(), f"offset_{tag}": off.numpy(),
                      f"mask_{tag}": mask.numpy(), f"cls_{tag}": cls.numpy()})
    np.savez_compressed(os.path.join(OUT, "detect_roi.npz"), iou_threshold=0.4, **cases)

    # ---- multibox_detection (NMS decode) on the tiny pyramid
    anc_t = torch.from_numpy(fix["anchors_tiny"])
    A = anc_t.shape[0]
    probs = torch.softmax(3 * torch.randn(2, A, 3, generator=gen), dim=2)
    offp = 0.5 * torch.randn(2, A, 4, generator=gen)
    det = box.multibox_detection(probs.clone(), offp.clone(), anc_t)
    np.savez_compressed(os.path.join(OUT, "detect_nms.npz"), probs=probs.numpy(), offsets=offp.numpy(),
                        detections=det.numpy())

    # ---- a mid-size NMS case (3 465 anchors, 3 classes + background, many overlapping confident boxes): pins the
    #      device NMS kernel (chunked greedy suppression, several 256-candidate chunks per class)
    per_map = []
    for idx, (h, w) in enumerate([(15, 19), (8, 10), (4, 5)]):
        g = anchors_mod.AnchorGenerator(sizes=sizes[idx].clone(), ratios=ratios.clone())
        per_map.append(g(torch.zeros(1, 1, h, w)).clone())
    anc_m = torch.cat(per_map)
    A = anc_m.shape[0]
    probs = torch.softmax(4 * torch.randn(2, A, 4, generator=gen), dim=2)
    offp = 0.3 * torch.randn(2, A, 4, generator=gen)
    det = box.multibox_detection(probs.clone(), offp.clone(), anc_m)
    np.savez_compressed(os.path.join(OUT, "detect_nms_mid.npz"), anchors=anc_m.numpy(), probs=probs.numpy(),
                        offsets=offp.numpy(), detections=det.numpy())
    convlstm_golden(gen)
    sli_synapse_golden(gen)
    events_golden()
    executor_golden()
    tiny_yolo_description()
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
