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
    tiny_yolo_description()
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
