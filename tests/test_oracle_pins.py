"""Pins that come from the REFERENCE itself (fixtures written by tests/golden/make_golden.py in the build container):

* ``convlstm.npz``       - outputs / gradients of the reference's own ``models/modules/conv_lstm.py:51-78`` on seeded
  inputs: pins ``oracle.net.ConvLSTM`` (the HIP ``LSTM()`` layer is checked against the same file in
  ``tests/test_gpu_next_ops.py``);
* ``tiny_yolo_desc.json`` - the nested layer lists the reference's ``models/tiny_yolo.py:16-89`` builds: pins the
  product's transcription of the net description independently of the oracle (which consumes the product's
  description object), together with the layer census of SURVEY Appendix A.
"""
import json
import os

import numpy as np
import torch

import snn_for_object_detection_amd as S
from oracle import net as ON


def test_oracle_convlstm_matches_reference_vectors(golden_dir):
    z = np.load(os.path.join(golden_dir, "convlstm.npz"))
    x = torch.from_numpy(z["x"]).requires_grad_()
    T, B, Cin = x.shape[:3]
    Ch = z["h"].shape[2]
    cell = ON.ConvLSTM(Cin, Ch)
    with torch.no_grad():
        cell.conv.weight.copy_(torch.from_numpy(z["weight"]))
    state, hs, cs = None, [], []
    for t in range(T):
        h, state = cell(x[t], state)
        hs.append(h)
        cs.append(state[1])
    hs, cs = torch.stack(hs), torch.stack(cs)
    assert torch.equal(hs.detach(), torch.from_numpy(z["h"])) and torch.equal(cs.detach(), torch.from_numpy(z["c"]))
    ((hs * torch.from_numpy(z["gh"])).sum() + (state[1] * torch.from_numpy(z["gc"])).sum()).backward()
    assert torch.allclose(x.grad, torch.from_numpy(z["gx"]), rtol=1e-6, atol=1e-7)
    assert torch.allclose(cell.conv.weight.grad, torch.from_numpy(z["gw"]), rtol=1e-6, atol=1e-6)


def _render(item):
    if isinstance(item, (list, tuple)):
        tag = type(item).__name__ if type(item).__name__ in ("Residual", "Dense") else "list"
        return {"merge": tag, "items": [_render(i) for i in item]}
    kind = type(item).__name__
    fields = {"Conv": ("out_channels", "kernel_size", "stride"), "Norm": ("bias",), "LIF": ("state_storage",),
              "LI": ("state_storage",), "SLI": ("state_storage",), "LSTM": ("hidden_size",),
              "Pool": ("type", "kernel_size", "stride"), "Up": ("scale", "mode")}.get(kind, ())
    return {"layer": kind, **{f: getattr(item, f) for f in fields}}


def test_tiny_yolo_description_equals_the_reference_description(golden_dir):
    want = json.load(open(os.path.join(golden_dir, "tiny_yolo_desc.json")))
    m = S.TinyYolo(num_classes=2, time_window=0)
    got = {"backbone": _render(m.backbone_cfgs()), "neck": _render(m.neck_cfgs()),
           "head": _render(m.head_cfgs(36, 27))}
    assert json.loads(json.dumps(got, sort_keys=True)) == want


def test_tiny_yolo_layer_census():
    """48 Conv, 22 Norm, 19 LIF, 3 LI, 14 Residual merges, 19 Dense merges, 28 Pass, 3 Tanh,
    3 Return - counted on the product's module tree, no oracle involved."""
    m = S.TinyYolo(num_classes=2, time_window=0)
    kinds = [type(x).__name__ for x in m.modules()]
    count = {k: kinds.count(k) for k in set(kinds)}
    assert count["HipConv2d"] == 48 and count["HipBatchNorm2d"] == 22
    assert count["LIFCell"] == 19 and count["LICell"] == 3
    assert count["Identity"] == 28 and count["HipTanh"] == 3 and count["Storage"] == 3
    merges = [b.merge for b in m.modules() if isinstance(b, S.BlockGen)]
    assert merges.count("residual") == 14 and merges.count("dense") == 19
    # conv census by (Cin, Cout, k, stride): the stage structure of the Appendix-A table
    convs = sorted((c.in_channels, c.out_channels, c.kernel_size[0], c.stride[0])
                   for c in m.modules() if isinstance(c, torch.nn.Conv2d))
    assert convs.count((2, 64, 3, 2)) == 1 and convs.count((32, 32, 3, 1)) == 2
    assert convs.count((64, 64, 3, 1)) == 3 and convs.count((128, 128, 3, 1)) == 9
    assert convs.count((64, 128, 3, 2)) == 1 and convs.count((128, 256, 3, 2)) == 1
    assert convs.count((256, 256, 3, 2)) == 2
    assert convs.count((768, 256, 1, 1)) == 1 and convs.count((640, 256, 1, 1)) == 1
    assert convs.count((512, 256, 1, 1)) == 1 and convs.count((320, 128, 1, 1)) == 1
    assert convs.count((256, 36, 1, 1)) == 3 and convs.count((256, 27, 1, 1)) == 3
    assert sum(1 for c in convs if c[2] == 3) == 19 and sum(1 for c in convs if c[2] == 1) == 29


def test_oracle_sli_matches_reference_vectors(golden_dir):
    """``sli.npz``: the reference's own ``sli_feed_forward_step`` (models/modules/sli.py:110-126) over a seeded
    sequence - the in-tree witness of norse's LI step ordering, pinned as data."""
    z = np.load(os.path.join(golden_dir, "sli.npz"))
    x = torch.from_numpy(z["x"]).requires_grad_()
    cell = ON.SLICell(dt=float(z["dt"]))
    state, vs = None, []
    for t in range(x.shape[0]):
        v, state = cell(x[t], state)   # state None at t = 0: the oracle builds SLICell.initial_state (0-dim v, zero i)
        vs.append(v)
    vs = torch.stack(vs)
    assert torch.equal(vs.detach(), torch.from_numpy(z["v"]))
    assert torch.equal(state[1].detach(), torch.from_numpy(z["i_final"]))
    ((vs * torch.from_numpy(z["gv"])).sum() + (state[1] * torch.from_numpy(z["gi"])).sum()).backward()
    assert torch.allclose(x.grad, torch.from_numpy(z["gx"]), rtol=1e-6, atol=1e-7)


def test_oracle_li_step_is_the_sli_step_without_the_gate(golden_dir):
    """The LI restatement (oracle/neurons.py, norse absent) against the reference-generated SLI vectors: with the
    saturation gate forced to 1 the SLI step IS norse's LI step (sli.py:110-126 vs SURVEY 8a-7), so feeding the LI
    restatement the gated inputs ``x * sigmoid(v_st - |v_prev|)`` recorded from the fixture must reproduce the fixture's
    potentials bit for bit - this pins the LI ordering (current jump first) and constants on reference data."""
    from oracle import neurons as NE
    z = np.load(os.path.join(golden_dir, "sli.npz"))
    x, v_ref = torch.from_numpy(z["x"]), torch.from_numpy(z["v"])
    state = None
    v_prev = torch.zeros_like(x[0])
    for t in range(x.shape[0]):
        gated = x[t] * torch.sigmoid(torch.as_tensor(1.0) - torch.abs(v_prev))
        v, state = NE.LICell(dt=float(z["dt"]))(gated, state)
        assert torch.equal(v.detach(), v_ref[t])
        v_prev = v.detach()
    assert torch.equal(state.i.detach(), torch.from_numpy(z["i_final"]))


def test_oracle_synapse_matches_reference_vectors(golden_dir):
    """``synapse.npz``: ``synapse_feed_forward_step`` (models/modules/synapse.py:73-103), sigma = 0 and 0.7."""
    z = np.load(os.path.join(golden_dir, "synapse.npz"))
    for tag in ("s0", "s07"):
        x = torch.from_numpy(z["x"]).clone().requires_grad_()
        cell = ON.SynapseCell(dt=float(z["dt"]), sigma_inhibition=float(z[f"sigma_{tag}"]))
        state, gs = None, []
        for t in range(x.shape[0]):
            g, state = cell(x[t], state)
            gs.append(g)
        gs = torch.stack(gs)
        assert torch.equal(gs.detach(), torch.from_numpy(z[f"g_{tag}"]))
        assert torch.equal(state[0].detach(), torch.from_numpy(z[f"p_final_{tag}"]))
        (gs * torch.from_numpy(z["gg"])).sum().backward()
        assert torch.allclose(x.grad, torch.from_numpy(z[f"gx_{tag}"]), rtol=1e-6, atol=1e-7)
