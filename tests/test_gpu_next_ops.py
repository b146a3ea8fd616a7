"""Remaining operator API on the device against the oracle restatements of the
reference's own cells: SLI (sli.py:80-126), Synapse (synapse.py:39-103), ConvLSTM (conv_lstm.py:10-78)."""
import os

import numpy as np
import pytest
import torch

from oracle import net as ON
from tests.util import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S(hip_lib):
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import snn_for_object_detection_amd as pkg
    return pkg


def _run_oracle(cell, x, bn=None):
    state, outs = None, []
    for t in range(x.shape[0]):
        xt = bn(x[t]) if bn is not None else x[t]
        o, state = cell(xt, state)
        outs.append(o)
    return torch.stack(outs), state


@pytest.mark.parametrize("with_bn", [False, True])
def test_sli_matches_reference_step(S, with_bn):
    from snn_for_object_detection_amd import BlockGen, Norm, SLI
    torch.manual_seed(3)
    T, B, C, H, W = 9, 2, 8, 5, 6
    x = 2.0 * torch.randn(T, B, C, H, W)
    blk = BlockGen(C, [Norm(), SLI()] if with_bn else [SLI()]).cuda()
    bn = torch.nn.BatchNorm2d(C) if with_bn else None
    if with_bn:
        bn.bias = None
    xr, xd = x.clone().requires_grad_(), x.cuda().requires_grad_()
    outr, str_ = _run_oracle(ON.SLICell(), xr, bn)
    outd, std = blk(xd)
    g = torch.randn_like(outr)
    (outr * g).sum().backward()
    (outd * g.cuda()).sum().backward()
    assert rel_err(outd, outr) < 1e-5
    st = std[0][-1]
    assert rel_err(st.v, str_[0]) < 1e-5 and rel_err(st.i, str_[1]) < 1e-5
    assert rel_err(xd.grad, xr.grad) < 1e-4
    if with_bn:
        assert rel_err(blk.net[0][0].weight.grad, bn.weight.grad) < 1e-4


@pytest.mark.parametrize("sigma", [0.0, 0.7])
def test_synapse_matches_reference_step(S, sigma):
    from snn_for_object_detection_amd.layer_gen import SynapseCell
    torch.manual_seed(4)
    T, B, C, H, W = 7, 2, 4, 3, 5
    x = torch.randn(T, B, C, H, W)
    xr, xd = x.clone().requires_grad_(), x.cuda().requires_grad_()
    outr, str_ = _run_oracle(ON.SynapseCell(sigma_inhibition=sigma), xr)
    outd, std = SynapseCell(sigma_inhibition=sigma).cuda()(xd)
    g = torch.randn_like(outr)
    (outr * g).sum().backward()
    (outd * g.cuda()).sum().backward()
    assert rel_err(outd, outr) < 1e-5 and rel_err(std.p, str_[0]) < 1e-5
    assert rel_err(xd.grad, xr.grad) < 1e-4
    # single-step protocol with carried state gives the same sequence
    cell, st, outs = SynapseCell(sigma_inhibition=sigma).cuda(), None, []
    for t in range(T):
        o, st = cell(x[t].cuda(), st)
        outs.append(o)
    assert torch.equal(torch.stack(outs), outd.detach())
    with pytest.raises(ValueError):
        SynapseCell(sigma_inhibition=0.3)


def test_conv_lstm_matches_reference(S):
    from snn_for_object_detection_amd import BlockGen, Conv, LSTM
    from oracle.net import BlockRef
    torch.manual_seed(6)
    T, B, H, W = 5, 2, 6, 7
    cfg = lambda: [Conv(8, 3), LSTM(12), LSTM()]  # noqa: E731
    blk, ref = BlockGen(3, cfg()), BlockRef(3, cfg())
    ref.load_state_dict(blk.state_dict())
    assert list(blk.state_dict()) == ["net.0.0.weight", "net.0.1.conv.weight", "net.0.2.conv.weight"]
    blk = blk.cuda()
    x = torch.randn(T, B, 3, H, W)
    outd, std = blk(x.cuda())
    state, outs = None, []
    for t in range(T):
        o, state = ref(x[t], state)
        outs.append(o)
    outr = torch.stack(outs)
    g = torch.randn_like(outr)
    (outr * g).sum().backward()
    (outd * g.cuda()).sum().backward()
    assert outd.shape == (T, B, 12, H, W) and rel_err(outd, outr) < 1e-5
    assert rel_err(std[0][2][1], state[0][2][1]) < 1e-5           # final cell state
    for pd, pr in zip(blk.parameters(), ref.parameters()):
        assert rel_err(pd.grad, pr.grad) < 1e-4


def test_conv_lstm_matches_reference_vectors(S, golden_dir):
    """The HIP ``LSTM()`` layer against vectors produced by the reference's own ``models/modules/conv_lstm.py:51-78``
    (tests/golden/make_golden.py): hidden / cell states of 3 steps and the gradients of a seeded loss."""
    from snn_for_object_detection_amd.layer_gen import ConvLSTM
    z = np.load(os.path.join(golden_dir, "convlstm.npz"))
    x = torch.from_numpy(z["x"]).cuda().requires_grad_()
    T, B, Cin = x.shape[:3]
    Ch = z["h"].shape[2]
    cell = ConvLSTM(Cin, Ch).cuda()
    with torch.no_grad():
        cell.conv.weight.copy_(torch.from_numpy(z["weight"]).cuda())
    hs, (h_last, c_last) = cell(x)                       # whole sequence
    assert rel_err(hs, torch.from_numpy(z["h"])) < 1e-5 and rel_err(c_last, torch.from_numpy(z["c"][-1])) < 1e-5
    ((hs * torch.from_numpy(z["gh"]).cuda()).sum() + (c_last * torch.from_numpy(z["gc"]).cuda()).sum()).backward()
    assert rel_err(x.grad, torch.from_numpy(z["gx"])) < 1e-4
    assert rel_err(cell.conv.weight.grad, torch.from_numpy(z["gw"])) < 1e-4
    state, cs = None, []                                 # reference protocol: one timestep at a time
    with torch.no_grad():
        for t in range(T):
            h, state = cell(x[t].detach(), state)
            cs.append(state[1])
    assert rel_err(torch.stack(cs), torch.from_numpy(z["c"])) < 1e-5


def test_sli_layer_matches_reference_generated_vectors(S, golden_dir):
    """HIP ``SLI()`` layer against outputs of the reference's own ``sli_feed_forward_step`` (tests/golden/sli.npz,
    written by make_golden.py from models/modules/sli.py:110-126)."""
    from snn_for_object_detection_amd import BlockGen, SLI
    z = np.load(os.path.join(golden_dir, "sli.npz"))
    x = torch.from_numpy(z["x"]).cuda().requires_grad_()
    blk = BlockGen(x.shape[2], [SLI()]).cuda()
    out, st = blk(x)
    ((out * torch.from_numpy(z["gv"]).cuda()).sum() + (st[0][-1].i * torch.from_numpy(z["gi"]).cuda()).sum()).backward()
    assert rel_err(out, torch.from_numpy(z["v"])) < 1e-6
    assert rel_err(st[0][-1].i, torch.from_numpy(z["i_final"])) < 1e-6
    assert rel_err(x.grad, torch.from_numpy(z["gx"])) < 1e-5


@pytest.mark.parametrize("tag", ["s0", "s07"])
def test_synapse_layer_matches_reference_generated_vectors(S, golden_dir, tag):
    """HIP ``Synapse`` cell against outputs of the reference's own ``synapse_feed_forward_step``
    (tests/golden/synapse.npz, from models/modules/synapse.py:73-103)."""
    from snn_for_object_detection_amd.layer_gen import SynapseCell
    z = np.load(os.path.join(golden_dir, "synapse.npz"))
    x = torch.from_numpy(z["x"]).cuda().requires_grad_()
    out, st = SynapseCell(dt=float(z["dt"]), sigma_inhibition=float(z[f"sigma_{tag}"])).cuda()(x)
    (out * torch.from_numpy(z["gg"]).cuda()).sum().backward()
    assert rel_err(out, torch.from_numpy(z[f"g_{tag}"])) < 1e-6
    assert rel_err(st.p, torch.from_numpy(z[f"p_final_{tag}"])) < 1e-6
    assert rel_err(x.grad, torch.from_numpy(z[f"gx_{tag}"])) < 1e-5
