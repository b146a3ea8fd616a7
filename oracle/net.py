"""Oracle restatement of the reference executor and step logic (TEST INFRASTRUCTURE ONLY).

Pure PyTorch, CPU, fp32, time-OUTER loop exactly as the reference runs it:

* ``BlockRef``      - ``models/generator.py:82-198`` (branch lists, Residual = stack+sum,
  Dense = cat(dim=1), per-layer state routing keyed on "forward has a ``state`` arg")
* ``make_layer``    - ``models/modules/layer_gen.py:96-347`` (module per LayerGen kind)
* ``SumPool2d`` / ``Storage`` / ``StateStorage`` - ``models/modules/common.py:18-123``
* ``ConvLSTM``      - ``models/modules/conv_lstm.py:10-78``;  ``SLICell`` - ``models/modules/sli.py:80-126``;
  ``SynapseCell``   - ``models/modules/synapse.py:39-103``
* ``BackboneRef`` / ``NeckRef`` / ``HeadRef`` / ``HeadGenRef`` - ``models/generator.py:220-538``
* ``SODaRef``       - ``models/soda.py:66-96,135-158,202-281`` without Lightning / torchmetrics

The module tree (attribute names, ModuleList nesting) mirrors the reference so a
``state_dict`` is interchangeable with the reference's and with the product's.

The network DESCRIPTION is not restated here: ``SODaRef`` consumes any object with
``backbone_cfgs() / neck_cfgs() / head_cfgs(box_out, cls_out)`` whose lists hold
LayerGen-like objects, dispatched by class NAME and read by attribute, so the same
description object drives the oracle and the product.

Conv2d / BatchNorm2d / pools / Upsample / losses are torch's own CPU kernels - the
very modules the reference instantiates.  LIF / LI come from ``oracle.neurons``
(PARITY UNPINNED, see there).

PINNED on a run of the reference itself since round 4: ``tests/golden/executor.npz`` holds what the reference's own
``SODa`` (``__init__ / forward / _forward_impl / _loss``), ``BlockGen``, ``BackboneGen``, ``NeckGen``, ``Head`` / ``HeadGen``
and layer modules produced for a description without spiking neurons (Conv, Norm, ReLU, SiLU, Tanh, Pool, Up, ConvLSTM,
Pass, Return, Residual, Dense; ``tests/golden/make_golden.py::executor_golden``): ``SODaRef`` / ``BlockRef`` reproduce its
state_dict keys, predictions, loss, every parameter gradient, the BatchNorm buffers and a block's ConvLSTM state tree
BIT FOR BIT (``tests/test_oracle_pins.py::test_oracle_executor_heads_and_loss_match_the_reference_run``).  ConvLSTM, SLI and
Synapse are pinned on their own fixtures as well; what stays unpinned in this file is nothing but the two norse cells.
"""

import inspect
from typing import Any, List, Optional, Tuple

import torch
from torch import nn
import torch.nn.functional as F

from . import detect
from .neurons import LICell, LIFCell


# --------------------------------------------------------------------------- small modules
class SumPool2d(nn.Module):
    def __init__(self, kernel_size: int, stride: int = 1, padding: int = 0):
        super().__init__()
        self.k, self.s, self.p = kernel_size, stride, padding

    def forward(self, x):
        return F.avg_pool2d(x, self.k, self.s, self.p) * self.k * self.k


class Storage(nn.Module):
    def __init__(self):
        super().__init__()
        self.storage = None

    def forward(self, x):
        self.storage = x
        return x

    def get_storage(self):
        kept, self.storage = self.storage, None
        return kept


class StateStorage(nn.Module):
    def __init__(self, m: nn.Module):
        super().__init__()
        self.module = m
        self.state_list: List[Any] = []
        self.spike_list: List[torch.Tensor] = []

    def get_spikes(self) -> torch.Tensor:
        return torch.stack(self.spike_list)

    def forward(self, x, state=None):
        if state is None:
            self.state_list.clear()
            self.spike_list.clear()
        out, new_state = self.module(x, state)
        if not self.training:
            self.state_list.append(new_state)
            self.spike_list.append(out)
        return out, new_state


class ConvLSTM(nn.Module):
    def __init__(self, in_channels: int, hidden_channels: int, kernel_size: int = 1, bias: bool = False):
        super().__init__()
        self.in_channels, self.hidden_channels = in_channels, hidden_channels
        self.conv = nn.Conv2d(in_channels + hidden_channels, 4 * hidden_channels, kernel_size, bias=bias)

    def forward(self, x, state=None):
        if state is None:
            b, _, h, w = x.shape
            state = (torch.zeros((b, self.hidden_channels, h, w), device=x.device),
                     torch.zeros((b, self.hidden_channels, h, w), device=x.device))
        hid, cell = state
        gates = self.conv(torch.cat([x, hid], dim=1))
        gi, gf, go, gc = torch.split(gates, self.hidden_channels, dim=1)
        cell_next = torch.sigmoid(gf) * cell + torch.sigmoid(gi) * torch.tanh(gc)
        hid_next = torch.sigmoid(go) * torch.tanh(cell_next)
        return hid_next, (hid_next, cell_next)


class SLICell(nn.Module):
    """Saturable LI: LI step with the input scaled by sigmoid(v_st - |v|)."""

    def __init__(self, dt: float = 0.001):
        super().__init__()
        self.dt = dt
        self.tau_syn_inv = torch.as_tensor(1.0 / 5e-3)
        self.tau_mem_inv = torch.as_tensor(1.0 / 1e-2)
        self.v_leak = torch.as_tensor(0.0)
        self.v_st = torch.as_tensor(1.0)

    def forward(self, x, state=None):
        if state is None:
            v = self.v_leak.detach().clone()
            v.requires_grad = True
            state = (v, torch.zeros(*x.shape, device=x.device, dtype=x.dtype))
        v, i = state
        i_jump = i + x * torch.sigmoid(self.v_st - torch.abs(v))
        v_new = v + self.dt * self.tau_mem_inv * ((self.v_leak - v) + i_jump)
        i_dec = i_jump + (-self.dt * self.tau_syn_inv * i_jump)
        return v_new, (v_new, i_dec)


class SynapseCell(nn.Module):
    def __init__(self, dt: float = 0.001, sigma_inhibition: float = 0.0):
        super().__init__()
        self.dt = dt
        self.tau_sec = torch.as_tensor(1.0 / 1e-3)
        self.tau_dis = torch.as_tensor(1.0 / 5e-3)
        self.sigma = torch.as_tensor(sigma_inhibition)
        if (self.sigma != 0) & (self.sigma < 0.5):
            raise ValueError("Valid values for sigma_inhibition are 0 or >= 0.5")

    def forward(self, x, state=None):
        if state is None:
            p0 = torch.zeros(*x.shape, device=x.device, dtype=x.dtype)
            p0.requires_grad = True
            state = (p0,)
        (p_old,) = state
        tau = torch.where(x > 0, self.tau_sec, self.tau_dis).to(x.dtype)
        p_new = p_old + (x - p_old) * tau * self.dt
        if self.sigma.is_nonzero():
            g = 4 * self.sigma * (p_new - self.sigma * p_new.square())
        else:
            g = p_new
        return g.clamp(0.0), (p_new,)


def _is_stateful(m: nn.Module) -> bool:
    # norse.torch.utils.state._is_module_stateful: "forward has a parameter named state"
    return "state" in inspect.signature(m.forward).parameters


# --------------------------------------------------------------------------- LayerGen -> module
def make_layer(gen: Any, in_channels: int) -> Tuple[nn.Module, int]:
    kind = type(gen).__name__
    if kind == "Pass":
        return nn.Identity(), in_channels
    if kind == "Conv":
        out = in_channels if gen.out_channels is None else gen.out_channels
        conv = nn.Conv2d(in_channels, out, kernel_size=gen.kernel_size, padding=int(gen.kernel_size / 2),
                         stride=gen.stride, bias=False)
        return conv, out
    if kind == "Norm":
        bn = nn.BatchNorm2d(in_channels)
        if not gen.bias:
            bn.bias = None
        return bn, in_channels
    if kind == "LIF":
        return (StateStorage(LIFCell()) if gen.state_storage else LIFCell()), in_channels
    if kind == "LI":
        return (StateStorage(LICell()) if gen.state_storage else LICell()), in_channels
    if kind == "SLI":
        return (StateStorage(SLICell()) if gen.state_storage else SLICell()), in_channels
    if kind == "Synapse":
        return SynapseCell(), in_channels
    if kind == "LSTM":
        hidden = in_channels if gen.hidden_size is None else gen.hidden_size
        return ConvLSTM(in_channels, hidden), hidden
    if kind == "Pool":
        ptype = gen.type
        if ptype == "A":
            return nn.AvgPool2d(gen.kernel_size, gen.stride), in_channels
        if ptype == "M":
            return nn.MaxPool2d(gen.kernel_size, gen.stride), in_channels
        if ptype == "S":
            return SumPool2d(gen.kernel_size, gen.stride), in_channels
        raise ValueError(f'[ERROR]: Non-existent pool type "{ptype}"!')
    if kind == "Up":
        return nn.Upsample(scale_factor=gen.scale, mode=gen.mode), in_channels
    if kind == "ReLU":
        return nn.ReLU(), in_channels
    if kind == "SiLU":
        return nn.SiLU(), in_channels
    if kind == "Tanh":
        return nn.Tanh(), in_channels
    if kind == "Return":
        gen.out_channels = in_channels
        return Storage(), in_channels
    raise TypeError(f"unknown layer description {kind}")


# --------------------------------------------------------------------------- block executor
class BlockRef(nn.Module):
    def __init__(self, in_channels: int, cfgs):
        super().__init__()
        merge = type(cfgs).__name__
        self.merge = merge if merge in ("Residual", "Dense") else "Forward"
        if self.merge == "Forward":
            cfgs = [cfgs]
        self.out_channels = 0
        branches, self.branch_state = [], []
        for bcfg in cfgs:
            layers, flags, ch = [], [], in_channels
            for gen in bcfg:
                if isinstance(gen, list):
                    layer = BlockRef(ch, gen)
                    ch = layer.out_channels
                else:
                    layer, ch = make_layer(gen, ch)
                layers.append(layer)
                flags.append(_is_stateful(layer))
            branches.append(nn.ModuleList(layers))
            self.branch_state.append(flags)
            if self.merge == "Residual":
                if not self.out_channels:
                    self.out_channels = ch
                elif self.out_channels != ch:
                    raise RuntimeError("[ERROR]: The number of channels in the residual network does not match! "
                                       "Check the configuration settings.")
            elif self.merge == "Dense":
                self.out_channels += ch
            else:
                self.out_channels = ch
        self.net = nn.ModuleList(branches)

    def forward(self, x, state=None):
        outs, new_state = [], []
        state = [None] * len(self.net) if state is None else state
        for branch, flags, bstate in zip(self.net, self.branch_state, state):
            bstate = [None] * len(branch) if bstate is None else bstate
            y = x
            for idx, (layer, stateful) in enumerate(zip(branch, flags)):
                if stateful:
                    y, bstate[idx] = layer(y, bstate[idx])
                else:
                    y = layer(y)
            outs.append(y)
            new_state.append(bstate)
        if self.merge == "Residual":
            merged = torch.stack(outs).sum(dim=0)
        elif self.merge == "Dense":
            merged = torch.cat(outs, dim=1)
        else:
            merged = outs[0]
        return merged, new_state


def _reference_init(root: nn.Module) -> None:
    # models/generator.py:245-256
    for m in root.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)


class BackboneRef(nn.Module):
    def __init__(self, cfg_fn, in_channels: int = 2, init_weights: bool = True):
        super().__init__()
        self.net_cfg = cfg_fn()
        self.net = BlockRef(in_channels, self.net_cfg)
        self.out_channels = self.net.out_channels
        if init_weights:
            _reference_init(self)

    def forward(self, x, state):
        return self.net(x, state)


class NeckRef(nn.Module):
    def __init__(self, cfg_fn, in_channels: int = 2, init_weights: bool = False):
        super().__init__()
        self.net_cfg = cfg_fn()
        self.net = BlockRef(in_channels, self.net_cfg)
        self.out_channels = self.net.out_channels
        if init_weights:
            _reference_init(self)
        self.out_shape = self._taps(self.net_cfg)

    def _taps(self, cfg) -> List[int]:
        found: List[int] = []
        for item in cfg:
            if type(item).__name__ == "Return":
                found.append(item.out_channels)
            elif isinstance(item, list):
                found += self._taps(item)
        return found

    def forward(self, x, state):
        _, state = self.net(x, state)
        taps = [m.get_storage() for m in self.net.modules() if isinstance(m, Storage)]
        return taps, state


class HeadGenRef(nn.Module):
    def __init__(self, cfg_fn, box_out: int, cls_out: int, in_channels: int = 2, init_weights: bool = False):
        super().__init__()
        self.box_out, self.cls_out = box_out, cls_out
        self.net_cfg = cfg_fn(box_out, cls_out)
        self.base_net = BlockRef(in_channels, [self.net_cfg[0]])
        self.box_net = BlockRef(self.base_net.out_channels, [self.net_cfg[1]])
        self.cls_net = BlockRef(self.base_net.out_channels, [self.net_cfg[2]])
        if init_weights:
            _reference_init(self)

    def forward(self, x, state):
        state = [None] * 3 if state is None else state
        y, state[0] = self.base_net(x, state[0])
        box, state[1] = self.box_net(y, state[1])
        cls, state[2] = self.cls_net(y, state[2])
        return box, cls, state


class _AnchorRef(nn.Module):
    def __init__(self, sizes, ratios):
        super().__init__()
        self.sizes = nn.Parameter(sizes, requires_grad=False)
        self.ratios = nn.Parameter(ratios, requires_grad=False)

    def forward(self, fmap):
        if not hasattr(self, "anchors"):
            h, w = fmap.shape[-2:]
            self.anchors = detect.anchor_boxes(h, w, self.sizes.data, self.ratios.data).to(fmap.device)
        return self.anchors


class HeadRef(nn.Module):
    def __init__(self, cfg_fn, num_classes: int, in_shape: List[int], init_weights: bool = True):
        super().__init__()
        self.num_classes = num_classes
        sizes, ratios = detect.head_anchor_sizes(len(in_shape))
        n_anchor = sizes.shape[1] * len(ratios)
        for idx, ch in enumerate(in_shape):
            setattr(self, f"anchor_gen_{idx}", _AnchorRef(sizes[idx], ratios))
            setattr(self, f"model_{idx}",
                    HeadGenRef(cfg_fn, n_anchor * 4, n_anchor * (num_classes + 1), ch, init_weights))

    @staticmethod
    def _flat(preds):
        return torch.cat([torch.flatten(p.permute(0, 2, 3, 1), start_dim=1) for p in preds], dim=1)

    def forward(self, maps, state):
        state = [None] * len(maps) if state is None else state
        anchors, cls_all, box_all = [], [], []
        for idx, fmap in enumerate(maps):
            anchors.append(getattr(self, f"anchor_gen_{idx}")(fmap))
            box, cls, state[idx] = getattr(self, f"model_{idx}")(fmap, state[idx])
            box_all.append(box)
            cls_all.append(cls)
        cls_flat = self._flat(cls_all)
        box_flat = self._flat(box_all)
        return (torch.cat(anchors), cls_flat.reshape(cls_flat.shape[0], -1, self.num_classes + 1),
                box_flat.reshape(box_flat.shape[0], -1, 4), state)


# --------------------------------------------------------------------------- detector
class SODaRef(nn.Module):
    """The reference's ``SODa`` step logic on a plain ``nn.Module``.

    ``desc`` supplies ``backbone_cfgs() / neck_cfgs() / head_cfgs(box_out, cls_out)``.
    """

    def __init__(self, desc, num_classes: int, loss_ratio: float = 0.04, time_window: int = 16,
                 iou_threshold: float = 0.4, learning_rate: float = 0.001, init_weights: bool = True):
        super().__init__()
        self.num_classes, self.loss_ratio, self.time_window = num_classes, loss_ratio, time_window
        self.iou_threshold, self.learning_rate = iou_threshold, learning_rate
        self.base_net = BackboneRef(desc.backbone_cfgs, in_channels=2, init_weights=init_weights)
        self.neck_net = NeckRef(desc.neck_cfgs, self.base_net.out_channels, init_weights=init_weights)
        self.head_net = HeadRef(desc.head_cfgs, num_classes, self.neck_net.out_shape, init_weights=init_weights)
        self.cls_loss = nn.CrossEntropyLoss(reduction="none")
        self.box_loss = nn.L1Loss(reduction="none")

    def configure_optimizers(self):
        return torch.optim.Adamax(self.parameters(), lr=self.learning_rate)

    def _forward_impl(self, x, state):
        state = [None] * 3 if state is None else state
        feat, state[0] = self.base_net(x, state[0])
        taps, state[1] = self.neck_net(feat, state[1])
        anchors, cls, box, state[2] = self.head_net(taps, state[2])
        return (anchors, cls, box), state

    def forward(self, X):
        state = None
        for frame in X:
            preds, state = self._forward_impl(frame, state)
        return preds

    def _rand_start_time(self):
        if not self.time_window:
            return 0
        return torch.randint(0, self.time_window, (1,), requires_grad=False, dtype=torch.uint32)

    def _loss(self, preds, labels):
        anchors, cls_preds, bbox_preds = preds
        offset, mask, cls_lab = detect.roi_targets(anchors, labels, self.iou_threshold)
        n_cls = cls_preds.shape[2]
        ce = self.cls_loss(cls_preds.reshape(-1, n_cls), cls_lab.reshape(-1))
        l1 = self.box_loss(bbox_preds * mask, offset * mask)
        positive = cls_lab.reshape(-1) > 0
        return ce[positive].mean() * self.loss_ratio + ce[~positive].mean() * (1 - self.loss_ratio) + l1.mean()

    def training_step(self, batch):
        preds = self.forward(batch[0][self._rand_start_time():])
        return self._loss(preds, batch[1])

    def predict(self, x, state):
        preds, state = self._forward_impl(x.unsqueeze(0), state)
        anchors, cls, bbox = preds
        det = detect.multibox_detection(F.softmax(cls, dim=2), bbox, anchors).squeeze(0)
        det = det[det[:, 0] >= 0]
        det[:, 2:] = torch.clamp(det[:, 2:], min=0.0, max=1.0)
        return det, state

    def spike_taps(self):
        """``{module path: [T,B,C,h,w]}`` for every StateStorage (eval mode only)."""
        return {name: m.get_spikes() for name, m in self.named_modules()
                if isinstance(m, StateStorage) and m.spike_list}
