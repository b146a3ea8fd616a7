"""Model generation tools (mirror of the reference's ``models/generator.py``).

Same description API and module-tree layout (``BlockGen.net`` is a ``ModuleList`` of per-branch
``ModuleList``s, ``generator.py:115,143``; heads are ``model_{i}.{base_net,box_net,cls_net}``,
``:403-413,522-525``) so ``state_dict`` keys match the reference's checkpoints.

What is re-designed is the EXECUTOR.  Every recurrence in the generated nets is per-neuron and a
block is a DAG per timestep (``generator.py:181-198``), so instead of the reference's time-outer
loop (``soda.py:141-144``) a block may be handed the whole sequence ``[T,B,C,H,W]`` and runs
layer-major: one conv launch per layer for all ``T*B`` frames and one fused
BatchNorm+LIF/LI temporal-scan launch per ``Norm -> neuron`` pair, with the membrane state kept in
registers across ``T``.  A single timestep ``[B,C,H,W]`` (+ carried state) uses the same kernels
with ``T = 1``, which keeps the reference's calling protocol for streaming ``predict``.

``ListGen``  : ``List[LayerGen | ListGen]``;  ``ListState``: ``List[Tensor | None | ListState]``.
"""

import inspect
import os
from typing import Any, List, Optional, Tuple, Union

import torch
from torch import nn

from . import _hip
from . import functional as HF
from .anchors import AnchorGenerator
from .layer_gen import *  # noqa: F401,F403  (the reference re-exports the layer generators here)
from .layer_gen import (ConvLSTM, Dense, HipBatchNorm2d, HipConv2d, HipReLU, HipSiLU, HipTanh, LayerGen,
                        LICell, LIFCell, Residual, Return, SLICell, StateStorage, Storage, SumPool2d, SynapseCell)

ListGen = List[Union[LayerGen, "ListGen"]]
ListState = List[Union[torch.Tensor, None, "ListState"]]


def _is_module_stateful(m: nn.Module) -> bool:
    """norse's rule (norse.torch.utils.state): a module is stateful iff ``forward`` takes ``state``."""
    return "state" in inspect.signature(m.forward).parameters


def _neuron_cell(layer: nn.Module) -> Optional[nn.Module]:
    cell = layer.module if isinstance(layer, StateStorage) else layer
    return cell if isinstance(cell, (LIFCell, LICell, SLICell, SynapseCell)) else None


#####################################################################
#                         Block Generators                          #
#####################################################################
def _is_plain_1x1(layer) -> bool:
    return (isinstance(layer, HipConv2d) and layer.kernel_size == (1, 1) and layer.stride == (1, 1)
            and layer.padding == (0, 0) and layer.bias is None and layer.groups == 1)


class BlockGen(nn.Module):
    """Builds a block from a (nested) configuration list and runs it (generator.py:35-198).

    A plain list is one sequential branch; ``Residual([...])`` sums its branches, ``Dense([...])``
    concatenates them on channels; nested lists become nested ``BlockGen``s.
    """

    def __init__(self, in_channels: int, cfgs: ListGen, in_unbounded: bool = False):
        """``in_unbounded`` (internal): the block input may exceed the fp16 x 3 range contract (it comes, possibly through
        convolutions, pools, merges or nested blocks, from ReLU / SiLU / SumPool / ConvLSTM with no BatchNorm, spiking neuron
        or Tanh in between); ``out_unbounded`` says the same of the block output."""
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = 0
        self.out_unbounded = False
        if isinstance(cfgs, Residual):
            self.merge = "residual"
        elif isinstance(cfgs, Dense):
            self.merge = "dense"
        else:
            self.merge = "forward"
            cfgs = [cfgs]

        branch_list: List[nn.ModuleList] = []
        self.branch_state: List[List[bool]] = []
        self._branch_channels: List[int] = []
        for branch_cfg in cfgs:
            layers, flags, channels, unb = self._make_branch(in_channels, branch_cfg, in_unbounded)
            self.out_unbounded = self.out_unbounded or unb   # a sum / concatenation is bounded only if every branch is
            branch_list.append(layers)
            self.branch_state.append(flags)
            self._branch_channels.append(channels)
            self._account_channels(channels)
        self.net = nn.ModuleList(branch_list)
        # fused execution plan per branch: list of (kind, first_index, n_layers)
        self._plan = [self._plan_branch(branch) for branch in self.net]
        # zero-copy Dense merge: channel offset of every branch inside the concat buffer, and the offset of
        # the first branch that is a bare Pass (its content = the block input, which the PRODUCER of that
        # input can write there directly when this block is the next layer of the parent branch)
        self._offsets = [sum(self._branch_channels[:k]) for k in range(len(self._branch_channels))]
        # Residual([[..., Norm, LIF], [Pass]]) (the YOLO bottleneck): the shortcut is added in the LIF kernel's
        # output store instead of a separate stack + sum pass: (main branch, pass branch) or None
        self._fused_shortcut: Optional[Tuple[int, int]] = None
        if self.merge == "residual" and len(self.net) == 2 and not os.environ.get("SNN_NO_FUSED_SHORTCUT"):
            for main, other in ((0, 1), (1, 0)):
                is_pass = len(self.net[other]) == 1 and isinstance(self.net[other][0], nn.Identity)
                plan = self._plan[main]
                if not (is_pass and plan and plan[-1][0] == "norm_neuron" and plan[-1][2] == 2):
                    continue
                holder = self.net[main][plan[-1][1] + 1]
                if isinstance(_neuron_cell(holder), LIFCell) and not isinstance(holder, StateStorage):
                    self._fused_shortcut = (main, other)
                    break
        self._pass_offset: Optional[int] = None
        if self.merge == "dense":
            for k, branch in enumerate(self.net):
                if len(branch) == 1 and isinstance(branch[0], nn.Identity):
                    self._pass_offset = self._offsets[k]
                    break
        self._siblings = self._plan_siblings()

    # ------------------------------------------------------------------ construction
    def _make_branch(self, in_channels: int, cfg: ListGen, unbounded: bool = False):
        state_layers: List[bool] = []
        layer_list: List[nn.Module] = []
        channels = in_channels
        for layer_gen in cfg:
            if isinstance(layer_gen, list):
                layer = BlockGen(channels, layer_gen, in_unbounded=unbounded)
                channels = layer.out_channels
            else:
                layer, channels = layer_gen.get(channels)
            # fp16x3 (the default forward arithmetic) has a range contract (|x| < 4094) that spikes, sums of spikes and
            # normalised activations meet by construction; a convolution whose input is not PROVABLY bounded - an
            # unbounded activation (ReLU / SiLU / SumPool / ConvLSTM) upstream with no BatchNorm, spiking neuron or Tanh
            # since, through any chain of convolutions, pools, passes, merges, nested blocks - takes the any-range bf16x6
            # arithmetic instead (unless the description set a precision itself)
            if isinstance(layer, HipConv2d) and layer.forward_precision is None and unbounded:
                layer.forward_precision = "bf16x6"
            if isinstance(layer, BlockGen):
                unbounded = layer.out_unbounded
            elif isinstance(layer, (HipReLU, HipSiLU, SumPool2d, ConvLSTM)):
                unbounded = True
            elif isinstance(layer, (HipBatchNorm2d, HipTanh)) or isinstance(_neuron_cell(layer), LIFCell):
                unbounded = False
            # everything else (convolutions, pools, up-sampling, pass, LI / SLI / Synapse integrators, Return) hands the
            # status of its input on
            layer_list.append(layer)
            state_layers.append(_is_module_stateful(layer))
        return nn.ModuleList(layer_list), state_layers, channels, unbounded

    def _account_channels(self, channels: int) -> None:
        if self.merge == "residual":
            if not self.out_channels:
                self.out_channels = channels
            elif self.out_channels != channels:
                raise RuntimeError(
                    "[ERROR]: The number of channels in the residual "
                    "network does not match! Check the configuration settings."
                )
        elif self.merge == "dense":
            self.out_channels += channels
        else:
            self.out_channels = channels

    @staticmethod
    def _plan_branch(branch: nn.ModuleList) -> List[Tuple[str, int, int]]:
        """Peephole fusion: Norm -> (LIF | LI [-> Tanh]) becomes one temporal-scan launch."""
        plan, idx, n = [], 0, len(branch)
        while idx < n:
            layer = branch[idx]
            nxt = branch[idx + 1] if idx + 1 < n else None
            if (_is_plain_1x1(layer) and isinstance(nxt, BlockGen) and nxt._opens_with_1x1(layer.out_channels)
                    and not os.environ.get("SNN_NO_COMPOSED_CONV")):
                # Conv(c,1) feeding only the branch-opening Conv(.,1)s of the next block (the C2f entry): the two
                # linear maps are composed and the intermediate tensor never exists (functional._ComposedConv1x1)
                plan.append(("conv_block", idx, 2))
                idx += 2
            elif isinstance(layer, HipBatchNorm2d) and nxt is not None and _neuron_cell(nxt) is not None:
                cell = _neuron_cell(nxt)
                tanh_follows = (isinstance(cell, LICell) and not isinstance(nxt, StateStorage)
                                and idx + 2 < n and isinstance(branch[idx + 2], HipTanh))
                plan.append(("norm_neuron", idx, 3 if tanh_follows else 2))
                idx += 3 if tanh_follows else 2
            else:
                plan.append(("layer", idx, 1))
                idx += 1
        return plan

    def _plan_siblings(self):
        """Dense block whose branches ALL open with a plain 1x1 convolution of the block input, and whose outputs land
        side by side in the concat buffer (the C2f split ``Dense([[Conv(c/2,1), rec_block], [Conv(c/2,1)]])``,
        ``models/tiny_yolo.py:84-85``: the first output goes to the pass-through slot at the END of the nested block's
        slice, the second right behind it): the convolutions run as ONE (``functional.sibling_conv1x1``).
        -> per branch ``(channel offset in this block's buffer, channels, index of the nested block or None)`` or None."""
        if self.merge != "dense" or len(self.net) < 2 or not HF.USE_SIBLING_FUSION:
            return None
        specs = []
        precs = set()
        for b, (branch, plan) in enumerate(zip(self.net, self._plan)):
            if not plan or plan[0] != ("layer", 0, 1) or not _is_plain_1x1(branch[0]):
                return None
            conv = branch[0]
            precs.add((conv.forward_precision, conv.backward_precision))
            if len(plan) == 1:
                specs.append((self._offsets[b], conv.out_channels, None))
            elif (len(plan) == 2 and plan[1][0] == "layer" and isinstance(branch[plan[1][1]], BlockGen)
                  and branch[plan[1][1]]._pass_offset is not None and branch[plan[1][1]].in_channels == conv.out_channels):
                nxt = branch[plan[1][1]]
                specs.append((self._offsets[b] + nxt._pass_offset, conv.out_channels, plan[1][1]))
            else:
                return None
        if len(precs) != 1:
            return None
        for (off, c, _), (nxt_off, _, _) in zip(specs, specs[1:]):
            if off + c != nxt_off:
                return None
        return specs

    def _run_siblings(self, X: torch.Tensor, promise, pre_conv):
        """The branch-opening convolutions as one launch: -> (branch inputs, promises of the nested blocks)."""
        specs = self._siblings
        convs = [branch[0] for branch in self.net]
        pendings = []
        for b, (_, _, nested) in enumerate(specs):
            if nested is None:
                pendings.append(None)
            else:
                nxt = self.net[b][nested]
                pendings.append(HF.ConcatPromise(nxt.out_channels,
                                                 parent=HF.Dest(promise, self._offsets[b], self._branch_channels[b])))
        first = convs[0]
        fp = (pre_conv.forward_precision if pre_conv is not None else None) or first.forward_precision
        bp = (pre_conv.backward_precision if pre_conv is not None else None) or first.backward_precision
        dest = HF.Dest(promise, specs[0][0], sum(c for _, c, _ in specs))
        whole = HF.sibling_conv1x1(X, pre_conv.weight if pre_conv is not None else None, [c.weight for c in convs],
                                   dest=dest, forward_precision=fp, backward_precision=bp)
        T, B, _, H, W = whole.shape
        for p in pendings:
            if p is not None:
                p.get(T, B, H, W, whole)   # the nested block's buffer = its slice of this block's, now holding its pass slot
        return HF.split_channels(whole, [c for _, c, _ in specs]), pendings

    def _opens_with_1x1(self, in_channels: int) -> bool:
        """Every branch starts with a plain 1x1 convolution of the block input (and nothing else reads it)."""
        if self.in_channels != in_channels or not len(self.net):
            return False
        for branch, plan in zip(self.net, self._plan):
            if not plan or plan[0] != ("layer", 0, 1) or not _is_plain_1x1(branch[0]):
                return False
        return True

    # ------------------------------------------------------------------ execution
    def forward(self, X: torch.Tensor, state: Optional[ListState] = None, dest=None, promise=None, pre_conv=None,
                last_only: bool = False):
        """``X`` is ``[B,C,h,w]`` (one timestep) or ``[T,B,C,h,w]`` (whole sequence).

        ``last_only`` (internal; the detection head): the caller keeps the last timestep only - when the block ends in
        ``Norm -> LIF`` / ``Norm -> LI [-> Tanh]`` the fused scan returns ``[B,C,h,w]`` of the last step and never writes the others.

        ``dest`` / ``promise`` are internal (``functional.Dest`` / ``ConcatPromise``): on sequences the
        Dense merge is zero-copy - each branch's last operator writes its channel slice of one shared
        buffer, nested blocks write slices of their parent's slice, and a bare ``Pass`` branch costs nothing
        when the layer that produced the block input was told to write it there (look-ahead below).
        """
        out = []
        out_state = []
        state = [None] * len(self.net) if state is None else state
        zero_copy = X.dim() == 5
        if not zero_copy:
            dest = promise = None
        if self.merge == "dense" and zero_copy and promise is None:
            promise = HF.ConcatPromise(self.out_channels, parent=dest)
        siblings = None
        spike_coded = getattr(X, "_snn_spike_threshold", None) is not None
        if (self._siblings is not None and zero_copy and HF.USE_SIBLING_FUSION and X.is_cuda
                and (X.dtype != torch.bfloat16 or all(c % 32 == 0 for c in (self.in_channels, *(s[1] for s in self._siblings))))):
            inputs, siblings = self._run_siblings(X, promise, pre_conv)
        elif spike_coded:
            raise RuntimeError("internal error: a tensor of saved potentials (spikes never written) reached a block that does "
                               "not open with the fused sibling convolution")
        else:
            inputs = HF.fanout(X, len(self.net))
        for b, (branch, flags, plan, branch_state) in enumerate(zip(self.net, self.branch_state, self._plan, state)):
            branch_state = [None] * len(branch) if branch_state is None else branch_state
            if self.merge == "dense":
                branch_dest = HF.Dest(promise, self._offsets[b], self._branch_channels[b]) if zero_copy else None
            elif self.merge == "forward":
                branch_dest = dest
            else:
                branch_dest = None  # residual: branches are summed, only the sum is placed
            fuse_here = self._fused_shortcut is not None and self._fused_shortcut[0] == b
            Y = inputs[b]
            pending = None  # promise prepared for the next step (a Dense block with a Pass branch)
            if siblings is not None:
                pending = siblings[b]   # (the fused convolution was step 0 of every branch)
            for k, (kind, idx, span) in enumerate(plan):
                if siblings is not None and k == 0:
                    continue
                last = k == len(plan) - 1
                step_dest = branch_dest if last else None
                step_promise, pending = pending, None
                if zero_copy and not last:
                    nkind, nidx, _ = plan[k + 1]
                    nxt = branch[nidx]
                    if nkind == "layer" and isinstance(nxt, BlockGen) and nxt._pass_offset is not None:
                        parent = branch_dest if (k + 1 == len(plan) - 1) else None
                        pending = HF.ConcatPromise(nxt.out_channels, parent=parent)
                        step_dest = HF.Dest(pending, nxt._pass_offset, nxt.in_channels)
                layer = branch[idx]
                if kind == "norm_neuron":
                    holder = branch[idx + 1]
                    cell = _neuron_cell(holder)
                    neuron = cell.kind
                    if span == 3:
                        neuron = _hip.NEURON_LI_TANH
                    old = branch_state[idx + 1]
                    # LI+Tanh keeps its own output for the backward pass and wants it dense: place by copy
                    direct = step_dest if neuron != _hip.NEURON_LI_TANH else None
                    shortcut = None
                    if fuse_here and last:  # the block's merged output: LIF(...) + shortcut, placed at `dest`
                        direct, shortcut = dest, inputs[self._fused_shortcut[1]]
                    only_last = (last_only and last and zero_copy and len(self.net) == 1 and direct is None
                                 and shortcut is None
                                 and neuron in (_hip.NEURON_LIF, _hip.NEURON_LI, _hip.NEURON_LI_TANH)
                                 and not (isinstance(holder, StateStorage) and not self.training))
                    # the only consumer is the fused sibling convolution of the next block (the stage-entry Conv -> Norm -> LIF
                    # in front of a C2f split): it can form the spikes from the saved potentials, none are written
                    # ... or a plain convolution (a Conv -> Norm -> LIF -> Conv stack: the deep backbones): HF.conv2d thresholds on
                    # load where its kernels cover the shape and writes the spikes itself where they do not
                    nxt_l = branch[plan[k + 1][1]] if (not last and plan[k + 1][0] == "layer") else None
                    spikes_ok = (zero_copy and not last and self.training and neuron == _hip.NEURON_LIF
                                 and direct is None and shortcut is None and not isinstance(holder, StateStorage)
                                 and HF.USE_SPIKES_FROM_VDEC
                                 and ((plan[k + 1][0] == "conv_block" and branch[plan[k + 1][1] + 1]._siblings is not None
                                       and HF.USE_SIBLING_FUSION)
                                      or (isinstance(nxt_l, HipConv2d) and nxt_l.in_channels % 32 == 0
                                          and nxt_l.out_channels % 4 == 0 and nxt_l.kernel_size[0] <= 5
                                          and nxt_l.forward_precision is None and nxt_l.backward_precision is None)))
                    Y, new = HF.affine_neuron(Y, neuron, old, bn=layer, params=cell.params, dest=direct,
                                              addend=shortcut, last_only=only_last, spikes_ok=spikes_ok)
                    if isinstance(holder, StateStorage):
                        holder.record(old, Y, new)
                    branch_state[idx + 1] = new
                elif kind == "conv_block":
                    blk = branch[idx + 1]
                    Y, branch_state[idx + 1] = blk(Y, branch_state[idx + 1], dest=step_dest, promise=step_promise,
                                                   pre_conv=layer)
                elif isinstance(layer, BlockGen):
                    Y, branch_state[idx] = layer(Y, branch_state[idx], dest=step_dest, promise=step_promise,
                                                 last_only=last_only and last and len(self.net) == 1)
                elif isinstance(layer, HipConv2d) and pre_conv is not None and k == 0:
                    Y = HF.composed_conv1x1(Y, pre_conv.weight, layer.weight, dest=step_dest,
                                            forward_precision=pre_conv.forward_precision or layer.forward_precision,
                                            backward_precision=(pre_conv.backward_precision
                                                                or layer.backward_precision))
                elif isinstance(layer, HipConv2d):
                    # a train-mode BatchNorm right behind: the convolution sums y, y^2 for it while storing y
                    feeds_bn = False
                    if not last and plan[k + 1][0] == "norm_neuron" and Y.is_cuda:
                        bn = branch[plan[k + 1][1]]
                        feeds_bn = bn.training or (bn.running_mean is None and bn.running_var is None)
                    Y = layer(Y, dest=step_dest, bn_stats=feeds_bn)
                elif isinstance(layer, (LIFCell, LICell, SLICell, SynapseCell)):
                    Y, branch_state[idx] = layer(Y, branch_state[idx], dest=step_dest)
                elif flags[idx]:
                    Y, branch_state[idx] = layer(Y, branch_state[idx])
                else:
                    Y = layer(Y)
                if step_dest is not None:
                    Y = HF.place(Y, step_dest)  # no-op when the operator wrote there itself
            if branch_dest is not None and not plan:
                Y = HF.place(Y, branch_dest)
            out.append(Y)
            out_state.append(branch_state)
        if self.merge == "residual" and self._fused_shortcut is not None:
            merged = out[self._fused_shortcut[0]]
        elif self.merge == "residual":
            merged = HF.sum_tensors(out, dest=dest)
        elif self.merge == "dense":
            merged = HF.assemble_channels(promise, out) if zero_copy else HF.concat_channels(out)
        else:
            merged = out[0]
        return merged, out_state


#####################################################################
#                         Model Generator                           #
#####################################################################
class ModelGen(nn.Module):
    """Base class of the backbone / neck / head generators (generator.py:206-276)."""

    def __init__(self, cfg, in_channels: int = 2, init_weights: bool = True) -> None:
        super().__init__()
        self.out_channels = 0
        self.net_cfg = self._load_cfg(cfg)
        self._net_generator(in_channels)
        if init_weights:
            # generator.py:245-256: Kaiming-normal (fan_out, relu) convs, BatchNorm weight 1
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                    if m.bias is not None:
                        nn.init.constant_(m.bias, 0)
                elif isinstance(m, nn.BatchNorm2d):
                    nn.init.constant_(m.weight, 1)
                    if m.bias is not None:
                        nn.init.constant_(m.bias, 0)

    def _net_generator(self, in_channels: int) -> None:
        self.net = BlockGen(in_channels, self.net_cfg)
        self.out_channels = self.net.out_channels

    def _load_cfg(self, cfg) -> ListGen:
        raise NotImplementedError

    def forward(self, X: torch.Tensor, state: Optional[ListState] = None):
        raise NotImplementedError


class BackboneGen(ModelGen):
    """Returns the tensor of the last layer (generator.py:283-295)."""

    def _load_cfg(self, cfg) -> ListGen:
        return cfg()

    def forward(self, X: torch.Tensor, state: Optional[ListState]) -> Tuple[torch.Tensor, ListState]:
        return self.net(X, state)


class NeckGen(ModelGen):
    """Returns the tensors stashed by the ``Return`` layers (generator.py:303-351)."""

    def __init__(self, cfg, in_channels: int = 2, init_weights: bool = False):
        super().__init__(cfg, in_channels, init_weights)
        self.out_shape = self._search_out(self.net_cfg)

    def _search_out(self, cfg) -> List[int]:
        out: List[int] = []
        for module in cfg:
            if isinstance(module, Return):
                out.append(module.out_channels)
            elif isinstance(module, list):
                out += self._search_out(module)
        return out

    def _load_cfg(self, cfg) -> ListGen:
        return cfg()

    def forward(self, X: torch.Tensor, state: Optional[ListState]) -> Tuple[List[torch.Tensor], ListState]:
        out = []
        _, state = self.net(X, state)
        for module in self.net.modules():
            if isinstance(module, Storage):
                out.append(module.get_storage())
        return out, state


#####################################################################
#                          Head descriptor                          #
#####################################################################
class Head(nn.Module):
    """One ``HeadGen`` + ``AnchorGenerator`` per feature map; merges predictions (generator.py:359-457)."""

    def __init__(self, cfg, num_classes: int, in_shape: List[int], init_weights: bool = True) -> None:
        super().__init__()
        self.num_classes = num_classes

        max = 0.75
        min = 0.08
        size_per_pix = 3
        sizes = torch.arange(min, max, (max - min) / (len(in_shape) * size_per_pix), dtype=torch.float32)
        sizes = sizes.reshape((-1, size_per_pix))
        ratios = torch.tensor((0.5, 1.0, 2), dtype=torch.float32)

        num_anchors = size_per_pix * len(ratios)
        num_class_out = num_anchors * (self.num_classes + 1)
        num_box_out = num_anchors * 4

        for idx, channels in enumerate(in_shape):
            setattr(self, f"anchor_gen_{idx}", AnchorGenerator(sizes=sizes[idx], ratios=ratios))
            setattr(self, f"model_{idx}", HeadGen(cfg, num_box_out, num_class_out, channels, init_weights))

    def forward(self, X: List[torch.Tensor], state: Optional[ListState]):
        """-> ``(anchors[A,4], cls_preds[B,A,C+1], bbox_preds[B,A,4], state)``.

        Given sequences the predictions are those of the LAST timestep (all the reference keeps,
        ``soda.py:141-144``) and ``state`` is the state after the last timestep.
        """
        state = [None] * len(X) if state is None else state
        anchors, cls_preds, bbox_preds = [], [], []
        # The heads are independent of each other and of the neck stages behind their tap: the first HEAD_STREAMS heads (the
        # largest maps) run on streams of their own (functional.aux_streams) that wait for THEIR tap only, so the 30x38
        # head overlaps the latency-bound 15x19 / 8x10 neck stages - forward and, since autograd runs a node's backward
        # on the stream of its forward, backward.  The other heads follow the end of the neck on the main stream.  (One
        # auxiliary stream by default: HIP has four hardware queues - main, weight gradients, gradient exchange, this.)
        streams = [None] * len(X)
        n_aux = min(HF.HEAD_STREAMS, len(X) - 1) if HF.USE_HEAD_STREAMS else 0
        if (n_aux > 0 and all(m.is_cuda and m.dim() == 5 for m in X[:n_aux])
                and all(getattr(m, "_snn_ready", None) is not None for m in X[:n_aux])):
            streams = HF.aux_streams(X[0].device, n_aux) + [None] * (len(X) - n_aux)
        main = torch.cuda.current_stream() if X and X[0].is_cuda else None
        for idx, map in enumerate(X):
            anchors.append(getattr(self, f"anchor_gen_{idx}")(map))
            side = streams[idx]
            if side is None:
                boxes, classes, state[idx] = getattr(self, f"model_{idx}")(map, state[idx])
            else:
                side.wait_event(map._snn_ready[1])
                with torch.cuda.stream(side):
                    boxes, classes, state[idx] = getattr(self, f"model_{idx}")(map, state[idx])
                for t in (boxes, classes, *_tensors_of(state[idx])):
                    t.record_stream(main)          # allocated on the head's stream, read on the main stream afterwards
            bbox_preds.append(boxes)
            cls_preds.append(classes)
        for side in streams:
            if side is not None:
                main.wait_stream(side)
        anchors = torch.cat(anchors)
        cls_preds = self._concat_preds(cls_preds)
        cls_preds = cls_preds.reshape(cls_preds.shape[0], -1, self.num_classes + 1)
        bbox_preds = self._concat_preds(bbox_preds)
        bbox_preds = bbox_preds.reshape(bbox_preds.shape[0], -1, 4)
        return anchors, cls_preds, bbox_preds, state

    def _flatten_pred(self, pred: torch.Tensor) -> torch.Tensor:
        # channels-last storage makes this permute + flatten a free view
        return torch.flatten(torch.permute(pred, (0, 2, 3, 1)), start_dim=1)

    def _concat_preds(self, preds: List[torch.Tensor]) -> torch.Tensor:
        return torch.cat([self._flatten_pred(p) for p in preds], dim=1)


class HeadGen(ModelGen):
    """``cfg(box_out, cls_out)`` -> three lists: preparation, box net, class net (generator.py:465-538)."""

    def __init__(self, cfg, box_out: int, cls_out: int, in_channels: int = 2, init_weights=False):
        self.box_out = box_out
        self.cls_out = cls_out
        super().__init__(cfg, in_channels, init_weights)

    def _net_generator(self, in_channels: int) -> None:
        self.base_net = BlockGen(in_channels, [self.net_cfg[0]])
        self.box_net = BlockGen(self.base_net.out_channels, [self.net_cfg[1]])
        self.cls_net = BlockGen(self.base_net.out_channels, [self.net_cfg[2]])

    def _load_cfg(self, cfg) -> ListGen:
        return cfg(self.box_out, self.cls_out)

    def forward(self, X: torch.Tensor, state: Optional[ListState]):
        state = [None] * 3 if state is None else state
        # sequence input, stateless prediction nets: only the last timestep's predictions survive
        keep_last = X.dim() == 5 and not (any(_has_state(m) for m in self.box_net.modules())
                                          or any(_has_state(m) for m in self.cls_net.modules()))
        Y, state[0] = self.base_net(X, state[0], last_only=keep_last)
        if Y.dim() == 5 and keep_last:
            Y = Y[-1]
        if keep_last and Y.dtype == torch.bfloat16:
            Y = HF.to_float32(Y)   # bf16-storage mode: the last-step read-out (a few frames) and the prediction nets run in fp32
        box, state[1] = self.box_net(Y, state[1])
        cls, state[2] = self.cls_net(Y, state[2])
        if box.dim() == 5:
            box, cls = box[-1], cls[-1]
        if box.dtype == torch.bfloat16:
            box, cls = HF.to_float32(box), HF.to_float32(cls)
        return box, cls, state


def _tensors_of(state):
    """Every tensor of a (nested) state list / tuple."""
    if isinstance(state, torch.Tensor):
        yield state
    elif isinstance(state, (list, tuple)):
        for s in state:
            yield from _tensors_of(s)


def _has_state(m: nn.Module) -> bool:
    """Modules that must see every timestep: stateful ones, and BatchNorm (running-stat updates)."""
    if isinstance(m, nn.BatchNorm2d):
        return True
    return not isinstance(m, BlockGen) and _is_module_stateful(m)
