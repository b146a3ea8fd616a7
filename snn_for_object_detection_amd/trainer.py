"""Flat-buffer training step: zero-copy gradient layout, one RCCL all-reduce, fused Adamax.

The reference distributes with Lightning DDP (``config/config.yaml:34-37``: one process per GPU,
bucketed NCCL gradient all-reduce) and optimises with ``torch.optim.Adamax`` (``soda.py:135-136``).
Here, per process (= per GPU):

* all trainable parameters live in ONE flat fp32 buffer (each parameter is a view, conv weights in
  their OHWI storage order) and all gradients in a second flat buffer;
* the backward kernels write weight gradients straight into their slice of the flat gradient
  (``functional.GradSlot``) - no per-parameter ``.grad`` tensors, no flatten / unflatten copies;
* data parallelism is ONE all-reduce (SUM) of the 16.9 MB flat gradient over RCCL / xGMI - the payload
  is latency-bound, so it is deliberately not bucketed - followed by ONE fused
  Adamax kernel over the flat buffers that also applies the ``1/world_size`` averaging.

``torch.distributed`` is the transport only (backend ``nccl`` = RCCL on ROCm, ``gloo`` in CPU tests).
"""

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist

from . import _hip
from .functional import GradSlot, wgrad_stream_sync


def _storage_view(flat: torch.Tensor, offset: int, like: torch.Tensor) -> torch.Tensor:
    """View of ``flat[offset:offset+n]`` with ``like``'s logical shape and dense storage order."""
    n = like.numel()
    chunk = flat[offset:offset + n]
    if like.dim() == 4:  # conv weight [O,I,KH,KW] stored channels-last = [O,KH,KW,I]
        o, i, kh, kw = like.shape
        return chunk.view(o, kh, kw, i).permute(0, 3, 1, 2)
    return chunk.view(like.shape)


class FlatTrainer:
    """Owns the flat parameter / gradient / optimiser-state buffers of a model.

    ``zero_grad()`` -> forward/backward (gradients land in ``flat_grad``) -> ``step()``
    (all-reduce across ``process_group`` when world_size > 1, then fused Adamax).
    """

    def __init__(self, model: torch.nn.Module, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 process_group=None, use_grad_slots: bool = True):
        self.params: List[torch.nn.Parameter] = [p for p in model.parameters() if p.requires_grad]
        if not self.params:
            raise RuntimeError("model has no trainable parameters")
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        total_padded = (total + 3) // 4 * 4
        self.flat_param = torch.zeros(total_padded, device=dev, dtype=torch.float32)
        self.flat_grad = torch.zeros(total_padded, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(total_padded, device=dev, dtype=torch.float32)
        self.exp_inf = torch.zeros(total_padded, device=dev, dtype=torch.float32)
        self.numel = total
        self.lr, self.betas, self.eps = lr, betas, eps
        self.step_count = 0
        self.group = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.slots: List[GradSlot] = []
        self.grad_views: List[torch.Tensor] = []
        # transposed copies [Cin][KH][KW][Cout] of every conv weight (operand of the data-gradient conv), refreshed
        # by ONE kernel after each optimiser step instead of one launch per layer inside the backward pass
        self.flat_wt = torch.empty(total_padded, device=dev, dtype=torch.float32) if dev.type == "cuda" else None
        self._conv_params: List[torch.nn.Parameter] = []
        table = []
        off = 0
        for p in self.params:
            if p.dtype != torch.float32:
                raise RuntimeError("FlatTrainer: float32 parameters only")
            view = _storage_view(self.flat_param, off, p.data)
            view.copy_(p.data)
            p.data = view
            gview = _storage_view(self.flat_grad, off, p.data)
            self.grad_views.append(gview)
            slot = GradSlot(self.flat_grad[off:off + p.numel()])
            self.slots.append(slot)
            if use_grad_slots:
                p._snn_grad_slot = slot
            if p.dim() == 4 and self.flat_wt is not None:
                o, i, kh, kw = p.shape
                table.append([off, o, kh * kw, i])
                p._snn_wt = self.flat_wt[off:off + p.numel()].view(i, kh, kw, o)
                p._snn_wt_version = -1  # not valid yet
                self._conv_params.append(p)
            off += p.numel()
        self._wt_table = torch.tensor(table, dtype=torch.int64, device=dev) if table else None
        self.refresh_transposed_weights()

    def refresh_transposed_weights(self) -> None:
        """Re-derive every ``p._snn_wt`` from the current weights (one launch).  ``_Conv2d.backward`` uses a cached
        transpose only while the parameter's version counter still matches, so a ``load_state_dict`` or any other
        torch-side in-place update simply falls back to the per-layer transpose until the next call."""
        if self._wt_table is None:
            return
        _hip.call("snn_weight_transpose_batched", self.flat_param.data_ptr(), self.flat_wt.data_ptr(),
                  self._wt_table.data_ptr(), len(self._conv_params), torch.cuda.current_stream().cuda_stream)
        for p in self._conv_params:
            p._snn_wt_version = p._version

    # ------------------------------------------------------------------
    def zero_grad(self) -> None:
        if self.flat_grad.is_cuda:
            wgrad_stream_sync()  # nothing may still be writing into the buffer that is about to be cleared
        self.flat_grad.zero_()
        for p, slot in zip(self.params, self.slots):
            slot.written = False
            p.grad = None

    def _collect_autograd_grads(self) -> None:
        """Parameters whose gradient came through autograd (ops without slot support) are folded in."""
        for p, slot, gview in zip(self.params, self.slots, self.grad_views):
            if p.grad is not None:
                if slot.written:
                    gview.add_(p.grad)
                else:
                    gview.copy_(p.grad)
                    slot.written = True
                p.grad = None

    def all_reduce(self) -> None:
        """One SUM all-reduce of the whole flat gradient (averaging is folded into the Adamax kernel)."""
        if self.world > 1:
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.group)

    def step(self) -> None:
        if self.flat_param.is_cuda:
            wgrad_stream_sync()  # weight-gradient kernels run on a side stream (functional._Conv2d.backward)
        self._collect_autograd_grads()
        self.all_reduce()
        self.step_count += 1
        if self.flat_param.is_cuda:
            _hip.call("snn_adamax_step", self.flat_param.data_ptr(), self.flat_grad.data_ptr(),
                      self.exp_avg.data_ptr(), self.exp_inf.data_ptr(), self.flat_param.numel(), self.lr,
                      self.betas[0], self.betas[1], self.eps, self.step_count, 1.0 / self.world,
                      torch.cuda.current_stream().cuda_stream)
            self.refresh_transposed_weights()
        else:
            raise RuntimeError("FlatTrainer.step: parameters are not on a HIP device; the optimiser kernel has no "
                               "CPU fallback")

    # ------------------------------------------------------------------ helpers for tests / checkpoints
    def synchronize(self) -> None:
        """Join the weight-gradient side stream: after this the current stream may read ``flat_grad``."""
        if self.flat_grad.is_cuda:
            wgrad_stream_sync()

    def averaged_grad(self) -> torch.Tensor:
        """The (all-reduced) flat gradient divided by world_size, as the optimiser sees it."""
        self.synchronize()
        return self.flat_grad[: self.numel] / self.world

    def grads_by_name(self, model: torch.nn.Module):
        self.synchronize()
        names = [n for n, p in model.named_parameters() if p.requires_grad]
        return {n: g for n, g in zip(names, self.grad_views)}


def convert_sync_batchnorm(model: torch.nn.Module, process_group=None) -> torch.nn.Module:
    """Opt-in equivalent of Lightning's ``sync_batchnorm: true`` (config/config.yaml:76).

    Every BatchNorm of ``model`` then normalises with statistics of the GLOBAL batch.  Thanks to the layer-major
    schedule the exchange is ONE all-reduce of a ``[T, C, 2]`` fp64 tensor per BatchNorm layer in forward and
    one in backward (44 per step for TinyYolo) instead of one per layer per timestep (1 408 at T = 32).
    Default (not converted): statistics and running buffers are rank-local.
    """
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("convert_sync_batchnorm needs an initialised process group")
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m._snn_sync_group = (process_group,)
    return model


def broadcast_parameters(trainer: FlatTrainer, src: int = 0) -> None:
    """Make every rank start from rank ``src``'s weights (DDP does this at construction)."""
    if trainer.world > 1:
        dist.broadcast(trainer.flat_param, src=src, group=trainer.group)
        if trainer.flat_param.is_cuda:
            trainer.refresh_transposed_weights()  # the weights changed behind torch's version counters
