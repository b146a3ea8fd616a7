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

import weakref
from typing import Dict, Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist

from . import _hip
from . import functional as _HF
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
                 process_group=None, use_grad_slots: bool = True, broadcast_buffers: bool = False,
                 overlap_grad_exchange: bool = True, exchange_single_rank: bool = False,
                 find_unused_parameters: bool = True):
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
        self.step_count = 0                       # optimiser steps taken by this trainer
        # torch.optim.Adamax keeps ``step`` PER PARAMETER and does not advance it for a parameter without a gradient
        self.param_steps: List[int] = [0] * len(self.params)
        self.group = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        # whether the step contains the exchange collectives.  ``exchange_single_rank`` keeps them in a one-rank group
        # (they sum over one rank): a rehearsal of the RCCL plumbing - streams, events, work handles - on a one-GPU box
        self.exchange = self.world > 1 or (exchange_single_rank and dist.is_available() and dist.is_initialized())
        # torch DDP's switch of the same name.  True: a parameter without a gradient on THIS rank still takes the other
        # ranks' (one small flag all-reduce per step, as DDP's used-parameter bitmap).  False (what the reference's
        # ``strategy: ddp`` means, config/config.yaml:35): every rank must produce every gradient in every step - no flag
        # exchange, and a rank that did not raises before it enters the gradient exchange, as DDP does.
        self.find_unused_parameters = find_unused_parameters
        self.slots: List[GradSlot] = []
        self.grad_views: List[torch.Tensor] = []
        # transposed copies [Cin][KH][KW][Cout] of every conv weight (operand of the data-gradient conv), refreshed
        # by ONE kernel after each optimiser step instead of one launch per layer inside the backward pass
        self.flat_wt = torch.zeros(total_padded, device=dev, dtype=torch.float32) if dev.type == "cuda" else None
        # pre-split images of both (snn_weight_presplit: fp16 hi/lo pieces of w * 2^8 for the forward convolutions, bf16
        # hi/lo pieces of w^T for the data gradients), refreshed with the transposes: the convolution kernels then read
        # ready-made pieces instead of converting the weight tile in every block
        self.flat_w16 = torch.empty(total_padded, device=dev, dtype=torch.float32) if dev.type == "cuda" else None
        self.flat_wt16 = (torch.empty(total_padded, device=dev, dtype=torch.float32)
                          if dev.type == "cuda" and _HF.USE_PRESPLIT_DGRAD else None)
        self._conv_params: List[torch.nn.Parameter] = []
        # weight images in MFMA-fragment order for the halo-resident 3x3 kernel (csrc/conv_halo.hip): forward image from
        # the OHWI weights (fp16 pieces), data-gradient image from the transposed weights with mirrored taps (bf16)
        self.flat_wfrag = self.flat_wtfrag = None
        frag_table = []
        table = []
        off = 0
        self._offset_of = {}   # id(parameter) -> float offset in the flat buffers
        for p in self.params:
            if p.dtype != torch.float32:
                raise RuntimeError("FlatTrainer: float32 parameters only")
            # an earlier trainer's cached images must not outlive it on the parameter (they would pass the version check)
            for name in ("_snn_wt", "_snn_w16", "_snn_wt16", "_snn_wfrag", "_snn_wtfrag", "_snn_grad_slot", "_snn_composed",
                         "_snn_compose_with", "_snn_sibling_group", "_snn_sibling_weight"):
                if hasattr(p, name):
                    delattr(p, name)
            p._snn_wt_version = -1
            self._offset_of[id(p)] = off
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
                if p.numel() >= 2 ** 31:
                    raise RuntimeError("FlatTrainer: a convolution weight with 2^31 or more elements (the batched transpose "
                                       "indexes a layer in 32 bits)")
                table.append([off, o, kh * kw, i])
                p._snn_wt = self.flat_wt[off:off + p.numel()].view(i, kh, kw, o)
                p._snn_wt_version = -1  # not valid yet
                # rows of both matrices must start on a 4-float group of the flat buffers
                # (tensor VIEWS, not addresses: the parameter keeps the image alive after the trainer is gone)
                if off % 4 == 0 and (kh * kw * i) % 4 == 0 and (kh * kw * o) % 4 == 0:
                    p._snn_w16 = self.flat_w16[off:off + p.numel()]
                    if self.flat_wt16 is not None:
                        p._snn_wt16 = self.flat_wt16[off:off + p.numel()]
                if (kh, kw) == (3, 3) and i % 32 == 0 and o % 32 == 0 and off % 4 == 0:
                    frag_table.append((p, off, o, i))
                self._conv_params.append(p)
            off += p.numel()
        self._wt_table = torch.tensor(table, dtype=torch.int64, device=dev) if table else None
        self._frag_tables = None
        if frag_table and self.flat_wt is not None:
            self.flat_wfrag = torch.empty(total_padded, device=dev, dtype=torch.float32)
            self.flat_wtfrag = torch.empty(total_padded, device=dev, dtype=torch.float32)
            fwd_rows = [[f_off, 4 * f_off, o, i] for _, f_off, o, i in frag_table]
            bwd_rows = [[f_off, 4 * f_off, i, o] for _, f_off, o, i in frag_table]   # w^T: [Cin][3][3][Cout]
            self._frag_tables = (torch.tensor(fwd_rows, dtype=torch.int64, device=dev),
                                 torch.tensor(bwd_rows, dtype=torch.int64, device=dev),
                                 max(9 * (i // 32) * (o // 32) * 128 for _, _, o, i in frag_table))
            self._frag_params = [p for p, _, _, _ in frag_table]
            for p, f_off, o, i in frag_table:
                p._snn_wfrag = self.flat_wfrag[f_off:f_off + p.numel()]
                p._snn_wtfrag = self.flat_wtfrag[f_off:f_off + p.numel()]
        self._offsets = [0]
        for p in self.params:
            self._offsets.append(self._offsets[-1] + p.numel())
        # BatchNorm buffers (running statistics, num_batches_tracked).  Without SyncBatchNorm every rank updates them
        # from its own shard; torch DDP (the reference's ``strategy: ddp``, config/config.yaml:35) broadcasts rank 0's
        # buffers to all ranks before every forward (``broadcast_buffers=True``), so a checkpoint written by any rank
        # holds the same statistics.  Training itself never reads them (train-mode BatchNorm uses batch statistics),
        # so the broadcast happens where it matters - ``sync_buffers()`` / ``checkpoint()`` before evaluating or
        # saving - instead of ~130 tiny launches in every step; ``broadcast_buffers=True`` restores DDP's cadence.
        self._float_buffers = [b for b in model.buffers() if b.is_floating_point()]
        self._int_buffers = [b for b in model.buffers() if not b.is_floating_point() and b.numel() == 1]
        self.broadcast_buffers = broadcast_buffers
        # gradient exchange overlapped with the backward pass: see ``overlap_from``
        self._early_lo: Optional[int] = None
        self._early_first: Optional[int] = None
        self._early_work = None
        self._comm_stream = None
        self._flags_work = None
        self._flags = None
        self._ones_flags = None
        if dev.type == "cuda" and _HF.USE_WGRAD_STREAM:
            # the weight-gradient stream is chosen NOW, with the device idle (its choice probes hardware-queue sharing)
            _HF._side_stream(dev)
        # an earlier trainer's hook must not outlive it on the model (a rebuilt trainer - checkpoint resume, another
        # ``overlap_grad_exchange`` setting, one rank - would inherit a stray all-reduce of the OLD trainer's buffers)
        if hasattr(model, "_snn_neck_grads_ready"):
            model._snn_neck_grads_ready = None
        self._grad_group = process_group   # the group the GRADIENT exchange runs on (see below)
        self.overlap_disabled_reason: Optional[str] = None
        if (overlap_grad_exchange and self.exchange and hasattr(model, "_snn_neck_grads_ready")
                and all(hasattr(model, a) for a in ("neck_net", "head_net"))):
            # a SODa detector: neck + head gradients go out while the backbone's backward pass still runs.  The hook
            # fires when the backward pass crosses the backbone / neck boundary; at that point only gradients written
            # through GradSlots are in ``flat_grad`` - a gradient that travels through autograd's ``p.grad`` is folded
            # in by ``step()``, AFTER the early exchange - so the overlap needs a slot on every early parameter.
            lo = self.overlap_from([model.neck_net, model.head_net])
            if not all(hasattr(p, "_snn_grad_slot") for p in self._early_params()):
                self._early_lo = None
                self.overlap_disabled_reason = ("parameters of the neck / head without a gradient slot "
                                                "(use_grad_slots=False): one all-reduce in step()")
            else:
                ref = weakref.ref(self)

                def _hook():
                    tr = ref()
                    if tr is not None:
                        tr.early_all_reduce()

                model._snn_neck_grads_ready = _hook
                if dev.type == "cuda":
                    self._comm_stream = _HF.concurrent_stream(torch.cuda.current_stream(dev),
                                                              avoid=tuple(_HF._SIDE_STREAMS.values()))
                    # ProcessGroupNCCL runs the collectives of ONE group on one internal stream: on the group that also
                    # carries the SyncBatchNorm exchanges, the backbone's first SyncBN backward all-reduce would queue
                    # behind the multi-MB tail exchange and the main stream would block on it.  The gradient exchange
                    # therefore gets a communicator of its own (every rank builds its trainer: a collective call).
                    if dist.get_backend(process_group) == "nccl":
                        self._grad_group = dist.new_group(ranks=dist.get_process_group_ranks(process_group)
                                                          if process_group is not None else None, backend="nccl")
                assert lo == self._early_lo
        self.refresh_transposed_weights()

    # ------------------------------------------------------------------ overlapped gradient exchange
    def overlap_from(self, modules: Iterable[torch.nn.Module]) -> int:
        """Arrange for the gradients of every parameter of ``modules`` to be all-reduced as soon as the backward pass
        has produced them (``early_all_reduce()``, called from a backward hook), while the rest of the backward pass
        still runs; ``step()`` then exchanges only what is left.  The parameters must form the TAIL of the flat buffer
        (for ``SODa``: neck + head = 93 % of TinyYolo, whose gradients are complete when the backward pass crosses the
        backbone / neck boundary).  Returns the first flat index of the early part.

        Assumes ONE backward pass per ``step()`` (the early part is summed over the ranks as soon as that pass has
        produced it; a second backward pass would add local gradients onto already reduced ones): accumulate
        micro-batches with ``FlatTrainer(..., overlap_grad_exchange=False)``."""
        ids = {id(p) for m in modules for p in m.parameters() if p.requires_grad}
        first = next((k for k, p in enumerate(self.params) if id(p) in ids), None)
        if first is None or any(id(p) not in ids for p in self.params[first:]) or len(ids) != len(self.params) - first:
            raise RuntimeError("overlap_from: the modules' parameters are not the tail of the trainer's parameter order")
        self._early_lo = self._offsets[first]
        self._early_first = first
        return self._early_lo

    def _early_params(self) -> List[torch.nn.Parameter]:
        return self.params[self._early_first:] if self._early_lo is not None else []

    def early_all_reduce(self) -> None:
        """Start the all-reduce of ``flat_grad[early_lo:]`` on a communication stream behind everything the main and
        the weight-gradient streams hold at this point.  No-op for a single rank or without ``overlap_from``.

        ONE backward pass per ``step()``: a second pass (micro-batch accumulation, several losses) would add local
        gradients onto a tail that is already summed over the ranks - possibly while the collective still runs - so the
        second call raises instead of returning silently wrong gradients."""
        if not self.exchange or self._early_lo is None:
            return
        if self._early_work is not None:
            raise RuntimeError("FlatTrainer: a second backward pass reached the backbone / neck boundary before step(): "
                               "the neck / head gradients of the first pass are already being summed over the ranks. "
                               "Accumulate micro-batches with FlatTrainer(..., overlap_grad_exchange=False)")
        for p in self._early_params():
            if p.grad is not None:
                raise RuntimeError("FlatTrainer: a neck / head parameter received its gradient through autograd's .grad "
                                   "instead of its gradient slot; the overlapped exchange would miss it. Build the "
                                   "trainer with overlap_grad_exchange=False")
        part = self.flat_grad[self._early_lo:]
        if part.is_cuda:
            if self._comm_stream is None:   # (normally taken in __init__, where probing it costs no pipeline drain)
                self._comm_stream = _HF.concurrent_stream(torch.cuda.current_stream(part.device),
                                                          avoid=tuple(_HF._SIDE_STREAMS.values()))
            comm = self._comm_stream
            comm.wait_stream(torch.cuda.current_stream())
            for st in list(_HF._SIDE_STREAMS.values()) + [a for ss in _HF._AUX_STREAMS.values() for a in ss]:
                comm.wait_stream(st)   # (weight gradients of the heads are written on their auxiliary streams)
            with torch.cuda.stream(comm):
                self._early_work = dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self._grad_group, async_op=True)
        else:
            self._early_work = dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self._grad_group, async_op=True)

    def _join_early(self) -> None:
        """Current stream waits for the early exchange (the handle stays: ``step()`` still has the head to reduce)."""
        if self._early_work is not None:
            self._early_work.wait()
            if self._comm_stream is not None:
                torch.cuda.current_stream().wait_stream(self._comm_stream)

    def refresh_transposed_weights(self) -> None:
        """Re-derive every ``p._snn_wt`` from the current weights (one launch).  ``_Conv2d.backward`` uses a cached
        transpose only while the parameter's version counter still matches, so a ``load_state_dict`` or any other
        torch-side in-place update simply falls back to the per-layer transpose until the next call."""
        if self._wt_table is None:
            return
        st = torch.cuda.current_stream().cuda_stream
        _hip.call("snn_weight_transpose_batched", self.flat_param.data_ptr(), self.flat_wt.data_ptr(),
                  self._wt_table.data_ptr(), len(self._conv_params), st)
        n = self.flat_param.numel()
        _hip.call("snn_weight_presplit", self.flat_param.data_ptr(), self.flat_w16.data_ptr(), n, _hip.PREC_FP16X3, st)
        if self.flat_wt16 is not None:
            _hip.call("snn_weight_presplit", self.flat_wt.data_ptr(), self.flat_wt16.data_ptr(), n, _hip.PREC_BF16X3, st)
        if self._frag_tables is not None:
            fwd_t, bwd_t, max_threads = self._frag_tables
            # forward image: fp16 pieces (fp16 x 3), or bf16 pieces when the activations are stored in bf16
            fwd_prec = _hip.PREC_BF16X3 if _HF.get_activation_storage() == "bf16" else _hip.PREC_FP16X3
            _hip.call("snn_weight_frag_image_batched", self.flat_param.data_ptr(), self.flat_wfrag.data_ptr(),
                      fwd_t.data_ptr(), fwd_t.shape[0], max_threads, 0, fwd_prec, st)
            for p in self._frag_params:
                p._snn_wfrag_prec = fwd_prec
            _hip.call("snn_weight_frag_image_batched", self.flat_wt.data_ptr(), self.flat_wtfrag.data_ptr(),
                      bwd_t.data_ptr(), bwd_t.shape[0], max_threads, 1, _hip.PREC_BF16X3, st)
        for p in self._conv_params:
            p._snn_wt_version = p._version
        self._refresh_composed(st)

    def _refresh_composed(self, st: int) -> None:
        """w2 w1 (and its transpose) of every composed 1x1 pair that registered itself during a forward pass
        (``functional._ComposedConv1x1``: ``w2._snn_compose_with = w1``) and the row-stacked ``[w2a; w2b] w1`` of every
        sibling group (``functional._SiblingConv1x1``: ``w2a._snn_sibling_group = (w1, (w2a, w2b))``), all products in
        ONE launch."""
        def plain_1x1(*ws):
            return all(id(w) in self._offset_of and w.dim() == 4 and tuple(w.shape[2:]) == (1, 1) for w in ws)

        pairs = [(p._snn_compose_with, p) for p in self._conv_params if getattr(p, "_snn_compose_with", None) is not None]
        pairs = [(w1, w2) for w1, w2 in pairs if plain_1x1(w1, w2) and w2.shape[1] == w1.shape[0]]
        groups = [p._snn_sibling_group for p in self._conv_params if getattr(p, "_snn_sibling_group", None) is not None]
        groups = [(w1, w2s) for w1, w2s in groups if plain_1x1(w1, *w2s) and all(w.shape[1] == w1.shape[0] for w in w2s)]
        if not pairs and not groups:
            return
        key = (tuple((id(w1), id(w2)) for w1, w2 in pairs), tuple((id(w1), tuple(id(w) for w in w2s)) for w1, w2s in groups))
        if getattr(self, "_composed_key", None) != key:
            rows, views, off = [], [], 0
            for w1, w2 in pairs:                       # {A = w2, B = w1, C, Ct, M, N, K, ldct}
                c2, c1, cin = w2.shape[0], w1.shape[0], w1.shape[1]
                rows.append([self._offset_of[id(w2)], self._offset_of[id(w1)], off, off, c2, cin, c1, 0])
                views.append(("pair", w1, w2, off, c2, cin))
                off += (c2 * cin + 3) // 4 * 4
            for w1, w2s in groups:                     # row blocks of ONE [sum c2][cin] matrix, column blocks of its transpose
                c1, cin = w1.shape[0], w1.shape[1]
                ct = sum(w.shape[0] for w in w2s)
                row0 = 0
                for w in w2s:
                    rows.append([self._offset_of[id(w)], self._offset_of[id(w1)], off + row0 * cin, off + row0, w.shape[0],
                                 cin, c1, ct])
                    row0 += w.shape[0]
                views.append(("group", w1, w2s, off, ct, cin))
                off += (ct * cin + 3) // 4 * 4
            dev = self.flat_param.device
            self._composed_table = torch.tensor(rows, dtype=torch.int64, device=dev)
            self._composed_c = torch.empty(off, device=dev, dtype=torch.float32)
            self._composed_ct = torch.empty(off, device=dev, dtype=torch.float32)
            self._composed_tiles = max(((r[4] + 31) // 32) * ((r[5] + 31) // 32) for r in rows)
            self._composed_views = views
            self._composed_key = key
        _hip.call("snn_small_gemm_batched", self.flat_param.data_ptr(), self.flat_param.data_ptr(),
                  self._composed_c.data_ptr(), self._composed_ct.data_ptr(), self._composed_table.data_ptr(),
                  int(self._composed_table.shape[0]), self._composed_tiles, st)
        for kind, w1, w2, off, rows_c, cin in self._composed_views:
            c = self._composed_c[off:off + rows_c * cin].view(rows_c, cin)
            ct = self._composed_ct[off:off + rows_c * cin].view(cin, rows_c)
            if kind == "pair":
                w2._snn_composed = (w1, (w1._version, w2._version), c, ct)
            else:
                w2[0]._snn_sibling_weight = (w1, w2, tuple(w._version for w in w2) + (w1._version,), c, ct)

    # ------------------------------------------------------------------
    def zero_grad(self) -> None:
        if self.flat_grad.is_cuda:
            wgrad_stream_sync()  # nothing may still be writing into the buffer that is about to be cleared
        if self._early_work is not None:   # a backward pass whose step() never came: let its collective finish first
            self._join_early()
            self._early_work = None
        _HF.reset_backward_state()
        self.flat_grad.zero_()
        for p, slot in zip(self.params, self.slots):
            slot.written = False
            p.grad = None

    def _collect_autograd_grads(self) -> None:
        """Parameters whose gradient came through autograd (ops without slot support) are folded in."""
        early = self._early_first if (self._early_work is not None and self._early_lo is not None) else len(self.params)
        for k, (p, slot, gview) in enumerate(zip(self.params, self.slots, self.grad_views)):
            if p.grad is not None:
                if k >= early:
                    raise RuntimeError("FlatTrainer.step: a neck / head parameter holds an autograd .grad while its slice "
                                       "of the flat gradient is being all-reduced (it would be added after - or during - "
                                       "the exchange); build the trainer with overlap_grad_exchange=False")
                if slot.written:
                    gview.add_(p.grad)
                else:
                    gview.copy_(p.grad)
                    slot.written = True
                p.grad = None

    def all_reduce(self) -> None:
        """SUM all-reduce of the flat gradient (averaging is folded into the Adamax kernel): ONE collective over the whole
        buffer, or - when ``early_all_reduce()`` already sent the tail during the backward pass - one over the head
        that was still being written then, joined with the early one."""
        if not self.exchange:
            return
        if self._early_work is not None:
            if self._early_lo > 0:
                dist.all_reduce(self.flat_grad[: self._early_lo], op=dist.ReduceOp.SUM, group=self._grad_group)
            self._join_early()                 # the current stream waits for the collective
            self._early_work = None
        else:
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self._grad_group)

    def _written_flags(self) -> List[bool]:
        """Which parameters received a gradient this step.  ``torch.optim.Adamax`` skips parameters whose ``.grad`` is
        None (their moments do not decay, the parameter does not move, its ``step`` does not advance).  With data
        parallelism a parameter counts as written when ANY rank wrote it (the all-reduce delivers the others'
        gradients).  EVERY rank enters the flag exchange in EVERY step (a rank-local condition in front of a collective
        hangs the ranks that took the other branch); only a rank that itself skipped a parameter reads the result."""
        written = [slot.written for slot in self.slots]
        if self.exchange and not self.find_unused_parameters:
            if not all(written):
                missing = [k for k, w in enumerate(written) if not w]
                raise RuntimeError(f"FlatTrainer(find_unused_parameters=False): {len(missing)} parameter(s) received no "
                                   f"gradient on this rank in this step (first: index {missing[0]}, shape "
                                   f"{tuple(self.params[missing[0]].shape)}); build the trainer with "
                                   "find_unused_parameters=True if parts of the model may stay unused")
            return written
        if self.exchange:
            if all(written):
                # the common case never touches the host: a device-side copy of a cached all-ones vector (building the
                # tensor from the Python list is a pageable host-to-device copy that stalls the launching thread until
                # the stream has drained - once per step, in front of the gradient exchange)
                if self._ones_flags is None:
                    self._ones_flags = torch.ones(len(self.slots), dtype=torch.int32, device=self.flat_grad.device)
                flags = self._ones_flags.clone()
            else:
                flags = torch.tensor(written, dtype=torch.int32, device=self.flat_grad.device)
            work = dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=self.group, async_op=True)
            if all(written):
                # whatever the other ranks did, the union is "all written": no host round trip.  The handle is kept
                # until the next step so the tensor outlives the collective.
                self._flags, self._flags_work = flags, work
            else:
                work.wait()
                written = [bool(f) for f in flags.tolist()]
        return written

    def _ranges(self, written: List[bool]) -> List[Tuple[int, int, int]]:
        """``(lo, hi, step)`` element ranges for the fused Adamax kernel: adjacent written parameters with the same
        (already advanced) step count coalesced; the common case is one launch over the padded buffer."""
        ranges: List[Tuple[int, int, int]] = []
        for k, w in enumerate(written):
            if not w:
                continue
            lo, hi, stp = self._offsets[k], self._offsets[k + 1], self.param_steps[k]
            if ranges and ranges[-1][1] == lo and ranges[-1][2] == stp:
                ranges[-1] = (ranges[-1][0], hi, stp)
            else:
                ranges.append((lo, hi, stp))
        if len(ranges) == 1 and ranges[0][:2] == (0, self.numel):
            ranges[0] = (0, self.flat_param.numel(), ranges[0][2])   # include the padding: whole 16-byte groups
        return ranges

    def _check_bindings(self) -> None:
        """``model.to(...)`` / ``model.float()`` / a manual ``p.data = ...`` would silently detach a parameter from
        the flat buffer (the optimiser would then update memory the model no longer reads)."""
        base = self.flat_param.data_ptr()
        for k, p in enumerate(self.params):
            if p.data_ptr() != base + 4 * self._offsets[k]:
                raise RuntimeError("FlatTrainer: a parameter no longer lives in the flat parameter buffer (the model "
                                   "was moved or re-cast after the trainer was built); build a new FlatTrainer")

    def sync_buffers(self, src: int = 0) -> None:
        """Rank ``src``'s BatchNorm buffers to every rank in one broadcast (torch DDP's ``broadcast_buffers``)."""
        if not self.exchange or not (self._float_buffers or self._int_buffers):
            return
        flat = torch.cat([b.detach().reshape(-1).float() for b in self._float_buffers]
                         + [b.detach().reshape(-1).float() for b in self._int_buffers])
        dist.broadcast(flat, src=src, group=self.group)
        off = 0
        with torch.no_grad():
            for b in self._float_buffers + self._int_buffers:
                n = b.numel()
                b.copy_(flat[off:off + n].view(b.shape).to(b.dtype))
                off += n

    def step(self) -> None:
        if not self.flat_param.is_cuda:
            raise RuntimeError("FlatTrainer.step: parameters are not on a HIP device; the optimiser kernel has no "
                               "CPU fallback")
        wgrad_stream_sync()  # weight-gradient kernels run on a side stream (functional._Conv2d.backward)
        self._check_bindings()
        self._collect_autograd_grads()
        if self._flags_work is not None:     # last step's flag exchange (never read): let it retire before the next one
            self._flags_work.wait()
            self._flags_work = self._flags = None
        written = self._written_flags()
        self.all_reduce()
        self.step_count += 1
        for k, w in enumerate(written):
            if w:
                self.param_steps[k] += 1
        st = torch.cuda.current_stream().cuda_stream
        for lo, hi, stp in self._ranges(written):
            _hip.call("snn_adamax_step", self.flat_param.data_ptr() + 4 * lo, self.flat_grad.data_ptr() + 4 * lo,
                      self.exp_avg.data_ptr() + 4 * lo, self.exp_inf.data_ptr() + 4 * lo, hi - lo, self.lr,
                      self.betas[0], self.betas[1], self.eps, stp, 1.0 / self.world, st)
        self.refresh_transposed_weights()
        if self.broadcast_buffers:
            self.sync_buffers()

    # ------------------------------------------------------------------ checkpoint / resume
    def checkpoint(self, model: torch.nn.Module) -> Dict:
        """``{"model": ..., "optimizer": ...}`` with the BatchNorm buffers made equal on all ranks first (rank 0's, as
        torch DDP leaves them) - what a reference Lightning checkpoint carries."""
        self.synchronize()
        self.sync_buffers()
        return {"model": {k: v.detach().clone() for k, v in model.state_dict().items()}, "optimizer": self.state_dict()}

    def state_dict(self) -> Dict:
        """Optimiser state in ``torch.optim.Adamax.state_dict()`` form: ``state[k] = {step, exp_avg, exp_inf}`` for the
        k-th trainable parameter that has received a gradient (tensors in the parameter's LOGICAL shape; ``step`` is kept
        per parameter, as torch does), one param group.  Interchangeable with a
        ``torch.optim.Adamax`` built over the same parameter order (the reference's Lightning checkpoints carry that
        state, ``models/soda.py:135-136``).  Weights and BatchNorm buffers are the model's own ``state_dict()``."""
        state = {}
        for k, p in enumerate(self.params):
            if self.param_steps[k] == 0:   # torch creates a parameter's state at its first gradient
                continue
            state[k] = {"step": torch.tensor(float(self.param_steps[k])),
                        "exp_avg": _storage_view(self.exp_avg, self._offsets[k], p.data).clone(),
                        "exp_inf": _storage_view(self.exp_inf, self._offsets[k], p.data).clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "foreach": None,
                 "maximize": False, "differentiable": False, "capturable": False,
                 "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd: Dict) -> None:
        group = sd["param_groups"][0]
        if len(group["params"]) != len(self.params):
            raise RuntimeError(f"optimizer state holds {len(group['params'])} parameters, the model has "
                               f"{len(self.params)}")
        if group.get("weight_decay", 0) or group.get("maximize", False):
            raise RuntimeError("FlatTrainer: weight_decay / maximize are not supported by the fused Adamax kernel")
        self.lr, self.betas, self.eps = group["lr"], tuple(group["betas"]), group["eps"]
        self.param_steps = [0] * len(self.params)
        self.exp_avg.zero_()
        self.exp_inf.zero_()
        for k, p in enumerate(self.params):
            st = sd["state"].get(k)
            if st is None:  # torch creates a parameter's state at its first gradient
                continue
            for name, flat in (("exp_avg", self.exp_avg), ("exp_inf", self.exp_inf)):
                t = st[name]
                if tuple(t.shape) != tuple(p.shape):
                    raise RuntimeError(f"optimizer state {name}[{k}] has shape {tuple(t.shape)}, parameter "
                                       f"{tuple(p.shape)}")
                _storage_view(flat, self._offsets[k], p.data).copy_(t)
            self.param_steps[k] = int(float(st["step"]))
        self.step_count = max(self.param_steps) if self.param_steps else 0

    # ------------------------------------------------------------------ helpers for tests / checkpoints
    def synchronize(self) -> None:
        """Join the weight-gradient side stream AND an early gradient exchange still in flight: after this the current
        stream may read (or clip, in place) ``flat_grad``.  With the overlapped exchange (N > 1, SODa) the neck / head
        part ``flat_grad[early_lo:]`` is then already SUMMED over the ranks while the backbone part is still local until
        ``step()``; ``FlatTrainer(overlap_grad_exchange=False)`` keeps everything local until ``step()``."""
        if self.flat_grad.is_cuda:
            wgrad_stream_sync()
        self._join_early()

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
    if trainer.exchange:
        dist.broadcast(trainer.flat_param, src=src, group=trainer.group)
        if trainer.flat_param.is_cuda:
            trainer.refresh_transposed_weights()  # the weights changed behind torch's version counters
