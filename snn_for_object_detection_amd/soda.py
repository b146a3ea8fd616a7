"""Basic object detector class (mirror of the reference's ``models/soda.py``).

Same constructor, hooks and step logic; what differs:

* ``forward`` hands the WHOLE event sequence ``X[T,B,2,H,W]`` to the generated networks, which run
  layer-major on the gfx950 kernels (``generator.py`` here).  ``forward(X, time_outer=True)`` runs the
  reference's literal loop ``for ts in X`` (``soda.py:141-144``) on the same kernels with ``T = 1``;
  both give the same result and the second exists for parity tests and streaming use.
* Lightning / torchmetrics are not dependencies: the class is a plain ``nn.Module`` exposing the
  Lightning hook names (``training_step``, ``configure_optimizers`` ...) so a trainer loop or a
  LightningModule shim can drive it.  mAP evaluation (``soda.py:283-321``) is out of scope.
"""

from types import SimpleNamespace
from typing import Optional, Tuple

import torch
from torch import nn
from torch.nn import functional as F

from . import box
from .generator import BackboneGen, Head, ListGen, ListState, NeckGen
from .roi import RoI


class SODa(nn.Module):
    """Base detector; subclasses supply ``backbone_cfgs / neck_cfgs / head_cfgs`` (soda.py:98-133)."""

    def __init__(
        self,
        num_classes: int,
        loss_ratio: float = 0.04,
        time_window: int = 16,
        iou_threshold: float = 0.4,
        learning_rate: float = 0.001,
        state_storage: bool = False,
        init_weights: bool = True,
        plotter=None,
    ):
        super().__init__()
        self.hparams = SimpleNamespace(
            num_classes=num_classes, loss_ratio=loss_ratio, time_window=time_window,
            iou_threshold=iou_threshold, learning_rate=learning_rate, state_storage=state_storage,
            init_weights=init_weights,
        )
        self.plotter = plotter
        self.logged = {}
        self._sync_logged = set()
        # set by trainer.FlatTrainer.attach(): called in the backward pass when it crosses the backbone / neck boundary
        # (every neck and head gradient is complete or enqueued there) to start their all-reduce early
        self._snn_neck_grads_ready = None

        self.base_net = BackboneGen(self.backbone_cfgs, in_channels=2, init_weights=self.hparams.init_weights)
        self.neck_net = NeckGen(self.neck_cfgs, self.base_net.out_channels, init_weights=self.hparams.init_weights)
        self.head_net = Head(self.head_cfgs, self.hparams.num_classes, self.neck_net.out_shape,
                             init_weights=self.hparams.init_weights)
        self.roi_blk = RoI(self.hparams.iou_threshold)
        self.cls_loss = nn.CrossEntropyLoss(reduction="none")
        self.box_loss = nn.L1Loss(reduction="none")

    # ------------------------------------------------------------------ description hooks
    def backbone_cfgs(self) -> ListGen:
        raise NotImplementedError

    def neck_cfgs(self) -> ListGen:
        raise NotImplementedError

    def head_cfgs(self, box_out: int, cls_out: int) -> ListGen:
        raise NotImplementedError

    # ------------------------------------------------------------------ optimisation
    def configure_optimizers(self) -> torch.optim.Optimizer:
        return torch.optim.Adamax(self.parameters(), lr=self.hparams.learning_rate)

    def log(self, name, value, sync_dist: bool = False, **kwargs) -> None:
        """Lightning's ``self.log`` reduced to a dict.  ``sync_dist=True`` (every loss the reference logs,
        ``soda.py:151-157``) marks the entry for the cross-rank mean, which ``synced_logs()`` takes with ONE all-reduce
        over all marked entries when the values are actually read - not one collective per logged value per step."""
        self.logged[name] = value.detach() if isinstance(value, torch.Tensor) else value
        if sync_dist:
            self._sync_logged.add(name)

    def synced_logs(self, process_group=None) -> dict:
        """The logged values with the ``sync_dist`` entries averaged over the ranks (Lightning's reduction)."""
        import torch.distributed as dist
        out = dict(self.logged)
        names = sorted(n for n in self._sync_logged if isinstance(out.get(n), torch.Tensor))
        if names and dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1:
            flat = torch.stack([out[n].float().reshape(()) for n in names])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=process_group)
            flat = flat / dist.get_world_size(process_group)
            for k, n in enumerate(names):
                out[n] = flat[k]
        return out

    # ------------------------------------------------------------------ forward
    def forward(self, X: torch.Tensor, time_outer: bool = False) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """``X[T,B,2,H,W]`` -> ``(anchors[A,4], cls_preds[B,A,C+1], bbox_preds[B,A,4])`` of the last step."""
        if time_outer:
            state = None
            for ts in X:
                preds, state = self._forward_impl(ts, state)
            return preds
        preds, _ = self._forward_impl(X, None)
        return preds

    def _forward_impl(self, X: torch.Tensor, state: Optional[ListState]):
        from . import functional as HF
        state = [None] * 3 if state is None else state
        HF.begin_counter_batch()
        try:
            base_out, state[0] = self.base_net.forward(X, state[0])
            if self._snn_neck_grads_ready is not None and base_out.dim() == 5:
                base_out = HF.grad_ready_hook(base_out, self._snn_neck_grads_ready)
            neck_out, state[1] = self.neck_net.forward(base_out, state[1])
            anchors, cls_preds, bbox_preds, state[2] = self.head_net.forward(neck_out, state[2])
        finally:
            HF.flush_counter_batch()
        return (anchors, cls_preds, bbox_preds), state

    # ------------------------------------------------------------------ steps
    def _step(self, batch: Tuple[torch.Tensor, torch.Tensor]) -> torch.Tensor:
        X, labels = batch[0][self._rand_start_time():], batch[1]
        preds = self.forward(X)
        return self._loss(preds, labels)

    def training_step(self, batch: Tuple[torch.Tensor, torch.Tensor], batch_idx: int = 0) -> torch.Tensor:
        loss = self._step(batch)
        self.log("train_loss", loss, prog_bar=True, batch_size=batch[0].shape[1], sync_dist=True)
        return loss

    def validation_step(self, batch: Tuple[torch.Tensor, torch.Tensor], batch_idx: int = 0) -> torch.Tensor:
        loss = self._step(batch)
        self.log("val_loss", loss, batch_size=batch[0].shape[1], sync_dist=True)
        return loss

    def test_step(self, batch: Tuple[torch.Tensor, torch.Tensor], batch_idx: int = 0) -> torch.Tensor:
        loss = self._step(batch)
        self.log("test_loss", loss, batch_size=batch[0].shape[1], sync_dist=True)
        return loss

    def predict(self, X: torch.Tensor, state: Optional[ListState]) -> Tuple[torch.Tensor, ListState]:
        """Streaming inference for one event frame ``X[2,H,W]`` (soda.py:202-233).

        Returns rows ``(class id, confidence, x1, y1, x2, y2)`` and the new detector state.
        """
        preds, state = self._forward_impl(X.unsqueeze(0), state)
        anchors, cls, bbox = preds
        prep_pred = box.multibox_detection(F.softmax(cls, dim=2), bbox, anchors).squeeze(0)
        prep_pred = prep_pred[prep_pred[:, 0] >= 0]
        prep_pred[:, 2:] = torch.clamp(prep_pred[:, 2:], min=0.0, max=1.0)
        return prep_pred, state

    def _rand_start_time(self) -> int:
        # soda.py:246-257: drop a random prefix of the sequence; the SAME draw (``requires_grad=False``,
        # ``dtype=torch.uint32``: a seeded run consumes the generator exactly as the reference does)
        if not self.hparams.time_window:
            return 0
        return int(torch.randint(0, self.hparams.time_window, (1,), requires_grad=False, dtype=torch.uint32).item())

    def _loss(self, preds: Tuple[torch.Tensor, torch.Tensor, torch.Tensor], labels: torch.Tensor) -> torch.Tensor:
        # soda.py:259-281
        anchors, cls_preds, bbox_preds = preds
        bbox_offset, bbox_mask, class_labels = self.roi_blk(anchors, labels)
        if cls_preds.is_cuda:
            # one launch for the anchor targets (RoI above), two for the loss, one for its gradient (csrc/targets.hip)
            from . import functional as HF
            return HF.detection_loss(cls_preds, bbox_preds, bbox_offset, bbox_mask, class_labels,
                                     self.hparams.loss_ratio)
        _, _, num_classes = cls_preds.shape
        cls = self.cls_loss.forward(cls_preds.reshape(-1, num_classes), class_labels.reshape(-1))
        bbox = self.box_loss.forward(bbox_preds * bbox_mask, bbox_offset * bbox_mask)
        mask = class_labels.reshape(-1) > 0
        gt_loss = cls[mask].mean()
        background_loss = cls[~mask].mean()
        return (gt_loss * self.hparams.loss_ratio + background_loss * (1 - self.hparams.loss_ratio) + bbox.mean())

    def spike_taps(self):
        """``{module path: spikes[T,B,C,h,w]}`` of every ``StateStorage`` (``state_storage=True``, eval mode)."""
        from .layer_gen import StateStorage
        return {name: m.get_spikes() for name, m in self.named_modules()
                if isinstance(m, StateStorage) and m.spike_list}
