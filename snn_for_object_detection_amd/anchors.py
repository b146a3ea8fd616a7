"""Anchor boxes per feature-map pixel (mirror of the reference's ``utils/anchors.py``)."""

import torch
from torch import nn


class AnchorGenerator(nn.Module):
    """``AnchorGenerator(sizes, ratios)(feature_map) -> [anchor, 4]`` corner boxes, normalised.

    Same contract as ``utils/anchors.py:8-85``: ``len(sizes) * len(ratios)`` boxes centred on every
    pixel, ordered pixel-major then ratio-major then size; computed on the first call and cached on
    the module (the reference caches the same way, ``:41-44``).  Host-side set-up, not hot path.
    """

    def __init__(self, sizes: torch.Tensor, ratios: torch.Tensor, step: int = 1) -> None:
        super().__init__()
        self.step = step
        self.sizes = nn.Parameter(sizes, requires_grad=False)
        self.ratios = nn.Parameter(ratios, requires_grad=False)

    def __call__(self, X: torch.Tensor) -> torch.Tensor:
        if not hasattr(self, "anchors"):
            self._cal_anchors(X)
        return self.anchors

    def _cal_anchors(self, X: torch.Tensor) -> None:
        rows, cols = X.shape[-2:]
        # one-off host-side set-up: evaluated on the CPU (bit-identical on every device), then moved
        dev = torch.device("cpu")
        sizes, ratios = self.sizes.detach().cpu(), self.ratios.detach().cpu()
        per_pixel = len(sizes) * len(ratios)
        # pixel centres in normalised image coordinates
        cy = (torch.arange(rows, device=dev) + 0.5) * (1.0 / rows)
        cx = (torch.arange(cols, device=dev) + 0.5) * (1.0 / cols)
        gy, gx = torch.meshgrid(cy, cx, indexing="ij")
        gy, gx = gy.reshape(-1), gx.reshape(-1)
        # box extents for every (ratio, size) pair, corrected for the map's aspect
        bw = torch.cat([sizes * r for r in ratios]) * rows / cols
        bh = torch.cat([sizes / r for r in ratios]) * cols / rows
        corners = torch.stack((-bw, -bh, bw, bh)).T.repeat(rows * cols, 1) / 2
        centres = torch.stack([gx, gy, gx, gy], dim=1).repeat_interleave(per_pixel, dim=0)
        self.anchors = (centres + corners).to(X.device)
