"""Shared helpers for the parity tests (oracle = checker, product = libsnn_hip.so path)."""
import torch

from oracle.net import SODaRef


def make_pair(model_cls, num_classes=2, seed=2, device="cuda", **kw):
    """Product model on `device` and the oracle on CPU, built from the SAME description, same weights."""
    torch.manual_seed(seed)
    product = model_cls(num_classes=num_classes, **kw)
    oracle = SODaRef(product, num_classes, loss_ratio=product.hparams.loss_ratio,
                     time_window=product.hparams.time_window, iou_threshold=product.hparams.iou_threshold)
    missing = oracle.load_state_dict(product.state_dict(), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return product.to(device), oracle


def synthetic_events(T, B, H, W, p=0.05, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(T, B, 2, H, W, generator=g) < p).float()


def synthetic_labels(B, n_boxes=2, n_classes=2, seed=1, pad_rows=0):
    g = torch.Generator().manual_seed(seed)
    out = torch.full((B, n_boxes + pad_rows, 5), -1.0)
    for b in range(B):
        for k in range(n_boxes):
            while True:
                xy = torch.rand(2, 2, generator=g)
                lo, hi = xy.min(0).values, xy.max(0).values
                if (hi - lo).prod() > 0.01:
                    break
            out[b, k, 0] = float(torch.randint(0, n_classes, (1,), generator=g))
            out[b, k, 1:3], out[b, k, 3:5] = lo, hi
    return out


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_rel(a: torch.Tensor, b: torch.Tensor, floor: float = 1e-6) -> float:
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float(((a - b).abs() / (b.abs().max() + floor)).max())


def executor_net(pkg):
    """The description of tests/golden/make_golden.py::executor_golden, written with THIS package's layer generators."""
    L = pkg

    class Net(pkg.SODa):
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
    return Net


def executor_block_cfg(L):
    return [L.Conv(6, 3), L.Norm(), L.Tanh(), L.Dense([[L.LSTM(5)], [L.Pass()], [L.Conv(3, 1), L.SiLU()]]),
            L.Residual([[L.Conv(kernel_size=1)], [L.LSTM()]])]
