"""Oracle restatement of the norse 1.1.0 LIF / LI cells (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED: norse is a third-party dependency of the reference
(``environment.yml:222`` pins norse==1.1.0) that is not installed and not on
disk, and the reference holds no tests/golden vectors for it.  The steps below
restate the published algorithm of

* ``norse/torch/functional/lif.py: lif_feed_forward_step`` and
  ``LIFParameters`` (tau_syn_inv=200, tau_mem_inv=100, v_leak=0, v_th=1,
  v_reset=0, method="super", alpha=100), ``norse/torch/module/lif.py: LIFCell``
  (initial state v=v_leak everywhere, i=0, dt=1e-3),
* ``norse/torch/functional/leaky_integrator.py: li_feed_forward_step`` /
  ``LICell`` (initial v = v_leak as a 0-dim tensor, i = zeros),
* ``norse/torch/functional/superspike.py: SuperSpike`` (forward Heaviside
  ``x > 0``; backward ``g / (alpha*|x| + 1)^2``),

anchored on the reference's call sites ``models/modules/layer_gen.py:232-235``
(``snn.LIFCell()`` with default parameters) and ``:252-254`` (``snn.LICell()``),
and on the in-tree witness ``models/modules/sli.py:110-126`` whose body is the LI
step (current jump first, then voltage, then current decay) with one extra
sigmoid factor, and ``sli.py:26-39,97-107`` for constants / initial state.

Every arithmetic statement keeps the operand order and the python-scalar /
0-dim-tensor typing of the original so that fp32 rounding is reproduced.
"""

from typing import NamedTuple, Optional, Tuple

import torch


class LIFParameters(NamedTuple):
    tau_syn_inv: torch.Tensor = torch.as_tensor(1.0 / 5e-3)
    tau_mem_inv: torch.Tensor = torch.as_tensor(1.0 / 1e-2)
    v_leak: torch.Tensor = torch.as_tensor(0.0)
    v_th: torch.Tensor = torch.as_tensor(1.0)
    v_reset: torch.Tensor = torch.as_tensor(0.0)
    method: str = "super"
    alpha: torch.Tensor = torch.as_tensor(100.0)


class LIParameters(NamedTuple):
    tau_syn_inv: torch.Tensor = torch.as_tensor(1.0 / 5e-3)
    tau_mem_inv: torch.Tensor = torch.as_tensor(1.0 / 1e-2)
    v_leak: torch.Tensor = torch.as_tensor(0.0)


class LIFFeedForwardState(NamedTuple):
    v: torch.Tensor
    i: torch.Tensor


class LIState(NamedTuple):
    v: torch.Tensor
    i: torch.Tensor


class _SuperSpike(torch.autograd.Function):
    """Heaviside forward, SuperSpike surrogate backward (Zenke & Ganguli 2018)."""

    @staticmethod
    def forward(ctx, u: torch.Tensor, alpha: torch.Tensor) -> torch.Tensor:
        ctx.save_for_backward(u, alpha)
        # norse heaviside: where(x <= 0, 0, 1)
        return torch.where(u <= torch.zeros_like(u), torch.zeros_like(u), torch.ones_like(u))

    @staticmethod
    def backward(ctx, g):
        u, alpha = ctx.saved_tensors
        return g / (alpha * torch.abs(u) + 1.0).pow(2), None


def superspike(u: torch.Tensor, alpha: torch.Tensor) -> torch.Tensor:
    return _SuperSpike.apply(u, alpha)


def lif_feed_forward_step(
    x: torch.Tensor,
    state: LIFFeedForwardState,
    p: LIFParameters = LIFParameters(),
    dt: float = 0.001,
) -> Tuple[torch.Tensor, LIFFeedForwardState]:
    # current jump
    i_new = state.i + x
    # membrane update
    dv = dt * p.tau_mem_inv * ((p.v_leak - state.v) + i_new)
    v_decayed = state.v + dv
    # synaptic current decay
    di = -dt * p.tau_syn_inv * i_new
    i_decayed = i_new + di
    # threshold + reset (reset term NOT detached -> gradient flows through z)
    z_new = superspike(v_decayed - p.v_th, p.alpha)
    v_new = (1 - z_new) * v_decayed + z_new * p.v_reset
    return z_new, LIFFeedForwardState(v=v_new, i=i_decayed)


def li_feed_forward_step(
    x: torch.Tensor,
    state: LIState,
    p: LIParameters = LIParameters(),
    dt: float = 0.001,
) -> Tuple[torch.Tensor, LIState]:
    i_jump = state.i + x
    dv = dt * p.tau_mem_inv * ((p.v_leak - state.v) + i_jump)
    v_new = state.v + dv
    di = -dt * p.tau_syn_inv * i_jump
    i_decayed = i_jump + di
    return v_new, LIState(v_new, i_decayed)


class LIFCell(torch.nn.Module):
    """``m(x, state=None) -> (z, state)`` for ONE timestep (norse SNNCell protocol)."""

    def __init__(self, p: LIFParameters = LIFParameters(), dt: float = 0.001):
        super().__init__()
        self.p, self.dt = p, dt

    def initial_state(self, x: torch.Tensor) -> LIFFeedForwardState:
        state = LIFFeedForwardState(
            v=torch.full(x.shape, self.p.v_leak.detach().item(), device=x.device, dtype=torch.float32),
            i=torch.zeros(*x.shape, device=x.device, dtype=torch.float32),
        )
        state.v.requires_grad = True
        return state

    def forward(self, x: torch.Tensor, state: Optional[LIFFeedForwardState] = None):
        state = state if state is not None else self.initial_state(x)
        return lif_feed_forward_step(x, state, self.p, self.dt)


class LICell(torch.nn.Module):
    def __init__(self, p: LIParameters = LIParameters(), dt: float = 0.001):
        super().__init__()
        self.p, self.dt = p, dt

    def initial_state(self, x: torch.Tensor) -> LIState:
        state = LIState(
            v=self.p.v_leak.detach().clone(),
            i=torch.zeros(*x.shape, device=x.device, dtype=torch.float32),
        )
        state.v.requires_grad = True
        return state

    def forward(self, x: torch.Tensor, state: Optional[LIState] = None):
        state = state if state is not None else self.initial_state(x)
        return li_feed_forward_step(x, state, self.p, self.dt)


def neuron_constants(dt: float = 0.001):
    """fp32 constants exactly as the torch expressions above round them.

    Returns ``(c_mem, c_syn, v_leak, v_th, v_reset, alpha)`` as python floats
    holding fp32 values; the HIP kernels receive these numbers as arguments.
    """
    p = LIFParameters()
    c_mem = (dt * p.tau_mem_inv).item()
    c_syn = (-dt * p.tau_syn_inv).item()  # negative
    return c_mem, c_syn, p.v_leak.item(), p.v_th.item(), p.v_reset.item(), p.alpha.item()
