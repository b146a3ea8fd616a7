"""Autograd operators over the HIP C ABI (``include/snn_hip.h``).

Every operator works on a whole layer-major sequence ``[T, B, C, H, W]`` (logical NCHW per
frame, channels-last memory ``[T][B][H][W][C]``), or on one timestep ``[B, C, H, W]`` which is
the ``T = 1`` case of the same kernels (streaming ``predict``, reference-style time-outer loops).

torch is used for device memory, the autograd tape and the current stream only; all arithmetic
of the hot path happens in ``libsnn_hip.so``.  No CPU / eager fallback exists: tensors that are
not on a HIP device raise.
"""

import collections
import ctypes
import os
from typing import List, NamedTuple, Optional, Sequence, Tuple

import torch
from torch.autograd import Function

from . import _hip
from ._hip import NeuronParams

# norse defaults as fp32 rounds them (dt*tau_mem_inv, -dt*tau_syn_inv, v_leak, v_th, v_reset, alpha);
# oracle/neurons.py:neuron_constants() evaluates the same torch expressions (tests pin the equality).
_F32 = torch.float32
DEFAULT_DT = 0.001


def neuron_params(dt: float = DEFAULT_DT) -> NeuronParams:
    tau_syn_inv = torch.as_tensor(1.0 / 5e-3)
    tau_mem_inv = torch.as_tensor(1.0 / 1e-2)
    # + SLI saturation potential (sli.py:38-39) and the synapse constants (synapse.py:26-36,77)
    return NeuronParams((dt * tau_mem_inv).item(), (-dt * tau_syn_inv).item(), 0.0, 1.0, 0.0, 100.0,
                        1.0, torch.as_tensor(1.0 / 1e-3).item(), torch.as_tensor(1.0 / 5e-3).item(), dt, 0.0)


# ------------------------------------------------------------------------------------------- precision
# The arithmetic of a convolution is an argument of every C-ABI call (include/snn_hip.h, SNN_PREC_*): the library keeps
# no arithmetic state.  What is kept HERE is the Python-side default that layers without their own setting use:
#   forward : "fp16x3" (default: two fp16 pieces per operand after exact power-of-two pre-scaling, three products;
#             fp32-grade for conv inputs |x| < 4094 and weights |w| < 255 - spikes and normalised activations),
#             "bf16x6" (three bf16 pieces, six products: fp32-grade for any range, half the speed),
#             "fp32"   (exact fp32 MFMA, an fmaf chain);
#   backward: "bf16x3" (default: bf16 hi + lo, hi*hi + hi*lo + lo*hi, relative error ~1e-5 of the exact product),
#             "fp32";
#   both    : "bf16"   - the opt-in THROUGHPUT mode: operands rounded once to bf16, ONE product per multiply-add, fp32
#             accumulation and fp32 tensors.  8 significant bits: not a parity mode and never a default (stated
#             tolerances: tests/test_gpu_bf16_mode.py).
# A layer overrides it with ``HipConv2d.forward_precision / .backward_precision`` (BlockGen sets "bf16x6" on
# convolutions fed by an unbounded activation - ReLU / SiLU / SumPool / ConvLSTM - where the fp16 range contract of
# "fp16x3" is not guaranteed by construction).
FORWARD_MODES = {"fp32": _hip.PREC_FP32, "bf16x6": _hip.PREC_BF16X6, "fp16x3": _hip.PREC_FP16X3,
                 "bf16": _hip.PREC_BF16X1}
BACKWARD_MODES = {"fp32": _hip.PREC_FP32, "bf16x3": _hip.PREC_BF16X3, "bf16": _hip.PREC_BF16X1}
DEFAULT_FORWARD_PRECISION = "fp16x3"
DEFAULT_BACKWARD_PRECISION = "bf16x3"


def _checked(mode: str, table: dict, what: str) -> str:
    if mode not in table:
        raise ValueError(f"{what} precision must be one of {sorted(table)}, got {mode!r}")
    return mode


# session defaults; SNN_FORWARD_PRECISION / SNN_BACKWARD_PRECISION preset them (validated here, not in the library)
_forward_precision = _checked(os.environ.get("SNN_FORWARD_PRECISION") or DEFAULT_FORWARD_PRECISION, FORWARD_MODES,
                              "SNN_FORWARD_PRECISION: forward")
_backward_precision = _checked(os.environ.get("SNN_BACKWARD_PRECISION") or DEFAULT_BACKWARD_PRECISION, BACKWARD_MODES,
                               "SNN_BACKWARD_PRECISION: backward")


def set_forward_precision(mode: str) -> None:
    """Default forward arithmetic of convolutions without their own ``forward_precision``."""
    global _forward_precision
    _forward_precision = _checked(mode, FORWARD_MODES, "forward")


def get_forward_precision() -> str:
    return _forward_precision


def set_backward_precision(mode: str) -> None:
    """Default arithmetic of the backward convolutions (data / weight gradients) without their own setting."""
    global _backward_precision
    _backward_precision = _checked(mode, BACKWARD_MODES, "backward")


def get_backward_precision() -> str:
    return _backward_precision


# ------------------------------------------------------------------------------------------- activation storage
# "fp32" (default, the parity path): every activation tensor is fp32 in HBM.
# "bf16": the opt-in bf16-STORAGE throughput mode (the dtype BASELINE configs[1] names).  The wide tensors of a layer-major
#   step - convolution outputs y, spikes / neuron outputs, saved decayed potentials, gradients gx / dy / dx - are bf16 in
#   HBM (half the bytes of every HBM-bound kernel); neuron state (v, i) stays fp32 in registers across T, BatchNorm
#   statistics / coefficients, weights, weight gradients and the optimiser stay fp32, accumulation is fp32 (spikes are
#   exact in bf16).  Convolutions multiply the stored bf16 activations by bf16-rounded weights: one MFMA product
#   (SNN_PREC_BF16S).  The event frames stay fp32 ({0,1}: two channels), and so does everything behind the head's
#   last-step read-out (a few frames).  Not a parity mode: 8 significant bits per stored value; stated tolerances in
#   tests/test_gpu_bf16_storage.py.  The mode is decided where the event frames enter (the Cin = 2 layer writes bf16) and
#   follows the tensors from there: an operator works in the storage type of its input.
_BF16 = torch.bfloat16
STORAGE_MODES = {"fp32": _F32, "bf16": _BF16}
_activation_storage = _checked(os.environ.get("SNN_ACTIVATION_STORAGE") or "fp32", STORAGE_MODES,
                               "SNN_ACTIVATION_STORAGE: storage")


def set_activation_storage(mode: str) -> None:
    """"fp32" (default) or "bf16" (opt-in throughput mode, see above)."""
    global _activation_storage
    if mode not in STORAGE_MODES:
        raise ValueError(f"activation storage must be one of {sorted(STORAGE_MODES)}, got {mode!r}")
    _activation_storage = mode


def get_activation_storage() -> str:
    return _activation_storage


def _prec_codes(forward: Optional[str], backward: Optional[str]) -> Tuple[int, int]:
    """Per-call ``precision`` arguments: the layer's own modes, else the session defaults."""
    return (FORWARD_MODES[_checked(forward or _forward_precision, FORWARD_MODES, "forward")],
            BACKWARD_MODES[_checked(backward or _backward_precision, BACKWARD_MODES, "backward")])


# ------------------------------------------------------------------------------------------- helpers
def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


# Weight gradients do not feed the rest of the backward pass, so they run on a second HIP stream beside the
# data-gradient chain (fills the CUs that small feature maps leave idle).  ``wgrad_stream_sync()`` joins the
# side stream; trainer.FlatTrainer.step() calls it before the all-reduce / optimiser read the gradients.
_SIDE_STREAMS = {}
USE_WGRAD_STREAM = not os.environ.get("SNN_NO_WGRAD_STREAM")   # tuning aid: everything on one stream


# HIP multiplexes its streams onto a few hardware queues (four by default), in the order the streams are first used.  Two
# streams that land on ONE queue run strictly one after the other: with an RCCL communicator created before the model
# (the normal order under torchrun) the weight-gradient stream shared the main stream's queue and the step lost the whole
# overlap - 25.9 instead of 24.7 ms, measured with a one-rank RCCL group (tools/dist_overhead.py).  So a side stream is
# not taken blindly from torch's pool: candidates are PROBED against the stream they are meant to run beside.
STREAM_PROBE_CANDIDATES = 12
_STREAM_PROBE_LOG = []   # (device, candidates tried, concurrent?) per chosen stream - bench.py reports it


def runs_concurrently(candidate: "torch.cuda.Stream", beside: "torch.cuda.Stream") -> bool:
    """True if work queued on ``candidate`` AFTER a long-running job on ``beside`` finishes before that job does (the two
    do not share a hardware queue).  Synchronises the device; about 2 ms."""
    dev = beside.device
    torch.cuda.synchronize(dev)
    big = torch.empty(64 << 20, dtype=torch.float32, device=dev)      # 256 MiB: eight fills are about a millisecond
    small = torch.empty(64, dtype=torch.float32, device=dev)
    e0, e1, c1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    with torch.cuda.stream(beside):
        big.fill_(0.0)                                                # first-touch cost stays outside the timed part
        e0.record(beside)
        for _ in range(8):
            big.fill_(1.0)
        e1.record(beside)
    with torch.cuda.stream(candidate):
        small.fill_(1.0)
        c1.record(candidate)
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(c1) < 0.5 * e0.elapsed_time(e1)


def concurrent_stream(beside: "torch.cuda.Stream", avoid=()) -> "torch.cuda.Stream":
    """A stream of ``beside``'s device that really runs beside it (and is none of ``avoid``): the first probed candidate
    from torch's pool that does; the first candidate if none does (the result is still correct, only serial)."""
    dev = beside.device
    tried = []
    for _ in range(STREAM_PROBE_CANDIDATES):
        cand = torch.cuda.Stream(device=dev)
        if any(cand.cuda_stream == o.cuda_stream for o in tuple(avoid) + tuple(tried) + (beside,)):
            continue
        tried.append(cand)
        if runs_concurrently(cand, beside):
            _STREAM_PROBE_LOG.append((str(dev), len(tried), True))
            return cand
    _STREAM_PROBE_LOG.append((str(dev), len(tried), False))
    return tried[0] if tried else torch.cuda.Stream(device=dev)


def _side_stream(device) -> "torch.cuda.Stream":
    st = _SIDE_STREAMS.get(device)
    if st is None:
        st = _SIDE_STREAMS[device] = concurrent_stream(torch.cuda.current_stream(device))
    return st


# Independent sub-networks (the detection heads behind the ``Return`` taps, generator.Head) CAN run on auxiliary streams
# beside the main stream: the 30x38 head then overlaps the latency-bound 15x19 / 8x10 neck stages, forward and - because
# autograd runs a node's backward on the stream of its forward - backward.  Built, bit-identical (tests/test_gpu_model.py)
# and measured in round 4: NOT faster - same-call A/B, three rounds: 23.37 / 23.15 / 23.24 ms with one head stream
# against 23.11 / 23.01 / 22.95 without, 22.97 with two (tools/ab.sh): what runs beside a kernel that fills the CUs only
# takes its share of them, and the cross-stream events cost more than the launch tails they fill.  So: OFF unless
# SNN_HEAD_STREAMS=<n> asks for n streams.
HEAD_STREAMS = int(os.environ.get("SNN_HEAD_STREAMS", "0") or 0)   # heads (largest maps first) with a stream of their own
USE_HEAD_STREAMS = HEAD_STREAMS > 0
_AUX_STREAMS = {}   # device -> [streams], probed like the weight-gradient stream
_AUX_HOME = {}      # device -> the stream the auxiliary streams run beside (the "main" stream of the step)


def aux_streams(device, n: int) -> List["torch.cuda.Stream"]:
    """``n`` streams that really run beside the current stream (and beside the weight-gradient stream)."""
    have = _AUX_STREAMS.setdefault(device, [])
    if len(have) < n:
        main = torch.cuda.current_stream(device)
        _AUX_HOME[device] = main
        while len(have) < n:
            have.append(concurrent_stream(main, avoid=tuple(_SIDE_STREAMS.values()) + tuple(have)))
    return have[:n]


def on_aux_stream(device=None) -> bool:
    """True while the current stream is one of the auxiliary streams (weight gradients then run inline on it: the
    side-stream bookkeeping below assumes ONE main stream)."""
    cur = torch.cuda.current_stream(device)
    return any(cur.cuda_stream == s.cuda_stream for ss in _AUX_STREAMS.values() for s in ss)


def aux_streams_sync() -> None:
    """Make the current stream wait for everything queued on the auxiliary streams (weight gradients of the heads are
    written there; autograd joins only the streams of gradient-accumulation leaves)."""
    for dev, ss in _AUX_STREAMS.items():
        cur = torch.cuda.current_stream(dev)
        for s in ss:
            if s.cuda_stream != cur.cuda_stream:
                cur.wait_stream(s)


# Operands of a side-stream kernel are kept alive HERE until the main stream has waited for that kernel, instead of
# being handed to ``Tensor.record_stream``: a record_stream'ed block returns to the caching allocator only once the
# GPU has really passed the event, and the host enqueues a whole backward pass ahead of the GPU - so every dy of the
# pass stayed reserved at once (257 GiB reserved for 113 GiB live on the 1280x720 B=4 step, and an allocator that
# frees and re-mallocs its cache each step from B=6 on).  Released this way the blocks are recycled in stream order.
_SIDE_PENDING = collections.deque()   # (event recorded on the side stream, tensors its kernels read)
# weight-gradient launches the side stream may lag behind the main stream before the main stream waits for the oldest
WGRAD_SIDE_DEPTH = int(os.environ.get("SNN_WGRAD_SIDE_DEPTH", "1"))


def _side_retire(main, keep: int) -> None:
    """Main stream waits for all but the ``keep`` most recent side-stream launches and drops their operands."""
    while len(_SIDE_PENDING) > keep:
        ev, _tensors = _SIDE_PENDING.popleft()
        main.wait_event(ev)


def _side_hold(side, *tensors) -> None:
    ev = torch.cuda.Event()
    ev.record(side)
    _SIDE_PENDING.append((ev, tensors))


def wgrad_stream_sync() -> None:
    """Make the current stream wait for every weight-gradient kernel queued on the side stream (and on the auxiliary
    streams of the detection heads)."""
    _SIDE_PENDING.clear()
    for dev, st in _SIDE_STREAMS.items():
        torch.cuda.current_stream(dev).wait_stream(st)
    aux_streams_sync()


def _wgrad_on_side() -> bool:
    return USE_WGRAD_STREAM and not (_AUX_STREAMS and on_aux_stream())


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _require_device(t: torch.Tensor, what: str, bf16_ok: bool = False) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{what}: tensor is on {t.device}; the MI355X path has no CPU fallback "
                           "(move the model and its inputs to a HIP device)")
    if t.dtype == _BF16 and not bf16_ok:
        raise RuntimeError(f"{what}: this operator has no bf16-storage form (convolutions, Norm + LIF / LI / LI+Tanh / "
                           "none, and the Dense / Residual merges do); run the model with activation storage \"fp32\"")
    if t.dtype != _F32 and t.dtype != _BF16:
        raise RuntimeError(f"{what}: expected float32, got {t.dtype}")


def cl_stride(x: torch.Tensor) -> Optional[int]:
    """Pixel stride ``ld`` when the logical ``[..., C, H, W]`` tensor is stored frame-contiguously as
    ``[..., H, W, ld >= C]`` with channel stride 1 (dense channels-last, or a channel slice of a wider
    channels-last buffer); ``None`` otherwise."""
    if x.dim() < 3:
        return None
    *lead, C, H, W = x.shape
    st = x.stride()
    sc, sh, sw = st[-3], st[-2], st[-1]
    if C > 1 and sc != 1:
        return None
    if W > 1:
        ld = sw
    elif H > 1:
        ld = sh
    else:
        ld = C
        for d in range(len(lead) - 1, -1, -1):
            if lead[d] > 1:
                ld = st[d]
                break
    if ld < C or (W > 1 and sw != ld) or (H > 1 and sh != W * ld):
        return None
    expect = H * W * ld
    for d in range(len(lead) - 1, -1, -1):
        if lead[d] > 1 and st[d] != expect:
            return None
        expect *= lead[d]
    return ld


def is_channels_last(x: torch.Tensor) -> bool:
    """True when the logical ``[..., C, H, W]`` tensor is stored DENSELY as ``[..., H, W, C]``."""
    nd = x.dim()
    perm = list(range(nd - 3)) + [nd - 2, nd - 1, nd - 3]
    return x.permute(perm).is_contiguous()


def _cl_view(buf: torch.Tensor) -> torch.Tensor:
    """dense ``[..., H, W, C]`` buffer -> logical ``[..., C, H, W]`` view."""
    nd = buf.dim()
    return buf.permute(list(range(nd - 3)) + [nd - 1, nd - 3, nd - 2])


def _new_cl(lead: Sequence[int], C: int, H: int, W: int, like: torch.Tensor, dtype=None) -> torch.Tensor:
    """Dense channels-last tensor on ``like``'s device, in ``like``'s storage type unless ``dtype`` says otherwise."""
    return _cl_view(torch.empty((*lead, H, W, C), device=like.device, dtype=dtype or like.dtype))


def _alias(root: torch.Tensor, storage_offset: int, T: int, B: int, C: int, H: int, W: int, ld: int) -> torch.Tensor:
    """A fresh tensor (no autograd view relation) over ``root``'s storage: logical ``[T,B,C,H,W]``,
    channels-last with pixel stride ``ld``.  Only the HIP kernels write through such aliases."""
    t = torch.empty(0, device=root.device, dtype=root.dtype)
    t.set_(root.untyped_storage(), storage_offset, (T, B, C, H, W), (B * H * W * ld, H * W * ld, 1, W * ld, ld))
    return t


class ConcatPromise:
    """A lazily allocated channels-last buffer ``[T,B,H,W,total_c]`` that several producers fill slice by
    slice (the Dense merge of generator.py:157-161 without the copy).  A promise nested in a parent
    destination is itself a channel slice of the parent's buffer."""

    def __init__(self, total_c: int, parent: "Optional[Dest]" = None):
        self.total_c, self.parent = total_c, parent
        self.buf: Optional[torch.Tensor] = None

    def get(self, T: int, B: int, H: int, W: int, like: torch.Tensor, dtype=None) -> torch.Tensor:
        if self.buf is None:
            if self.parent is not None:
                self.buf = self.parent.tensor(T, B, self.total_c, H, W, like, dtype)
            else:
                self.buf = _new_cl((T, B), self.total_c, H, W, like, dtype)
        if self.buf.dtype != (dtype or like.dtype):
            raise RuntimeError("Dense merge: branches store their outputs in different types (fp32 / bf16 storage mixed)")
        if tuple(self.buf.shape) != (T, B, self.total_c, H, W):
            raise RuntimeError(f"Dense merge: branch outputs differ in shape: {tuple(self.buf.shape)} vs "
                               f"{(T, B, self.total_c, H, W)}")
        return self.buf


class Dest:
    """Where an operator must put its ``[T,B,c,H,W]`` result: channels ``off .. off+c`` of a promise."""

    def __init__(self, promise: ConcatPromise, off: int, c: int):
        self.promise, self.off, self.c = promise, off, c

    def tensor(self, T: int, B: int, C: int, H: int, W: int, like: torch.Tensor, dtype=None) -> torch.Tensor:
        if C != self.c:
            raise RuntimeError(f"destination slice holds {self.c} channels, operator produces {C}")
        buf = self.promise.get(T, B, H, W, like, dtype)
        return _alias(buf, buf.storage_offset() + self.off, T, B, C, H, W, cl_stride(buf))

    def holds(self, x: torch.Tensor) -> bool:
        """True when ``x`` already IS this destination (same storage, offset and pixel stride)."""
        buf = self.promise.buf
        if buf is None or x.dim() != 5 or x.shape[2] != self.c:
            return False
        return (x.untyped_storage().data_ptr() == buf.untyped_storage().data_ptr()
                and x.storage_offset() == buf.storage_offset() + self.off and cl_stride(x) == cl_stride(buf)
                and tuple(x.shape[:2]) == tuple(buf.shape[:2]) and tuple(x.shape[3:]) == tuple(buf.shape[3:]))


def _out_tensor(dest: Optional[Dest], T: int, B: int, C: int, H: int, W: int, like: torch.Tensor, dtype=None) -> torch.Tensor:
    return dest.tensor(T, B, C, H, W, like, dtype) if dest is not None else _new_cl((T, B), C, H, W, like, dtype)


def _raw_to_cl(x: torch.Tensor) -> torch.Tensor:
    """Non-differentiable layout change to (possibly channel-sliced) channels-last memory."""
    if cl_stride(x) is not None:
        return x
    _require_device(x, "layout", bf16_ok=True)
    if x.dtype == _BF16:   # (rare: bf16 tensors are born channels-last) torch's strided copy
        out = _new_cl(x.shape[:-3], *x.shape[-3:], x)
        out.copy_(x)
        return out
    xc = x.contiguous()
    lead, (C, H, W) = xc.shape[:-3], xc.shape[-3:]
    n = 1
    for d in lead:
        n *= d
    out = _new_cl(lead, C, H, W, xc)
    _hip.call("snn_nchw_to_nhwc", xc.data_ptr(), out.data_ptr(), n, C, H, W, _stream())
    return out


def _raw_dense_cl(x: torch.Tensor) -> torch.Tensor:
    """Dense channels-last copy of a channel-sliced tensor (operators without a pixel-stride argument)."""
    x = _raw_to_cl(x)
    if is_channels_last(x):
        return x
    lead, (C, H, W) = x.shape[:-3], x.shape[-3:]
    n = 1
    for d in lead:
        n *= d
    out = _new_cl(lead, C, H, W, x)
    _hip.call("snn_copy_channels_bf16" if x.dtype == _BF16 else "snn_copy_channels", x.data_ptr(), cl_stride(x),
              out.data_ptr(), C, n * H * W, C, _stream())
    return out


def _raw_to_nchw(x: torch.Tensor) -> torch.Tensor:
    if x.is_contiguous():
        return x
    if x.dtype == _BF16:
        return x.contiguous()
    x = _raw_dense_cl(x)
    lead, (C, H, W) = x.shape[:-3], x.shape[-3:]
    n = 1
    for d in lead:
        n *= d
    out = torch.empty(x.shape, device=x.device, dtype=_F32)
    _hip.call("snn_nhwc_to_nchw", x.data_ptr(), out.data_ptr(), n, C, H, W, _stream())
    return out


class _ToChannelsLast(Function):
    @staticmethod
    def forward(ctx, x):
        return _raw_to_cl(x)

    @staticmethod
    def backward(ctx, g):
        return _raw_to_nchw(g)


def to_channels_last(x: torch.Tensor) -> torch.Tensor:
    """Differentiable entry adapter for callers holding NCHW-contiguous frames (soda.py:138-144)."""
    if cl_stride(x) is not None:
        return x
    return _ToChannelsLast.apply(x)


def as_sequence(x: torch.Tensor) -> Tuple[torch.Tensor, bool]:
    """``[B,C,H,W]`` or ``[T,B,C,H,W]`` -> channels-last ``[T,B,C,H,W]`` and "was a single step"."""
    if x.dim() == 4:
        return to_channels_last(x).unsqueeze(0), True
    if x.dim() == 5:
        return to_channels_last(x), False
    raise RuntimeError(f"expected [B,C,H,W] or [T,B,C,H,W], got shape {tuple(x.shape)}")


def _dims5(x: torch.Tensor):
    T, B, C, H, W = x.shape
    return T, B, C, H, W


# ------------------------------------------------------------------------------------------- gradient slots
class GradSlot:
    """Destination of a parameter's gradient inside a flat gradient buffer (``trainer.FlatTrainer``).

    When a parameter carries ``_snn_grad_slot`` the backward kernels write (first use in a step) or
    accumulate (later uses) its gradient straight into ``buf`` - the buffer the data-parallel all-reduce
    and the fused Adamax step work on - and autograd receives no gradient for it (zero copies).
    ``buf`` is dense in the parameter's STORAGE order (OHWI for conv weights).
    """

    __slots__ = ("buf", "written")

    def __init__(self, buf: torch.Tensor):
        self.buf = buf
        self.written = False

    def claim(self) -> int:
        """-> the ``accumulate`` flag for the kernel and mark the slot as holding data."""
        acc = 1 if self.written else 0
        self.written = True
        return acc


_NBT_BATCH = None  # while a model-level forward runs: deferred ``num_batches_tracked`` increments (see soda.py)


def begin_counter_batch() -> None:
    global _NBT_BATCH
    _NBT_BATCH = []


def flush_counter_batch() -> None:
    """Apply the deferred BatchNorm ``num_batches_tracked`` increments in one multi-tensor launch (22 BatchNorms
    = 22 one-element kernels per step otherwise)."""
    global _NBT_BATCH
    pending, _NBT_BATCH = _NBT_BATCH, None
    if pending:
        by_inc = {}
        for t, inc in pending:
            by_inc.setdefault(inc, []).append(t)
        for inc, tensors in by_inc.items():
            torch._foreach_add_(tensors, inc)


FUSE_OUTER_ADDEND = not os.environ.get("SNN_NO_OUTER_ADDEND")  # bisecting aid
# sibling 1x1 convolutions of one input (the C2f split) as ONE convolution (generator.BlockGen._plan_siblings)
USE_SIBLING_FUSION = not os.environ.get("SNN_NO_SIBLING_FUSION")
# a Norm -> LIF layer whose only consumer is such a convolution writes NO spike tensor: the consumer thresholds the saved
# potentials itself (snn_conv1x1_spikes_*; SNN_NO_SPIKES_FROM_VDEC: tuning / bisecting aid)
USE_SPIKES_FROM_VDEC = not os.environ.get("SNN_NO_SPIKES_FROM_VDEC")


class GradAccumulator:
    """Shared by the aliases a block hands to its branches (``fanout``): lets the FIRST data-gradient
    convolution that runs for the fanned-out tensor add the gradients other branches have already produced
    in its epilogue (``snn_conv2d_dgrad(addend=...)``) instead of a separate add pass afterwards."""

    __slots__ = ("deposits", "result", "fused", "outer", "exclusive", "grad_in_slot", "marks")

    def __init__(self):
        self.marks = {}       # key (or "result") -> (stream, event): where and when that tensor was produced - consulted
        #                       only while auxiliary streams exist (the consumers of a Return tap run on different streams,
        #                       and these tensors travel behind autograd's back)
        self.deposits = {}    # alias index -> gradient produced for that alias by a pass-through consumer
        self.result = None    # dx written by the fusing convolution
        self.fused = {}       # alias index -> deposit that went into ``result``
        self.outer = None     # (accumulator, alias index) of the enclosing fanout when the fanned-out tensor is
        #                       itself an alias (Residual inside Dense: the YOLO bottleneck)
        self.exclusive = set()    # keys whose deposit nobody else reads (a channel slice of a concat gradient handed to
        #                           exactly this consumer): the accumulated gradient may be written over it
        self.grad_in_slot = False  # the fanned-out tensor wants its total gradient left IN such a slice (split_channels)

    def deposit(self, key, g: torch.Tensor, exclusive: bool = False) -> None:
        if self.result is None and g is not None:
            self.deposits[key] = g
            if exclusive:
                self.exclusive.add(key)
            if _AUX_STREAMS:
                self.marks[key] = _stream_mark()

    def set_result(self, dx: torch.Tensor) -> None:
        self.result = dx
        if _AUX_STREAMS:
            self.marks["result"] = _stream_mark()


def _stream_mark():
    cur = torch.cuda.current_stream()
    ev = torch.cuda.Event()
    ev.record(cur)
    return cur, ev


def _await_mark(acc: "GradAccumulator", key, t: Optional[torch.Tensor]) -> None:
    """The current stream waits for the producer of a tensor handed over through a GradAccumulator on another stream."""
    mark = acc.marks.get(key) if _AUX_STREAMS else None
    if mark is None:
        return
    cur = torch.cuda.current_stream()
    if mark[0].cuda_stream != cur.cuda_stream:
        cur.wait_event(mark[1])
        if t is not None:
            t.record_stream(cur)   # allocated on the producer's stream, read by a kernel of this one


def _acc_of(x):
    return getattr(x, "_snn_acc", None)


def _same_tensor(a: torch.Tensor, b: torch.Tensor) -> bool:
    return (a.data_ptr() == b.data_ptr() and tuple(a.shape) == tuple(b.shape) and a.stride() == b.stride())


def _slot_of(param) -> Optional[GradSlot]:
    return getattr(param, "_snn_grad_slot", None) if param is not None else None


# ------------------------------------------------------------------------------------------- conv
def _dgrad_accumulate(acc, gy, ldg, wt, x, geom, st, prec, wt_split=None, wt_image=None, pend=None):
    """Data gradient of a convolution with the gradient accumulation of its input folded into the epilogue
    (up to two addends; see ``GradAccumulator``).  ``wt`` is the transposed weight ``[Cin][KH][KW][Cout]``."""
    T, B, Cin, H, W, Cout, KH, KW, Ho, Wo, stride, pad = geom
    dx, dx_shape = None, (T, B, Cin, H, W)
    addend, ld_add, addend2, ld_add2 = None, 0, None, 0
    outer_fused = None
    if acc is not None and acc[0].result is None and acc[0].deposits:
        # another branch of the block already produced its gradient for this tensor: add it here
        key, other = next(iter(acc[0].deposits.items()))
        _await_mark(acc[0], key, other)
        other = _raw_to_cl(other)
        if tuple(other.shape) == dx_shape:
            addend, ld_add = other.data_ptr(), cl_stride(other)
            acc[0].fused[key] = acc[0].deposits.pop(key)
    if (acc is not None and acc[0].result is None and not acc[0].deposits and acc[0].outer is not None
            and FUSE_OUTER_ADDEND):
        # the tensor is itself one alias of an enclosing fanout whose other consumer already deposited
        # its gradient (bottleneck input: conv + residual shortcut + Dense pass-through): second addend
        o_acc, o_key = acc[0].outer
        if o_acc.result is None and len(o_acc.deposits) == 1 and o_key not in o_acc.deposits:
            key2, other2 = next(iter(o_acc.deposits.items()))
            _await_mark(o_acc, key2, other2)
            other2 = _raw_to_cl(other2)
            if tuple(other2.shape) == dx_shape:
                addend2, ld_add2 = other2.data_ptr(), cl_stride(other2)
                outer_fused = (o_acc, o_key, key2)
                if (o_acc.grad_in_slot and key2 in o_acc.exclusive and other2.dtype == x.dtype and ld_add2 % 4 == 0
                        and USE_SIBLING_FUSION):
                    # the total is written over the concat-gradient slice it contains (every kernel's epilogue reads an
                    # addend element and stores the sum from the same lane): the gradient of a split_channels part then
                    # lies next to its siblings' and the fused convolution reads them as ONE channel-sliced tensor
                    dx = _alias(other2, other2.storage_offset(), T, B, Cin, H, W, ld_add2)
    chained = False
    if (acc is not None and acc[0].result is not None and addend is None and acc[1] not in acc[0].fused
            and tuple(acc[0].result.shape) == dx_shape):
        # a sibling convolution already produced (its gradient + the fused deposits) for this tensor: add
        # that here and become the accumulated result (two convolutions on one input: the C2f split)
        prev = acc[0].result
        _await_mark(acc[0], "result", prev)
        addend, ld_add = prev.data_ptr(), cl_stride(prev)
        chained = True
    if dx is None:
        dx = _new_cl((T, B), Cin, H, W, x)
    if pend is not None:
        # (gy is gx here) dy = A*gx + B*y + C inside the halo-resident kernel, stored to pend[1] for the weight gradient
        rec, dy_out = pend
        if wt_image is None:
            wt_image = _frag_image(wt, Cin, Cout, 1, _hip.PREC_BF16X3)
        _hip.call("snn_conv3x3_halo_bn", rec.gx.data_ptr(), rec.y.data_ptr(), rec.coef.data_ptr(), B, dy_out.data_ptr(),
                  wt_image.data_ptr(), dx.data_ptr(), cl_stride(dx), T * B, H, W, Cout, Cin, addend, ld_add, addend2, ld_add2,
                  st)
    elif (prec in _HALO_BWD_PRECS and ldg % 4 == 0 and USE_HALO_CONV and (KH, KW, stride, pad) == (3, 3, 2, 1)
            and _hip.query("snn_conv3x3_s2_dgrad_supported", T * B, H, W, Cin, Ho, Wo, Cout)):
        # stride 2: all four phase classes of dx from ONE staged pass over dy (k_conv_s2dgrad3)
        if wt_image is None:
            wt_image = _frag_image(wt, Cin, Cout, 1, _hip.PREC_BF16X3)
        _hip.call("snn_conv3x3_s2_dgrad", gy.data_ptr(), ldg, wt_image.data_ptr(), dx.data_ptr(), cl_stride(dx), T * B, H, W, Cin, Ho,
                  Wo, Cout, addend, ld_add, addend2, ld_add2, prec, st)
    elif (prec in _HALO_BWD_PRECS and ldg % 4 == 0 and _halo_ok(T * B, H, W, Cout, Cin, KH, KW, stride, pad)):
        # dx = conv3x3(dy, mirrored taps of w^T): the halo-resident kernel with the data gradient's weight image
        if wt_image is None:
            wt_image = _frag_image(wt, Cin, Cout, 1, _hip.PREC_BF16X3)
        _hip.call("snn_conv3x3_halo", gy.data_ptr(), ldg, wt_image.data_ptr(), dx.data_ptr(), cl_stride(dx), T * B, H, W, Cout,
                  Cin, addend, ld_add, addend2, ld_add2, None, 0, None, prec, st)
    else:
        _hip.call("snn_conv2d_dgrad", gy.data_ptr(), ldg, wt.data_ptr(), wt_split, dx.data_ptr(), cl_stride(dx), T * B, H, W,
                  Cin, Ho, Wo, Cout, KH, KW, stride, pad, addend, ld_add, addend2, ld_add2, prec, st)
    if acc is not None and (acc[0].result is None or chained):
        if chained and acc[0].outer is not None and acc[0].outer[0].result is acc[0].result:
            acc[0].outer[0].set_result(dx)                 # the enclosing fanout expects what this
            acc[0].outer[0].fused[acc[0].outer[1]] = dx    # fanout will hand back: the new total
        acc[0].set_result(dx)
        acc[0].fused[acc[1]] = dx
    if outer_fused is not None:
        o_acc, o_key, key2 = outer_fused
        o_acc.fused[key2] = o_acc.deposits.pop(key2)
        o_acc.fused[o_key] = dx   # what the inner fanout will hand back for this alias
        o_acc.set_result(dx)
    return dx


# Halo-resident 3x3 kernel (csrc/conv_halo.hip) for the 64 / 128-channel stride-1 layers, forward and data gradient
# (SNN_NO_HALO_CONV: tuning / bisecting aid - the implicit GEMM everywhere)
USE_HALO_CONV = not os.environ.get("SNN_NO_HALO_CONV")
# backward modes the halo kernels take; both read the bf16 x 3 image of the transposed weights (bf16 storage: hi pieces only)
_HALO_BWD_PRECS = (_hip.PREC_BF16X3, _hip.PREC_BF16S)


def _halo_ok(N: int, H: int, W: int, Cin: int, Cout: int, KH: int, KW: int, stride: int, pad: int) -> bool:
    return (USE_HALO_CONV and KH == 3 and KW == 3 and stride == 1 and pad == 1
            and bool(_hip.query("snn_conv3x3_halo_supported", N, H, W, Cin, Cout)))


def _frag_image(src: torch.Tensor, O: int, I: int, flip: int, prec: int) -> torch.Tensor:
    """Weight image in MFMA-fragment order of ONE ``[O][3][3][I]`` matrix (snn_weight_frag_image_batched with a
    one-row table): what a layer without a FlatTrainer-kept image builds per call (one small launch)."""
    img = torch.empty((9 * O * I,), device=src.device, dtype=_F32)
    table = torch.tensor([[0, 0, O, I]], dtype=torch.int64, device=src.device)
    _hip.call("snn_weight_frag_image_batched", src.data_ptr(), img.data_ptr(), table.data_ptr(), 1,
              9 * (I // 32) * (O // 32) * 128, flip, prec, _stream())
    return img


def _cached_image(weight, name: str):
    """FlatTrainer's per-step weight image (a tensor view kept on the parameter), while the version counter matches."""
    img = getattr(weight, name, None)
    if img is not None and weight._snn_wt_version == weight._version:
        return img
    return None


class PendingBnApply(NamedTuple):
    """Second phase of a train-mode BatchNorm backward that has NOT been run: dy = A[t,c]*gx + B[t,c]*y + C[t,c]
    (snn_bn_bwd_apply).  ``_AffineNeuron.backward`` hands its producer convolution gx together with this record when that
    convolution announced (``y._snn_defer_apply``) that its backward forms dy while reading it (snn_conv2d_wgrad_bn):
    the 12-bytes-per-element apply pass and the dy tensor disappear."""
    gx: torch.Tensor      # dense [T,B,H,W,C]
    y: torch.Tensor       # the convolution output saved for the BatchNorm backward ([T,B,C,H,W], channels-last)
    coef: torch.Tensor    # [3, T, C]
    dims: Tuple[int, int, int, int, int]   # T, B, C, H, W


_PENDING_APPLY = {}   # gx.data_ptr() -> PendingBnApply
USE_DEFERRED_BN_APPLY = not os.environ.get("SNN_NO_DEFERRED_BN_APPLY")   # tuning / bisecting aid
# The same inside the halo-resident DATA gradient (snn_conv3x3_halo_bn) is built and parity-tested but OFF by default:
# measured (rocprofv3, profiles/r03_*) the fused kernel takes 390 us where data gradient + apply take ~300 us on the
# 64-channel layers (both are near the HBM rate there and the fused form reads y through the halo overlap as well) and
# 144 vs ~117 us on the 128-channel ones (the per-tap combine sits in front of every k-step's barrier).  DESIGN section 5.
USE_DEFERRED_BN_APPLY_DGRAD = bool(os.environ.get("SNN_DEFERRED_BN_APPLY_DGRAD"))


def reset_backward_state() -> None:
    """Drop records of a backward pass that did not finish (an exception between the two nodes)."""
    _PENDING_APPLY.clear()


def _apply_pending(pend: PendingBnApply) -> None:
    """The classic second phase, in place: gx becomes dy."""
    T, B, C, H, W = pend.dims
    _hip.call("snn_bn_bwd_apply_bf16" if pend.gx.dtype == _BF16 else "snn_bn_bwd_apply", pend.gx.data_ptr(), pend.y.data_ptr(), cl_stride(pend.y), pend.coef[0].data_ptr(),
              pend.coef[1].data_ptr(), pend.coef[2].data_ptr(), pend.gx.data_ptr(), C, T, B * H * W, C, 0, _stream())


def _wgrad_bn_ok(x: torch.Tensor, weight: torch.Tensor, stride: int, pad: int) -> bool:
    """The convolution's backward can apply the BatchNorm-backward affine itself: its weight gradient when nothing else
    needs dy (the event-frame layer), or its halo-resident data gradient (which also leaves dy for the weight gradient)."""
    if not (USE_DEFERRED_BN_APPLY and weight.requires_grad and x.dim() == 5):
        return False
    T, B, Cin, H, W = x.shape
    Cout, _, KH, KW = weight.shape
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    if not x.requires_grad:
        return bool(_hip.query("snn_conv2d_wgrad_bn_supported", T * B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad))
    return (USE_DEFERRED_BN_APPLY_DGRAD and USE_HALO_CONV and (KH, KW, stride, pad) == (3, 3, 1, 1)
            and _backward_precision == "bf16x3"
            and bool(_hip.query("snn_conv3x3_halo_bn_supported", T * B, H, W, Cout, Cin, B)))


class _Conv2d(Function):
    """nn.Conv2d(bias=False, padding=int(k/2), stride) over all T*B frames (layer_gen.py:129-136)."""

    @staticmethod
    def forward(ctx, x, weight, stride: int, pad: int, slot=None, dest=None, acc=None, prec=None, bn_out=None, x_th=None):
        # x_th (a float): x holds the saved POTENTIALS of a LIF layer that wrote no spike tensor; the operand is
        # z = (x > x_th), formed by the kernels while they read (snn_conv2d_spikes_* / snn_conv3x3_halo_spikes)
        _require_device(x, "conv2d input", bf16_ok=True)
        fwd_prec, bwd_prec = prec if prec is not None else _prec_codes(None, None)
        _require_device(weight, "conv2d weight")
        T, B, Cin, H, W = _dims5(x)
        Cout, Cin_w, KH, KW = weight.shape
        # bf16 storage: follows the input; enters at the event-frame layer (fp32 {0,1} frames in, bf16 out)
        sb = x.dtype == _BF16 or (_activation_storage == "bf16" and Cin == 2 and (KH, KW) == (3, 3))
        if sb:
            fwd_prec = bwd_prec = _hip.PREC_BF16S
        if Cin_w != Cin:
            raise RuntimeError(f"conv2d: input has {Cin} channels, weight expects {Cin_w}")
        Ho = (H + 2 * pad - KH) // stride + 1
        Wo = (W + 2 * pad - KW) // stride + 1
        x = _raw_to_cl(x)
        if x_th is not None and (sb or not _hip.query("snn_conv2d_spikes_supported", T * B, H, W, Cin, Ho, Wo, Cout, KH, KW,
                                                       stride, pad, cl_stride(x), fwd_prec, bwd_prec)):
            # arithmetic or shape the thresholding kernels do not cover: the spikes are written after all
            x = _cl_view((x.permute(0, 1, 3, 4, 2) > x_th).to(_F32).contiguous())
            x_th = None
        w = weight.detach()
        w_ohwi = w if is_channels_last(w) else _raw_dense_cl(w)
        y = _out_tensor(dest, T, B, Cout, Ho, Wo, x, _BF16 if sb else None)
        partial = layout = None
        if bn_out is not None:  # a train-mode BatchNorm follows: its statistics come out of this kernel's epilogue
            n_part = _hip.query("snn_conv2d_fwd_bn_partial_size", T * B, B, Ho, Wo, Cout)
            partial = torch.empty((n_part,), device=x.device, dtype=torch.float64)
            layout = (ctypes.c_int * 2)()
        # FlatTrainer keeps a pre-split image of the weights (valid while the version counter matches): ready-made pieces
        w16 = None
        if (fwd_prec == _hip.PREC_FP16X3 and USE_PRESPLIT_WEIGHTS and w_ohwi is w
                and getattr(weight, "_snn_w16", None) is not None and weight._snn_wt_version == weight._version):
            w16 = weight._snn_w16.data_ptr()   # a tensor view on the parameter: alive as long as the parameter is
        halo = _halo_ok(T * B, H, W, Cin, Cout, KH, KW, stride, pad) and cl_stride(x) % 4 == 0
        if halo and fwd_prec in (_hip.PREC_FP16X3, _hip.PREC_BF16S):
            # the image holds fp16 pieces (fp16 x 3) or bf16 pieces (bf16 storage: the hi pieces are the rounded weights)
            img_prec = _hip.PREC_BF16X3 if sb else _hip.PREC_FP16X3
            img = _cached_image(weight, "_snn_wfrag") if w_ohwi is w else None
            if img is not None and getattr(weight, "_snn_wfrag_prec", _hip.PREC_FP16X3) != img_prec:
                img = None
            if img is None:
                img = _frag_image(w_ohwi, Cout, Cin, 0, img_prec)
            if x_th is not None:
                _hip.call("snn_conv3x3_halo_spikes", x.data_ptr(), cl_stride(x), x_th, img.data_ptr(), y.data_ptr(),
                          cl_stride(y), T * B, H, W, Cin, Cout, _ptr(partial), B, layout, _stream())
            else:
                _hip.call("snn_conv3x3_halo", x.data_ptr(), cl_stride(x), img.data_ptr(), y.data_ptr(), cl_stride(y), T * B,
                          H, W, Cin, Cout, None, 0, None, 0, _ptr(partial), B, layout, fwd_prec, _stream())
        elif x_th is not None:
            _hip.call("snn_conv2d_spikes_fwd", x.data_ptr(), cl_stride(x), x_th, w_ohwi.data_ptr(), y.data_ptr(), cl_stride(y),
                      T * B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, _ptr(partial), B, layout, _stream())
        else:
            _hip.call("snn_conv2d_fwd", x.data_ptr(), cl_stride(x), w_ohwi.data_ptr(), w16, y.data_ptr(), cl_stride(y),
                      T * B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, None, 0, _ptr(partial), B, layout, fwd_prec,
                      _stream())
        if layout is not None and layout[0] > 0:
            bn_out.append(BnPartial(partial, int(layout[0]), int(layout[1]), y.data_ptr(), (T, B * Ho * Wo, Cout)))
        ctx.prec = bwd_prec
        ctx.x_th = x_th
        ctx.save_for_backward(x, w_ohwi)
        ctx.weight_ref = weight if getattr(weight, "_snn_wt", None) is not None else None  # FlatTrainer's cached w^T
        ctx.geom = (T, B, Cin, H, W, Cout, KH, KW, Ho, Wo, stride, pad)
        ctx.slot = slot
        ctx.acc = acc
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w_ohwi = ctx.saved_tensors
        T, B, Cin, H, W, Cout, KH, KW, Ho, Wo, stride, pad = ctx.geom
        gy = _raw_to_cl(gy)
        ldg, ldx = cl_stride(gy), cl_stride(x)
        st = _stream()
        dx = dw = None
        pend = _PENDING_APPLY.pop(gy.data_ptr(), None)
        if pend is not None:
            fused = (not ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and ctx.slot is not None
                     and pend.gx.dtype == _F32 and pend.dims == (T, B, Cout, Ho, Wo) and cl_stride(pend.y) % 4 == 0
                     and _hip.query("snn_conv2d_wgrad_bn_supported", T * B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad))
            if fused:
                # dy is formed while the weight-gradient kernel reads gx and y: no apply pass, no dy tensor
                splitk = _hip.query("snn_conv2d_wgrad_splitk", T * B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, ctx.prec)
                main = torch.cuda.current_stream()
                on_side = _wgrad_on_side()
                stream = _side_stream(x.device) if on_side else main
                if on_side:
                    _side_retire(main, WGRAD_SIDE_DEPTH)
                    stream.wait_stream(main)
                with torch.cuda.stream(stream):
                    ws = torch.empty((splitk, Cout * KH * KW * Cin), device=x.device, dtype=_F32)
                    _hip.call("snn_conv2d_wgrad_bn", x.data_ptr(), ldx, pend.gx.data_ptr(), Cout, pend.y.data_ptr(),
                              cl_stride(pend.y), pend.coef.data_ptr(), T, B, ctx.slot.buf.data_ptr(), T * B, H, W, Cin, Ho,
                              Wo, Cout, KH, KW, stride, pad, ctx.slot.claim(), ws.data_ptr(), splitk, stream.cuda_stream)
                if on_side:
                    _side_hold(stream, x, pend.gx, pend.y, pend.coef)
                return None, None, None, None, None, None, None, None, None, None
            fused_dgrad = (ctx.needs_input_grad[0] and ctx.prec == _hip.PREC_BF16X3 and USE_HALO_CONV
                           and (KH, KW, stride, pad) == (3, 3, 1, 1) and pend.dims == (T, B, Cout, Ho, Wo)
                           and is_channels_last(pend.y)
                           and _hip.query("snn_conv3x3_halo_bn_supported", T * B, H, W, Cout, Cin, B))
            if not fused_dgrad:
                _apply_pending(pend)   # this convolution takes a materialised dy after all
                pend = None
        if ctx.needs_input_grad[0]:
            wref = ctx.weight_ref
            wt16 = wt_img = None
            if wref is not None and wref._snn_wt_version == wref._version:
                wt = wref._snn_wt  # transposed once per optimiser step for all layers (trainer.FlatTrainer)
                wt_img = getattr(wref, "_snn_wtfrag", None)   # ... and arranged for the halo-resident data gradient
                if ctx.prec == _hip.PREC_BF16X3 and USE_PRESPLIT_WEIGHTS and USE_PRESPLIT_DGRAD:
                    wt16 = _ptr(getattr(wref, "_snn_wt16", None))   # ... and pre-split into its bf16 pieces
            else:
                wt = torch.empty((Cin, KH, KW, Cout), device=x.device, dtype=_F32)
                _hip.call("snn_weight_transpose", w_ohwi.data_ptr(), wt.data_ptr(), Cout, KH, KW, Cin, st)
            if pend is not None:
                # dy is formed inside the data-gradient kernel from (gx, y, coef) and left in `gy_new` for the weight gradient
                gy_new = torch.empty_like(pend.gx)
                dx = _dgrad_accumulate(ctx.acc, gy, ldg, wt, x, ctx.geom, st, ctx.prec, wt_split=wt16, wt_image=wt_img,
                                       pend=(pend, gy_new))
                gy = _cl_view(gy_new)
                ldg = cl_stride(gy)
            else:
                dx = _dgrad_accumulate(ctx.acc, gy, ldg, wt, x, ctx.geom, st, ctx.prec, wt_split=wt16, wt_image=wt_img)
        if ctx.needs_input_grad[1]:
            splitk = _hip.query("snn_conv2d_wgrad_splitk", T * B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, ctx.prec)

            def wgrad(dst_ptr, accumulate, ws_, stream_ptr):
                if ctx.x_th is not None:   # x holds potentials: thresholded on load
                    _hip.call("snn_conv2d_spikes_wgrad", x.data_ptr(), ldx, ctx.x_th, gy.data_ptr(), ldg, dst_ptr, T * B, H,
                              W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, accumulate, ws_.data_ptr(), splitk, stream_ptr)
                else:
                    _hip.call("snn_conv2d_wgrad", x.data_ptr(), ldx, gy.data_ptr(), ldg, dst_ptr, T * B, H, W, Cin, Ho, Wo,
                              Cout, KH, KW, stride, pad, accumulate, ws_.data_ptr(), splitk, ctx.prec, stream_ptr)
            if ctx.slot is not None and _wgrad_on_side():
                # gradient goes straight into the flat buffer: nothing downstream in autograd needs it, so
                # the kernel runs on the side stream, concurrently with the data-gradient chain
                main, side = torch.cuda.current_stream(), _side_stream(x.device)
                _side_retire(main, WGRAD_SIDE_DEPTH)   # the weight gradient before the previous one is joined; its operands go
                side.wait_stream(main)  # gy (and every earlier use of the slot) is complete
                with torch.cuda.stream(side):
                    ws = torch.empty((splitk, Cout * KH * KW * Cin), device=x.device, dtype=_F32)
                    wgrad(ctx.slot.buf.data_ptr(), ctx.slot.claim(), ws, side.cuda_stream)
                _side_hold(side, x, gy)
            elif ctx.slot is not None:
                ws = torch.empty((splitk, Cout * KH * KW * Cin), device=x.device, dtype=_F32)
                wgrad(ctx.slot.buf.data_ptr(), ctx.slot.claim(), ws, st)
            else:
                ws = torch.empty((splitk, Cout * KH * KW * Cin), device=x.device, dtype=_F32)
                dw_ohwi = torch.empty((Cout, KH, KW, Cin), device=x.device, dtype=_F32)
                wgrad(dw_ohwi.data_ptr(), 0, ws, st)
                dw = dw_ohwi.permute(0, 3, 1, 2)
        return dx, dw, None, None, None, None, None, None, None, None


def _small_gemm(a: torch.Tensor, trans_a: bool, b: torch.Tensor, trans_b: bool, c: torch.Tensor, accumulate: int,
                ct: Optional[torch.Tensor] = None, ct_col: int = 0) -> None:
    """``c (+)= op(a) @ op(b)`` on dense row-major fp32 matrices (``snn_small_gemm``, current stream); ``ct``: the
    transposed result from the same launch - ``[n, m]``, or columns ``ct_col .. ct_col + m`` of a wider ``[n, M]`` matrix."""
    m, n = c.shape
    k = a.shape[0] if trans_a else a.shape[1]
    ct_ptr, ldct = (None, 0) if ct is None else (ct.data_ptr() + 4 * ct_col, ct.shape[1])
    _hip.call("snn_small_gemm", a.data_ptr(), a.shape[1], int(trans_a), b.data_ptr(), b.shape[1], int(trans_b),
              c.data_ptr(), n, m, n, k, int(accumulate), ct_ptr, ldct, _stream())


class _ComposedConv1x1(Function):
    """``conv1x1(conv1x1(x, w1), w2)`` without the intermediate tensor (no Norm / neuron between the two: the C2f
    entry ``Conv(c, 1)`` followed by the branch-opening ``Conv(c/2, 1)`` of ``models/tiny_yolo.py:76-82``).

    Both maps are linear, so ``y = (w2 w1) x``.  Forward runs ONE 1x1 convolution with the composed weight; backward
    needs neither ``conv(x, w1)`` nor its gradient: with ``G = sum_pixels gy x^T`` (one weight-gradient kernel),
    ``dw2 = G w1^T``, ``dw1 = w2^T G`` and ``dx = conv^T(gy, w2 w1)``.  The composed weight is rounded once in fp32
    (a re-association of the reference's two fp32 sums, same error level as any other summation order).
    """

    @staticmethod
    def forward(ctx, x, w1, w2, slot1, slot2, dest, acc, prec=None):
        _require_device(x, "conv2d input", bf16_ok=True)
        fwd_prec, bwd_prec = prec if prec is not None else _prec_codes(None, None)
        if x.dtype == _BF16:
            fwd_prec = bwd_prec = _hip.PREC_BF16S
        T, B, Cin, H, W = _dims5(x)
        C1, C2 = w1.shape[0], w2.shape[0]
        if w1.shape[1] != Cin or w2.shape[1] != C1 or tuple(w1.shape[2:]) != (1, 1) or tuple(w2.shape[2:]) != (1, 1):
            raise RuntimeError("composed 1x1 convolution: weight shapes do not chain")
        x = _raw_to_cl(x)
        w1m = w1.detach().reshape(C1, Cin)
        w2m = w2.detach().reshape(C2, C1)
        if not (w1m.is_contiguous() and w2m.is_contiguous()):
            w1m, w2m = w1m.contiguous(), w2m.contiguous()
        # FlatTrainer composes every registered pair once per optimiser step in ONE launch (snn_small_gemm_batched); the
        # pair registers itself here on first use (w2._snn_compose_with) and is served from the next refresh on, while the
        # version counters of both weights still match.  Same fmaf chains, same bits as the per-call product.
        cached = getattr(w2, "_snn_composed", None)
        if (cached is not None and cached[0] is w1 and cached[1] == (w1._version, w2._version)
                and tuple(cached[2].shape) == (C2, Cin)):
            wc, wct = cached[2], cached[3]
        else:
            if getattr(w1, "_snn_wt", None) is not None and getattr(w2, "_snn_wt", None) is not None:
                w2._snn_compose_with = w1   # both live in a FlatTrainer's flat buffer
            wc = torch.empty((C2, Cin), device=x.device, dtype=_F32)   # w2 w1: [C2, Cin] = OHWI of a 1x1 kernel
            # ... and (w2 w1)^T, the operand of the data gradient, out of the same launch (same fmaf chains: the bits a
            # separate w1^T w2^T product would give)
            wct = torch.empty((Cin, C2), device=x.device, dtype=_F32) if ctx.needs_input_grad[0] else None
            _small_gemm(w2m, False, w1m, False, wc, 0, ct=wct)
        y = _out_tensor(dest, T, B, C2, H, W, x)
        _hip.call("snn_conv2d_fwd", x.data_ptr(), cl_stride(x), wc.data_ptr(), None, y.data_ptr(), cl_stride(y), T * B,
                  H, W, Cin, H, W, C2, 1, 1, 1, 0, None, 0, None, 0, None, fwd_prec, _stream())
        ctx.prec = bwd_prec
        ctx.save_for_backward(x, w1m, w2m, wc)
        ctx.wct = wct
        ctx.geom = (T, B, Cin, H, W, C2, 1, 1, H, W, 1, 0)
        ctx.c1 = C1
        ctx.slots = (slot1, slot2)
        ctx.acc = acc
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w1m, w2m, wc = ctx.saved_tensors
        T, B, Cin, H, W, C2 = ctx.geom[:6]
        C1 = ctx.c1
        gy = _raw_to_cl(gy)
        ldg, ldx = cl_stride(gy), cl_stride(x)
        st = _stream()
        dx = dw1 = dw2 = None
        if ctx.needs_input_grad[0]:
            wct = ctx.wct   # (w2 w1)^T = w1^T w2^T: transposed 1x1 weight, left by the forward pass
            dx = _dgrad_accumulate(ctx.acc, gy, ldg, wct, x, ctx.geom, st, ctx.prec)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            slot1, slot2 = ctx.slots
            slotted = slot1 is not None and slot2 is not None
            splitk = _hip.query("snn_conv2d_wgrad_splitk", T * B, H, W, Cin, H, W, C2, 1, 1, 1, 0, ctx.prec)
            side_ok = slotted and _wgrad_on_side()
            main = torch.cuda.current_stream()
            stream = _side_stream(x.device) if side_ok else main
            if side_ok:
                _side_retire(main, WGRAD_SIDE_DEPTH)
                stream.wait_stream(main)
            with torch.cuda.stream(stream):
                ws = torch.empty((splitk, C2 * Cin), device=x.device, dtype=_F32)
                G = torch.empty((C2, Cin), device=x.device, dtype=_F32)
                _hip.call("snn_conv2d_wgrad", x.data_ptr(), ldx, gy.data_ptr(), ldg, G.data_ptr(), T * B, H, W, Cin, H,
                          W, C2, 1, 1, 1, 0, 0, ws.data_ptr(), splitk, ctx.prec, stream.cuda_stream)
                # dw2 = G w1^T [C2, C1], dw1 = w2^T G [C1, Cin]: straight into the flat gradient slots when there are any
                if slotted:
                    g2, g1 = slot2.buf.view(C2, C1), slot1.buf.view(C1, Cin)
                    acc2, acc1 = slot2.claim(), slot1.claim()
                else:
                    g2 = torch.empty((C2, C1), device=x.device, dtype=_F32)
                    g1 = torch.empty((C1, Cin), device=x.device, dtype=_F32)
                    acc2 = acc1 = 0
                _small_gemm(G, False, w1m, True, g2, acc2)
                _small_gemm(w2m, True, G, False, g1, acc1)
            if side_ok:
                _side_hold(stream, x, gy)
            if not slotted:
                dw1, dw2 = g1.view(C1, Cin, 1, 1), g2.view(C2, C1, 1, 1)
        return dx, dw1, dw2, None, None, None, None, None


def composed_conv1x1(x: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor, dest: Optional[Dest] = None,
                     forward_precision: Optional[str] = None, backward_precision: Optional[str] = None) -> torch.Tensor:
    if x.dtype == _BF16 and (w1.shape[1] % 32 or w2.shape[0] % 32):   # (see conv2d)
        y = to_bfloat16(composed_conv1x1(to_float32(x), w1, w2, None, forward_precision, backward_precision))
        return place(y, dest) if dest is not None else y
    seq, single = as_sequence(x)
    y = _ComposedConv1x1.apply(seq, w1, w2, _slot_of(w1), _slot_of(w2), dest, _acc_of(seq),
                               _prec_codes(forward_precision, backward_precision))
    return y[0] if single else y


class _SplitChannels(Function):
    """Channel ranges of one channels-last tensor as separate tensors (aliases, no copy): the branch inputs behind a
    fused sibling convolution.  Backward wants ONE gradient for the whole tensor: the parts ask (``_snn_grad_in_slot``)
    that their accumulated gradients be left in the concat-gradient slices they contain (``_dgrad_accumulate``), which
    for sibling outputs that sit side by side in a Dense merge are side by side too - then the gradients ARE one
    channel-sliced tensor; otherwise they are gathered by the strided copy kernel."""

    @staticmethod
    def forward(ctx, x, widths):
        T, B, C, H, W = _dims5(x)
        if sum(widths) != C:
            raise RuntimeError(f"split_channels: widths {widths} do not add up to {C} channels")
        ctx.widths, ctx.like = tuple(widths), (T, B, H, W)
        ld, outs, off = cl_stride(x), [], 0
        for c in widths:
            t = _alias(x, x.storage_offset() + off, T, B, c, H, W, ld)
            t._snn_grad_in_slot = True
            outs.append(t)
            off += c
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        T, B, H, W = ctx.like
        live = [g for g in gs if g is not None]
        if not live:
            return None, None
        gs = [None if g is None else _raw_to_cl(g) for g in gs]
        first = gs[0]
        joined = first is not None
        if joined:
            ld, base, off = cl_stride(first), first.untyped_storage().data_ptr(), first.storage_offset()
            for g, c in zip(gs, ctx.widths):
                if (g is None or g.untyped_storage().data_ptr() != base or g.storage_offset() != off or cl_stride(g) != ld
                        or g.dtype != first.dtype or tuple(g.shape) != (T, B, c, H, W)):
                    joined = False
                    break
                off += c
        if joined:
            return _alias(first, first.storage_offset(), T, B, sum(ctx.widths), H, W, ld), None
        like = live[0]
        out = _new_cl((T, B), sum(ctx.widths), H, W, like)
        off = 0
        for g, c in zip(gs, ctx.widths):
            dst = out.narrow(2, off, c)
            if g is None:
                dst.zero_()
            else:
                _copy_cl(g, dst)
            off += c
        return out, None


def split_channels(x: torch.Tensor, widths: Sequence[int]) -> List[torch.Tensor]:
    return list(_SplitChannels.apply(x, tuple(int(c) for c in widths)))


class _SiblingConv1x1(Function):
    """Several plain 1x1 convolutions of ONE input as one convolution with the row-stacked weight - the two
    ``Conv(c/2, 1)`` that open the branches of a C2f block (``models/tiny_yolo.py:84-85``), whose outputs sit side by side
    in the block's Dense merge - optionally each composed with a shared 1x1 convolution in front (``w1``: the C2f entry
    ``Conv(c, 1)``, see ``_ComposedConv1x1``).  The input is read once instead of once per branch; backward is ONE data
    gradient (instead of two chained through an addend: one read and one write of dx less) and ONE pixel reduction
    ``G = sum gy x^T`` whose row blocks give the branches' weight gradients (``dw2_b = G_b w1^T``, ``dw1 = sum_b w2_b^T G_b``).
    Arithmetic per output element is that of the separate convolutions (the same k-ordered products); only ``dw1`` sums its
    branch contributions in another order."""

    @staticmethod
    def forward(ctx, x, w1, dest, acc, prec, slot1, slots2, x_th, *w2s):
        """``x_th`` (not None): ``x`` holds the saved potentials of the LIF layer in front (``affine_neuron(spikes_ok=True)``),
        the operand is ``z = (x > x_th)``."""
        _require_device(x, "conv2d input", bf16_ok=True)
        fwd_prec, bwd_prec = prec if prec is not None else _prec_codes(None, None)
        if x.dtype == _BF16:
            fwd_prec = bwd_prec = _hip.PREC_BF16S
        T, B, Cin, H, W = _dims5(x)
        C1 = w1.shape[0] if w1 is not None else Cin
        widths = [int(w.shape[0]) for w in w2s]
        Ct = sum(widths)
        for w in w2s:
            if w.shape[1] != C1 or tuple(w.shape[2:]) != (1, 1):
                raise RuntimeError("sibling 1x1 convolutions: weight shapes do not match the input")
        if w1 is not None and (w1.shape[1] != Cin or tuple(w1.shape[2:]) != (1, 1)):
            raise RuntimeError("sibling 1x1 convolutions: the composed weight does not chain")
        x = _raw_to_cl(x)
        w1m = None if w1 is None else w1.detach().reshape(C1, Cin).contiguous()
        w2ms = [w.detach().reshape(c, C1).contiguous() for w, c in zip(w2s, widths)]
        # FlatTrainer builds the stacked (composed) weight and its transpose once per optimiser step, in its batched GEMM
        # launch; the group registers itself here on first use (w2s[0]._snn_sibling_group)
        need_t = ctx.needs_input_grad[0]
        cached = getattr(w2s[0], "_snn_sibling_weight", None)
        versions = tuple(w._version for w in w2s) + ((w1._version,) if w1 is not None else ())
        if (cached is not None and cached[0] is w1 and len(cached[1]) == len(w2s)
                and all(a is b for a, b in zip(cached[1], w2s)) and cached[2] == versions
                and tuple(cached[3].shape) == (Ct, Cin)):
            wc, wct = cached[3], cached[4]
        else:
            if w1 is not None and all(getattr(w, "_snn_wt", None) is not None for w in (w1, *w2s)):
                w2s[0]._snn_sibling_group = (w1, tuple(w2s))
            wc = torch.empty((Ct, Cin), device=x.device, dtype=_F32)
            wct = torch.empty((Cin, Ct), device=x.device, dtype=_F32) if need_t else None
            row = 0
            for w2m, c in zip(w2ms, widths):
                if w1m is not None:
                    _small_gemm(w2m, False, w1m, False, wc[row:row + c], 0, ct=wct, ct_col=row)
                else:
                    wc[row:row + c].copy_(w2m)
                row += c
            if w1m is None and need_t:
                wct.copy_(wc.t())
        y = _out_tensor(dest, T, B, Ct, H, W, x)
        if x_th is not None and not _hip.query("snn_conv1x1_spikes_supported", T * B, H, W, Cin, Ct, cl_stride(x), fwd_prec,
                                               bwd_prec):
            # (arithmetic or shape the thresholding kernels do not cover: the spikes are materialised after all)
            x = _cl_view((x.permute(0, 1, 3, 4, 2) > x_th).to(_F32).contiguous())
            x_th = None
        if x_th is not None:
            _hip.call("snn_conv1x1_spikes_fwd", x.data_ptr(), cl_stride(x), x_th, wc.data_ptr(), y.data_ptr(), cl_stride(y),
                      T * B, H, W, Cin, Ct, _stream())
        else:
            _hip.call("snn_conv2d_fwd", x.data_ptr(), cl_stride(x), wc.data_ptr(), None, y.data_ptr(), cl_stride(y), T * B,
                      H, W, Cin, H, W, Ct, 1, 1, 1, 0, None, 0, None, 0, None, fwd_prec, _stream())
        ctx.prec = bwd_prec
        ctx.x_th = x_th
        ctx.save_for_backward(x, w1m, *w2ms)
        ctx.wct = wct
        ctx.geom = (T, B, Cin, H, W, Ct, 1, 1, H, W, 1, 0)
        ctx.widths, ctx.c1 = widths, C1
        ctx.slot1, ctx.slots2 = slot1, slots2
        ctx.acc = acc
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w1m, *w2ms = ctx.saved_tensors
        T, B, Cin, H, W, Ct = ctx.geom[:6]
        C1, widths = ctx.c1, ctx.widths
        composed = w1m is not None
        gy = _raw_to_cl(gy)
        ldg, ldx = cl_stride(gy), cl_stride(x)
        st = _stream()
        dx = dw1 = None
        dw2s = [None] * len(widths)
        if ctx.needs_input_grad[0]:
            dx = _dgrad_accumulate(ctx.acc, gy, ldg, ctx.wct, x, ctx.geom, st, ctx.prec)
        if any(ctx.needs_input_grad[8:]) or (composed and ctx.needs_input_grad[1]):
            slot1, slots2 = ctx.slot1, ctx.slots2
            slotted = all(s_ is not None for s_ in slots2) and (slot1 is not None or not composed)
            splitk = _hip.query("snn_conv2d_wgrad_splitk", T * B, H, W, Cin, H, W, Ct, 1, 1, 1, 0, ctx.prec)
            side_ok = slotted and _wgrad_on_side()
            main = torch.cuda.current_stream()
            stream = _side_stream(x.device) if side_ok else main
            if side_ok:
                _side_retire(main, WGRAD_SIDE_DEPTH)
                stream.wait_stream(main)
            with torch.cuda.stream(stream):
                ws = torch.empty((splitk, Ct * Cin), device=x.device, dtype=_F32)
                G = torch.empty((Ct, Cin), device=x.device, dtype=_F32)
                if ctx.x_th is not None:   # x holds the potentials of the LIF layer in front: thresholded on load
                    _hip.call("snn_conv1x1_spikes_wgrad", x.data_ptr(), ldx, ctx.x_th, gy.data_ptr(), ldg, G.data_ptr(),
                              T * B, H, W, Cin, Ct, 0, ws.data_ptr(), splitk, stream.cuda_stream)
                else:
                    _hip.call("snn_conv2d_wgrad", x.data_ptr(), ldx, gy.data_ptr(), ldg, G.data_ptr(), T * B, H, W, Cin, H,
                              W, Ct, 1, 1, 1, 0, 0, ws.data_ptr(), splitk, ctx.prec, stream.cuda_stream)
                g1 = None
                if composed:
                    g1 = slot1.buf.view(C1, Cin) if slotted else torch.empty((C1, Cin), device=x.device, dtype=_F32)
                    acc1 = slot1.claim() if slotted else 0
                row = 0
                for b, (w2m, c) in enumerate(zip(w2ms, widths)):
                    Gb = G[row:row + c]
                    row += c
                    if slotted:
                        g2, acc2 = slots2[b].buf.view(c, C1), slots2[b].claim()
                    else:
                        g2, acc2 = torch.empty((c, C1), device=x.device, dtype=_F32), 0
                    if composed:
                        _small_gemm(Gb, False, w1m, True, g2, acc2)      # dw2_b = G_b w1^T
                        _small_gemm(w2m, True, Gb, False, g1, acc1)      # dw1 (+)= w2_b^T G_b
                        acc1 = 1
                    elif acc2:
                        g2.add_(Gb)
                    else:
                        g2.copy_(Gb)
                    if not slotted:
                        dw2s[b] = g2.view(c, C1, 1, 1)
                if composed and not slotted:
                    dw1 = g1.view(C1, Cin, 1, 1)
            if side_ok:
                _side_hold(stream, x, gy)
        return (dx, dw1, None, None, None, None, None, None, *dw2s)


def sibling_conv1x1(x: torch.Tensor, w1: Optional[torch.Tensor], w2s: Sequence[torch.Tensor], dest: Optional[Dest] = None,
                    forward_precision: Optional[str] = None, backward_precision: Optional[str] = None) -> torch.Tensor:
    """``cat([conv1x1(h, w) for w in w2s], channels)`` with ``h = conv1x1(x, w1)`` (``h = x`` when ``w1`` is None), as one
    convolution; ``x`` is a sequence ``[T,B,C,H,W]``."""
    return _SiblingConv1x1.apply(x, w1, dest, _acc_of(x), _prec_codes(forward_precision, backward_precision),
                                 _slot_of(w1), tuple(_slot_of(w) for w in w2s), getattr(x, "_snn_spike_threshold", None),
                                 *w2s)


class BnPartial(NamedTuple):
    """Statistics partials a forward convolution left for the BatchNorm behind it (``snn_conv2d_fwd`` ``bn_partial``)."""
    partial: torch.Tensor
    chunks: int
    rows_per_chunk: int
    data_ptr: int          # the tensor they describe
    dims: Tuple[int, int, int]   # (T, pixels per timestep, channels)


def conv2d(x: torch.Tensor, weight: torch.Tensor, stride: int = 1, padding: int = 0, dest: Optional[Dest] = None,
           forward_precision: Optional[str] = None, backward_precision: Optional[str] = None,
           bn_stats: bool = False) -> torch.Tensor:
    """``forward_precision`` / ``backward_precision``: this call's arithmetic (None = the session default).

    ``bn_stats``: the result feeds a train-mode BatchNorm - the kernel also emits that layer's statistics partials
    (attached to the result as ``_snn_bn_partial``; ``affine_neuron`` picks them up and skips its own pass over y).
    """
    if x.dtype == _BF16 and (weight.shape[1] % 32 or weight.shape[0] % 32):
        # bf16 storage: the kernels want whole 32-channel k-steps (forward: Cin, data gradient: Cout); other layers - a
        # prediction head applied to every timestep - take the fp32 kernels between two conversions
        y = to_bfloat16(conv2d(to_float32(x), weight, stride, padding, None, forward_precision, backward_precision, False))
        return place(y, dest) if dest is not None else y
    x_th = getattr(x, "_snn_spike_threshold", None)   # x holds a LIF layer's saved potentials (affine_neuron spikes_ok)
    seq, single = as_sequence(x)
    bn_out = [] if bn_stats else None
    y = _Conv2d.apply(seq, weight, int(stride), int(padding), _slot_of(weight), dest, _acc_of(seq),
                      _prec_codes(forward_precision, backward_precision), bn_out, x_th)
    y = y[0] if single else y
    if bn_out:
        y._snn_bn_partial = bn_out[0]
    if bn_stats and not single and _slot_of(weight) is not None and _wgrad_bn_ok(seq, weight, int(stride), int(padding)):
        y._snn_defer_apply = True   # (bn_stats: the BatchNorm behind is y's only consumer)
    return y


# ------------------------------------------------------------------------------------------- norm + neuron
class NeuronState(NamedTuple):
    """State of a LIF / LI layer: membrane potential and synaptic current, each ``[B,C,h,w]``.

    Same field names / order as norse's ``LIFFeedForwardState`` and ``LIState``.
    """
    v: torch.Tensor
    i: torch.Tensor


class SynapseState(NamedTuple):
    """State of a ``Synapse`` layer: mediator concentration (synapse.py:18-22)."""
    p: torch.Tensor


_SAVES_STEP = (_hip.NEURON_LIF, _hip.NEURON_SLI, _hip.NEURON_SYNAPSE)
SCAN_SEGMENT_T = 32   # backward scans of longer sequences run in segments of this many steps (None: one launch)
# BatchNorm statistics from the producing convolution's epilogue when it offers them (SNN_NO_CONV_BN_STATS: tuning
# aid, the separate snn_bn_stats pass everywhere)
USE_CONV_BN_STATS = not os.environ.get("SNN_NO_CONV_BN_STATS")
# pre-split weight images kept by FlatTrainer (SNN_NO_PRESPLIT_WEIGHTS: tuning aid, conversion in every block)
USE_PRESPLIT_WEIGHTS = not os.environ.get("SNN_NO_PRESPLIT_WEIGHTS")
# ... for the data gradient too: measured neutral (its bf16 split is three bit operations per pair, the forward's fp16 split a
# scale, two conversions and a subtraction), so off by default: one launch and one weight-sized buffer less per step
USE_PRESPLIT_DGRAD = bool(os.environ.get("SNN_PRESPLIT_DGRAD"))
SCAN_FLAGS = 0   # flags of snn_affine_neuron_bwd; tests set _hip.SCAN_WIDE_ADDRESSING to cover the 64-bit-pointer scan
# the reverse LIF scan of a train-mode Norm -> LIF layer does not read the convolution output y: the BatchNorm statistic it
# needs y for is formed from the neuron input rebuilt from the saved potentials (snn_affine_neuron_bwd
# SNN_SCAN_SUMS_FROM_STATE; SNN_NO_SUMS_FROM_STATE: tuning / bisecting aid)
USE_SUMS_FROM_STATE = not os.environ.get("SNN_NO_SUMS_FROM_STATE")

# Opt-in memory lever: a LIF layer whose per-step saved state ([T,B,H,W,C] fp32) is at least this many bytes stores
# checkpoints of (v, i) every snn_lif_ckpt_interval() steps instead and recomputes in the backward scan
# (snn_lif_fwd_ckpt / snn_lif_bwd_ckpt: bit-identical results, half the saved-state memory, speed-neutral).
# None (default) = never, 0 = every LIF layer.  Measured on 1280x720 B=4 T=32: 125 -> 113 GiB live.  Off by default
# because the half-size buffers fragment torch's caching allocator once the live set approaches the whole HBM
# (B=8: 222 GiB live but 287 GiB reserved and a thrashing allocator, against 250 GiB live at full speed without).
LIF_CHECKPOINT_BYTES: Optional[int] = None


class _AffineNeuron(Function):
    """[BatchNorm2d (per-timestep batch statistics)] -> [LIF | LI | LI+Tanh | nothing], fused.

    inputs : y[T,B,C,H,W], gamma[C]|None, bias[C]|None, v0|None, i0|None, addend[T,B,C,H,W]|None, + non-tensor config
    outputs: out[T,B,C,H,W], vT[B,C,H,W], iT[B,C,H,W]  (vT/iT are dummies when neuron == NONE)

    ``addend`` is a residual shortcut folded into the output store: out = neuron(norm(y)) + addend
    (generator.py:145-146 does stack + sum as a separate pass); its gradient is g_out itself.
    """

    @staticmethod
    def forward(ctx, y, gamma, bias, v0, i0, addend, cfg):
        (neuron, has_bn, training, eps, momentum, running_mean, running_var, params, g_slot, b_slot, dest,
         sync_group, bn_hint, last_only, defer_apply, spikes_out) = cfg
        _require_device(y, "norm/neuron input", bf16_ok=True)
        sb = y.dtype == _BF16   # bf16 storage: y, out, the saved per-step state and the gradients; (v, i) and all sums fp32
        if sb and (neuron not in (_hip.NEURON_NONE, _hip.NEURON_LIF, _hip.NEURON_LI, _hip.NEURON_LI_TANH) or y.shape[-3] % 4):
            raise RuntimeError("bf16 storage: Norm + none / LIF / LI / LI+Tanh with a multiple of 4 channels only")
        sb_flag = _hip.SCAN_BF16_STORAGE if sb else 0
        ctx.set_materialize_grads(False)  # unused final-state outputs must arrive as None, not as zero tensors
        y = _raw_to_cl(y)
        ldy = cl_stride(y)
        T, B, C, H, W = _dims5(y)
        M = B * H * W
        st = _stream()
        dev = y.device
        alpha = beta = mean = invstd = None
        use_running = has_bn and not training
        if has_bn:
            mean = torch.empty((T, C), device=dev, dtype=_F32)
            invstd = torch.empty((T, C), device=dev, dtype=_F32)
            alpha = torch.empty((T, C), device=dev, dtype=_F32)
            beta = torch.empty((T, C), device=dev, dtype=_F32)
            g_ptr = _ptr(gamma.detach()) if gamma is not None else None
            b_ptr = _ptr(bias.detach()) if bias is not None else None
            if use_running:
                if running_mean is None or running_var is None:
                    raise RuntimeError("BatchNorm in eval mode needs running statistics")
                _hip.call("snn_bn_stats_finalize", None, 0, 0, T, M, C, g_ptr, b_ptr, eps, momentum,
                          running_mean.data_ptr(), running_var.data_ptr(), 1, mean.data_ptr(), invstd.data_ptr(),
                          alpha.data_ptr(), beta.data_ptr(), st)
            else:
                if (bn_hint is not None and bn_hint.data_ptr == y.data_ptr() and bn_hint.dims == (T, M, C)
                        and USE_CONV_BN_STATS):
                    # the producing convolution summed y and y^2 on the way out (snn_conv2d_fwd bn_partial)
                    partial, chunks, rpc = bn_hint.partial, bn_hint.chunks, bn_hint.rows_per_chunk
                else:
                    n_part = _hip.query("snn_bn_stats_partial_size", T, M, C)
                    partial = torch.empty((n_part,), device=dev, dtype=torch.float64)
                    chunks = rpc = 0
                    _hip.call("snn_bn_stats_bf16" if sb else "snn_bn_stats", y.data_ptr(), ldy, T, M, C,
                              partial.data_ptr(), st)
                if sync_group is None:
                    _hip.call("snn_bn_stats_finalize", partial.data_ptr(), chunks, rpc, T, M, C, g_ptr, b_ptr, eps, momentum,
                              _ptr(running_mean), _ptr(running_var), 0, mean.data_ptr(), invstd.data_ptr(),
                              alpha.data_ptr(), beta.data_ptr(), st)
                else:
                    # SyncBatchNorm (config.yaml:76): ONE all-reduce of [T, C, 2] sums per layer for all timesteps
                    import torch.distributed as dist
                    world = dist.get_world_size(sync_group[0])
                    sums = torch.empty((T, C, 2), device=dev, dtype=torch.float64)
                    _hip.call("snn_bn_stats_reduce", partial.data_ptr(), chunks, rpc, T, M, C, sums.data_ptr(), st)
                    dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=sync_group[0])
                    scratch = torch.empty((T * C,), device=dev, dtype=torch.float64)
                    _hip.call("snn_bn_stats_from_sums", sums.data_ptr(), T, M * world, C, g_ptr, b_ptr, eps, momentum,
                              _ptr(running_mean), _ptr(running_var), mean.data_ptr(), invstd.data_ptr(),
                              alpha.data_ptr(), beta.data_ptr(), scratch.data_ptr(), st)
        need_grad = any(ctx.needs_input_grad[:5])
        # spikes_out (a list the caller reads afterwards): the consumer can form the spikes from the saved potentials
        # (snn_conv1x1_spikes_*), so when those are saved anyway no output tensor is written at all
        no_out = (spikes_out is not None and USE_SPIKES_FROM_VDEC and neuron == _hip.NEURON_LIF and need_grad and has_bn
                  and addend is None and dest is None and not last_only and not sb and C % 4 == 0 and ldy % 4 == 0
                  and params.v_th >= 0.0
                  and not (LIF_CHECKPOINT_BYTES is not None and T * M * C * 4 >= LIF_CHECKPOINT_BYTES))
        if last_only:
            # only the last timestep's output is kept (snn_affine_neuron_fwd SNN_SCAN_LAST_STEP_ONLY): out is [B,C,H,W]
            if (neuron not in (_hip.NEURON_LIF, _hip.NEURON_LI, _hip.NEURON_LI_TANH) or addend is not None
                    or dest is not None):
                raise RuntimeError("last_only is for LIF / LI / LI+Tanh without shortcut or concat destination")
            out = _new_cl((B,), C, H, W, y)
        elif no_out:
            out = None
        else:
            out = _out_tensor(dest, T, B, C, H, W, y)
        has_state = neuron != _hip.NEURON_NONE
        vT = _new_cl((B,), C, H, W, y, _F32) if has_state else torch.empty(0, device=dev)
        iT = _new_cl((B,), C, H, W, y, _F32) if has_state else torch.empty(0, device=dev)
        vdec = None
        ckpt = False
        if neuron in _SAVES_STEP and need_grad:
            # (the checkpointed kernels write / read all T outputs: not for the last-step-only read-out)
            ckpt = (neuron == _hip.NEURON_LIF and LIF_CHECKPOINT_BYTES is not None and not last_only and not sb
                    and T * M * C * 4 >= LIF_CHECKPOINT_BYTES)
            if ckpt:
                k = _hip.query("snn_lif_ckpt_interval")
                vdec = torch.empty(((T + k - 1) // k, 2, B, H, W, C), device=dev, dtype=_F32)
            else:
                vdec = torch.empty((T, B, H, W, C), device=dev, dtype=y.dtype)
        if v0 is not None:
            v0 = _expand_state(v0, (B, C, H, W), dev)
        if i0 is not None:
            i0 = _expand_state(i0, (B, C, H, W), dev)
        ctx.addend_acc = None
        ad_ptr, ld_ad = None, 0
        if addend is not None:
            if neuron == _hip.NEURON_LI_TANH:
                raise RuntimeError("a fused shortcut is not supported after LI+Tanh (its backward reads the output)")
            ctx.addend_acc = _acc_of(addend)
            addend = _raw_to_cl(addend)
            if tuple(addend.shape) != (T, B, C, H, W):
                raise RuntimeError("Residual merge: branch outputs differ in shape")
            if addend.dtype != y.dtype:
                raise RuntimeError("Residual merge: shortcut and branch are stored in different types (fp32 / bf16)")
            ad_ptr, ld_ad = addend.data_ptr(), cl_stride(addend)
        if ckpt:
            _hip.call("snn_lif_fwd_ckpt", y.data_ptr(), ldy, _ptr(alpha), _ptr(beta), _ptr(v0), _ptr(i0),
                      out.data_ptr(), cl_stride(out), ad_ptr, ld_ad, _ptr(vT), _ptr(iT), vdec.data_ptr(), T, M, C,
                      params, st)
        else:
            _hip.call("snn_affine_neuron_fwd", neuron, y.data_ptr(), ldy, _ptr(alpha), _ptr(beta), _ptr(v0), _ptr(i0),
                      _ptr(out), C if out is None else cl_stride(out), ad_ptr, ld_ad, _ptr(vT) if has_state else None,
                      _ptr(iT) if has_state else None, _ptr(vdec), T, M, C, params,
                      (_hip.SCAN_LAST_STEP_ONLY if last_only else 0) | sb_flag
                      | (_hip.SCAN_SPIKES_FROM_VDEC if no_out else 0), st)
            if no_out:
                # what travels to the consumer is an alias of the saved potentials, marked with the threshold that turns
                # them into this layer's output
                out = _alias(vdec, vdec.storage_offset(), T, B, C, H, W, C)
                spikes_out.append(float(params.v_th))
        ctx.ckpt = ckpt
        ctx.sb = sb
        ctx.defer_apply = defer_apply
        ctx.last_only = last_only
        ctx.cfg = (neuron, has_bn, use_running, params, (T, B, C, H, W))
        ctx.slots = (g_slot, b_slot)
        ctx.sync_group = sync_group if (has_bn and not use_running) else None
        ctx.has_v0 = v0 is not None
        ctx.has_i0 = i0 is not None
        if neuron == _hip.NEURON_LI_TANH and need_grad and not is_channels_last(out):
            raise RuntimeError("LI+Tanh output placed in a concat slice is not supported for training")
        state = vdec if neuron in _SAVES_STEP else (out if neuron == _hip.NEURON_LI_TANH else None)
        ctx.save_for_backward(y, gamma, mean, invstd, alpha, beta, state, bias if has_bn else None)
        if not has_state:
            ctx.mark_non_differentiable(vT, iT)
        # with a neuron, vT / iT stay differentiable (time-outer BPTT through the carried state)
        if sb and last_only:
            out = _raw_convert(out, _F32)   # the read-out of the last step (a few frames) leaves the bf16 domain here
        return out, vT, iT

    @staticmethod
    def backward(ctx, g_out, g_vT, g_iT):
        y, gamma, mean, invstd, alpha, beta, state, bn_bias = ctx.saved_tensors
        neuron, has_bn, use_running, params, (T, B, C, H, W) = ctx.cfg
        M = B * H * W
        st = _stream()
        dev = y.device
        has_state = neuron != _hip.NEURON_NONE
        g_addend = None
        if g_out is not None and ctx.needs_input_grad[5]:
            g_addend = g_out
            if ctx.addend_acc is not None:  # lets the shortcut's producer-side dgrad add it in its epilogue
                ctx.addend_acc[0].deposit(ctx.addend_acc[1], g_out)
        sb = ctx.sb
        if g_out is None:
            g_out = torch.zeros((B, H, W, C) if ctx.last_only else (T, B, H, W, C), device=dev, dtype=y.dtype)
            g_out = _cl_view(g_out)
        if g_out.dtype != y.dtype:
            g_out = _raw_convert(g_out, y.dtype)   # (bf16 storage: the fp32 gradient of the last-step read-out)
        g_out = _raw_to_cl(g_out)
        es = y.element_size()
        scan_flags = SCAN_FLAGS | (_hip.SCAN_LAST_STEP_ONLY if ctx.last_only else 0) | (_hip.SCAN_BF16_STORAGE if sb else 0)
        ldg, ldy = cl_stride(g_out), cl_stride(y)
        if not has_state:
            g_vT = g_iT = None
        if g_vT is not None:
            g_vT = _raw_dense_cl(g_vT)
        if g_iT is not None:
            g_iT = _raw_dense_cl(g_iT)
        need_y = ctx.needs_input_grad[0]
        need_gamma = has_bn and gamma is not None and ctx.needs_input_grad[1]
        need_bias = has_bn and ctx.needs_input_grad[2]
        need_sums = has_bn and ((need_y and not use_running) or need_gamma or need_bias)
        gx = torch.empty((T, B, H, W, C), device=dev, dtype=y.dtype)
        g_v0 = _new_cl((B,), C, H, W, y, _F32) if (has_state and ctx.has_v0 and ctx.needs_input_grad[3]) else None
        g_i0 = _new_cl((B,), C, H, W, y, _F32) if (has_state and ctx.has_i0 and ctx.needs_input_grad[4]) else None
        sums = None
        if need_sums:
            n_sums = _hip.query("snn_affine_neuron_bwd_sums_size", T, M, C)
            sums = torch.empty((n_sums,), device=dev, dtype=torch.float64)
        # eval-mode BN has no batch coupling: dy = alpha * gx, applied while gx is written
        apply_scale = 1 if (has_bn and use_running) else 0
        dy = dgamma = dbias = None
        coef = None
        dg_ptr = db_ptr = None
        acc_flag = 0
        if need_sums:
            coef = torch.empty((3, T, C), device=dev, dtype=_F32)
            g_slot, b_slot = ctx.slots
            slotted = (g_slot is not None or not need_gamma) and (b_slot is not None or not need_bias)
            if slotted and (need_gamma or need_bias):
                # gradients go straight into the flat gradient buffer; both share one accumulate flag
                acc_flag = 1 if (g_slot or b_slot).written else 0
                for s_ in (g_slot, b_slot):
                    if s_ is not None:
                        s_.claim()
                dg_ptr = g_slot.buf.data_ptr() if (g_slot is not None and need_gamma) else None
                db_ptr = b_slot.buf.data_ptr() if (b_slot is not None and need_bias) else None
            else:
                dgamma = torch.empty((C,), device=dev, dtype=_F32) if need_gamma else None
                dbias = torch.empty((C,), device=dev, dtype=_F32) if need_bias else None
                dg_ptr, db_ptr = _ptr(dgamma), _ptr(dbias)
        # Long sequences in SEGMENTS of SCAN_SEGMENT_T steps, last segment first.  The scan kernel keeps its per-(t, c)
        # BatchNorm sums for ALL T steps in LDS (one slab per wave); at T = 128 that leaves room for 16 channels per
        # block only, i.e. 64-byte runs per pixel - measured 2.6 TB/s against 4.6 at T = 32.  The recurrence crosses a
        # segment boundary through (g_v, g_i), which the kernel already takes and returns: same values bit for bit.
        segmented = (need_sums and has_state and not ctx.ckpt and ctx.sync_group is None and SCAN_SEGMENT_T
                     and T > SCAN_SEGMENT_T)
        # (segments: every segment must be covered; a segment behind the first one finds the neuron state it starts from
        # in the two saved potentials in front of it - SNN_SCAN_STATE_LOOKBACK)
        sums_from_state = bool(
            USE_SUMS_FROM_STATE and need_sums and not ctx.ckpt and not apply_scale and neuron == _hip.NEURON_LIF
            and not ctx.has_v0 and not ctx.has_i0 and g_v0 is None and g_i0 is None
            and not (segmented and T % SCAN_SEGMENT_T == 1)    # (a second segment starting at step 1 has one step to look back on)
            and all(_hip.query("snn_affine_neuron_bwd_sums_from_state", neuron, ts_, M, C, ldg, params, scan_flags)
                    for ts_ in ({SCAN_SEGMENT_T, T % SCAN_SEGMENT_T or SCAN_SEGMENT_T} if segmented else {T})))
        if segmented:
            fr_g, fr_y, fr_c = M * ldg * es, M * ldy * es, M * C * es    # bytes per timestep of g_out / y / dense tensors
            gv_in, gi_in = g_vT, g_iT
            first = True
            g_none = None   # last_only: the output gradient of every segment but the last one is zero
            for t1 in range(T, 0, -SCAN_SEGMENT_T):
                t0 = max(0, t1 - SCAN_SEGMENT_T)
                ts = t1 - t0
                last = t0 == 0
                gv_out = g_v0 if (last and g_v0 is not None) else torch.empty((B, H, W, C), device=dev, dtype=_F32)
                gi_out = g_i0 if (last and g_i0 is not None) else torch.empty((B, H, W, C), device=dev, dtype=_F32)
                n_sums = _hip.query("snn_affine_neuron_bwd_sums_size", ts, M, C)
                seg_sums = torch.empty((n_sums,), device=dev, dtype=torch.float64)
                tc = t0 * C * 4
                if not ctx.last_only:
                    g_seg = g_out.data_ptr() + t0 * fr_g
                elif t1 == T:
                    g_seg = g_out.data_ptr()          # [B,H,W,C] of the last step = the last step of this segment
                else:
                    if g_none is None:
                        g_none = torch.zeros_like(g_out)
                    g_seg = g_none.data_ptr()
                st_off = 0 if (ctx.last_only and neuron == _hip.NEURON_LI_TANH) else t0 * fr_c   # saved output: last step only
                seg_flags = scan_flags
                if sums_from_state:
                    seg_flags |= _hip.SCAN_SUMS_FROM_STATE | (_hip.SCAN_STATE_LOOKBACK if t0 > 0 else 0)
                _hip.call("snn_affine_neuron_bwd", neuron, g_seg, ldg,
                          None if state is None else state.data_ptr() + st_off, y.data_ptr() + t0 * fr_y, ldy,
                          _ptr(gv_in), _ptr(gi_in), None if alpha is None else alpha.data_ptr() + tc,
                          None if beta is None else beta.data_ptr() + tc, apply_scale, gx.data_ptr() + t0 * fr_c,
                          gv_out.data_ptr(), gi_out.data_ptr(), seg_sums.data_ptr(), ts, M, C, params, seg_flags, st)
                if sums_from_state:
                    _hip.call("snn_bn_bwd_finalize_from_state", seg_sums.data_ptr(), ts, M, C, _ptr(gamma), _ptr(bn_bias),
                              mean.data_ptr() + tc, invstd.data_ptr() + tc, gx.data_ptr() + t0 * fr_c,
                              y.data_ptr() + t0 * fr_y, ldy, coef[0].data_ptr() + tc, coef[1].data_ptr() + tc,
                              coef[2].data_ptr() + tc, dg_ptr, db_ptr, acc_flag if first else 1, st)
                else:
                    _hip.call("snn_bn_bwd_finalize", seg_sums.data_ptr(), ts, M, C, _ptr(gamma), mean.data_ptr() + tc,
                              invstd.data_ptr() + tc, coef[0].data_ptr() + tc, coef[1].data_ptr() + tc,
                              coef[2].data_ptr() + tc, dg_ptr, db_ptr, acc_flag if first else 1, st)
                gv_in, gi_in, first = gv_out, gi_out, False
        if segmented:
            pass
        elif sums_from_state:
            # y is not read: the statistic comes from the neuron input rebuilt from the saved potentials
            _hip.call("snn_affine_neuron_bwd", neuron, g_out.data_ptr(), ldg, _ptr(state), None, ldy,
                      _ptr(g_vT), _ptr(g_iT), _ptr(alpha), _ptr(beta), 0, gx.data_ptr(), None, None, _ptr(sums), T, M, C,
                      params, scan_flags | _hip.SCAN_SUMS_FROM_STATE, st)
        elif ctx.ckpt:
            _hip.call("snn_lif_bwd_ckpt", g_out.data_ptr(), ldg, state.data_ptr(), y.data_ptr(), ldy, _ptr(g_vT),
                      _ptr(g_iT), _ptr(alpha), _ptr(beta), apply_scale, gx.data_ptr(), _ptr(g_v0), _ptr(g_i0),
                      _ptr(sums), T, M, C, params, st)
        else:
            _hip.call("snn_affine_neuron_bwd", neuron, g_out.data_ptr(), ldg, _ptr(state), y.data_ptr(), ldy,
                      _ptr(g_vT), _ptr(g_iT), _ptr(alpha), _ptr(beta), apply_scale, gx.data_ptr(), _ptr(g_v0),
                      _ptr(g_i0), _ptr(sums), T, M, C, params, scan_flags, st)
        if need_sums:
            if segmented:
                pass   # coefficients and parameter gradients were finalised per segment
            elif sums_from_state and ctx.sync_group is None:
                _hip.call("snn_bn_bwd_finalize_from_state", sums.data_ptr(), T, M, C, _ptr(gamma), _ptr(bn_bias),
                          mean.data_ptr(), invstd.data_ptr(), gx.data_ptr(), y.data_ptr(), ldy, coef[0].data_ptr(),
                          coef[1].data_ptr(), coef[2].data_ptr(), dg_ptr, db_ptr, acc_flag, st)
            elif ctx.sync_group is None and not sums_from_state:
                _hip.call("snn_bn_bwd_finalize", sums.data_ptr(), T, M, C, _ptr(gamma), mean.data_ptr(),
                          invstd.data_ptr(), coef[0].data_ptr(), coef[1].data_ptr(), coef[2].data_ptr(), dg_ptr,
                          db_ptr, acc_flag, st)
            else:
                # SyncBatchNorm backward: dy needs the GLOBAL sums (one all-reduce of [T, C, 2]); the parameter
                # gradients use the rank-local sums - the data-parallel all-reduce averages them afterwards
                import torch.distributed as dist
                world = dist.get_world_size(ctx.sync_group[0])
                raw_local = torch.empty((T, C, 2), device=dev, dtype=torch.float64)
                if sums_from_state:
                    _hip.call("snn_bn_bwd_reduce_from_state", sums.data_ptr(), T, M, C, _ptr(gamma), _ptr(bn_bias),
                              mean.data_ptr(), invstd.data_ptr(), gx.data_ptr(), y.data_ptr(), ldy, raw_local.data_ptr(), st)
                else:
                    _hip.call("snn_bn_bwd_reduce", sums.data_ptr(), T, M, C, raw_local.data_ptr(), st)
                raw = raw_local.clone()
                dist.all_reduce(raw, op=dist.ReduceOp.SUM, group=ctx.sync_group[0])
                param_sums = torch.empty((T, C, 2), device=dev, dtype=torch.float64)
                _hip.call("snn_bn_bwd_coef", raw.data_ptr(), raw_local.data_ptr(), param_sums.data_ptr(), T, M * world,
                          C, _ptr(gamma), mean.data_ptr(), invstd.data_ptr(), coef[0].data_ptr(), coef[1].data_ptr(),
                          coef[2].data_ptr(), dg_ptr, db_ptr, acc_flag, st)
            if need_y and not use_running and ctx.defer_apply and ctx.sync_group is None:
                # the producing convolution forms dy itself while it reads gx (see PendingBnApply)
                _PENDING_APPLY[gx.data_ptr()] = PendingBnApply(gx, y, coef, (T, B, C, H, W))
            elif need_y and not use_running:
                # in place: dy overwrites gx
                _hip.call("snn_bn_bwd_apply_bf16" if sb else "snn_bn_bwd_apply", gx.data_ptr(), y.data_ptr(), ldy,
                          coef[0].data_ptr(), coef[1].data_ptr(), coef[2].data_ptr(), gx.data_ptr(), C, T, M, C, 0, st)
        if need_y:
            dy = _cl_view(gx)
        return dy, dgamma, dbias, g_v0, g_i0, g_addend, None


def _expand_state(s: torch.Tensor, shape, dev) -> torch.Tensor:
    """State tensors may be 0-dim (LICell's initial v) or NCHW; the kernel wants dense [B,H,W,C]."""
    s = s.detach()
    if s.dim() == 0 or tuple(s.shape) != tuple(shape):
        s = s.to(device=dev, dtype=_F32).expand(shape)
    return _raw_dense_cl(s)


def affine_neuron(y: torch.Tensor, neuron: int, state: Optional[NeuronState] = None, bn=None,
                  params: Optional[NeuronParams] = None, dest: Optional[Dest] = None,
                  addend: Optional[torch.Tensor] = None, last_only: bool = False, spikes_ok: bool = False):
    """Fused ``[Norm] -> [neuron] [+ addend]`` over a sequence or a single step.

    ``spikes_ok`` (LIF on a sequence; the caller guarantees that the ONLY consumer is ``sibling_conv1x1``): the result may be
    the layer's saved potentials instead of its spikes, marked ``_snn_spike_threshold`` - no spike tensor is written.

    ``bn`` is an ``nn.BatchNorm2d``-like module (weight, bias, running stats, eps, momentum, training)
    or None; ``addend`` (same shape as the output) is a residual shortcut added in the output store.
    ``last_only`` (LI / LI+Tanh on a sequence): return the output of the LAST timestep only, ``[B,C,H,W]`` - the other
    T-1 outputs are never written and the backward pass reads no output gradient for them.
    Returns ``(out, NeuronState | None)``.
    """
    if y.dtype == _BF16 and (neuron not in (_hip.NEURON_NONE, _hip.NEURON_LIF, _hip.NEURON_LI, _hip.NEURON_LI_TANH)
                             or y.shape[-3] % 4):
        # no bf16-storage form of this scan (SLI / Synapse, channel counts that are not a multiple of 4): see _through_fp32
        out, new_state = affine_neuron(to_float32(y), neuron, state, bn, params, None,
                                       None if addend is None else to_float32(addend), last_only, False)
        out = out if (last_only and y.dim() == 5) else to_bfloat16(out)
        return (place(out, dest) if dest is not None else out), new_state
    bn_hint = getattr(y, "_snn_bn_partial", None)
    defer_apply = bool(getattr(y, "_snn_defer_apply", False))
    seq, single = as_sequence(y)
    if addend is not None:
        acc = _acc_of(addend)
        addend, _ = as_sequence(addend)
        if acc is not None:
            addend._snn_acc = acc
    params = params or neuron_params()
    has_bn = bn is not None
    gamma = bias = rm = rv = None
    training, eps, momentum = False, 1e-5, 0.1
    if has_bn:
        gamma, bias = bn.weight, bn.bias
        training = bn.training or (bn.running_mean is None and bn.running_var is None)
        rm, rv = bn.running_mean, bn.running_var
        eps = bn.eps
        if bn.momentum is None:
            raise RuntimeError("cumulative-average BatchNorm (momentum=None) is not supported")
        momentum = bn.momentum
        if training and seq.shape[1] * seq.shape[3] * seq.shape[4] <= 1:
            # torch.nn.functional.batch_norm's own check (the reference's per-timestep BatchNorm2d raises it): one value
            # per channel has no variance to normalise with
            raise ValueError("Expected more than 1 value per channel when training, got input size "
                             f"{torch.Size((seq.shape[1], seq.shape[2], seq.shape[3], seq.shape[4]))}")
        if training and bn.num_batches_tracked is not None:
            if _NBT_BATCH is not None:
                _NBT_BATCH.append((bn.num_batches_tracked, int(seq.shape[0])))  # one fused update per forward
            else:
                bn.num_batches_tracked.add_(seq.shape[0])
    v0 = i0 = None
    if state is not None:
        if neuron == _hip.NEURON_SYNAPSE:
            (v0,) = state
        else:
            v0, i0 = state
    sync_group = getattr(bn, "_snn_sync_group", None) if has_bn else None
    spikes_out = [] if (spikes_ok and not single and dest is None and addend is None) else None
    cfg = (neuron, has_bn, training, float(eps), float(momentum), rm, rv, params, _slot_of(gamma), _slot_of(bias),
           dest, sync_group, bn_hint, bool(last_only) and not single, defer_apply, spikes_out)
    out, vT, iT = _AffineNeuron.apply(seq, gamma, bias, v0, i0, addend, cfg)
    if spikes_out:
        out._snn_spike_threshold = spikes_out[0]   # `out` holds v_dec: its consumer thresholds on load
    if neuron == _hip.NEURON_NONE:
        new_state = None
    elif neuron == _hip.NEURON_SYNAPSE:
        new_state = SynapseState(vT)
    else:
        new_state = NeuronState(vT, iT)
    if last_only and not single:
        return out, new_state
    return (out[0] if single else out), new_state


# ------------------------------------------------------------------------------------------- merges
def _copy_cl(src: torch.Tensor, dst: torch.Tensor) -> None:
    """dst = src, both (possibly channel-sliced) channels-last ``[T,B,C,H,W]``."""
    if src.dtype != dst.dtype:
        raise RuntimeError("merge: operands are stored in different types (fp32 / bf16 storage mixed)")
    T, B, C, H, W = _dims5(src)
    _hip.call("snn_copy_channels_bf16" if src.dtype == _BF16 else "snn_copy_channels", src.data_ptr(), cl_stride(src),
              dst.data_ptr(), cl_stride(dst), T * B * H * W, C, _stream())


def _add_cl(a: torch.Tensor, b: torch.Tensor, dst: torch.Tensor) -> None:
    """dst = a + b over channels-last ``[T,B,C,H,W]`` operands with their own pixel strides (dst may be a)."""
    if not (a.dtype == b.dtype == dst.dtype):
        raise RuntimeError("merge: operands are stored in different types (fp32 / bf16 storage mixed)")
    T, B, C, H, W = _dims5(a)   # (bf16 storage: the sum is formed in fp32 and rounded once)
    _hip.call("snn_add_bf16" if a.dtype == _BF16 else "snn_add", a.data_ptr(), cl_stride(a), b.data_ptr(), cl_stride(b),
              dst.data_ptr(), cl_stride(dst), T * B * H * W, C, _stream())


class _Concat(Function):
    """Dense merge: torch.cat(out, dim=1) per timestep (generator.py:157-158), copying form."""

    @staticmethod
    def forward(ctx, dest, *xs):
        xs = [_raw_to_cl(x) for x in xs]
        T, B, _, H, W = _dims5(xs[0])
        widths = [x.shape[2] for x in xs]
        Ct = sum(widths)
        out = _out_tensor(dest, T, B, Ct, H, W, xs[0])
        off = 0
        for x, c in zip(xs, widths):
            _require_device(x, "concat input", bf16_ok=True)
            if x.shape[0] != T or x.shape[1] != B or x.shape[3] != H or x.shape[4] != W:
                raise RuntimeError("Dense merge: branch outputs differ in shape")
            _copy_cl(x, out.narrow(2, off, c))
            off += c
        ctx.widths = widths
        return out

    @staticmethod
    def backward(ctx, g):
        g = _raw_to_cl(g)
        grads, off = [None], 0
        for k, c in enumerate(ctx.widths):
            grads.append(g.narrow(2, off, c) if ctx.needs_input_grad[k + 1] else None)  # channel-slice views
            off += c
        return tuple(grads)


class _ConcatAssemble(Function):
    """Dense merge without a copy: every branch output already IS its channel slice of ``whole``
    (``ConcatPromise``); this node only ties the autograd graph together.  Backward hands each branch
    the matching channel-slice VIEW of the incoming gradient (the kernels take a pixel stride)."""

    @staticmethod
    def forward(ctx, whole, *xs):
        ctx.widths = [x.shape[2] for x in xs]
        ctx.accs = [_acc_of(x) for x in xs]
        T, B, C, H, W = _dims5(whole)
        return _alias(whole, whole.storage_offset(), T, B, C, H, W, cl_stride(whole))

    @staticmethod
    def backward(ctx, g):
        g = _raw_to_cl(g)
        grads, off = [None], 0
        for k, c in enumerate(ctx.widths):
            gk = g.narrow(2, off, c) if ctx.needs_input_grad[k + 1] else None
            if gk is not None and ctx.accs[k] is not None:
                # a channel range of g that only branch k is handed: whoever accumulates that branch's gradient may
                # write the total over it
                ctx.accs[k][0].deposit(ctx.accs[k][1], gk, exclusive=True)
            grads.append(gk)
            off += c
        return tuple(grads)


class _Sum(Function):
    """Residual merge: torch.stack(out).sum(0) (generator.py:145-146)."""

    @staticmethod
    def forward(ctx, dest, *xs):
        xs = [_raw_to_cl(x) for x in xs]
        for x in xs:
            _require_device(x, "residual input", bf16_ok=True)
            if x.shape != xs[0].shape:
                raise RuntimeError("Residual merge: branch outputs differ in shape")
        T, B, C, H, W = _dims5(xs[0])
        ctx.accs = [_acc_of(x) for x in xs]
        out = _out_tensor(dest, T, B, C, H, W, xs[0])
        _add_cl(xs[0], xs[1], out)
        for x in xs[2:]:
            _add_cl(out, x, out)
        return out

    @staticmethod
    def backward(ctx, g):
        for acc, need in zip(ctx.accs, ctx.needs_input_grad[1:]):
            if acc is not None and need:
                acc[0].deposit(acc[1], g)
        return (None,) + tuple(g if need else None for need in ctx.needs_input_grad[1:])


class _Fanout(Function):
    """One tensor consumed by ``n`` branches of a block (generator.py:181-187 hands every branch the same X).

    Forward returns ``n`` aliases (no copy); backward adds the branch gradients with the strided add
    kernel, so gradient accumulation stays on this library's kernels and accepts channel-sliced grads."""

    @staticmethod
    def forward(ctx, x, n: int):
        ctx.acc = GradAccumulator()
        ctx.outer = ctx.acc.outer = _acc_of(x)
        ctx.acc.grad_in_slot = bool(getattr(x, "_snn_grad_in_slot", False))
        outs = []
        for k in range(n):
            t = torch.empty(0, device=x.device, dtype=x.dtype)
            t.set_(x.untyped_storage(), x.storage_offset(), x.size(), x.stride())
            t._snn_acc = (ctx.acc, k)
            outs.append(t)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        acc = ctx.acc
        total = None
        if acc.result is not None:
            # a data-gradient convolution already holds (its own + the fused deposits'); add only what is missing
            total = acc.result
            _await_mark(acc, "result", total)   # (it may have been written on another stream than this node's)
            for k, g in enumerate(gs):
                if g is None:
                    continue
                if k in acc.fused:
                    if not _same_tensor(g, acc.fused[k]):
                        raise RuntimeError("fanout: a branch input received a gradient the fused accumulation did "
                                           "not expect (the alias was consumed more than once)")
                    continue
                gk = _raw_to_cl(g)
                if total.dim() == 4:
                    _add_cl(total.unsqueeze(0), gk.unsqueeze(0), total.unsqueeze(0))
                else:
                    _add_cl(total, gk, total)
        else:
            live = [g for g in gs if g is not None]
            if not live:
                return None, None
            if len(live) == 1:
                total = live[0]
            else:
                single = live[0].dim() == 4
                seqs = [_raw_to_cl(g.unsqueeze(0) if single else g) for g in live]
                T, B, C, H, W = _dims5(seqs[0])
                out = _new_cl((T, B), C, H, W, seqs[0])
                _add_cl(seqs[0], seqs[1], out)
                for g in seqs[2:]:
                    _add_cl(out, g, out)
                total = out[0] if single else out
        acc.deposits.clear()
        acc.fused.clear()
        acc.exclusive.clear()
        acc.marks.clear()
        acc.result = None
        if ctx.outer is not None:
            ctx.outer[0].deposit(ctx.outer[1], total)
        return total, None


def fanout(x: torch.Tensor, n: int):
    """``n`` aliases of ``x`` for the ``n`` branches of a block; their gradients are summed by a HIP kernel."""
    if n == 1 or not x.requires_grad:
        return [x] * n
    return list(_Fanout.apply(x, n))


class _Place(Function):
    """Copy a tensor into a destination slice (fallback when a producer could not write there itself)."""

    @staticmethod
    def forward(ctx, x, dest):
        ctx.acc = _acc_of(x)
        x = _raw_to_cl(x)
        T, B, C, H, W = _dims5(x)
        out = dest.tensor(T, B, C, H, W, x)
        _copy_cl(x, out)
        return out

    @staticmethod
    def backward(ctx, g):
        if ctx.acc is not None:
            ctx.acc[0].deposit(ctx.acc[1], g)
        return g, None


def place(x: torch.Tensor, dest: Dest) -> torch.Tensor:
    """``x`` as the channel slice ``dest`` of a concat buffer (no-op when it already lives there)."""
    if dest.holds(x):
        return x
    return _Place.apply(x, dest)


def concat_channels(xs: List[torch.Tensor], dest: Optional[Dest] = None) -> torch.Tensor:
    if len(xs) == 1 and dest is None:
        return xs[0]
    seqs = [as_sequence(x) for x in xs]
    out = _Concat.apply(dest, *[s for s, _ in seqs])
    return out[0] if seqs[0][1] else out


def assemble_channels(promise: ConcatPromise, xs: List[torch.Tensor]) -> torch.Tensor:
    """Zero-copy Dense merge of branch outputs that were produced inside ``promise``'s buffer."""
    return _ConcatAssemble.apply(promise.buf, *xs)


def sum_tensors(xs: List[torch.Tensor], dest: Optional[Dest] = None) -> torch.Tensor:
    if len(xs) == 1:
        return xs[0] if dest is None else place(as_sequence(xs[0])[0], dest)
    seqs = [as_sequence(x) for x in xs]
    out = _Sum.apply(dest, *[s for s, _ in seqs])
    return out[0] if seqs[0][1] else out


class _GradReady(Function):
    """Identity whose backward first calls ``fn()``: marks the point of the backward pass at which every gradient
    produced downstream of ``x`` (in forward order) is complete or enqueued - ``FlatTrainer.early_all_reduce``."""

    @staticmethod
    def forward(ctx, x, fn):
        ctx.fn = fn
        acc = _acc_of(x)
        out = torch.empty(0, device=x.device, dtype=x.dtype)
        out.set_(x.untyped_storage(), x.storage_offset(), x.size(), x.stride())
        if acc is not None:
            out._snn_acc = acc
        return out

    @staticmethod
    def backward(ctx, g):
        ctx.fn()
        return g, None


def grad_ready_hook(x: torch.Tensor, fn) -> torch.Tensor:
    """``x`` unchanged; during the backward pass ``fn()`` runs when the gradient of ``x`` arrives."""
    if fn is None or not x.requires_grad:
        return x
    return _GradReady.apply(x, fn)


# ------------------------------------------------------------------------------------------- pointwise / pooling
class _Act(Function):
    @staticmethod
    def forward(ctx, x, act: int):
        _require_device(x, "activation input")
        x = _raw_dense_cl(x)
        y = _cl_view(torch.empty(x.permute(0, 1, 3, 4, 2).shape, device=x.device, dtype=_F32))
        _hip.call("snn_act_fwd", act, x.data_ptr(), y.data_ptr(), x.numel(), _stream())
        ctx.act = act
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y = ctx.saved_tensors
        gy = _raw_dense_cl(gy)
        gx = _cl_view(torch.empty(x.permute(0, 1, 3, 4, 2).shape, device=x.device, dtype=_F32))
        _hip.call("snn_act_bwd", ctx.act, x.data_ptr(), y.data_ptr(), gy.data_ptr(), gx.data_ptr(), x.numel(),
                  _stream())
        return gx, None


def _raw_convert(x: torch.Tensor, dtype) -> torch.Tensor:
    """fp32 <-> bf16 copy of a tensor in the same memory order (snn_convert_bf16; round to nearest even).  Dense
    channels-last or contiguous tensors are converted in place order; anything else is made dense channels-last first."""
    if x.dtype == dtype:
        return x
    _require_device(x, "storage conversion", bf16_ok=True)
    if x.dim() >= 3 and not x.is_contiguous():
        x = _raw_dense_cl(x)
        out = _new_cl(x.shape[:-3], *x.shape[-3:], x, dtype)
    else:
        x = x.contiguous()
        out = torch.empty(x.shape, device=x.device, dtype=dtype)
    if x.numel():
        _hip.call("snn_convert_bf16", x.data_ptr(), out.data_ptr(), x.numel(), 1 if dtype == _BF16 else 0, _stream())
    return out


class _Convert(Function):
    """Leaves / enters the bf16-storage domain: the gradient crosses the boundary the other way."""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src_dtype = x.dtype
        return _raw_convert(x, dtype)

    @staticmethod
    def backward(ctx, g):
        return _raw_convert(g, ctx.src_dtype), None


def to_float32(x: torch.Tensor) -> torch.Tensor:
    return x if x.dtype == _F32 else _Convert.apply(x, _F32)


def to_bfloat16(x: torch.Tensor) -> torch.Tensor:
    return x if x.dtype == _BF16 else _Convert.apply(x, _BF16)


def _through_fp32(op, x: torch.Tensor, *args):
    """Operators without a bf16-storage kernel (activations, pooling, up-sampling, ConvLSTM, SLI / Synapse: none of them is
    on the TinyYolo path) still work in that mode: the tensor is widened to fp32 for the operator and its result is
    narrowed again (two conversion passes - correct, not fast)."""
    if x.dtype != _BF16:
        return op(x, *args)
    return to_bfloat16(op(to_float32(x), *args))


def activation(x: torch.Tensor, act: int) -> torch.Tensor:
    def run(t, a):
        seq, single = as_sequence(t)
        y = _Act.apply(seq, a)
        return y[0] if single else y
    return _through_fp32(run, x, act)


class _Pool(Function):
    @staticmethod
    def forward(ctx, x, kind: int, k: int, stride: int):
        _require_device(x, "pool input")
        x = _raw_dense_cl(x)
        T, B, C, H, W = _dims5(x)
        Ho, Wo = (H - k) // stride + 1, (W - k) // stride + 1
        if Ho <= 0 or Wo <= 0:
            raise RuntimeError(f"pool: window {k} larger than input {H}x{W}")
        y = _new_cl((T, B), C, Ho, Wo, x)
        _hip.call("snn_pool_fwd", kind, x.data_ptr(), y.data_ptr(), T * B, H, W, C, Ho, Wo, k, stride, _stream())
        ctx.geom = (kind, k, stride, T, B, C, H, W, Ho, Wo)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        kind, k, stride, T, B, C, H, W, Ho, Wo = ctx.geom
        gy = _raw_dense_cl(gy)
        gx = _new_cl((T, B), C, H, W, x)
        _hip.call("snn_pool_bwd", kind, x.data_ptr(), gy.data_ptr(), gx.data_ptr(), T * B, H, W, C, Ho, Wo, k, stride,
                  _stream())
        return gx, None, None, None


def pool2d(x: torch.Tensor, kind: int, kernel_size: int, stride: int) -> torch.Tensor:
    def run(t, *a):
        seq, single = as_sequence(t)
        y = _Pool.apply(seq, *a)
        return y[0] if single else y
    return _through_fp32(run, x, kind, int(kernel_size), int(stride))


class _Upsample(Function):
    @staticmethod
    def forward(ctx, x, scale: int):
        _require_device(x, "upsample input")
        x = _raw_dense_cl(x)
        T, B, C, H, W = _dims5(x)
        y = _new_cl((T, B), C, H * scale, W * scale, x)
        _hip.call("snn_upsample_fwd", x.data_ptr(), y.data_ptr(), T * B, H, W, C, scale, _stream())
        ctx.geom = (scale, T, B, C, H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        scale, T, B, C, H, W = ctx.geom
        gy = _raw_dense_cl(gy)
        gx = _new_cl((T, B), C, H, W, gy)
        _hip.call("snn_upsample_bwd", gy.data_ptr(), gx.data_ptr(), T * B, H, W, C, scale, _stream())
        return gx, None


def upsample_nearest(x: torch.Tensor, scale: int) -> torch.Tensor:
    def run(t, sc):
        seq, single = as_sequence(t)
        y = _Upsample.apply(seq, sc)
        return y[0] if single else y
    return _through_fp32(run, x, int(scale))


# ------------------------------------------------------------------------------------------- ConvLSTM
class _LstmCell(Function):
    """Pointwise LSTM update on the gate pre-activations (conv_lstm.py:66-76); one timestep ``[B,*,H,W]``."""

    @staticmethod
    def forward(ctx, gates, c_prev):
        _require_device(gates, "lstm gates")
        gates = _raw_dense_cl(gates)
        B, C4, H, W = gates.shape
        C = C4 // 4
        if c_prev is not None:
            c_prev = _raw_dense_cl(c_prev)
        h = _new_cl((B,), C, H, W, gates)
        c = _new_cl((B,), C, H, W, gates)
        _hip.call("snn_lstm_cell_fwd", gates.data_ptr(), _ptr(c_prev), h.data_ptr(), c.data_ptr(), B * H * W, C,
                  _stream())
        ctx.save_for_backward(gates, c_prev, c)
        return h, c

    @staticmethod
    def backward(ctx, gh, gc):
        gates, c_prev, c = ctx.saved_tensors
        B, C4, H, W = gates.shape
        C = C4 // 4
        gh = _raw_dense_cl(gh) if gh is not None else None
        gc = _raw_dense_cl(gc) if gc is not None else None
        g_gates = _new_cl((B,), C4, H, W, gates)
        g_cp = _new_cl((B,), C, H, W, gates) if (c_prev is not None and ctx.needs_input_grad[1]) else None
        _hip.call("snn_lstm_cell_bwd", gates.data_ptr(), _ptr(c_prev), c.data_ptr(), _ptr(gh), _ptr(gc),
                  g_gates.data_ptr(), _ptr(g_cp), B * H * W, C, _stream())
        return g_gates, g_cp


def lstm_cell(gates: torch.Tensor, c_prev: Optional[torch.Tensor]):
    if gates.dtype == _BF16:   # (see _through_fp32) the cell state c stays fp32 across the steps
        h, c = _LstmCell.apply(to_float32(gates), None if c_prev is None else to_float32(c_prev))
        return to_bfloat16(h), c
    return _LstmCell.apply(gates, c_prev)


class _StackTime(Function):
    """``torch.stack`` over per-timestep ``[B,C,H,W]`` results into one channels-last ``[T,B,C,H,W]``."""

    @staticmethod
    def forward(ctx, *xs):
        xs = [_raw_to_cl(x) for x in xs]
        B, C, H, W = xs[0].shape
        out = _new_cl((len(xs), B), C, H, W, xs[0])
        M, st = B * H * W, _stream()
        for t, x in enumerate(xs):
            _hip.call("snn_copy_channels", x.data_ptr(), cl_stride(x), out[t].data_ptr(), C, M, C, st)
        return out

    @staticmethod
    def backward(ctx, g):
        return tuple(g[t] for t in range(g.shape[0]))


def stack_time(xs: List[torch.Tensor]) -> torch.Tensor:
    if xs and xs[0].dtype == _BF16:
        return torch.stack(list(xs))   # (time-outer loops are off the layer-major path) torch's copy
    return _StackTime.apply(*xs)


# ------------------------------------------------------------------------------------------- detection loss
class _DetectionLoss(Function):
    """``loss_ratio * mean(CE[pos]) + (1 - loss_ratio) * mean(CE[neg]) + mean(L1(bbox*mask, offset*mask))``
    (models/soda.py:259-281) and its gradient: two kernel launches forward, one backward, no host round trip."""

    @staticmethod
    def forward(ctx, cls_preds, bbox_preds, bbox_offset, bbox_mask, class_labels, loss_ratio: float):
        _require_device(cls_preds, "loss: class predictions")
        _require_device(bbox_preds, "loss: box predictions")
        K = cls_preds.shape[-1]
        logits, boxes = cls_preds.detach().contiguous(), bbox_preds.detach().contiguous()
        off, msk = bbox_offset.contiguous(), bbox_mask.contiguous()
        lab = class_labels.contiguous()
        rows = lab.numel()
        if logits.numel() != rows * K or boxes.numel() != rows * 4 or off.numel() != rows * 4 or lab.dtype != torch.int64:
            raise RuntimeError("detection loss: predictions and targets differ in shape")
        dev = logits.device
        ws = torch.empty(_hip.query("snn_det_loss_workspace_size", rows), device=dev, dtype=torch.uint8)
        stats = torch.empty(5, device=dev, dtype=torch.float64)
        loss = torch.empty((), device=dev, dtype=_F32)
        _hip.call("snn_det_loss_fwd", logits.data_ptr(), boxes.data_ptr(), off.data_ptr(), msk.data_ptr(),
                  lab.data_ptr(), rows, K, float(loss_ratio), ws.data_ptr(), stats.data_ptr(), loss.data_ptr(), _stream())
        ctx.save_for_backward(logits, boxes, off, msk, lab, stats)
        ctx.loss_ratio = float(loss_ratio)
        ctx.shapes = (cls_preds.shape, bbox_preds.shape)
        return loss

    @staticmethod
    def backward(ctx, g):
        logits, boxes, off, msk, lab, stats = ctx.saved_tensors
        rows, K = lab.numel(), logits.shape[-1]
        g = g.detach().to(_F32).contiguous()
        g_logits, g_boxes = torch.empty_like(logits), torch.empty_like(boxes)
        _hip.call("snn_det_loss_bwd", logits.data_ptr(), boxes.data_ptr(), off.data_ptr(), msk.data_ptr(),
                  lab.data_ptr(), rows, K, ctx.loss_ratio, stats.data_ptr(), g.data_ptr(), g_logits.data_ptr(),
                  g_boxes.data_ptr(), _stream())
        return g_logits.view(ctx.shapes[0]), g_boxes.view(ctx.shapes[1]), None, None, None, None


def detection_loss(cls_preds: torch.Tensor, bbox_preds: torch.Tensor, bbox_offset: torch.Tensor,
                   bbox_mask: torch.Tensor, class_labels: torch.Tensor, loss_ratio: float) -> torch.Tensor:
    return _DetectionLoss.apply(cls_preds, bbox_preds, bbox_offset, bbox_mask, class_labels, loss_ratio)


# ------------------------------------------------------------------------------------------- events
def events_to_frames(t_bin: torch.Tensor, x: torch.Tensor, y: torch.Tensor, p: torch.Tensor, T: int, H: int,
                     W: int) -> torch.Tensor:
    """Scatter events into binary frames ``[T, 2, H, W]`` (utils/datasets.py:378-435), on device."""
    for t in (t_bin, x, y, p):
        if not t.is_cuda or t.dtype != torch.int32:
            raise RuntimeError("events_to_frames: int32 HIP tensors required")
    buf = torch.empty((T, H, W, 2), device=t_bin.device, dtype=_F32)
    _hip.call("snn_events_to_frames", t_bin.data_ptr(), x.data_ptr(), y.data_ptr(), p.data_ptr(), t_bin.numel(),
              buf.data_ptr(), T, H, W, _stream())
    return buf.permute(0, 3, 1, 2)
