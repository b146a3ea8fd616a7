"""Per-kernel timing with HIP events + the algorithmic work of every C-ABI launch.

``bench.py`` installs a ``KernelProfiler`` for a few extra steps after its timed region: each
``_hip.call`` is bracketed by two events recorded on the stream the kernel is launched on (torch's
current stream), and its algorithmic FLOPs / bytes are derived from the call's own arguments with
the work model of SURVEY section 8(d):

* conv (fwd, dgrad, wgrad): ``2 * M * Cout * KH*KW*Cin`` FLOPs each (M = output pixels; dgrad and
  wgrad each count as one forward); bytes = every operand read / written once;
* fused norm+neuron scans: ``4 B`` per element per tensor the ideal fused kernel touches.

Labels name the kernel template instance that ``csrc/conv.hip`` dispatches for the shape, so the
rows can be matched with ``rocprofv3 --kernel-trace --stats`` output.
"""

from collections import defaultdict
from typing import Dict

import torch


WGRAD_LABELS = ("k_conv_wgrad_pipe", "k_conv_wgrad_halo", "k_conv_first<wgrad>")   # snn_conv2d_wgrad_kernel's 0, 1, 2


def _wgrad_label(n, h, w, cin, ho, wo, cout, kh, kw, stride, pad, prec) -> str:
    """The three weight-gradient kernels are different programs with different roofs (the implicit GEMM of the 1x1 layers is
    HBM-bound, the halo-resident 3x3 kernel matrix-bound, the event-frame row kernel streams): one label each (+ its
    ordered split-K reduce).  Round 3 priced them as ONE family."""
    from . import _hip
    return WGRAD_LABELS[int(_hip.query("snn_conv2d_wgrad_kernel", n, h, w, cin, ho, wo, cout, kh, kw, stride, pad, prec))]


def _conv_label(name: str, a) -> str:
    if name == "snn_conv2d_wgrad":
        return _wgrad_label(a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15], a[19])
    # fwd / dgrad share k_conv_gather<BN, WM, WN, DGRAD, VEC>
    dgrad = name == "snn_conv2d_dgrad"
    cin, cout = a[8], a[11]
    oc, ic = (cin, cout) if dgrad else (cout, cin)
    if not dgrad and cin == 2 and a[12] == 3 and a[13] == 3 and cout % 4 == 0 and (cout // 4) & (cout // 4 - 1) == 0 \
            and cout <= 256 and a[16] is None:
        return "k_conv_first<2, 3, false>"  # direct row kernel of the event-frame layer
    if a[12] == 3 and a[13] == 3 and a[14] == 1 and a[15] == 1 and oc <= 32 and oc % 4 == 0 and ic % 32 == 0:
        return f"k_conv_direct3<32, 4, 1, {'true' if dgrad else 'false'}>"  # halo-resident 3x3 kernel
    if oc <= 32:
        tile = "32, 4, 1"
    elif oc <= 64:
        tile = "64, 2, 2"
    else:
        tile = "128, 2, 2"
    return f"k_conv_gather<{tile}, {'true' if dgrad else 'false'}, {'true' if ic % 4 == 0 else 'false'}>"


PREC_BF16S, SCAN_BF16_STORAGE = 6, 4   # include/snn_hip.h: the bf16-storage mode (activation tensors 2 bytes per element)


def work_of(name: str, a):
    """-> (label, flops, bytes) for one launch, or (name, 0, 0) for bookkeeping kernels.  Activation tensors count 4 bytes
    per element, or 2 in the bf16-storage mode (labels carry "bf16s" then); weights and weight gradients are fp32."""
    if name in ("snn_conv2d_fwd", "snn_conv2d_dgrad"):
        a = tuple(a[:3]) + tuple(a[4:])   # without the pre-split weight pointer: the positions snn_conv2d_wgrad has
        n, h, w, cin, ho, wo, cout, kh, kw = a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13]
        sb = a[21 if name == "snn_conv2d_fwd" else 20] == PREC_BF16S
        es = 2.0 if sb else 4.0
        es_in = 4.0 if (cin == 2 and name == "snn_conv2d_fwd") else es   # the event frames stay fp32
        flops = 2.0 * n * ho * wo * cout * kh * kw * cin
        byts = es_in * n * h * w * cin + es * n * ho * wo * cout + 4.0 * cout * kh * kw * cin
        label = _conv_label(name, a)
        if sb and not label.startswith("k_conv_first"):   # bf16 storage: always the implicit GEMM (no direct-3x3 form)
            oc = cin if name == "snn_conv2d_dgrad" else cout
            label = (f"k_conv_gather<{'32, 4, 1' if oc <= 32 else ('64, 2, 2' if oc <= 64 else '128, 2, 2')}, "
                     f"{'true' if name == 'snn_conv2d_dgrad' else 'false'}, true>")
        return label + (", bf16s" if sb else ""), flops, byts
    if name == "snn_conv2d_wgrad":
        n, h, w, cin, ho, wo, cout, kh, kw = a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13]
        sb = a[19] == PREC_BF16S
        es = 2.0 if sb else 4.0
        flops = 2.0 * n * ho * wo * cout * kh * kw * cin
        byts = (4.0 if cin == 2 else es) * n * h * w * cin + es * n * ho * wo * cout + 4.0 * cout * kh * kw * cin
        return _conv_label(name, a) + (", bf16s" if sb else ""), flops, byts
    if name in ("snn_conv1x1_spikes_fwd", "snn_conv1x1_spikes_wgrad"):   # 1x1 over spikes formed from saved potentials
        fwd = name.endswith("_fwd")
        n, h, w, cin, cout = (a[6], a[7], a[8], a[9], a[10])
        flops = 2.0 * n * h * w * cout * cin
        byts = 4.0 * (n * h * w * cin + n * h * w * cout + cout * cin)
        if not fwd:
            return WGRAD_LABELS[0], flops, byts
        tile = "32, 4, 1" if cout <= 32 else ("64, 2, 2" if cout <= 64 else "128, 2, 2")
        return f"k_conv_gather<{tile}, false, true>", flops, byts
    if name in ("snn_conv2d_spikes_fwd", "snn_conv2d_spikes_wgrad"):   # any kernel size over spikes formed from saved potentials
        # (vdec, ld, v_th, w | dy, y | lddy, ldy | dw, N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, ...)
        fwd = name.endswith("_fwd")
        n, h, w, cin, ho, wo, cout, kh, kw, stride, pad = a[6:17]
        flops = 2.0 * n * ho * wo * cout * kh * kw * cin
        byts = 4.0 * (n * h * w * cin + n * ho * wo * cout + cout * kh * kw * cin)
        if not fwd:
            return _wgrad_label(n, h, w, cin, ho, wo, cout, kh, kw, stride, pad, 1) + ", spikes", flops, byts
        tile = "32, 4, 1" if cout <= 32 else ("64, 2, 2" if cout <= 64 else "128, 2, 2")
        return f"k_conv_gather<{tile}, false, true>", flops, byts
    if name == "snn_conv3x3_halo_spikes":   # halo-resident forward over spikes formed from saved potentials (two products)
        n, h, w, cin, cout = a[6], a[7], a[8], a[9], a[10]
        flops = 2.0 * n * h * w * cout * 9 * cin
        byts = 4.0 * (n * h * w * cin + n * h * w * cout) + 4.0 * cout * 9 * cin
        return f"k_conv_halo3<{128 if cout % 128 == 0 else (64 if cout % 64 == 0 else 32)}, fwd, spikes>", flops, byts
    if name == "snn_conv3x3_halo":   # halo-resident 3x3 / stride 1 (csrc/conv_halo.hip): forward (fp16 x 3) or data gradient
        n, h, w, cin, cout, prec = a[5], a[6], a[7], a[8], a[9], a[17]
        es = 2.0 if prec == PREC_BF16S else 4.0
        flops = 2.0 * n * h * w * cout * 9 * cin
        byts = es * (n * h * w * cin + n * h * w * cout) + 4.0 * cout * 9 * cin
        kind = "fwd" if prec == 4 else ("dgrad" if prec == 1 else "bf16s")   # (bf16 storage: one instance serves both)
        return f"k_conv_halo3<{128 if cout % 128 == 0 else (64 if cout % 64 == 0 else 32)}, {kind}>", flops, byts
    if name == "snn_conv3x3_halo_bn":   # the same data gradient with the BatchNorm-backward affine applied while staging
        n, h, w, cin, cout = a[8], a[9], a[10], a[11], a[12]
        flops = 2.0 * n * h * w * cout * 9 * cin
        byts = 4.0 * (3 * n * h * w * cin + n * h * w * cout + cout * 9 * cin)   # gx, y read; dy, dx written
        return f"k_conv_halo3<{128 if cout % 128 == 0 else 64}, dgrad>", flops, byts
    if name == "snn_conv3x3_s2_dgrad":   # one-pass stride-2 data gradient
        n, h, w, cin, ho, wo, cout = a[5], a[6], a[7], a[8], a[9], a[10], a[11]
        sb = a[16] == PREC_BF16S
        flops = 2.0 * n * ho * wo * cout * 9 * cin
        byts = (2.0 if sb else 4.0) * (n * h * w * cin + n * ho * wo * cout) + 4.0 * cout * 9 * cin
        return "k_conv_s2dgrad3<dgrad>" + (", bf16s" if sb else ""), flops, byts
    if name == "snn_conv2d_wgrad_bn":   # event-frame weight gradient with the BatchNorm-backward affine applied on the fly
        n, h, w, cin, ho, wo, cout, kh, kw = a[10], a[11], a[12], a[13], a[14], a[15], a[16], a[17], a[18]
        flops = 2.0 * n * ho * wo * cout * kh * kw * cin
        byts = 4.0 * (n * h * w * cin + 2 * n * ho * wo * cout + cout * kh * kw * cin)   # x, gx and y are read
        return WGRAD_LABELS[2], flops, byts
    if name == "snn_affine_neuron_fwd":
        neuron, T, M, C = a[0], a[14], a[15], a[16]
        last_only = bool(a[18] & 2)   # SNN_SCAN_LAST_STEP_ONLY: the output of ONE step is written
        elems = float(T) * M * C
        # y read; out written (one step of it with last_only); + vdec, + fused shortcut
        wrote_out = 0 if a[7] is None else (1.0 / T if last_only else 1)   # (SNN_SCAN_SPIKES_FROM_VDEC: no output tensor)
        tensors = 1 + wrote_out + (1 if a[13] is not None else 0) + (1 if a[9] is not None else 0)
        sb = bool(a[18] & SCAN_BF16_STORAGE)
        return f"k_affine_neuron_fwd<{neuron}>" + (", bf16s" if sb else ""), 12.0 * elems, (2.0 if sb else 4.0) * elems * tensors
    if name == "snn_affine_neuron_bwd":
        neuron, T, M, C = a[0], a[15], a[16], a[17]
        last_only = bool(a[19] & 2)   # output gradient (and a saved OUTPUT, LI+Tanh) exist for the last step only
        elems = float(T) * M * C
        one = 1.0 / T if last_only else 1
        saved = 0 if a[3] is None else (one if neuron == 3 else 1)   # LIF & co. save v_dec per step, LI+Tanh its output
        from_state = bool(a[19] & 16)   # SNN_SCAN_SUMS_FROM_STATE: the statistic comes from the saved state, y is not read
        tensors = one + 1 + saved + (1 if (a[14] is not None and not from_state) else 0)   # g_out, gx, saved state, y (BN sums)
        sb = bool(a[19] & SCAN_BF16_STORAGE)
        return f"k_affine_neuron_bwd<{neuron}>" + (", bf16s" if sb else ""), 16.0 * elems, (2.0 if sb else 4.0) * elems * tensors
    if name == "snn_lif_fwd_ckpt":  # y, out (+ shortcut), checkpoints = 2/K of a tensor
        T, M, C = a[13], a[14], a[15]
        elems = float(T) * M * C
        return "k_affine_neuron_fwd<1,ckpt>", 12.0 * elems, 4.0 * elems * (2.5 + (1 if a[8] is not None else 0))
    if name == "snn_lif_bwd_ckpt":  # g_out, y, gx, checkpoints
        T, M, C = a[14], a[15], a[16]
        elems = float(T) * M * C
        return "k_lif_bwd_ckpt", 28.0 * elems, 4.0 * elems * 3.5
    if name in ("snn_bn_stats", "snn_bn_stats_bf16"):
        T, M, C = a[2], a[3], a[4]
        sb = name.endswith("_bf16")
        return "k_bn_stats" + (", bf16s" if sb else ""), 3.0 * T * M * C, (2.0 if sb else 4.0) * T * M * C
    if name in ("snn_bn_bwd_apply", "snn_bn_bwd_apply_bf16"):
        T, M, C = a[8], a[9], a[10]
        sb = name.endswith("_bf16")
        return "k_bn_bwd_apply" + (", bf16s" if sb else ""), 4.0 * T * M * C, (6.0 if sb else 12.0) * T * M * C
    if name in ("snn_copy_channels", "snn_add_channels"):
        M, C = a[4], a[5]
        return "k_channels", 0.0, (8.0 if name == "snn_copy_channels" else 12.0) * M * C
    if name == "snn_add":
        return "k_add", float(a[6]) * a[7], 12.0 * a[6] * a[7]
    if name in ("snn_nchw_to_nhwc", "snn_nhwc_to_nchw"):
        n = float(a[2]) * a[3] * a[4] * a[5]
        return "k_layout", 0.0, 8.0 * n
    if name == "snn_adamax_step":
        return "k_adamax", 8.0 * a[4], 28.0 * a[4]
    return name, 0.0, 0.0


class KernelProfiler:
    """Install with ``_hip.PROFILER = KernelProfiler()``; call ``summary()`` after a device sync."""

    def __init__(self):
        self.records = []
        self.neuron_steps = 0.0   # neuron-timesteps (elements x T) of the forward scans seen: the unit SURVEY 8(d) prices

    def before(self, name, args):
        label, flops, byts = work_of(name, args)
        if name in ("snn_affine_neuron_fwd", "snn_lif_fwd_ckpt"):
            self.neuron_steps += flops / 12.0   # work_of counts 12 FLOP per neuron-timestep
        start = torch.cuda.Event(enable_timing=True)
        end = torch.cuda.Event(enable_timing=True)
        start.record()
        return (label, flops, byts, start, end)

    def after(self, token):
        token[4].record()
        self.records.append(token)

    def summary(self) -> Dict[str, dict]:
        torch.cuda.synchronize()
        agg = defaultdict(lambda: {"calls": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
        for label, flops, byts, start, end in self.records:
            row = agg[label]
            row["calls"] += 1
            row["ms"] += start.elapsed_time(end)
            row["flops"] += flops
            row["bytes"] += byts
        for row in agg.values():
            sec = max(row["ms"], 1e-9) * 1e-3
            row["avg_us"] = 1e3 * row["ms"] / row["calls"]
            row["tflops"] = row["flops"] / sec / 1e12
            row["gbs"] = row["bytes"] / sec / 1e9
        return dict(agg)
