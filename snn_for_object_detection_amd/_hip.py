"""ctypes binding of ``libsnn_hip.so`` (the C ABI declared in ``include/snn_hip.h``).

There is NO fallback: if the library is missing or a call fails the product raises.
Device pointers come from torch storages (``tensor.data_ptr()``), the stream from
``torch.cuda.current_stream().cuda_stream``; torch is plumbing only.
"""

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SNN_HIP_LIB") or os.path.join(_HERE, "libsnn_hip.so")  # SNN_HIP_LIB: tuning aid (ablation / -DSNN_TUNING builds)

NEURON_NONE, NEURON_LIF, NEURON_LI, NEURON_LI_TANH, NEURON_SLI, NEURON_SYNAPSE = 0, 1, 2, 3, 4, 5
POOL_AVG, POOL_MAX, POOL_SUM = 0, 1, 2
ACT_RELU, ACT_SILU, ACT_TANH = 0, 1, 2
ABI_VERSION = 11
PREC_FP32, PREC_BF16X3, PREC_BF16X6, PREC_FP16X3, PREC_BF16X1, PREC_BF16S = 0, 1, 3, 4, 5, 6   # SNN_PREC_* of include/snn_hip.h
SCAN_WIDE_ADDRESSING, SCAN_LAST_STEP_ONLY, SCAN_BF16_STORAGE, SCAN_SPIKES_FROM_VDEC, SCAN_SUMS_FROM_STATE = 1, 2, 4, 8, 16
SCAN_STATE_LOOKBACK = 32


class NeuronParams(Structure):
    _fields_ = [("c_mem", c_float), ("c_syn", c_float), ("v_leak", c_float), ("v_th", c_float),
                ("v_reset", c_float), ("alpha", c_float), ("v_st", c_float), ("tau_sec", c_float),
                ("tau_dis", c_float), ("dt", c_float), ("sigma", c_float)]


_P, _I, _L, _F = c_void_p, c_int, c_int64, c_float

# name -> (restype, argtypes); mirrors include/snn_hip.h one to one
SIGNATURES = {
    "snn_abi_version": (c_int, []),
    "snn_last_error": (c_char_p, []),
    "snn_nchw_to_nhwc": (c_int, [_P, _P, _L, _I, _I, _I, _P]),
    "snn_nhwc_to_nchw": (c_int, [_P, _P, _L, _I, _I, _I, _P]),
    "snn_weight_transpose": (c_int, [_P, _P, _I, _I, _I, _I, _P]),
    "snn_weight_transpose_batched": (c_int, [_P, _P, _P, _I, _P]),
    "snn_detect_decode": (c_int, [_P, _P, _P, _I, _I, _P, _P, _P, _P]),
    "snn_nms_sorted": (c_int, [_P, _P, _P, _I, _F, _P, _P, _P, _P, _P]),
    "snn_conv2d_fwd": (c_int, [_P, _L, _P, _P, _P, _L, _L, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _L, _P, _I, _P, _I,
                               _P]),
    "snn_weight_presplit": (c_int, [_P, _P, _L, _I, _P]),
    "snn_conv2d_fwd_bn_partial_size": (c_size_t, [_L, _I, _I, _I, _I]),
    "snn_conv2d_wgrad_bn_supported": (c_int, [_L, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I]),
    "snn_conv2d_wgrad_bn": (c_int, [_P, _L, _P, _L, _P, _L, _P, _I, _I, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I,
                                    _P, _I, _P]),
    "snn_conv3x3_s2_dgrad_supported": (c_int, [_L, _I, _I, _I, _I, _I, _I]),
    "snn_conv3x3_s2_dgrad": (c_int, [_P, _L, _P, _P, _L, _L, _I, _I, _I, _I, _I, _I, _P, _L, _P, _L, _I, _P]),
    "snn_conv3x3_halo_bn_supported": (c_int, [_L, _I, _I, _I, _I, _I]),
    "snn_conv3x3_halo_bn": (c_int, [_P, _P, _P, _I, _P, _P, _P, _L, _L, _I, _I, _I, _I, _P, _L, _P, _L, _P]),
    "snn_conv3x3_halo_supported": (c_int, [_L, _I, _I, _I, _I]),
    "snn_conv3x3_halo_bn_chunks": (c_int64, [_I, _I, _I]),
    "snn_weight_frag_image_bytes": (c_size_t, [_I, _I]),
    "snn_weight_frag_image_batched": (c_int, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "snn_conv3x3_halo": (c_int, [_P, _L, _P, _P, _L, _L, _I, _I, _I, _I, _P, _L, _P, _L, _P, _I, _P, _I, _P]),
    "snn_conv2d_dgrad": (c_int, [_P, _L, _P, _P, _P, _L, _L, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _L, _P, _L,
                                 _I, _P]),
    "snn_conv2d_wgrad": (c_int, [_P, _L, _P, _L, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _I,
                                 _P]),
    "snn_conv2d_wgrad_splitk": (c_int, [_L, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I]),
    "snn_conv2d_wgrad_kernel": (c_int, [_L, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I]),
    "snn_conv1x1_spikes_supported": (c_int, [_L, _I, _I, _I, _I, _L, _I, _I]),
    "snn_conv1x1_spikes_fwd": (c_int, [_P, _L, _F, _P, _P, _L, _L, _I, _I, _I, _I, _P]),
    "snn_conv1x1_spikes_wgrad": (c_int, [_P, _L, _F, _P, _L, _P, _L, _I, _I, _I, _I, _I, _P, _I, _P]),
    "snn_conv2d_spikes_supported": (c_int, [_L, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _L, _I, _I]),
    "snn_conv2d_spikes_fwd": (c_int, [_P, _L, _F, _P, _P, _L, _L, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P]),
    "snn_conv2d_spikes_wgrad": (c_int, [_P, _L, _F, _P, _L, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P]),
    "snn_conv3x3_halo_spikes": (c_int, [_P, _L, _F, _P, _P, _L, _L, _I, _I, _I, _I, _P, _I, _P, _P]),
    "snn_bn_stats_partial_size": (c_size_t, [_I, _L, _I]),
    "snn_bn_stats": (c_int, [_P, _L, _I, _L, _I, _P, _P]),
    "snn_bn_stats_finalize": (c_int, [_P, _I, _I, _I, _L, _I, _P, _P, _F, _F, _P, _P, _I, _P, _P, _P, _P, _P]),
    "snn_bn_stats_reduce": (c_int, [_P, _I, _I, _I, _L, _I, _P, _P]),
    "snn_bn_stats_from_sums": (c_int, [_P, _I, _L, _I, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P]),
    "snn_bn_bwd_reduce": (c_int, [_P, _I, _L, _I, _P, _P]),
    "snn_bn_bwd_coef": (c_int, [_P, _P, _P, _I, _L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "snn_affine_neuron_fwd": (c_int, [_I, _P, _L, _P, _P, _P, _P, _P, _L, _P, _L, _P, _P, _P, _I, _L, _I,
                                      POINTER(NeuronParams), _I, _P]),
    "snn_lif_ckpt_interval": (c_int, []),
    "snn_lif_fwd_ckpt": (c_int, [_P, _L, _P, _P, _P, _P, _P, _L, _P, _L, _P, _P, _P, _I, _L, _I,
                                 POINTER(NeuronParams), _P]),
    "snn_lif_bwd_ckpt": (c_int, [_P, _L, _P, _P, _L, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _L, _I,
                                 POINTER(NeuronParams), _P]),
    "snn_affine_neuron_bwd_sums_size": (c_size_t, [_I, _L, _I]),
    "snn_affine_neuron_bwd": (c_int, [_I, _P, _L, _P, _P, _L, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _L, _I,
                                      POINTER(NeuronParams), _I, _P]),
    "snn_bn_bwd_finalize": (c_int, [_P, _I, _L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "snn_bn_bwd_finalize_from_state": (c_int, [_P, _I, _L, _I, _P, _P, _P, _P, _P, _P, _L, _P, _P, _P, _P, _P, _I, _P]),
    "snn_bn_bwd_reduce_from_state": (c_int, [_P, _I, _L, _I, _P, _P, _P, _P, _P, _P, _L, _P, _P]),
    "snn_affine_neuron_bwd_sums_from_state": (c_int, [_I, _I, _L, _I, _L, POINTER(NeuronParams), _I]),
    "snn_bn_bwd_apply": (c_int, [_P, _P, _L, _P, _P, _P, _P, _L, _I, _L, _I, _I, _P]),
    "snn_bn_bwd_apply_bf16": (c_int, [_P, _P, _L, _P, _P, _P, _P, _L, _I, _L, _I, _I, _P]),
    "snn_bn_stats_bf16": (c_int, [_P, _L, _I, _L, _I, _P, _P]),
    "snn_copy_channels": (c_int, [_P, _L, _P, _L, _L, _I, _P]),
    "snn_add_channels": (c_int, [_P, _L, _P, _L, _L, _I, _P]),
    "snn_add": (c_int, [_P, _L, _P, _L, _P, _L, _L, _I, _P]),
    "snn_roi_workspace_size": (c_size_t, [_I, _I, _I]),
    "snn_roi_assign": (c_int, [_P, _P, _I, _I, _I, _F, _P, _P, _P, _P, _P]),
    "snn_det_loss_workspace_size": (c_size_t, [_L]),
    "snn_det_loss_fwd": (c_int, [_P, _P, _P, _P, _P, _L, _I, _F, _P, _P, _P, _P]),
    "snn_det_loss_bwd": (c_int, [_P, _P, _P, _P, _P, _L, _I, _F, _P, _P, _P, _P, _P]),
    "snn_small_gemm": (c_int, [_P, _L, _I, _P, _L, _I, _P, _L, _I, _I, _I, _I, _P, _L, _P]),
    "snn_small_gemm_batched": (c_int, [_P, _P, _P, _P, _P, _I, _I, _P]),
    "snn_copy_channels_bf16": (c_int, [_P, _L, _P, _L, _L, _I, _P]),
    "snn_add_bf16": (c_int, [_P, _L, _P, _L, _P, _L, _L, _I, _P]),
    "snn_convert_bf16": (c_int, [_P, _P, _L, _I, _P]),
    "snn_act_fwd": (c_int, [_I, _P, _P, _L, _P]),
    "snn_act_bwd": (c_int, [_I, _P, _P, _P, _P, _L, _P]),
    "snn_lstm_cell_fwd": (c_int, [_P, _P, _P, _P, _L, _I, _P]),
    "snn_lstm_cell_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _L, _I, _P]),
    "snn_pool_fwd": (c_int, [_I, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _P]),
    "snn_pool_bwd": (c_int, [_I, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _P]),
    "snn_upsample_fwd": (c_int, [_P, _P, _L, _I, _I, _I, _I, _P]),
    "snn_upsample_bwd": (c_int, [_P, _P, _L, _I, _I, _I, _I, _P]),
    "snn_adamax_step": (c_int, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _I, _F, _P]),
    "snn_events_to_frames": (c_int, [_P, _P, _P, _P, _L, _P, _I, _I, _I, _P]),
}

_lib = None


def load():
    """Load the HIP library once; raise (never fall back) when it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the gfx950 HIP extension has not been built "
            "(run `python -m snn_for_object_detection_amd._build`); there is no CPU / eager fallback")
    if not os.environ.get("SNN_HIP_LIB"):
        # a binary shipped next to edited sources must not run silently (it is keyed on the sources' sha256, _build.py)
        from . import _build
        if not _build.library_is_current():
            raise RuntimeError(
                f"{LIB_PATH} was not built from the kernel sources as they are now (csrc/*.hip, include/snn_hip.h "
                "changed, or the build stamp is missing): run `python -m snn_for_object_detection_amd._build`")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    if lib.snn_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libsnn_hip.so ABI {lib.snn_abi_version()} != binding {ABI_VERSION}: rebuild")
    _lib = lib
    return lib


PROFILER = None  # optional object with .before(name, args) -> token and .after(token); see profiler.py


def call(name: str, *args) -> None:
    """Invoke an int-returning entry point; non-zero -> RuntimeError with the library's message."""
    lib = load()
    if PROFILER is not None:
        token = PROFILER.before(name, args)
        rc = getattr(lib, name)(*args)
        PROFILER.after(token)
    else:
        rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.snn_last_error()
        raise RuntimeError(f"{name} failed (rc={rc}): {msg.decode() if msg else '?'}")


def query(name: str, *args):
    return getattr(load(), name)(*args)
