"""The HOST side of the C ABI under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers on the
CPU build only).  ``_build.build_sanitized()`` compiles every ``csrc/*.hip`` with ``hipcc --offload-host-only
-fsanitize=address,undefined``; a child process (ASan runtime preloaded) then sweeps the host-only entry points - shape
planning, workspace / partial sizes, split-K plans, the argument checks of the launchers that return before any HIP call -
over edge-case geometries.  Any report (heap overflow in a planning table, signed overflow in a size product, ...)
aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes, itertools, sys
sys.path.insert(0, sys.argv[1])
from snn_for_object_detection_amd import _hip
lib = _hip.load()                       # SNN_HIP_LIB points at the sanitized host build
assert lib.snn_abi_version() == _hip.ABI_VERSION
calls = 0
sizes = [(1, 1), (1, 7), (8, 10), (15, 19), (30, 38), (60, 76), (120, 152), (240, 304), (720, 1280), (3, 78), (2, 79)]
chans = [1, 2, 3, 8, 27, 32, 36, 64, 96, 128, 256, 320, 768]
for (H, W), Cin, Cout in itertools.product(sizes, chans, chans):
    for k, s in ((1, 1), (3, 1), (3, 2), (5, 1), (7, 2)):
        pad = k // 2
        Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
        if Ho <= 0 or Wo <= 0:
            continue
        for N in (1, 5, 160, 4096):
            for prec in (_hip.PREC_FP32, _hip.PREC_BF16X3, _hip.PREC_BF16X1):
                sk = lib.snn_conv2d_wgrad_splitk(N, H, W, Cin, Ho, Wo, Cout, k, k, s, pad, prec)
                assert sk >= 1, (N, H, W, Cin, Cout, k, s, prec, sk)
                calls += 1
            for fps in (1, 5, 7):
                n = lib.snn_conv2d_fwd_bn_partial_size(N, fps, Ho, Wo, Cout)
                assert (n == 0) == (N % fps != 0), (N, fps, n)
                calls += 1
        ok = lib.snn_conv3x3_halo_supported(160, H, W, Cin, Cout)
        assert ok in (0, 1)
        assert lib.snn_conv3x3_halo_bn_chunks(5, H, W) >= 1
        calls += 2
for T, M, C in itertools.product((1, 4, 32, 128), (1, 80, 5700, 91200, 4 * 720 * 1280), (1, 27, 64, 256)):
    assert lib.snn_bn_stats_partial_size(T, M, C) > 0
    assert lib.snn_affine_neuron_bwd_sums_size(T, M, C) > 0
    calls += 2
assert lib.snn_roi_workspace_size(5, 13545, 2) > 0 and lib.snn_det_loss_workspace_size(5 * 13545) > 0
assert lib.snn_weight_frag_image_bytes(128, 64) == 9 * 128 * 64 * 4
assert lib.snn_lif_ckpt_interval() >= 2
# launchers: argument checks that return (with a message) before any HIP call
def refused(name, *args):
    rc = getattr(lib, name)(*args)
    assert rc != 0 and lib.snn_last_error(), name
    return rc
refused("snn_conv2d_fwd", None, 4, None, None, None, 4, 1, 4, 4, 4, 4, 4, 4, 3, 3, 1, 1, None, 0, None, 0, None, _hip.PREC_FP16X3, None)
refused("snn_conv2d_dgrad", None, 4, None, None, None, 4, 1, 4, 4, 4, 4, 4, 4, 3, 3, 1, 1, None, 0, None, 0, _hip.PREC_BF16X3, None)
refused("snn_conv3x3_halo", None, 64, None, None, 64, 1, 4, 4, 64, 64, None, 0, None, 0, None, 0, None, _hip.PREC_FP16X3, None)
refused("snn_weight_presplit", None, None, 7, _hip.PREC_FP16X3, None)
refused("snn_weight_frag_image_batched", None, None, None, 0, 0, 0, _hip.PREC_FP16X3, None)
print("sanitized host sweep ok:", calls, "planning calls")
'''


@pytest.mark.timeout(900)
def test_c_abi_host_side_under_asan_and_ubsan(tmp_path):
    from snn_for_object_detection_amd import _build
    info = _build.build_sanitized()
    assert os.path.exists(info["lib"]) and os.path.exists(info["asan_runtime"]), info
    script = tmp_path / "sweep.py"
    script.write_text(CHILD)
    env = dict(os.environ, SNN_HIP_LIB=info["lib"], LD_PRELOAD=info["asan_runtime"],
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    res = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=850)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    assert "sanitized host sweep ok" in res.stdout
    assert "runtime error" not in res.stderr and "AddressSanitizer" not in res.stderr, res.stderr[-3000:]
