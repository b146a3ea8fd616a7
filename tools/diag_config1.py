#!/usr/bin/env python3
"""Layer-by-layer spike comparison with the oracle at BASELINE configs[1] size (GEN1, B=5, T=32), train-mode forward.
usage: diag_config1.py [SNN_NO_CONV_BN_STATS=1 in the environment to use the separate statistics pass]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import snn_for_object_detection_amd as S  # noqa: E402
from tests.test_gpu_configs import _train_mode_layerwise  # noqa: E402
from tests.util import rel_err  # noqa: E402

preds, preds_r, layers = _train_mode_layerwise(S, 32, 240, 304, 2, B=5)
for row in layers:
    if row[1] is None:
        print(f"{row[0]:70s} LI rel {row[2]:.3e}")
    else:
        per_t = row[2]
        first_t = next((t for t, n in enumerate(per_t) if n > 0), None)
        print(f"{row[0]:70s} spikes {row[1]:10.0f} prod {row[3]:10.0f} mismatches {sum(per_t):8d} first t {first_t} "
              f"n@first {per_t[first_t] if first_t is not None else 0}")
print("preds", rel_err(preds[1], preds_r[1]), rel_err(preds[2], preds_r[2]))
