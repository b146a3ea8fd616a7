"""CPU oracle for the SODa / TinyYolo spiking-CNN training step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / the reported CPU
baseline.  The product package (``snn_for_object_detection_amd``) never imports
this package and fails loudly when its HIP extension is missing.

What it restates (all file:line relative to the upstream reference tree):

* ``neurons.py``  - norse 1.1.0 ``LIFCell`` / ``LICell`` feed-forward steps and
  the SuperSpike surrogate gradient (third-party, NOT on disk here; pinned
  version ``environment.yml:222``; call sites ``models/modules/layer_gen.py:232-235,
  252-254``; in-tree structural witness ``models/modules/sli.py:97-126``).
* ``net.py``      - the time-outer executor of ``models/generator.py:82-198,
  220-538``, the layer set of ``models/modules/layer_gen.py:96-347`` and
  ``common.py:18-123``, the step logic of ``models/soda.py:138-158,202-281``.
* ``detect.py``   - ``utils/anchors.py:46-85``, ``utils/roi.py:18-109``,
  ``utils/box.py:9-153``.

Pinning status
--------------
* ``detect.py`` is PINNED: checked against outputs of the reference's own
  ``utils/{box,anchors,roi}.py`` executed in the build container; vectors and
  the generating script live in ``tests/golden/``.
* ``net.py`` (executor, merges, state threading, taps, heads, the time loop, ``_loss``) is PINNED since round 4: bit for
  bit against a run of the reference's own ``SODa`` / ``BlockGen`` / ``NeckGen`` / ``Head`` on a description without
  spiking neurons (``tests/golden/executor.npz``); ``events.py`` (voxelisation, label selection, collate) likewise against
  the reference's own ``utils/datasets.py`` methods (``tests/golden/events.npz``).
* Conv2d / BatchNorm2d / pooling / losses are torch's own CPU kernels (the
  reference calls the same ``torch.nn`` modules), so they are the reference
  arithmetic by construction.
* The LIF / LI steps are "PARITY UNPINNED": norse is not installed and not on
  disk, the reference ships no tests or golden vectors, so the restatement
  follows the published norse 1.1.0 algorithm and is pinned only by
  closed-form known-answer tests (``tests/test_oracle_neurons.py``).
"""
