"""MI355X-native spiking-CNN detector step behind the SODa / TinyYolo operator API.

Public names follow the reference (``models``, ``models.generator``, ``models.modules``,
``utils.{anchors,roi,box}``): ``SODa``, ``TinyYolo``, ``BlockGen``, ``BackboneGen``, ``NeckGen``,
``Head``, ``HeadGen``, ``ListGen``, ``ListState`` and the layer generators ``Conv, Norm, LIF, LI,
Pool, Up, Pass, Return, ReLU, SiLU, Tanh, LSTM, Synapse, SLI, Residual, Dense``.

All arithmetic of the hot path runs in ``libsnn_hip.so`` (hand-written gfx950 kernels, C ABI in
``include/snn_hip.h``); importing the package never touches the GPU, the first forward does and
raises if the library is missing - there is no CPU / eager fallback.
"""

from . import box, functional  # noqa: F401
from .anchors import AnchorGenerator  # noqa: F401
from .data import EventBatcher  # noqa: F401
from .generator import BackboneGen, BlockGen, Head, HeadGen, ListGen, ListState, ModelGen, NeckGen  # noqa: F401
from .layer_gen import *  # noqa: F401,F403
from .roi import RoI  # noqa: F401
from .soda import SODa  # noqa: F401
from .tiny_yolo import TinyYolo  # noqa: F401

__version__ = "0.1.0"
