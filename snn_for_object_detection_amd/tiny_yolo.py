"""Network configuration similar to YOLOv8 (mirror of the reference's ``models/tiny_yolo.py``)."""

from .generator import ListGen
from .layer_gen import *  # noqa: F401,F403
from .soda import SODa


class TinyYolo(SODa):
    """YOLOv8-like spiking detector (tiny_yolo.py:10-89): 48 convs, 22 norms, 19 LIF, 3 LI."""

    def backbone_cfgs(self) -> ListGen:
        return [
            *self._conv(64, 3, 2),
            *self._c2f(64, 2),
            *self._conv(128, 3, 2),
            *self._c2f(128, 3),
        ]

    def neck_cfgs(self) -> ListGen:
        stages = []
        for depth in (4, 3, 2):
            stages += [*self._conv(256, 3, 2), *self._c2f(256, depth), Return()]
        return stages

    def head_cfgs(self, box_out: int, cls_out: int) -> ListGen:
        prepare = [Conv(kernel_size=1), Norm(), LI(state_storage=self.hparams.state_storage), Tanh()]
        return [prepare, [Conv(box_out, 1)], [Conv(cls_out, 1)]]

    def _conv(self, out_channels: int = None, kernel: int = 3, stride: int = 1):
        return (
            Conv(out_channels, stride=stride, kernel_size=kernel),
            Norm(),
            LIF(state_storage=self.hparams.state_storage),
        )

    def _bottleneck(self, shortcut: bool = True):
        net = (*self._conv(),)
        return Residual([[*net], [Pass()]]) if shortcut else net

    def _rec_block(self, n: int, shortcut: bool):
        if n == 0:
            return []
        inner = [self._bottleneck(shortcut), *self._rec_block(n - 1, shortcut)]
        return (Dense([inner, [Pass()]]),)

    def _c2f(self, out_channels: int, n: int, shortcut: bool = True):
        half = int(out_channels / 2)
        return (
            Conv(out_channels, 1),
            Dense([[Conv(half, 1), *self._rec_block(n, shortcut)], [Conv(half, 1)]]),
            Conv(out_channels, 1),
        )
