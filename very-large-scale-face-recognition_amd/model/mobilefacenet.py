"""MobileFaceNet (reference model/mobilefacenet_def.py:77-123) — not built yet in this round: the
depthwise / pointwise kernels are the next row of the hot-path table."""
from .. import _lib


class MobileFaceNet(object):
    def __init__(self, feat_dim=128, fp16=False, **kwargs):
        raise _lib.VlsfrError("MobileFaceNet is not implemented on the gfx950 path yet (DESIGN.md §next); "
                              "use an iResNet backbone ('ir18' ... 'ir200')")
