"""MI355X-native implementation of the 4M (Massively Multimodal Masked Modeling) training hot path.

Same import surface as the upstream ``fourm`` package for the path it covers
(``fourm.models.fm.FM / FourM``, ``fourm.vq``, ``fourm.utils.create_model``); every tensor operation of
the train step runs in hand-written gfx950 HIP kernels reached through ``libfourm_hip.so``.
"""
__version__ = "0.1.0"

from . import _upstream

_upstream.extend_path(__name__, __path__)          # sub-packages this package lacks resolve upstream
__getattr__ = _upstream.fallthrough(__name__, is_package=True)
