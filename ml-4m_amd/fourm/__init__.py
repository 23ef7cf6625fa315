"""MI355X-native implementation of the 4M (Massively Multimodal Masked Modeling) training hot path.

Same import surface as the upstream ``fourm`` package for the path it covers
(``fourm.models.fm.FM / FourM``, ``fourm.vq``, ``fourm.utils.create_model``); every tensor operation of
the train step runs in hand-written gfx950 HIP kernels reached through ``libfourm_hip.so``.
"""
__version__ = "0.1.0"
