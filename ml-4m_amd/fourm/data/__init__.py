"""Data side.  Implemented here: the modality registry (``modality_info``), synthetic batches of the loader's output contract
(``synthetic``), the device-side masking and the compact host-to-device formats (``masking``, ``h2d``, ``token_shards``: upstream's
pre-tokenised shards as raw int16 id batches).  The CPU pipeline
(webdataset / HF / folder loaders, augmenters, ``UnifiedMasking``, ``build_mixture_dataloader``, ``get_train_dataloader``, ...) is
upstream's and is reached through the fall-through of ``fourm._upstream`` when a checkout is configured."""
from .. import _upstream
_upstream.extend_path(__name__, __path__)          # first: sub-modules loaded below may import upstream-only siblings
from .synthetic import synthetic_batch, modality_shapes, SyntheticLoader

__getattr__ = _upstream.fallthrough(__name__, is_package=True)
