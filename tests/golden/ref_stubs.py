"""Import shim for the upstream reference tree (container-only; never used on the GPU box).

The reference package pulls in torchvision / diffusers / webdataset / ... at import time even
though none of them take part in the arithmetic of the hot path.  This module
registers inert stand-ins for those distributions so that ``fourm.models.fm`` and ``fourm.vq`` can be
imported from ``/root/reference`` and used to (a) validate the ``oracle/`` restatement and
(b) generate the golden fixtures committed under ``tests/golden/``.
"""
import importlib.machinery
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("FOURM_REFERENCE_ROOT", "/root/reference")

_STUBS = [
    "torchvision", "torchvision.transforms", "torchvision.transforms.functional", "torchvision.ops",
    "torchvision.ops.misc", "torchvision.datasets", "torchvision.datasets.vision",
    "boto3", "boto3.s3", "boto3.s3.transfer",
    "webdataset", "webdataset.handlers", "webdataset.filters",
    "cv2", "albumentations", "braceexpand", "wandb",
    "diffusers", "diffusers.utils", "diffusers.schedulers", "diffusers.schedulers.scheduling_utils",
    "diffusers.configuration_utils", "diffusers.models", "diffusers.models.modeling_utils",
    "diffusers.models.embeddings", "diffusers.models.unet_2d_blocks", "diffusers.models.resnet",
    "diffusers.models.controlnet", "diffusers.pipelines", "diffusers.pipelines.pipeline_utils",
]


class _Inert(types.ModuleType):
    """Module whose every attribute is a fresh, plain class (usable as a base class)."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (), {"__init__": lambda self, *a, **k: None,
                              "__call__": lambda self, *a, **k: self})
        setattr(self, name, cls)
        return cls


def _install_diffusers_helpers():
    """The handful of ``diffusers`` helpers upstream's DiVAE decoder calls at run time (vq/models/unet/unet.py, vq/scheduling/*.py):
    none of them computes anything - a config namespace filled from the constructor arguments, ``nn.Module`` as the model base class,
    attribute registration of the pipeline's parts, and ``torch.randn`` with a generator.  Restated here so that the UNMODIFIED upstream
    classes run and the DiVAE oracle can be pinned to them."""
    import functools
    import inspect

    import torch

    class _Config(dict):
        __getattr__ = dict.__getitem__

    def register_to_config(init):
        @functools.wraps(init)
        def wrapped(self, *args, **kwargs):
            sig = inspect.signature(init)
            bound = sig.bind(self, *args, **kwargs)
            bound.apply_defaults()
            cfg = {k: v for k, v in bound.arguments.items() if k != "self"}
            object.__setattr__(self, "config", _Config(cfg)) if not isinstance(self, torch.nn.Module) else self.__dict__.__setitem__("config", _Config(cfg))
            init(self, *args, **kwargs)
        return wrapped

    class ModelMixin(torch.nn.Module):
        @property
        def device(self):
            return next(self.parameters()).device

        @property
        def dtype(self):
            return next(self.parameters()).dtype

    class ConfigMixin:
        pass

    class SchedulerMixin:
        pass

    class DiffusionPipeline:
        def __init__(self, *a, **k):
            pass

        def register_modules(self, **kw):
            for k, v in kw.items():
                setattr(self, k, v)

    def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
        gen_dev = generator.device if generator is not None else (device or "cpu")
        return torch.randn(tuple(shape), generator=generator, device=gen_dev, dtype=dtype).to(device or gen_dev)

    sys.modules["diffusers.configuration_utils"].register_to_config = register_to_config
    sys.modules["diffusers.configuration_utils"].ConfigMixin = ConfigMixin
    sys.modules["diffusers.models.modeling_utils"].ModelMixin = ModelMixin
    sys.modules["diffusers.schedulers.scheduling_utils"].SchedulerMixin = SchedulerMixin
    sys.modules["diffusers"].DiffusionPipeline = DiffusionPipeline
    sys.modules["diffusers.utils"].randn_tensor = randn_tensor


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "fourm"))


def install():
    """Register the stubs and put the reference tree first on sys.path."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    for name in _STUBS:
        if name in sys.modules:
            continue
        m = _Inert(name)
        m.__path__ = []
        m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None, is_package=True)
        sys.modules[name] = m
        if "." in name:
            parent, child = name.rsplit(".", 1)
            setattr(sys.modules[parent], child, m)
    sys.modules["diffusers.schedulers.scheduling_utils"].KarrasDiffusionSchedulers = []
    sys.modules["diffusers.utils"].BaseOutput = type("BaseOutput", (dict,), {})
    _install_diffusers_helpers()
    sys.dont_write_bytecode = True
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
