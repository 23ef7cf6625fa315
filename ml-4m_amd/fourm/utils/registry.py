"""Name -> factory registry (the slice of the vendored timm registry the trainer uses:
fourm/utils/timm/registry.py:25, fourm/utils/timm/model_builder.py:27-74)."""
_FACTORIES = {}


def register_model(fn):
    _FACTORIES[fn.__name__] = fn
    return fn


def is_model(name: str) -> bool:
    return name in _FACTORIES


def model_entrypoint(name: str):
    return _FACTORIES[name]


def list_models(filter: str = ""):
    import fnmatch
    names = sorted(_FACTORIES)
    return fnmatch.filter(names, filter) if filter else names


def create_model(model_name: str, pretrained: bool = False, checkpoint_path: str = "", **kwargs):
    """Look a factory up by name and call it; ``None`` keyword values are dropped (so factories keep
    their defaults), as the upstream builder does."""
    import fourm.models.fm  # noqa: F401  (populates the registry)
    if not is_model(model_name):
        raise RuntimeError(f"Unknown model ({model_name})")
    kwargs = {k: v for k, v in kwargs.items() if v is not None}
    model = _FACTORIES[model_name](**kwargs)
    if checkpoint_path:
        from .checkpoint import load_state_dict
        model.load_state_dict(load_state_dict(checkpoint_path))
    return model
