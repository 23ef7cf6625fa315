"""Fall-through to an upstream apple/ml-4m checkout for everything this package does not implement itself.

This package replaces the *hot path* of 4M (model forward / backward, optimizer step, gradient exchange, tokenizer encode); the
CPU data pipeline (``fourm.data.*`` loaders, augmenters, masking classes), the vendored timm / CLIP / HMR2 helpers, S3 IO,
generation datasets etc. stay upstream's.  With ``ml-4m_amd/`` ahead of the upstream checkout on ``PYTHONPATH`` this package
shadows ``fourm``; to keep ``import fourm.data.unified_datasets`` or ``from fourm.data import build_mixture_dataloader`` working,
each sub-package here
  * appends the matching upstream directory to its ``__path__`` (sub-modules this package lacks are found there), and
  * defines a module ``__getattr__`` that resolves names it lacks from upstream's ``__init__`` / same-named module, loaded under a
    private alias inside the SAME package (so upstream's relative imports keep resolving to this package first).

The checkout is located through ``FOURM_UPSTREAM`` (a directory holding ``fourm/`` and ``run_training_4m.py``), else through the
first ``sys.path`` entry other than this package's parent that holds ``fourm/models/fm.py``.  Without a checkout everything this
package implements still works (the trainer then runs on ``fourm.data.synthetic``); a name that is upstream-only raises an
``AttributeError`` that says so."""
import importlib.util
import os
import sys
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))            # .../ml-4m_amd/fourm
_cache = {"root": False}


def upstream_root() -> Optional[str]:
    """Directory of the upstream checkout (the one containing ``fourm/``), or None."""
    if _cache["root"] is not False:
        return _cache["root"]
    cands = []
    if os.environ.get("FOURM_UPSTREAM"):
        cands.append(os.environ["FOURM_UPSTREAM"])
    cands += [p for p in sys.path if p]
    root = None
    for c in cands:
        c = os.path.abspath(c)
        f = os.path.join(c, "fourm")
        if os.path.samefile(f, _HERE) if os.path.isdir(f) else True:
            continue
        if os.path.isfile(os.path.join(f, "models", "fm.py")):
            root = c
            if os.path.abspath(os.environ.get("FOURM_UPSTREAM", "") or os.devnull) != c and not os.environ.get("FOURM_UPSTREAM_QUIET"):
                # found on sys.path rather than named explicitly: say so once (which data loaders / vendored helpers run depends on it)
                print(f"[fourm] out-of-scope modules (fourm.data, vendored helpers) fall through to the upstream checkout at {c} "
                      f"(found on sys.path; set FOURM_UPSTREAM to choose explicitly)", file=sys.stderr)
            break
    _cache["root"] = root
    return root


def extend_path(package_name: str, path_list) -> None:
    """Append upstream's directory of ``package_name`` (e.g. 'fourm.data') to that package's ``__path__``."""
    root = upstream_root()
    if root is None:
        return
    d = os.path.join(root, *package_name.split("."))
    if os.path.isdir(d) and d not in path_list:
        path_list.append(d)


def _load_alias(package_name: str, file_path: str, alias: str):
    """Execute an upstream source file as module ``<package_name>.<alias>`` (its relative imports resolve inside package_name)."""
    full = f"{package_name}.{alias}"
    if full in sys.modules:
        return sys.modules[full]
    spec = importlib.util.spec_from_file_location(full, file_path)
    mod = importlib.util.module_from_spec(spec)
    mod.__package__ = package_name
    sys.modules[full] = mod
    try:
        spec.loader.exec_module(mod)
    except BaseException:
        sys.modules.pop(full, None)
        raise
    return mod


def fallthrough(module_name: str, is_package: bool):
    """-> a module-level ``__getattr__`` for ``module_name`` ('fourm.data' or 'fourm.data.masking')."""
    def __getattr__(name):
        if name.startswith("__"):
            raise AttributeError(name)
        root = upstream_root()
        if root is None:
            raise AttributeError(f"module {module_name!r} has no attribute {name!r} (not part of the MI355X hot path; no upstream "
                                 f"apple/ml-4m checkout found - set FOURM_UPSTREAM to fall through to it)")
        parts = module_name.split(".")
        if is_package:
            src, pkg, alias = os.path.join(root, *parts, "__init__.py"), module_name, "_upstream_init"
            sub = os.path.join(root, *parts, name)
            if os.path.isfile(sub + ".py") or os.path.isdir(sub):       # a sub-module / sub-package upstream has and we lack
                return importlib.import_module(f"{module_name}.{name}")
        else:
            src, pkg, alias = os.path.join(root, *parts) + ".py", ".".join(parts[:-1]), "_upstream_" + parts[-1]
        if not os.path.isfile(src):
            raise AttributeError(f"module {module_name!r} has no attribute {name!r} (upstream has no {src})")
        mod = _load_alias(pkg, src, alias)
        try:
            return getattr(mod, name)
        except AttributeError:
            raise AttributeError(f"module {module_name!r} has no attribute {name!r} (neither here nor in {src})") from None
    return __getattr__


def merge(module_name: str, g: dict) -> None:
    """For a module of this package that shares its name with an upstream module (``fourm.utils.misc``, ``fourm.utils.dist``, ...):
    install the lazy fall-through and, when a checkout is configured and the upstream file imports cleanly, copy the public names
    this module does not define into its namespace - upstream's ``from .misc import *`` must keep delivering upstream's names.
    What this module defines always wins."""
    g["__getattr__"] = fallthrough(module_name, is_package=False)
    root = upstream_root()
    if root is None:
        return
    parts = module_name.split(".")
    src = os.path.join(root, *parts) + ".py"
    if not os.path.isfile(src):
        return
    try:
        mod = _load_alias(".".join(parts[:-1]), src, "_upstream_" + parts[-1])
    except Exception as e:     # an optional third-party package of the upstream module is missing: names resolve lazily (and fail there)
        if os.environ.get("FOURM_UPSTREAM_DEBUG"):
            print(f"[fourm._upstream] merge of {src} skipped: {type(e).__name__}: {e}", file=sys.stderr)
        return
    public = getattr(mod, "__all__", None) or [k for k in vars(mod) if not k.startswith("_")]
    for k in public:
        if k not in g:
            g[k] = getattr(mod, k)
    if "__all__" in g:
        g["__all__"] = list(g["__all__"]) + [k for k in public if k not in g["__all__"]]


def preload(package_name: str) -> None:
    """Execute upstream's ``__init__`` of ``package_name`` now (under its private alias) instead of at the first upstream-only name.
    ``fourm.utils`` does this at the end of its own ``__init__``: upstream's utils and data packages import each other, and the cycle
    only resolves in upstream's own order - utils first, completely (run_training_4m.py:33-34).  Failures (a missing optional
    third-party package) leave the lazy path in place."""
    root = upstream_root()
    if root is None:
        return
    src = os.path.join(root, *package_name.split("."), "__init__.py")
    if not os.path.isfile(src):
        return
    try:
        _load_alias(package_name, src, "_upstream_init")
    except Exception as e:
        if os.environ.get("FOURM_UPSTREAM_DEBUG"):
            print(f"[fourm._upstream] preload of {src} skipped: {type(e).__name__}: {e}", file=sys.stderr)
