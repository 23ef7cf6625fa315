"""``run_name: auto`` -> a name derived from the config path (upstream ``fourm/utils/run_name.py:14-29``): the path below ``cfgs/``
minus its first directory and the extension; 'auto' inside ``output_dir`` / ``s3_save_dir`` is replaced by it and the wandb run name
drops one more leading directory."""


def _after(s: str, sep: str) -> str:
    return s.partition(sep)[2]


def setup_run_name(args):
    if args.run_name == "auto":
        args.run_name = _after(_after(args.config_path, "cfgs/"), "/").replace(".yaml", "")
    if "wandb_run_name" in args and args.wandb_run_name == "auto":
        args.wandb_run_name = _after(args.run_name, "/")
    for key in ("output_dir", "s3_save_dir"):
        if key in args and "auto" in getattr(args, key):
            setattr(args, key, getattr(args, key).replace("auto", args.run_name))


# names only upstream's same-named module defines (see fourm/_upstream.py)
from fourm import _upstream as _up
_up.merge(__name__, globals())
