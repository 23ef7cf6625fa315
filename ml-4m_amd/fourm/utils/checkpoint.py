"""Checkpoint files in the upstream layout (``fourm/utils/checkpoint.py:91-191``):
``<output_dir>/checkpoint-<epoch|name>.pth`` = {'model', 'epoch', 'args', 'scaler', 'optimizer'} and
safetensors files whose metadata carries the FM config."""
import glob
import io
import os
from pathlib import Path

import torch

from .dist import is_main_process


def load_state_dict(path: str):
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    return ckpt["model"] if isinstance(ckpt, dict) and "model" in ckpt else ckpt


def save_model(args, epoch, model, model_without_ddp, optimizer, loss_scaler, loss_balancer=None, model_ema=None, ckpt_name=None,
               use_s3=False, all_nodes=False):
    if use_s3:
        raise NotImplementedError("S3 upload belongs to the trainer's storage layer, not to the hot path")
    if not (is_main_process() or (all_nodes and getattr(args, "gpu", 0) == 0)):
        return
    out = Path(args.output_dir)
    out.mkdir(parents=True, exist_ok=True)
    blob = {"model": model_without_ddp.state_dict(), "epoch": epoch, "args": args, "scaler": loss_scaler.state_dict()}
    if optimizer is not None:
        blob["optimizer"] = optimizer.state_dict()
    if loss_balancer is not None:
        blob["loss_balancer"] = loss_balancer.state_dict()
    if model_ema is not None:      # upstream stores the EMA module's weights under 'model_ema' (checkpoint.py:113-114)
        ema = getattr(model_ema, "ema", getattr(model_ema, "module", model_ema))
        blob["model_ema"] = ema.state_dict()
    torch.save(blob, out / f"checkpoint-{ckpt_name or str(epoch)}.pth")


def auto_load_model(args, model, model_without_ddp, optimizer, loss_scaler, model_ema=None):
    """Resume from ``args.resume`` or, with ``args.auto_resume``, from the highest-numbered checkpoint."""
    if getattr(args, "auto_resume", False) and len(getattr(args, "resume", "") or "") == 0:
        best = -1
        for f in glob.glob(os.path.join(args.output_dir, "checkpoint-*.pth")):
            tag = f.split("-")[-1].split(".")[0]
            if tag.isdigit():
                best = max(best, int(tag))
        if best >= 0:
            args.resume = os.path.join(args.output_dir, "checkpoint-%d.pth" % best)
        print("Auto resume checkpoint: %s" % args.resume)
    if not getattr(args, "resume", None):
        return
    ckpt = torch.load(args.resume, map_location="cpu", weights_only=False)
    model_without_ddp.load_state_dict(ckpt["model"])
    print("Resume checkpoint %s" % args.resume)
    if "optimizer" in ckpt and "epoch" in ckpt:
        optimizer.load_state_dict(ckpt["optimizer"])
        args.start_epoch = ckpt["epoch"] + 1
        if "scaler" in ckpt:
            loss_scaler.load_state_dict(ckpt["scaler"])
        print("With optim & sched!")
    if getattr(args, "model_ema", False):       # upstream checkpoint.py:154-156
        if model_ema is None or "model_ema" not in ckpt:
            raise ValueError("args.model_ema is set but " + ("no model_ema object was passed" if model_ema is None
                                                               else f"{args.resume} holds no 'model_ema' entry"))
        ema = getattr(model_ema, "ema", getattr(model_ema, "module", model_ema))
        ema.load_state_dict(ckpt["model_ema"])
        print("With EMA!")


def _parse_meta(meta):
    from yaml import YAMLError, safe_load
    out = {}
    for k, v in meta.items():
        if not isinstance(v, str) or len(v) > 10_000:
            out[k] = v
            continue
        try:
            out[k] = safe_load(v.replace("None", "null"))
        except YAMLError:
            out[k] = v
    return out


def load_safetensors(safetensors_path, return_metadata=True):
    from safetensors import safe_open
    with safe_open(safetensors_path, framework="pt", device="cpu") as f:
        tensors = {k: f.get_tensor(k) for k in f.keys()}
        meta = f.metadata()
    return (tensors, _parse_meta(meta or {})) if return_metadata else tensors


# names only upstream's same-named module defines (see fourm/_upstream.py)
from fourm import _upstream as _up
_up.merge(__name__, globals())
