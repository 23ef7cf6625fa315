#!/usr/bin/env python3
"""4M pre-training on MI355X: the same command line, YAML keys, schedules, log / checkpoint files as upstream
``run_training_4m.py`` (argument surface :42-239, data :256-351, model :354-387, main loop :390-673, epoch :676-795, eval :798-832),
driving this package's HIP hot path:

  * the model is wrapped in ``fourm.parallel.DataParallel`` (one process per GPU, RCCL mean of the flat gradient store overlapped
    with the hand-written backward) where upstream wraps it in ``DistributedDataParallel`` (:512);
  * ``bfloat16`` needs no autocast context: the engine computes with autocast's rounding points itself; ``float32`` selects the
    engine's fp32 verification mode; ``float16`` is rejected (no loss-scaled path);
  * the data side is upstream's (``fourm.data`` falls through to an upstream checkout, see ``fourm/_upstream.py``) unless every
    dataset of the data config has ``type: synthetic`` (or ``--data_config synthetic``): then batches come from
    ``fourm.data.SyntheticLoader`` and no dataset, tokenizer file or third-party loader package is touched.

torchrun --nproc_per_node=8 run_training_4m.py -c cfgs/default/4m/models/main/4m-b_mod7_500b.yaml [--flag value ...]
"""
import argparse
import datetime
import json
import math
import os
import sys
import time
import warnings
from contextlib import nullcontext
from pathlib import Path

import numpy as np
import torch
import yaml

import fourm.utils as utils
from fourm.data.modality_info import MODALITY_INFO
from fourm.models import fm  # noqa: F401  (registers the model factories)
from fourm.utils import NativeScalerWithGradNormCount as NativeScaler
from fourm.utils import create_model
from fourm.utils.optim_factory import create_optimizer

_DTYPES = {"float16": torch.float16, "fp16": torch.float16, "bfloat16": torch.bfloat16, "bf16": torch.bfloat16,
           "float32": torch.float32, "fp32": torch.float32}

# ---- the command line -----------------------------------------------------------------------------------------------------------
# (flag, type, default, extra)      extra: choices=[...] | nargs='+'
_VALUE_ARGS = [
    ("run_name", str, "auto", {}),
    ("batch_size", int, 256, {"help": "per GPU; effective batch = batch_size * accum_iter * GPUs"}),
    ("epochs", int, 100, {}),
    ("total_tokens", int, -1, {"help": "billions of tokens; sets the epoch count when epochs < 0"}),
    ("accum_iter", int, 1, {"help": "micro-batches per optimizer step"}),
    ("save_ckpt_freq", int, 20, {"help": "epochs between checkpoints"}),
    # model
    ("model", str, "fm_base_12e_12d_swiglu_nobias", {}),
    ("patch_size", int, 16, {}),
    ("input_size", int, 224, {}),
    ("num_register_tokens", int, 0, {}),
    ("dtype", str, "bfloat16", {"choices": ["float16", "bfloat16", "float32", "bf16", "fp16", "fp32"]}),
    ("num_input_tokens", int, 128, {"help": "encoder token budget"}),
    ("num_target_tokens", int, 128, {"help": "decoder token budget"}),
    ("min_input_tokens", int, None, {}),
    ("min_target_tokens", int, None, {}),
    ("loss_type", str, "mod", {"choices": ["mod", "token"]}),
    ("finetune", None, "", {"help": "checkpoint to start from (positional embeddings are dropped)"}),
    # optimizer
    ("opt", str, "adamw", {}),
    ("opt_eps", float, 1e-8, {}),
    ("opt_betas", float, [0.9, 0.95], {"nargs": "+"}),
    ("clip_grad", float, None, {}),
    ("skip_grad", float, None, {"help": "skip the update when the gradient norm exceeds this"}),
    ("momentum", float, 0.9, {}),
    ("weight_decay", float, 0.05, {}),
    ("weight_decay_end", float, None, {}),
    ("blr", float, 1e-4, {"help": "base lr; lr = blr * total batch / 256"}),
    ("min_blr", float, 0.0, {}),
    ("frozen_model_blr", float, -1, {}),
    ("scheduler", str, "cosine", {"choices": ["cosine", "inverse_sqrt-10000"]}),
    ("warmup_epochs", int, 10, {}),
    ("warmup_steps", int, -1, {}),
    ("warmup_tokens", int, -1, {}),
    ("cooldown_epochs", int, 10, {}),
    ("cooldown_steps", int, -1, {}),
    ("cooldown_tokens", int, -1, {}),
    ("frozen_model_epochs", int, 0, {}),
    ("frozen_model_tokens", int, 0, {}),
    ("frozen_embedding_domain", str, None, {}),
    # data
    ("data_config", str, "", {"help": "YAML with the dataset mixture, or 'synthetic'"}),
    ("epoch_size", int, None, {"help": "samples per epoch"}),
    ("s3_endpoint", str, "", {}),
    ("s3_data_endpoint", str, None, {}),
    ("s3_multipart_chunksize_mb", int, 512, {}),
    ("s3_multipart_threshold_mb", int, 512, {}),
    ("s3_max_io_queue", int, 100, {}),
    ("text_tokenizer_path", None, "fourm/utils/tokenizer/trained/text_tokenizer_4m_wordpiece_30k.json", {}),
    # evaluation
    ("eval_freq", int, 10, {}),
    ("fixed_eval_input_tokens", int, 128, {}),
    ("fixed_eval_target_tokens", int, 128, {}),
    ("fixed_eval_batch_size", int, 32, {}),
    # misc
    ("output_dir", None, "", {}),
    ("device", None, "cuda", {}),
    ("seed", int, 0, {}),
    ("resume", None, "", {}),
    ("start_epoch", int, 0, {}),
    ("num_workers", int, 10, {}),
    ("rlimit", int, 4096, {}),
    ("s3_save_dir", str, "", {}),
    ("dist_url", None, "env://", {}),
    ("wandb_project", str, None, {}),
    ("wandb_entity", str, None, {}),
    ("wandb_run_name", str, "auto", {}),
]
# (flag, negated flag or None, default)
_SWITCHES = [
    ("compute_grad_norm", "no_compute_grad_norm", True), ("dist_eval", "no_dist_eval", True), ("fixed_eval", "no_fixed_eval", True),
    ("eval", None, False), ("auto_resume", "no_auto_resume", True), ("pin_mem", "no_pin_mem", True),
    ("find_unused_params", "no_find_unused_params", False), ("print_all", None, False), ("show_user_warnings", None, False),
    ("log_wandb", "no_log_wandb", False),
]


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser("4M pre-training (data parallel, MI355X hot path)", add_help=True)
    for name, typ, default, extra in _VALUE_ARGS:
        kw = dict(default=default, **extra)
        if typ is not None:
            kw["type"] = typ
        p.add_argument("--" + name, **kw)
    for name, neg, default in _SWITCHES:
        p.add_argument("--" + name, action="store_true", default=default)
        if neg:
            p.add_argument("--" + neg, action="store_false", dest=name)
    return p


def get_args(argv=None):
    """Defaults < YAML given with -c / --config < command line; unknown YAML keys become attributes too (upstream :224-234)."""
    pre = argparse.ArgumentParser(add_help=False)
    pre.add_argument("-c", "--config", default="", type=str, metavar="FILE")
    known, rest = pre.parse_known_args(argv)
    parser = build_parser()
    if known.config:
        with open(known.config) as f:
            parser.set_defaults(**(yaml.safe_load(f) or {}))
    args = parser.parse_args(rest)
    args.config_path = known.config
    return args


# ---- data ---------------------------------------------------------------------------------------------------------------------------
def setup_modality_info(args):
    info = {m: MODALITY_INFO[m] for m in args.all_domains}
    for m, d in info.items():
        if d["type"] == "img":
            size, patch = d.get("input_size", args.input_size), d.get("patch_size", args.patch_size)
            d["max_tokens"] = (size // patch) ** 2
    return info


def _domains(cfgs, key):
    return sorted(set().union(*[set(c[key].split("-")) for c in cfgs.values()]))


def load_data_config(args):
    if args.data_config in ("synthetic", "synthetic:dirichlet"):       # ':dirichlet' = batches masked on the device (DeviceUnifiedMasking)
        mods = "rgb@224-tok_rgb@224-tok_depth@224-tok_semseg@224-tok_normal@224-tok_clip@224-caption-det"
        return {"train": {"datasets": {"synthetic": {"type": "synthetic", "in_domains": mods, "out_domains": mods.partition("-")[2],
                                                     "masking": "dirichlet" if args.data_config.endswith(":dirichlet") else "uniform"}}}}
    print(f"Loading data config from: {args.data_config}")
    with open(args.data_config) as f:
        return yaml.safe_load(f)


def setup_data(args):
    """-> (modality_info, train loader, steps per epoch, {name: val loader} | None, {name: fixed-eval loader} | None).
    Loaders are built by upstream's ``fourm.data`` functions with upstream's arguments; synthetic datasets are deferred to
    ``attach_synthetic_loaders`` (they are shaped by the model)."""
    if args.min_input_tokens is None:
        args.min_input_tokens = args.num_input_tokens
    if args.min_target_tokens is None:
        args.min_target_tokens = args.num_target_tokens
    cfg = load_data_config(args)
    train_cfg = cfg["train"]["datasets"]
    args.in_domains, args.out_domains = _domains(train_cfg, "in_domains"), _domains(train_cfg, "out_domains")
    args.all_domains = sorted(set(args.in_domains) | set(args.out_domains))
    info = setup_modality_info(args)
    steps = (args.epoch_size or 0) // (args.batch_size * args.num_tasks)
    args.synthetic_data = all(c.get("type") == "synthetic" for c in train_cfg.values())
    args.synthetic_masking = next((c.get("masking", "uniform") for c in train_cfg.values() if c.get("type") == "synthetic"), "uniform")
    if args.synthetic_data:
        return info, None, steps, None, None

    from tokenizers import Tokenizer
    from fourm.data import build_mixture_dataloader, get_train_dataloader, get_val_dataloader, setup_sampling_mod_info
    tok = Tokenizer.from_file(args.text_tokenizer_path)
    if any(c["data_path"].startswith("s3") for c in train_cfg.values()):
        utils.s3_utils.override_wds_s3_tar_loading(args.s3_data_endpoint, args.s3_multipart_threshold_mb, args.s3_multipart_chunksize_mb,
                                                   args.s3_max_io_queue)
    common = dict(text_tokenizer=tok, input_size=args.input_size, num_input_tokens=args.num_input_tokens,
                  num_target_tokens=args.num_target_tokens, min_input_tokens=args.min_input_tokens,
                  min_target_tokens=args.min_target_tokens, num_tasks=args.num_tasks, num_workers=args.num_workers)
    iters, shards = [], []
    for name, dcfg in train_cfg.items():
        print(f"Setting up dataset {name} / train")
        mod_info, weights = setup_sampling_mod_info(dcfg, info)
        it = get_train_dataloader(dataset_config=dcfg, modality_info=mod_info, sampling_weights=weights, dataset_batch_size=None,
                                  epoch_size=None, **common)
        iters.append(it)
        if hasattr(it, "n_shards"):
            shards.append(it.n_shards)
    workers = min(min(shards), args.num_workers) if shards else args.num_workers
    train = build_mixture_dataloader(data_iters=iters, weights=cfg["train"].get("weights", [1.0] * len(iters)), modality_info=info,
                                     batch_size=args.batch_size, num_workers=workers, epoch_size=args.epoch_size, num_gpus=args.num_tasks)
    val = fixed = None
    if "val" in cfg:
        val, fixed = {}, {}
        for name, dcfg in cfg["val"]["datasets"].items():
            mod_info, weights = setup_sampling_mod_info(train_cfg[name], info)
            kw = dict(dataset_config=dcfg, dataset_name=name, train_configs=train_cfg, modality_info=mod_info, sampling_weights=weights,
                      fixed_eval_input_tokens=args.fixed_eval_input_tokens, fixed_eval_target_tokens=args.fixed_eval_target_tokens,
                      dist_eval=args.dist_eval, batch_size=int(1.5 * args.batch_size), pin_mem=args.pin_mem, **common)
            val[name] = get_val_dataloader(fixed_eval=False, **kw)
            if args.fixed_eval:
                fixed[name] = get_val_dataloader(fixed_eval=True, **kw)
        fixed = fixed or None
    return info, train, steps, val, fixed


def attach_synthetic_loaders(args, model, steps, device):
    """Loaders for ``type: synthetic`` data configs, shaped by the model's embeddings: batches generated on the device once
    (``masking: uniform``, the default), or raw modalities masked on the device every step (``masking: dirichlet``: token budgets, image
    masks and span masking by fourm.data.masking.DeviceUnifiedMasking - the loader's UnifiedMasking moved onto the GPU)."""
    from fourm.data import SyntheticLoader
    seed = args.seed + 1000 * utils.get_rank()
    train = SyntheticLoader(model, args.batch_size, args.num_input_tokens, args.num_target_tokens, steps, device=device, seed=seed,
                            masking=getattr(args, "synthetic_masking", "uniform"))
    val = {"synthetic": SyntheticLoader(model, args.batch_size, args.num_input_tokens, args.num_target_tokens, max(1, min(2, steps)),
                                        device=device, seed=seed + 500)}
    return train, val


# ---- model --------------------------------------------------------------------------------------------------------------------------
def get_model(args, modality_info):
    print(f"Creating model: {args.model} for modalities {list(modality_info.keys())}")

    def build(domains, key):
        out = {}
        for m in domains:
            d = modality_info[m]
            ctor = d.get(key)
            if ctor is None:
                continue
            if d["type"] == "img":
                out[m] = ctor(patch_size=d.get("patch_size", args.patch_size), image_size=d.get("input_size", args.input_size))
            else:
                out[m] = ctor()
        return out
    return create_model(args.model, encoder_embeddings=build(args.in_domains, "encoder_embedding"),
                        decoder_embeddings=build(args.out_domains, "decoder_embedding"), modality_info=modality_info,
                        num_register_tokens=args.num_register_tokens)


# ---- schedule bookkeeping -------------------------------------------------------------------------------------------------------------
def resolve_lengths(args):
    """Epochs / warm-up / cool-down / frozen phase given in tokens (billions) -> epochs or steps (upstream :432-472)."""
    tok_per_sample = args.num_input_tokens + args.num_target_tokens
    tok_per_step = tok_per_sample * args.batch_size * utils.get_world_size()
    if args.epochs < 0:
        if args.total_tokens < 0:
            sys.exit("Epochs and total tokens are both set to negative values, stopping training.")
        args.epochs = math.ceil(args.total_tokens * 1e9 / (tok_per_sample * args.epoch_size))
        print(f"Total tokens: {args.total_tokens}B\nSetting the number of epochs accordingly to {args.epochs}")
    elif args.total_tokens > 0:
        sys.exit("Epochs and total tokens are both non-negative, stopping training.")
    if args.warmup_epochs < 0 and args.warmup_steps < 0:
        if args.warmup_tokens < 0:
            sys.exit("Warmup epochs, steps and total tokens all set to negative values, stopping training.")
        args.warmup_steps = math.ceil(args.warmup_tokens * 1e9 / tok_per_step)
    if args.cooldown_epochs < 0 and args.cooldown_steps < 0:
        if args.cooldown_tokens < 0 and "inverse_sqrt" in args.scheduler:
            sys.exit("Cooldown epochs, steps and total tokens all set to negative values, stopping training.")
        args.cooldown_steps = math.ceil(args.cooldown_tokens * 1e9 / tok_per_step)
    if args.frozen_model_epochs <= 0:
        if args.frozen_model_tokens > 0:
            args.frozen_model_epochs = math.ceil(args.frozen_model_tokens * 1e9 / (tok_per_sample * args.epoch_size))
        else:
            print("No frozen models during training.")
    elif args.frozen_model_tokens > 0:
        sys.exit("Frozen_model_epochs and frozen_model_tokens are both non-negative, stopping training.")


def build_schedules(args, steps_per_epoch):
    """-> (lr per step, wd per step), the frozen-embedding phase first (upstream :518-559)."""
    if args.weight_decay_end is None:
        args.weight_decay_end = args.weight_decay
    frozen = max(args.frozen_model_epochs, 0)
    lr_head = utils.constant_scheduler(args.frozen_model_lr, frozen, steps_per_epoch) if frozen > 0 else np.array([])
    wd_head = utils.constant_scheduler(args.weight_decay, frozen, steps_per_epoch) if frozen > 0 else np.array([])
    epochs = args.epochs - frozen
    if args.scheduler == "cosine":
        lr = utils.cosine_scheduler(args.lr, args.min_lr, epochs, steps_per_epoch, warmup_epochs=args.warmup_epochs, warmup_steps=args.warmup_steps)
        wd = utils.cosine_scheduler(args.weight_decay, args.weight_decay_end, epochs, steps_per_epoch)
    elif "inverse_sqrt" in args.scheduler:
        tail = args.scheduler.split("-")[-1]
        ts = int(tail) if tail.isdigit() else 10_000
        cool = dict(cooldown_epochs=args.cooldown_epochs, cooldown_steps=args.cooldown_steps, timescale=ts)
        lr = utils.inverse_sqrt_scheduler(args.lr, args.min_lr, epochs, steps_per_epoch, warmup_epochs=args.warmup_epochs,
                                          warmup_steps=args.warmup_steps, **cool)
        wd = utils.inverse_sqrt_scheduler(args.weight_decay, args.weight_decay_end, epochs, steps_per_epoch, **cool)
    else:
        raise NotImplementedError(f"Scheduler {args.scheduler} not implemented.")
    lr, wd = np.concatenate((lr_head, lr)), np.concatenate((wd_head, wd))
    print("Max WD = %.7f, Min WD = %.7f" % (max(wd), min(wd)))
    return lr, wd


def _tokens_seen(steps, args, total_batch_size):
    per = steps * (total_batch_size / args.accum_iter) / 1e9
    return {"input_tokens_seen_b": per * args.num_input_tokens, "target_tokens_seen_b": per * args.num_target_tokens,
            "total_tokens_seen_b": per * (args.num_input_tokens + args.num_target_tokens)}


# ---- main ---------------------------------------------------------------------------------------------------------------------------
def main(args):
    utils.init_distributed_mode(args)
    device = torch.device(args.device)
    seed = args.seed + utils.get_rank()
    torch.manual_seed(seed)
    np.random.seed(seed)
    if not args.show_user_warnings:
        warnings.filterwarnings("ignore", category=UserWarning)
    if args.dtype not in _DTYPES:
        raise ValueError(f"Invalid dtype: {args.dtype}")
    dtype = _DTYPES[args.dtype]
    if dtype == torch.float16:
        raise NotImplementedError("float16 with loss scaling is not implemented on the HIP path: use bfloat16 (upstream's default) or float32")
    args.num_tasks, rank = utils.get_world_size(), utils.get_rank()

    modality_info, loader, steps_per_epoch, val_loaders, fixed_loaders = setup_data(args)
    model = get_model(args, modality_info)
    log_writer = utils.WandbLogger(args) if (rank == 0 and args.log_wandb) else None
    resolve_lengths(args)
    print(args)

    if args.finetune:
        ckpt = torch.hub.load_state_dict_from_url(args.finetune, map_location="cpu") if args.finetune.startswith("https") \
            else torch.load(args.finetune, map_location="cpu", weights_only=False)
        sd = {k: v for k, v in ckpt["model"].items() if ".pos_emb" not in k}
        print(model.load_state_dict(sd, strict=False))

    model.to(device)
    if dtype == torch.float32:
        model.compute_precision = "fp32"
    model_without_ddp = model
    n_parameters = sum(p.numel() for p in model.parameters() if p.requires_grad)
    print("Model = %s" % str(model_without_ddp))
    print(f"Number of params: {n_parameters / 1e6} M")
    if args.synthetic_data:
        loader, val_loaders = attach_synthetic_loaders(args, model, steps_per_epoch, device)

    world = utils.get_world_size()
    total_batch_size = args.batch_size * args.accum_iter * world
    args.lr, args.min_lr = args.blr * total_batch_size / 256, args.min_blr * total_batch_size / 256
    args.frozen_model_lr = (args.frozen_model_blr if args.frozen_model_blr > 0 else args.blr) * total_batch_size / 256
    print("LR = %.8f\nMin LR = %.8f" % (args.lr, args.min_lr))
    print("Total (effective) batch size = %d\nAccumulate grad iterations = %d" % (total_batch_size, args.accum_iter))
    print("Number of training steps = %d" % steps_per_epoch)
    print("Number of training examples per epoch = %d" % (args.batch_size * world * steps_per_epoch))

    from fourm.parallel import DataParallel
    model = DataParallel(model, device_ids=[args.gpu], find_unused_parameters=args.find_unused_params)
    model_without_ddp = model.module
    optimizer = create_optimizer(args, model_without_ddp)
    loss_scaler = NativeScaler(enabled=False)
    lr_values, wd_values = build_schedules(args, steps_per_epoch)
    utils.auto_load_model(args=args, model=model, model_without_ddp=model_without_ddp, optimizer=optimizer, loss_scaler=loss_scaler)

    def run_eval(loaders, n_in, n_out, tag):
        stats = {}
        for name, dl in (loaders or {}).items():
            prefix = f"[{tag}] " if not name else f"[{tag} ({name})] "
            stats.update(evaluate(model, dl, device, num_input_tokens=n_in, num_target_tokens=n_out, all_domains=args.all_domains,
                                  dtype=dtype, prefix=prefix, loss_type=args.loss_type))
        return stats

    if args.eval:
        print("Eval Stats:", run_eval(val_loaders, args.num_input_tokens, args.num_target_tokens, "Eval"))
        print("Fixed Eval Stats:", run_eval(fixed_loaders, args.fixed_eval_input_tokens, args.fixed_eval_target_tokens, "Fixed Eval"))
        return

    print(f"Start training for {args.epochs} epochs")
    t0 = time.time()
    for epoch in range(args.start_epoch, args.epochs):
        if log_writer is not None:
            log_writer.set_step(epoch * steps_per_epoch)
        train_stats = train_one_epoch(
            model=model, data_loader=loader, optimizer=optimizer, device=device, epoch=epoch, frozen_model_epochs=args.frozen_model_epochs,
            loss_scaler=loss_scaler, accum_iter=args.accum_iter, max_norm=args.clip_grad, max_skip_norm=args.skip_grad, log_writer=log_writer,
            start_steps=epoch * steps_per_epoch, lr_schedule_values=lr_values, wd_schedule_values=wd_values,
            num_input_tokens=args.num_input_tokens, num_target_tokens=args.num_target_tokens, all_domains=args.all_domains, dtype=dtype,
            loader_len=steps_per_epoch, output_dir=args.output_dir, compute_grad_norm=args.compute_grad_norm, loss_type=args.loss_type,
            total_batch_size=total_batch_size, frozen_embedding_domain=args.frozen_embedding_domain)
        last = epoch + 1 == args.epochs
        if args.output_dir and ((epoch + 1) % args.save_ckpt_freq == 0 or last):
            utils.save_model(args=args, model=model, model_without_ddp=model_without_ddp, optimizer=optimizer, loss_scaler=loss_scaler, epoch=epoch)
            if last:
                utils.save_model(args=args, model=model, model_without_ddp=model_without_ddp, optimizer=optimizer, loss_scaler=loss_scaler,
                                 epoch=epoch, ckpt_name="final", use_s3=len(args.s3_save_dir) > 0)
        log_stats = {**train_stats, "epoch": epoch, "n_parameters": n_parameters, **_tokens_seen((epoch + 1) * steps_per_epoch, args, total_batch_size)}
        if (epoch + 1) % args.eval_freq == 0 or last:
            log_stats.update(run_eval(val_loaders, args.num_input_tokens, args.num_target_tokens, "Eval"))
            log_stats.update(run_eval(fixed_loaders, args.fixed_eval_input_tokens, args.fixed_eval_target_tokens, "Fixed Eval"))
        if log_writer is not None:
            log_writer.update(log_stats)
        if args.output_dir and utils.is_main_process():
            with open(os.path.join(args.output_dir, "log.txt"), mode="a", encoding="utf-8") as f:
                f.write(json.dumps(log_stats) + "\n")
    print("Training time {}".format(datetime.timedelta(seconds=int(time.time() - t0))))


def _to_device(batch, device, all_domains):
    return {m: {k: v.to(device, non_blocking=True) for k, v in d.items()} for m, d in batch.items() if m in all_domains}


def train_one_epoch(model, data_loader, optimizer, num_input_tokens, num_target_tokens, loss_type, device, epoch, frozen_model_epochs,
                    loss_scaler, accum_iter, max_norm=None, max_skip_norm=None, log_writer=None, lr_scheduler=None, start_steps=None,
                    lr_schedule_values=None, wd_schedule_values=None, all_domains=(), dtype=torch.bfloat16, loader_len=None,
                    output_dir=None, compute_grad_norm=True, total_batch_size=None, frozen_embedding_domain=None):
    """One pass over ``data_loader`` (upstream :676-795): per-step lr / wd from the tables, forward + backward (the gradient mean
    across GPUs runs inside the backward; ``no_sync`` while accumulating), optimizer step, meters."""
    model.train()
    core = model.module
    if frozen_model_epochs > 0 and epoch < frozen_model_epochs:
        if frozen_embedding_domain is None:
            core.freeze_shared_params()
        else:
            core.freeze_params_except_specific_embeddings(frozen_embedding_domain)
    else:
        core.unfreeze_all()
    meters = utils.MetricLogger(delimiter="  ")
    meters.add_meter("lr", utils.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    meters.add_meter("min_lr", utils.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    for step, batch in enumerate(meters.log_every(data_loader, 10, iter_len=loader_len, header=f"Epoch: [{epoch}]")):
        it = start_steps + step
        update_grad = (step + 1) % accum_iter == 0
        if step % accum_iter == 0:
            for group in optimizer.param_groups:
                if lr_schedule_values is not None:
                    group["lr"] = lr_schedule_values[it] * group["lr_scale"]
                if wd_schedule_values is not None and group["weight_decay"] > 0:
                    group["weight_decay"] = wd_schedule_values[it]
        mod_dict = _to_device(batch, device, all_domains)
        with nullcontext() if update_grad else model.no_sync():
            loss, mod_loss = model(mod_dict, num_encoder_tokens=num_input_tokens, num_decoder_tokens=num_target_tokens, loss_type=loss_type)
            loss_value = loss.item()
            mod_loss_values = {f"{m}_loss": l.item() for m, l in mod_loss.items()}
            if not math.isfinite(loss_value):
                if output_dir:
                    torch.save(mod_dict, os.path.join(output_dir, "debug_mod_dict.pt"))
                print(f"Loss is {loss_value}, stopping training", file=sys.stderr)
                sys.exit(1)
            grad_norm = loss_scaler(loss / accum_iter, optimizer, clip_grad=max_norm, skip_grad=max_skip_norm, parameters=model.parameters(),
                                    compute_grad_norm=compute_grad_norm, update_grad=update_grad)
            if update_grad:
                optimizer.zero_grad()
        if device.type == "cuda":
            torch.cuda.synchronize()
        lrs = [g["lr"] for g in optimizer.param_groups]
        wds = [g["weight_decay"] for g in optimizer.param_groups if g["weight_decay"] > 0]
        meters.update(loss=loss_value, **mod_loss_values)
        meters.update(lr=max(lrs + [0.0]), min_lr=min(lrs + [1.0]), weight_decay=wds[-1] if wds else None, grad_norm=grad_norm)
        if log_writer is not None:
            log_writer.update({"loss": loss_value, "lr": max(lrs + [0.0]), "weight_decay": wds[-1] if wds else None, "grad_norm": grad_norm})
            log_writer.update(mod_loss_values)
            if total_batch_size is not None:
                per = it * (total_batch_size / accum_iter) / 1e9
                log_writer.update({"input_tokens_seen_b": per * num_input_tokens, "target_tokens_seen_b": per * num_target_tokens,
                                   "total_tokens_seen_b": per * (num_input_tokens + num_target_tokens)})
            log_writer.set_step()
        if lr_scheduler is not None:
            lr_scheduler.step_update(start_steps + step)
    meters.synchronize_between_processes()
    print("Averaged stats:", meters)
    return {"[Epoch] " + k: m.global_avg for k, m in meters.meters.items()}


@torch.no_grad()
def evaluate(model, data_loader, device, num_input_tokens, num_target_tokens, loss_type, all_domains=(), dtype=torch.bfloat16, prefix="[Eval] "):
    meters = utils.MetricLogger(delimiter="  ")
    model.eval()
    n = len(data_loader) if hasattr(data_loader, "__len__") else -1
    for batch in meters.log_every(data_loader, 10, iter_len=n, header=prefix):
        loss, mod_loss = model(_to_device(batch, device, all_domains), num_encoder_tokens=num_input_tokens,
                               num_decoder_tokens=num_target_tokens, loss_type=loss_type)
        meters.update(loss=loss.item(), **{f"{m}_loss": l.item() for m, l in mod_loss.items()})
    meters.synchronize_between_processes()
    print("Eval averaged stats:", meters)
    return {prefix + k: m.global_avg for k, m in meters.meters.items()}


if __name__ == "__main__":
    import resource
    args = get_args()
    soft, hard = resource.getrlimit(resource.RLIMIT_NOFILE)
    resource.setrlimit(resource.RLIMIT_NOFILE, (min(args.rlimit, hard) if hard > 0 else args.rlimit, hard))
    utils.setup_run_name(args)
    utils.setup_s3_args(args)
    if args.output_dir:
        Path(args.output_dir).mkdir(parents=True, exist_ok=True)
    main(args)
