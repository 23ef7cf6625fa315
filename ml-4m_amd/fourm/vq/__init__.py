"""VQ tokenizers: ViT encoder + cosine-similarity codebook search (``VQ``), plus the ViT decoder and the training path (``VQVAE``)."""
import os

from .vqvae import VQ, VQVAE, DiVAE


def tokenize_sub_batches(model, sub_batches, n_streams: int = 2):
    """upstream's tokenization loop - ``for sub_batch in imgs_batch.split(batch_size): tokens = model.tokenize(sub_batch)``
    (save_vq_tokens.py:262-288) - with up to ``n_streams`` sub-batches in flight, each on its own HIP stream.  Returns the token tensors in
    order; the CURRENT stream has waited for all of them on return.

    Why: at the reference's sub-batch of 64 a GEMM of the ViT-B tokenizer has 12 544 rows = 147 - 196 output tiles for 256 CUs (less than one
    round), and a second sub-batch's kernels run on the CUs the first leaves idle: 12.7 -> 14.7 k images/s with two streams
    (tools/vq_two_streams.py).  Safe because the tokenizer engines keep one scratch set per stream (Workspace(per_stream=True)); weight images and
    the normalised codebook are built by the first sub-batch, which therefore runs alone."""
    import torch
    sub_batches = list(sub_batches)
    if not sub_batches:
        return []
    with torch.no_grad():
        first = model.tokenize(sub_batches[0])                 # builds every cached operand on the current stream
        if n_streams <= 1 or len(sub_batches) == 1:
            return [first] + [model.tokenize(x) for x in sub_batches[1:]]
        cur = torch.cuda.current_stream()
        pool = getattr(model, "_tok_streams", None)
        if pool is None or len(pool) < n_streams or pool[0].device != cur.device:
            pool = [torch.cuda.Stream(device=cur.device) for _ in range(n_streams)]
            object.__setattr__(model, "_tok_streams", pool)
        ready = torch.cuda.Event()
        ready.record(cur)                                      # inputs and cached operands are complete up to here
        out = [first]
        for i, x in enumerate(sub_batches[1:]):
            s = pool[i % n_streams]
            if i < n_streams:
                s.wait_event(ready)
            with torch.cuda.stream(s):
                out.append(model.tokenize(x))
                x.record_stream(s)
                out[-1].record_stream(cur)
        for s in pool[:min(n_streams, len(sub_batches) - 1)]:
            cur.wait_stream(s)
    return out


def decode_token_batches(model, token_batches, n_streams: int = 2, **decode_kwargs):
    """``[model.decode_tokens(t, **decode_kwargs) for t in token_batches]`` for the diffusion detokenizer with up to ``n_streams`` batches in
    flight, each sampling loop on its own HIP stream.  At batch 8 the UNet's kernels are small (4 - 400 workgroups): a second decode's kernels
    run beside them.  The host enqueues one whole decode after the other (the scheduler object's host state is never shared mid-loop); the UNet
    engine keeps one scratch set per stream.  ``generator`` in decode_kwargs may be a list (one per batch)."""
    import torch
    token_batches = list(token_batches)
    if not token_batches:
        return []
    gens = decode_kwargs.pop("generator", None)
    gen_of = (lambda i: gens[i]) if isinstance(gens, (list, tuple)) else (lambda i: gens)
    with torch.no_grad():
        first = model.decode_tokens(token_batches[0], generator=gen_of(0), **decode_kwargs)      # builds the cached weight images
        if n_streams <= 1 or len(token_batches) == 1:
            return [first] + [model.decode_tokens(t, generator=gen_of(i + 1), **decode_kwargs) for i, t in enumerate(token_batches[1:])]
        cur = torch.cuda.current_stream()
        pool = getattr(model, "_dec_streams", None)
        if pool is None or len(pool) < n_streams or pool[0].device != cur.device:
            pool = [torch.cuda.Stream(device=cur.device) for _ in range(n_streams)]
            object.__setattr__(model, "_dec_streams", pool)
        ready = torch.cuda.Event()
        ready.record(cur)
        out = [first]
        for i, t in enumerate(token_batches[1:]):
            s = pool[i % n_streams]
            if i < n_streams:
                s.wait_event(ready)
            with torch.cuda.stream(s):
                out.append(model.decode_tokens(t, generator=gen_of(i + 1), **decode_kwargs))
                t.record_stream(s)
                out[-1].record_stream(cur)
        for s in pool[:min(n_streams, len(token_batches) - 1)]:
            cur.wait_stream(s)
    return out


def get_image_tokenizer(tokenizer_id: str, tokenizers_root: str = "./tokenizer_ckpts", encoder_only: bool = False, device: str = "cuda",
                        verbose: bool = True, return_None_on_fail: bool = False):
    """Load a tokenizer checkpoint saved by the upstream trainers: ``{root}/{id}.pth`` = {'model': state_dict, 'args': Namespace}.

    Follows upstream's loader step by step (``fourm/vq/__init__.py:8-79``):
      * renamed arguments: feature-map tokenizers (CLIP / DINO / ImageBind domains) take no patch projection; SAM-instance tokenizers run
        at ``mask_size``; ``quantizer_type / encoder_type / decoder_type / input_size* / quantizer_ema_decay`` get their constructor names;
      * ``n_labels, n_channels`` come from ``cls_emb.weight`` (semantic segmentation), else ``n_channels`` from the encoder's input layer;
      * ``encoder_only`` builds ``VQ`` from the checkpoint minus every decoder / post_quant_proj tensor, otherwise the model type is read
        off the checkpoint (controlnet keys -> VQControlNet, ``beta_schedule`` -> DiVAE, else VQVAE) and every argument of the run is
        forwarded to the constructor (``out_conv``, ``patch_size_dec``, ``image_size_enc`` ... included).
    DiVAE checkpoints (conditional-UNet diffusion decoder) build ``fourm.vq.DiVAE``; VQControlNet loads with ``encoder_only=True`` only.
    Unlike upstream (strict=False and a printed message) unexpected MISSING keys raise: a tokenizer with random weights is never returned."""
    import torch
    path = os.path.join(tokenizers_root, f"{tokenizer_id}.pth")
    if return_None_on_fail and not os.path.exists(path):
        return None
    if verbose:
        print(f"Loading tokenizer {tokenizer_id} ... ", end="")
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    a, sd = ckpt["args"], ckpt["model"]
    domain = getattr(a, "domain", "") or ""
    if any(t in domain for t in ("CLIP", "DINO", "ImageBind")):
        a.patch_proj = False
    elif "sam" in domain:
        a.input_size_min = a.input_size_max = a.input_size = a.mask_size
    renames = dict(quant_type="quantizer_type", enc_type="encoder_type", dec_type="decoder_type", image_size_enc="input_size_enc",
                   image_size_dec="input_size_dec", image_size_sd="input_size_sd", ema_decay="quantizer_ema_decay", enable_xformer="use_xformer")
    for new_name, old_name in renames.items():
        setattr(a, new_name, getattr(a, old_name, None))
    a.image_size = getattr(a, "input_size", None) or getattr(a, "input_size_max", None)
    if "cls_emb.weight" in sd:
        a.n_labels, a.n_channels = sd["cls_emb.weight"].shape
    elif "encoder.linear_in.weight" in sd:
        a.n_channels = sd["encoder.linear_in.weight"].shape[1]
    else:
        a.n_channels = sd["encoder.proj.weight"].shape[1]
    a.sync_codebook = False
    if encoder_only:
        model_type = VQ
        sd = {k: v for k, v in sd.items() if "decoder" not in k and "post_quant_proj" not in k}
    else:
        a.model_type = "VQControlNet" if any("controlnet" in k for k in sd) else "DiVAE" if hasattr(a, "beta_schedule") else "VQVAE"
        if a.model_type == "VQControlNet":
            raise NotImplementedError("VQControlNet (a Stable-Diffusion ControlNet from `diffusers`) has no HIP decoder - load it with encoder_only=True")
        model_type = DiVAE if a.model_type == "DiVAE" else VQVAE
    kw = {k: v for k, v in vars(a).items() if v is not None or k in ("n_labels", "image_size_enc", "image_size_dec")}
    if not isinstance(kw.get("config"), dict):
        kw.pop("config", None)        # (a trainer's config FILE path is not a model configuration)
    model = model_type(**kw)
    msg = model.load_state_dict(sd, strict=False)
    if verbose:
        print(msg)
    missing = [k for k in msg.missing_keys if not k.endswith("pos_emb")]      # (fixed sin-cos tables are rebuilt by the constructor)
    if missing:
        raise RuntimeError(f"tokenizer checkpoint {path} lacks {len(missing)} tensors the model needs, e.g. {missing[:4]}")
    return model.to(device).eval(), a


from fourm import _upstream as _up

_up.extend_path(__name__, __path__)
