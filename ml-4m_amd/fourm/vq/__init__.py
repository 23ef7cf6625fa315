"""VQ tokenizers: ViT encoder + cosine-similarity codebook search (``VQ``), plus the ViT decoder and the training path (``VQVAE``)."""
import os

from .vqvae import VQ, VQVAE, DiVAE


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
