"""VQ tokenizers: ViT encoder + cosine-similarity codebook search (``VQ``), plus the ViT decoder and the training path (``VQVAE``)."""
import os

from .vqvae import VQ, VQVAE


def get_image_tokenizer(tokenizer_id: str, tokenizers_root: str = "./tokenizer_ckpts", encoder_only: bool = False, device: str = "cuda",
                        verbose: bool = True, return_None_on_fail: bool = False):
    """Load a tokenizer checkpoint saved by the upstream trainers (``fourm/vq/__init__.py:8-79``):
    ``{root}/{id}.pth`` = {'model': state_dict, 'args': Namespace}.  ``encoder_only`` (or a checkpoint without ``decoder_type``) builds
    ``VQ``; otherwise ``VQVAE`` with its ViT decoder.  Diffusion-decoder checkpoints (DiVAE / VQControlNet) are rejected."""
    import torch
    path = os.path.join(tokenizers_root, f"{tokenizer_id}.pth")
    if return_None_on_fail and not os.path.exists(path):
        return None
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    a = ckpt["args"]
    if not encoder_only and (hasattr(a, "beta_schedule") or any("controlnet" in k for k in ckpt["model"])):
        raise NotImplementedError("diffusion-decoder tokenizers (DiVAE / VQControlNet) are not implemented: pass encoder_only=True")
    with_dec = not encoder_only and getattr(a, "decoder_type", None) and any(k.startswith("decoder.") for k in ckpt["model"])
    extra = dict(dec_type=a.decoder_type, image_size_dec=getattr(a, "input_size_dec", None)) if with_dec else {}
    model = (VQVAE if with_dec else VQ)(**extra, image_size=a.input_size[getattr(a, "domain", None)] if isinstance(a.input_size, dict) else a.input_size,
               n_channels=getattr(a, "n_channels", 3), enc_type=a.encoder_type, patch_proj=getattr(a, "patch_proj", True),
               post_mlp=getattr(a, "post_mlp", False), patch_size=a.patch_size, quant_type=getattr(a, "quantizer_type", "lucid"),
               codebook_size=a.codebook_size, num_codebooks=getattr(a, "num_codebooks", 1), latent_dim=a.latent_dim,
               norm_codes=getattr(a, "norm_codes", True), norm_latents=getattr(a, "norm_latents", False), sync_codebook=False)
    keep = ("encoder.", "quant_proj.", "quantize.") + (("decoder.", "post_quant_proj.") if with_dec else ())
    sd = {k: v for k, v in ckpt["model"].items() if k.startswith(keep)}
    msg = model.load_state_dict(sd, strict=False)
    if verbose:
        print(msg)
    return model.to(device).eval(), a

from fourm import _upstream as _up

_up.extend_path(__name__, __path__)
