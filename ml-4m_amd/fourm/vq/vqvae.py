"""``VQ``: tokenizer front half (image -> ViT encoder -> 1x1 projection -> nearest code), API of upstream
``fourm/vq/vqvae.py`` (``VQ`` :39-331: constructor arguments, ``encode`` / ``tokenize`` / ``tokens_to_embedding``,
state_dict keys), and ``VQVAE`` (:396-495): the ViT decoder behind the codebook (``decode_quant`` / ``decode_tokens`` / ``autoencode``)
and the gradient path of tokenizer training (``forward`` in training mode returns autograd-connected ``dec, code_loss``; the backward is
hand-written, fourm/vq/engine.py), and ``DiVAE`` (:498-764): the conditional-UNet diffusion detokenizer, inference (``decode_quant`` /
``decode_tokens`` / ``autoencode`` / ``forward`` with given noised inputs) on fourm.vq.models.unet + fourm.vq.scheduling.  VQControlNet
(a Stable-Diffusion ControlNet from ``diffusers``) stays upstream's.

Precision = upstream's autocast arithmetic: the 12 ViT blocks with bf16 GEMM operands (fp32 accumulate, fp32 residual /
LayerNorm / softmax); the tanh post-MLP, the 1x1 projection and the codebook search in exact fp32 (upstream disables
autocast there, vit_models.py:494-496, quantize_lucid.py:388-390).  Code assignment given identical latents is bit-identical
(tests); end-to-end token agreement against the all-fp32 upstream run is measured and asserted by the tests."""
import copy
from typing import Any, Dict, List, Optional, Tuple, Union

import torch
import torch.nn as nn

from .models import vit_models
from .quantizers import VectorQuantizerLucid

try:
    from huggingface_hub import PyTorchModelHubMixin
except Exception:  # pragma: no cover
    class PyTorchModelHubMixin:  # type: ignore
        pass


class VQ(nn.Module, PyTorchModelHubMixin):
    def __init__(self, image_size: int = 224, image_size_enc: Optional[int] = None, n_channels: str = 3, n_labels: Optional[int] = None,
                 enc_type: str = "vit_b_enc", patch_proj: bool = True, post_mlp: bool = False, patch_size: int = 16, quant_type: str = "lucid",
                 codebook_size: Union[int, str] = 16384, num_codebooks: int = 1, latent_dim: int = 32, norm_codes: bool = True,
                 norm_latents: bool = False, sync_codebook: bool = True, ema_decay: float = 0.99, threshold_ema_dead_code: float = 0.25,
                 code_replacement_policy: str = "batch_random", commitment_weight: float = 1.0, kmeans_init: bool = False,
                 ckpt_path: Optional[str] = None,
                 ignore_keys: List[str] = ["decoder", "loss", "post_quant_conv", "post_quant_proj", "encoder.pos_emb"],
                 freeze_enc: bool = False, undo_std: bool = False, config: Optional[Dict[str, Any]] = None, **kwargs):
        if config is not None:
            self.__init__(**copy.deepcopy(config))
            return
        super().__init__()
        if undo_std and (n_channels != 3 or n_labels is not None):
            raise ValueError("undo_std expects ImageNet-standardised RGB input")
        if quant_type != "lucid":
            raise NotImplementedError(f"quant_type {quant_type!r} has no HIP kernel")
        if "vit" not in enc_type or not hasattr(vit_models, enc_type):
            raise NotImplementedError(f"{enc_type} not implemented.")
        for k, v in dict(image_size=image_size, n_channels=n_channels, n_labels=n_labels, enc_type=enc_type, patch_proj=patch_proj,
                         post_mlp=post_mlp, patch_size=patch_size, quant_type=quant_type, codebook_size=codebook_size,
                         num_codebooks=num_codebooks, latent_dim=latent_dim, norm_codes=norm_codes, norm_latents=norm_latents,
                         sync_codebook=sync_codebook, ema_decay=ema_decay, threshold_ema_dead_code=threshold_ema_dead_code,
                         code_replacement_policy=code_replacement_policy, commitment_weight=commitment_weight, kmeans_init=kmeans_init,
                         ckpt_path=ckpt_path, ignore_keys=ignore_keys, freeze_enc=freeze_enc, undo_std=undo_std).items():
            setattr(self, k, v)
        # semantic segmentation (vqvae.py:141-146): class maps (B, H, W) are embedded by a learned table before the patch projection
        self.cls_emb = nn.Embedding(num_embeddings=n_labels, embedding_dim=n_channels) if n_labels is not None else None
        if n_labels is not None:
            self.colorize = torch.randn(3, n_labels, 1, 1)
        self.encoder = getattr(vit_models, enc_type)(in_channels=n_channels, patch_size=patch_size, resolution=image_size_enc or image_size,
                                                     patch_proj=patch_proj, post_mlp=post_mlp)
        self.enc_dim = self.encoder.dim_tokens
        self.quant_proj = torch.nn.Conv2d(self.enc_dim, self.latent_dim, 1)
        self.quantize = VectorQuantizerLucid(dim=latent_dim, codebook_size=codebook_size, codebook_dim=latent_dim, heads=num_codebooks,
                                             use_cosine_sim=norm_codes, threshold_ema_dead_code=threshold_ema_dead_code,
                                             code_replacement_policy=code_replacement_policy, sync_codebook=sync_codebook, decay=ema_decay,
                                             commitment_weight=commitment_weight, norm_latents=norm_latents, kmeans_init=kmeans_init)
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys)
        if freeze_enc:
            for name in ("encoder", "quant_proj", "quantize", "cls_emb"):
                mod = getattr(self, name, None)
                if mod is not None:
                    for p in mod.parameters():
                        p.requires_grad = False

    def init_from_ckpt(self, path: str, ignore_keys: List[str] = list()) -> "VQ":
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
        sd = ckpt["model"] if "model" in ckpt else ckpt["state_dict"]
        for k in list(sd.keys()):
            if any(k.startswith(ik) for ik in ignore_keys):
                del sd[k]
        print(self.load_state_dict(sd, strict=False))
        return self

    def prepare_input(self, x: torch.Tensor) -> torch.Tensor:
        """Upstream (vqvae.py:269-286) denormalises (undo_std) and embeds class maps (n_labels) here; this package folds both into the
        patch gather of the encoder (fm_vq_patchify_ex), so the input passes through unchanged - see ``_prep``."""
        return x

    def _prep(self):
        """What the patch gather applies to the raw input: class-embedding table and / or a per-channel affine map."""
        if self.cls_emb is None and not self.undo_std:
            return None
        prep = dict(cls_emb=self.cls_emb.weight if self.cls_emb is not None else None, scale=None, shift=None)
        if self.undo_std:      # 2 * denormalize(x) - 1 with the ImageNet statistics (fourm/utils/misc.py:23-37): (2 std) x + (2 mean - 1)
            dev = self.quant_proj.weight.device
            mean, std = torch.tensor((0.485, 0.456, 0.406), device=dev), torch.tensor((0.229, 0.224, 0.225), device=dev)
            prep["scale"], prep["shift"] = (2.0 * std).contiguous(), (2.0 * mean - 1.0).contiguous()
        return prep

    def to_rgb(self, x: torch.Tensor) -> torch.Tensor:
        """Class scores / embeddings (B, n_labels, H, W) -> a pseudo-colour image for visualisation (vqvae.py:288-300)."""
        x = torch.nn.functional.conv2d(x, weight=self.colorize.to(x))
        return (x - x.min()) / (x.max() - x.min())

    def encode(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.LongTensor]:
        """(quant (B, latent_dim, h, w) f32, code_loss (1,), tokens (B, h, w) int64)   [vqvae.py:302-318]"""
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("VQ.encode has no gradient path by itself (upstream trains the tokenizer through VQVAE.forward): use "
                                      "VQVAE.forward, run under torch.no_grad(), freeze the parameters or call .eval()")
        with torch.no_grad():
            return self._encode(x)

    def _encode(self, x):
        from .engine import vq_encode
        quant, loss, tokens = vq_encode(self, self.prepare_input(x))
        if self.training and self.quantize.training:
            # upstream's training-mode quantizer: EMA codebook update + the commitment term's VALUE (quantize_lucid.py:409-426, :540-548)
            z = self._last_latents                                    # (B, G, d) f32, as fed to the quantizer
            cb = self.quantize._codebook
            cb.ema_update_(z, tokens)
            if self.quantize.commitment_weight > 0:
                zq = torch.nn.functional.normalize(z, dim=-1) if self.quantize.norm_latents else z
                loss = (torch.nn.functional.mse_loss(quant.flatten(2).transpose(1, 2), zq) * self.quantize.commitment_weight).reshape(1)
        return quant, loss, tokens

    def tokenize(self, x: torch.Tensor) -> torch.LongTensor:
        return self.encode(x)[2]

    def tokens_to_embedding(self, tokens: torch.LongTensor) -> torch.Tensor:
        return self.quantize.indices_to_embedding(tokens)

    def forward(self, x: torch.Tensor):
        return self.encode(x)


class VQVAE(VQ):
    """Encoder + discrete bottleneck + ViT decoder (upstream ``VQVAE``, vqvae.py:396-495): same constructor, state_dict keys
    (``decoder.*``, ``post_quant_proj.*``) and methods.  In training mode ``forward`` is differentiable: ``dec`` and ``code_loss`` carry a
    hand-written backward (straight-through estimator + commitment term into the encoder, vq/quantizers/quantize_lucid.py:533-541)."""

    def __init__(self, dec_type: str = "vit_b_dec", out_conv: bool = False, image_size_dec: int = None, patch_size_dec: int = None,
                 config: Optional[Dict[str, Any]] = None, *args, **kwargs):
        if config is not None:
            self.__init__(**copy.deepcopy(config))
            return
        ckpt_path = kwargs.get("ckpt_path", None)                 # (loaded once the decoder exists)
        kwargs["ckpt_path"] = None
        super().__init__(*args, **kwargs)
        self.ckpt_path = ckpt_path
        if "vit" not in dec_type or not hasattr(vit_models, dec_type):
            raise NotImplementedError(f"{dec_type} not implemented.")
        self.dec_type, self.out_conv = dec_type, out_conv
        self.decoder = getattr(vit_models, dec_type)(out_channels=self.n_channels if self.n_labels is None else self.n_labels, patch_size=patch_size_dec or self.patch_size,
                                                     resolution=image_size_dec or self.image_size, out_conv=out_conv, post_mlp=self.post_mlp,
                                                     patch_proj=self.patch_proj)
        self.dec_dim = self.decoder.dim_tokens
        self.post_quant_proj = torch.nn.Conv2d(self.latent_dim, self.dec_dim, 1)
        if self.ckpt_path is not None:
            self.init_from_ckpt(self.ckpt_path, ignore_keys=self.ignore_keys)

    @torch.no_grad()
    def decode_quant(self, quant: torch.Tensor, **kwargs) -> torch.Tensor:
        from .engine import vqvae_decode_quant
        return vqvae_decode_quant(self, quant)

    @torch.no_grad()
    def decode_tokens(self, tokens: torch.LongTensor, **kwargs) -> torch.Tensor:
        from .engine import vqvae_decode_tokens
        return vqvae_decode_tokens(self, tokens)

    def forward(self, x: torch.Tensor, **kwargs) -> Tuple[torch.Tensor, torch.Tensor]:
        """(dec (B, C, H, W), code_loss (1,))   [vqvae.py:467-481]"""
        from .engine import VQVAEStep, vqvae_train_forward
        x = self.prepare_input(x)
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if needs_grad:
            # the autograd anchor is a parameter that DOES require grad (post_quant_proj may be frozen while the encoder or decoder trains)
            anchor = next(p for p in self.parameters() if p.requires_grad)
            return VQVAEStep.apply(anchor, self, x)
        with torch.no_grad():
            dec, code_loss, _ = vqvae_train_forward(self, x) if self.training else (*self._eval_forward(x), None)
        return dec, code_loss

    def _eval_forward(self, x):
        _, code_loss, tokens = self.encode(x)
        return self.decode_tokens(tokens), code_loss

    def autoencode(self, x: torch.Tensor, **kwargs) -> torch.Tensor:
        return self.forward(x)[0]


class DiVAE(VQ):
    """Encoder + discrete bottleneck + diffusion decoder (upstream ``DiVAE``, vqvae.py:498-764): same constructor, state_dict keys
    (``decoder.*`` = the conditional UNet) and methods.  Inference runs on the HIP kernels: ``decode_quant`` / ``decode_tokens`` /
    ``autoencode`` sample with the pipeline of fourm.vq.scheduling, ``forward(input_clean, input_noised, timesteps)`` evaluates the decoder
    once (no gradient path: the detokenizers are trained upstream).  ``uvit_*`` decoders are not built."""

    def __init__(self, dec_type: str = "unet_patched", num_train_timesteps: int = 1000, cls_free_guidance_dropout: float = 0.0, masked_cfg: bool = False,
                 masked_cfg_low: int = 0, masked_cfg_high: Optional[int] = None, scheduler: str = "ddpm", beta_schedule: str = "squaredcos_cap_v2",
                 prediction_type: str = "v_prediction", clip_sample: bool = False, thresholding: bool = True, conditioning: str = "concat",
                 dec_transformer_dropout: float = 0.2, zero_terminal_snr: bool = True, image_size_dec: Optional[int] = None,
                 config: Optional[Dict[str, Any]] = None, *args, **kwargs):
        if config is not None:
            self.__init__(**copy.deepcopy(config))
            return
        ckpt_path = kwargs.get("ckpt_path", None)                 # (loaded once the decoder exists)
        kwargs["ckpt_path"] = None
        super().__init__(*args, **kwargs)
        self.ckpt_path = ckpt_path
        from .models import unet
        from .scheduling import DDIMScheduler, DDPMScheduler, PipelineCond
        self.dec_type, self.num_train_timesteps, self.beta_schedule, self.prediction_type = dec_type, num_train_timesteps, beta_schedule, prediction_type
        self.clip_sample, self.thresholding, self.zero_terminal_snr = clip_sample, thresholding, zero_terminal_snr
        self.cfg_dist = torch.distributions.Bernoulli(probs=cls_free_guidance_dropout) if cls_free_guidance_dropout > 0.0 else None
        self.masked_cfg, self.masked_cfg_low, self.masked_cfg_high = masked_cfg, masked_cfg_low, masked_cfg_high
        if "unet_" not in dec_type or not hasattr(unet, dec_type):
            raise NotImplementedError(f"dec_type {dec_type} not implemented (HIP decoders: unet_patched).")
        self.decoder = getattr(unet, dec_type)(in_channels=self.n_channels, out_channels=self.n_channels, cond_channels=self.latent_dim,
                                               image_size=image_size_dec or self.image_size)
        cls = DDPMScheduler if scheduler == "ddpm" else DDIMScheduler
        self.noise_scheduler = cls(num_train_timesteps=num_train_timesteps, thresholding=thresholding, clip_sample=clip_sample, beta_schedule=beta_schedule,
                                   prediction_type=prediction_type, zero_terminal_snr=zero_terminal_snr)
        self.pipeline = PipelineCond(model=self.decoder, scheduler=self.noise_scheduler)
        if self.ckpt_path is not None:
            self.init_from_ckpt(self.ckpt_path, ignore_keys=self.ignore_keys)

    def sample_mask(self, quant: torch.Tensor, low: int = 0, high: Optional[int] = None) -> torch.BoolTensor:
        """(B, H_Q, W_Q) bool, True = conditioning masked out: a uniform number of tokens in [low, high] per sample (vqvae.py:618-638)."""
        B, _, hq, wq = quant.shape
        n = hq * wq
        high = high if high is not None else n
        zero_idxs = torch.randint(low=low, high=high + 1, size=(B,), device=quant.device)
        order = torch.argsort(torch.rand(B, n, device=quant.device), dim=1)
        return torch.where(order < zero_idxs.unsqueeze(1), 0, 1).reshape(B, hq, wq).bool()

    def _get_pipeline(self, scheduler=None):
        from .scheduling import PipelineCond
        return PipelineCond(model=self.decoder, scheduler=scheduler) if scheduler is not None else self.pipeline

    @torch.no_grad()
    def decode_quant(self, quant: torch.Tensor, timesteps: Optional[int] = None, scheduler=None, generator: Optional[torch.Generator] = None,
                     image_size=None, verbose: bool = False, scheduler_timesteps_mode: str = "trailing", orig_res=None) -> torch.Tensor:
        """quant (B, latent_dim, h, w) -> image (B, C, H, W): ``timesteps`` denoising steps of the pipeline (vqvae.py:640-672)."""
        return self._get_pipeline(scheduler)(quant, timesteps=timesteps, generator=generator, image_size=image_size, verbose=verbose,
                                             scheduler_timesteps_mode=scheduler_timesteps_mode, orig_res=orig_res)

    @torch.no_grad()
    def decode_tokens(self, tokens: torch.LongTensor, **kwargs) -> torch.Tensor:
        return self.decode_quant(self.tokens_to_embedding(tokens), **kwargs)

    @torch.no_grad()
    def autoencode(self, input_clean: torch.Tensor, timesteps: Optional[int] = None, scheduler=None, generator: Optional[torch.Generator] = None,
                   verbose: bool = True, scheduler_timesteps_mode: str = "trailing", orig_res=None, **kwargs) -> torch.Tensor:
        quant, _, _ = self.encode(input_clean)
        return self._get_pipeline(scheduler)(quant, timesteps=timesteps, generator=generator, image_size=input_clean.shape[-1], verbose=verbose,
                                             scheduler_timesteps_mode=scheduler_timesteps_mode, orig_res=orig_res)

    def forward(self, input_clean: torch.Tensor, input_noised: torch.Tensor, timesteps, cond_mask: Optional[torch.Tensor] = None, orig_res=None):
        """(dec, code_loss): encode the clean input, evaluate the diffusion decoder on the noised one (vqvae.py:716-764).  No gradient path."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.decoder.parameters()) and self.training:
            raise NotImplementedError("DiVAE.forward has no backward here (the diffusion decoder is inference-only): call under torch.no_grad() / .eval()")
        with torch.no_grad():
            quant, code_loss, _ = self._encode(input_clean)
            if cond_mask is None and self.cfg_dist is not None and self.training:
                B, _, hq, wq = quant.shape
                cond_mask = self.cfg_dist.sample((B,)).to(quant.device, dtype=torch.bool)[:, None, None].expand(B, hq, wq)
                if self.masked_cfg:
                    cond_mask = self.sample_mask(quant, low=self.masked_cfg_low, high=self.masked_cfg_high) * cond_mask
            dec = self.decoder(input_noised, timesteps, quant, cond_mask=cond_mask, orig_res=orig_res)
        return dec, code_loss


# names only upstream's same-named module defines resolve lazily (see fourm/_upstream.py)
from fourm import _upstream as _up
__getattr__ = _up.fallthrough(__name__, is_package=False)
