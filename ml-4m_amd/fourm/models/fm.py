"""4M masked encoder-decoder model on MI355X.

API-compatible with upstream ``fourm/models/fm.py`` (``FourM`` :54, ``FM`` :783, factories :839-1130):
same constructor arguments, attribute names, parameter tree / state_dict keys, ``forward`` contract and
sub-methods.  The computation is different in kind: ``forward`` hands the whole step to
``fourm.hip.engine.FourMEngine`` — fused select+embed, bf16 MFMA GEMMs with fused epilogues, fused masked
attention with in-kernel masks, segmented heads — and gradients come from a hand-written backward, not
from an autograd graph.
"""
import copy
import math
import weakref
from functools import partial
from typing import Any, Dict, Optional, Tuple, Union

import torch
import torch.nn as nn

from fourm.utils.registry import register_model
from .fm_utils import Block, DecoderBlock, LayerNorm

try:  # the hub mixin only adds from_pretrained / push_to_hub
    from huggingface_hub import PyTorchModelHubMixin
except Exception:  # pragma: no cover
    class PyTorchModelHubMixin:  # type: ignore
        pass

__all__ = ["FourM", "FM"]


class _TrainStep(torch.autograd.Function):
    """Bridges the hand-written backward into ``loss.backward()``: forward returns the loss computed by
    the engine, backward receives d(objective)/d(loss) and launches ``train_backward``."""

    @staticmethod
    def forward(ctx, anchor, engine, total):
        ctx.engine = engine
        return total.clone()

    @staticmethod
    def backward(ctx, grad_out):
        eng = ctx.engine
        # a fresh accumulation window = the trainer cleared the gradients (zero_grad(set_to_none=True)) since the last
        # backward; judged on the TRAINABLE parameters this engine attached last time (frozen ones never get a grad)
        eng.attach_grads(zero=eng.grads_were_cleared(), untouched=eng.untouched_params())
        eng.train_backward(grad_out.reshape(1).float().contiguous())
        return None, None, None


class FourM(nn.Module):
    """See upstream ``FourM`` (fm.py:54-105) for the meaning of the arguments; they are identical."""

    def __init__(self,
                 encoder_embeddings: Dict[str, nn.Module],
                 decoder_embeddings: Dict[str, nn.Module],
                 modality_info: Dict[str, Any],
                 dim: int = 768,
                 encoder_depth: int = 12,
                 decoder_depth: int = 12,
                 num_heads: int = 12,
                 mlp_ratio: float = 4.0,
                 qkv_bias: bool = True,
                 proj_bias: bool = True,
                 mlp_bias: bool = True,
                 drop_path_rate_encoder: float = 0.0,
                 drop_path_rate_decoder: float = 0.0,
                 shared_drop_path: bool = False,
                 act_layer: nn.Module = nn.GELU,
                 norm_layer: Union[partial, nn.Module] = partial(LayerNorm, eps=1e-6),
                 gated_mlp: bool = False,
                 qk_norm: bool = False,
                 decoder_causal_mask: bool = False,
                 decoder_sep_mask: bool = True,
                 num_register_tokens: int = 0,
                 use_act_checkpoint: bool = False,
                 share_modality_embeddings: bool = True,
                 ):
        super().__init__()
        self.modality_info = modality_info
        self.dim = dim
        self.decoder_causal_mask = decoder_causal_mask
        self.decoder_sep_mask = decoder_sep_mask
        self.init_std = 0.02
        self.use_act_checkpoint = use_act_checkpoint
        self.num_register_tokens = num_register_tokens
        # "bf16": the hot path (upstream's autocast arithmetic); "fp32": verification kernels without any rounding (set before the
        # first forward, or afterwards followed by ``model._engine = None``)
        self.compute_precision = "bf16"

        self.encoder_modalities = set(encoder_embeddings.keys())
        self.decoder_modalities = set(decoder_embeddings.keys())
        for emb in list(encoder_embeddings.values()) + list(decoder_embeddings.values()):
            emb.init(dim_tokens=dim, init_std=self.init_std)
        self.encoder_embeddings = nn.ModuleDict(encoder_embeddings)
        self.decoder_embeddings = nn.ModuleDict(decoder_embeddings)
        if share_modality_embeddings:
            self.share_modality_embeddings()

        def rates(rate, depth, offset, total):
            if shared_drop_path:
                return [x.item() for x in torch.linspace(0, rate, total)][offset:offset + depth]
            return [x.item() for x in torch.linspace(0, rate, depth)]
        blk = dict(dim=dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, proj_bias=proj_bias, mlp_bias=mlp_bias,
                   act_layer=act_layer, norm_layer=norm_layer, gated_mlp=gated_mlp, qk_norm=qk_norm)
        total = encoder_depth + decoder_depth
        self.encoder = nn.ModuleList([Block(drop_path=r, **blk) for r in rates(drop_path_rate_encoder, encoder_depth, 0, total)])
        self.encoder_norm = norm_layer(dim)
        self.decoder_proj_context = nn.Linear(dim, dim)
        self.decoder = nn.ModuleList([DecoderBlock(drop_path=r, **blk)
                                      for r in rates(drop_path_rate_decoder, decoder_depth, encoder_depth, total)])
        self.decoder_norm = norm_layer(dim)

        self.mask_token = nn.Parameter(torch.zeros(1, 1, dim))
        nn.init.normal_(self.mask_token, std=self.init_std)
        if num_register_tokens > 0:
            self.register_tokens = nn.Parameter(torch.zeros(1, num_register_tokens, dim))
            nn.init.normal_(self.register_tokens, std=self.init_std)
        else:
            self.register_tokens = None
        self.init_weights()
        self._engine = None
        try:        # stand-alone ``blk(x, mask)`` calls find the engine through this registry (fourm/hip/functional.py)
            from fourm.hip.functional import register_blocks
            register_blocks(self)
        except (ImportError, OSError):      # no (loadable) libfourm_hip.so: the model is a parameter container only (state_dict tools)
            pass

    # ------------------------------------------------------------------------------------------
    def share_modality_embeddings(self):
        """One ``mod_emb`` Parameter per modality, shared by its encoder and decoder embedder (fm.py:176-180)."""
        for mod in self.encoder_modalities & self.decoder_modalities:
            self.decoder_embeddings[mod].mod_emb = self.encoder_embeddings[mod].mod_emb

    def init_weights(self):
        """MAE-style initialisation: Xavier-uniform Linears with the fused qkv / kv matrices treated as
        separate square blocks, unit LayerNorms, N(0, 0.02) embeddings (fm.py:182-216)."""
        for name, mod in self.named_modules():
            if "tokenizer" in name:
                continue
            if isinstance(mod, nn.Linear):
                fused = 3 if "qkv" in name else 2 if "kv" in name else 1
                if fused > 1:
                    bound = math.sqrt(6. / float(mod.weight.shape[0] // fused + mod.weight.shape[1]))
                    nn.init.uniform_(mod.weight, -bound, bound)
                else:
                    nn.init.xavier_uniform_(mod.weight)
                if mod.bias is not None:
                    nn.init.constant_(mod.bias, 0)
            elif isinstance(mod, (nn.LayerNorm, LayerNorm)):
                nn.init.constant_(mod.weight, 1.0)
                if mod.bias is not None:
                    nn.init.constant_(mod.bias, 0)
            elif isinstance(mod, nn.Embedding):
                nn.init.normal_(mod.weight, std=self.init_std)

    def get_num_layers_encoder(self):
        return len(self.encoder)

    def get_num_layers_decoder(self):
        return len(self.decoder)

    def get_num_layers(self):
        return len(self.encoder) + len(self.decoder)

    @torch.jit.ignore
    def no_weight_decay(self):
        skip = set()
        for side in ("encoder_embeddings", "decoder_embeddings"):
            for mod, emb in getattr(self, side).items():
                if hasattr(emb, "no_weight_decay"):
                    skip |= {f"{side}.{mod}.{n}" for n in emb.no_weight_decay()}
        return skip

    # ------------------------------------------------------------------------------------------
    # engine plumbing
    # ------------------------------------------------------------------------------------------
    @property
    def engine(self):
        from fourm.hip.engine import FourMEngine
        if self._engine is None:
            if not self.mask_token.is_cuda:
                raise RuntimeError("FourM computes on an MI355X through libfourm_hip.so; move the model to the GPU first "
                                   "(there is no CPU implementation of the hot path)")
            self._engine = FourMEngine(self)
            from fourm.hip.functional import register_blocks
            register_blocks(self)
        return self._engine

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._engine = None          # parameters moved / changed dtype: rebuild stores and shadows lazily
        return out

    def __getstate__(self):
        d = self.__dict__.copy()
        d["_engine"] = None
        return d

    # ------------------------------------------------------------------------------------------
    # selection (upstream sub-API)
    # ------------------------------------------------------------------------------------------
    def cat_encoder_tensors(self, mod_dict):
        """Every position of every encoder modality, concatenated, nothing zeroed   [fm.py:245-277]:
        (tokens (B,O,D), emb (B,O,D), mask (B,O) bool, mod_mask (B,O) int16).  Embeds from ``d['tensor']`` (any
        precomputed ``d['x']`` / ``d['emb']`` are not read)."""
        from fourm.hip import functional as Fh
        Fh._no_grad_only("FourM.cat_encoder_tensors")
        eng = self.engine
        eng.prepare()
        names = [n for n in mod_dict if n in self.encoder_embeddings]
        s = eng.select(mod_dict, 0, False, names, "api.cat_enc.", want_x0=False, raw=1)
        B, Nt, D, R = s["B"], s["Nt"], self.dim, s["R"]
        return (s["tokens"][:R].view(B, Nt, D).clone(), s["emb"][:R].view(B, Nt, D).clone(), s["mask"].clone(), s["mod_mask"].clone())

    def cat_decoder_tensors(self, mod_dict):
        """The decoder-side concatenation   [fm.py:279-336]: modalities in the shuffled order, sequences shifted for
        teacher forcing, grid modalities queried with the mask token:
        (tokens (B,P,D), emb (B,P,D), mask (B,P) bool, target_ids (B,P) int64, attention mask (B,P) int32, mod_mask (B,P) int16)."""
        from fourm.hip import functional as Fh, _lib as _L
        Fh._no_grad_only("FourM.cat_decoder_tensors")
        eng = self.engine
        eng.prepare()
        order = eng.dec_order(mod_dict)
        s = eng.select(mod_dict, 0, True, order, "api.cat_dec.", want_x0=False, raw=1)
        B, Nt, D, R = s["B"], s["Nt"], self.dim, s["R"]
        dams = []
        for n in s["names"]:
            dam = mod_dict[n]["decoder_attention_mask"].reshape(B, -1)
            dams.append(dam[:, :-1] if self.decoder_embeddings[n].kind == _L.KIND_SEQ else dam)      # sequences: teacher forcing
        return (s["tokens"][:R].view(B, Nt, D).clone(), s["emb"][:R].view(B, Nt, D).clone(), s["mask"].clone(),
                s["target_ids"].clone(), torch.cat(dams, 1), s["mod_mask"].clone())

    def forward_mask_encoder(self, mod_dict, num_encoder_tokens: int):
        """(tokens (B,N,D), emb (B,N,D), mask (B,1,N) bool, mod_mask (B,N) int16)   [fm.py:338-390]"""
        eng = self.engine
        eng.prepare()
        names = [n for n in mod_dict if n in self.encoder_embeddings]
        s = eng.select(mod_dict, num_encoder_tokens, False, names, "api.enc.", want_x0=False)
        B, Nt, D, R = s["B"], s["Nt"], self.dim, s["R"]
        return (s["tokens"][:R].view(B, Nt, D).clone(), s["emb"][:R].view(B, Nt, D).clone(), s["mask"].view(B, 1, Nt).clone(),
                s["mod_mask"].clone())

    def forward_mask_decoder(self, mod_dict, num_decoder_tokens: int):
        """(tokens, emb, mask (B,1,M), target_ids (B,M) int64, attention mask (B,M,M) bool, mod_mask)   [fm.py:392-438]"""
        from fourm.hip import ops
        eng = self.engine
        eng.prepare()
        s = eng.select(mod_dict, num_decoder_tokens, True, eng.dec_order(mod_dict), "api.dec.", want_x0=False)
        B, Nt, D, R = s["B"], s["Nt"], self.dim, s["R"]
        dense = ops.dense_decoder_mask(None if self.decoder_causal_mask else s["cs"], s["mod_pre"], B, Nt,
                                       self.decoder_causal_mask, self.decoder_sep_mask)
        return (s["tokens"][:R].view(B, Nt, D).clone(), s["emb"][:R].view(B, Nt, D).clone(), s["mask"].view(B, 1, Nt).clone(),
                s["target_ids"].clone(), dense, s["mod_mask"].clone())

    def adapt_decoder_attention_mask(self, decoder_attention_mask: torch.Tensor, mod_mask: Optional[torch.Tensor] = None):
        """Compressed (B,M) mask -> dense (B,M,M) bool, True = blocked   [fm.py:440-475]"""
        from fourm.hip import ops
        B, M = decoder_attention_mask.shape
        cs = None if self.decoder_causal_mask else decoder_attention_mask.int().cumsum(-1).int().contiguous()
        mod = mod_mask.to(torch.int16).contiguous() if (self.decoder_sep_mask and mod_mask is not None) else None
        if cs is None and mod is None and not self.decoder_causal_mask:
            return torch.zeros(B, M, M, dtype=torch.bool, device=decoder_attention_mask.device)
        if cs is None and mod is None:
            mod = torch.zeros(B, M, dtype=torch.int16, device=decoder_attention_mask.device)
        return ops.dense_decoder_mask(cs, mod, B, M, self.decoder_causal_mask, mod is not None and self.decoder_sep_mask)

    # ------------------------------------------------------------------------------------------
    # trunk (upstream sub-API; inference only — training goes through forward())
    # ------------------------------------------------------------------------------------------
    def forward_encoder(self, x: torch.Tensor, encoder_mask: torch.Tensor) -> torch.Tensor:
        from fourm.hip import functional as Fh, ops
        Fh._no_grad_only("FourM.forward_encoder")
        eng = self.engine
        eng.prepare()
        B, N, D = x.shape
        buf = eng.ws.get("api.x", (ops.ru(B * N, 128), D), torch.float32)
        buf[: B * N] = x.reshape(B * N, D).float()
        mask = Fh._mask_args(encoder_mask, B, N, N)
        cur = buf
        for i, blk in enumerate(self.encoder):
            cur = eng.encoder_block_fwd(blk, cur, B, N, mask, None, f"enc{i % 2}")
        out = torch.empty(B * N, D, dtype=torch.float32, device=x.device)
        ops.layernorm_fwd(cur, self.encoder_norm.weight, self.encoder_norm.bias, out, eps=self.encoder_norm.eps, R=B * N)
        return out.view(B, N, D)

    def forward_decoder(self, y, context, encoder_mask, decoder_attention_mask):
        from fourm.hip import functional as Fh, ops
        Fh._no_grad_only("FourM.forward_decoder")
        eng = self.engine
        eng.prepare()
        B, M, D = y.shape
        N = context.shape[1]
        yb = eng.ws.get("api.y", (ops.ru(B * M, 128), D), torch.float32)
        cb = eng.ws.get("api.c", (ops.ru(B * N, 128), D), torch.float32)
        yb[: B * M] = y.reshape(B * M, D).float()
        cb[: B * N] = context.reshape(B * N, D).float()
        sa, xa = Fh._mask_args(decoder_attention_mask, B, M, M), Fh._mask_args(encoder_mask, B, M, N)
        cur = yb
        for i, blk in enumerate(self.decoder):
            cur = eng.decoder_block_fwd(blk, cur, cb, B, M, N, sa, xa, None, f"dec{i % 2}")
        out = torch.empty(B * M, D, dtype=torch.float32, device=y.device)
        ops.layernorm_fwd(cur, self.decoder_norm.weight, self.decoder_norm.bias, out, eps=self.decoder_norm.eps, R=B * M)
        return out.view(B, M, D)

    def forward_logits(self, y, decoder_mod_dict, decoder_mod_mask, return_all_logits: bool = False):
        """{mod: logits}; per-modality rows (``y[mod_mask == id]``) unless ``return_all_logits``   [fm.py:521-545]"""
        out = {}
        for mod in decoder_mod_dict:
            rows = y if return_all_logits else y[decoder_mod_mask == self.modality_info[mod]["id"]]
            out[mod] = self.decoder_embeddings[mod].forward_logits(rows)
        return out

    # ------------------------------------------------------------------------------------------
    def forward(self, mod_dict, num_encoder_tokens: int, num_decoder_tokens: int, loss_type: str = "mod", return_logits: bool = False):
        """One masked-modeling step   [fm.py:640-691].

        ``mod_dict[mod]`` = {'tensor', 'input_mask', 'target_mask', 'decoder_attention_mask'} (device tensors).
        Returns ``(loss, {mod: loss})`` — ``loss.backward()`` runs the HIP backward and leaves fp32 gradients
        in ``param.grad`` — or, with ``return_logits``, ``{mod: logits (B, M, vocab)}``."""
        eng = self.engine
        if return_logits:
            return self._forward_logits_all(mod_dict, num_encoder_tokens, num_decoder_tokens)
        train = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        total, head_loss, heads, counts = eng.train_forward(mod_dict, num_encoder_tokens, num_decoder_tokens, loss_type, save=train)
        present = [m for m in mod_dict if m in self.decoder_embeddings]
        hl = head_loss.clone()
        mod_loss = {m: hl[heads.index(m)] for m in present}
        if train:
            anchor = next(p for p in self.parameters() if p.requires_grad)
            loss = _TrainStep.apply(anchor, eng, total)
        else:
            loss = total.clone()
        # upstream's loss has shape (1,) when some modality had no target token (zeros(1) enters the sum,
        # fm.py:593-600) and is 0-dim otherwise; the value is what matters downstream (loss.item()).
        return loss.reshape(()), mod_loss

    def _forward_logits_all(self, mod_dict, n_enc, n_dec):
        from fourm.hip import functional as Fh, ops
        Fh._no_grad_only("FourM.forward(return_logits=True)")
        eng = self.engine
        eng.prepare()
        enc = eng.select(mod_dict, n_enc, False, [n for n in mod_dict if n in self.encoder_embeddings], "enc.")
        dec = eng.select(mod_dict, n_dec, True, eng.dec_order(mod_dict), "dec.")
        y, _ = eng.trunk_forward(enc, dec, save=False)
        B, Mt, D = dec["B"], dec["Nt"], self.dim
        R = B * Mt
        yn = eng.ws.get("api.yn", (y.shape[0], D), eng.adt)
        ops.layernorm_fwd(y, self.decoder_norm.weight, self.decoder_norm.bias, yn, eps=self.decoder_norm.eps, R=R)
        out = {}
        for mod in mod_dict:
            if mod not in self.decoder_embeddings:
                continue
            w = self.decoder_embeddings[mod].to_logits.weight
            V = w.shape[0]
            lg = torch.empty(R, ops.ru(V, 4), dtype=eng.adt, device=y.device)
            ops.gemm_nt(yn, eng.w(w), lg, M=R, N=V, K=D)
            out[mod] = lg[:, :V].reshape(B, Mt, V)
        return out

    # ------------------------------------------------------------------------------------------
    # freezing helpers (fm.py:694-776)
    # ------------------------------------------------------------------------------------------
    def _set_trainable(self, modules, flag):
        for m in modules:
            for p in m.parameters():
                p.requires_grad = flag

    def freeze_encoder(self, freeze_embeddings=True):
        self._set_trainable([self.encoder, self.encoder_norm] + ([self.encoder_embeddings] if freeze_embeddings else []), False)

    def unfreeze_encoder(self, unfreeze_embeddings=True):
        self._set_trainable([self.encoder, self.encoder_norm] + ([self.encoder_embeddings] if unfreeze_embeddings else []), True)

    def freeze_decoder(self, freeze_embeddings=True):
        self._set_trainable([self.decoder, self.decoder_norm] + ([self.decoder_embeddings] if freeze_embeddings else []), False)

    def unfreeze_decoder(self, unfreeze_embeddings=True):
        self._set_trainable([self.decoder, self.decoder_norm] + ([self.decoder_embeddings] if unfreeze_embeddings else []), True)

    def _freeze_except(self, trunk, norm, embeddings, frozen_embedding_domain):
        domains = frozen_embedding_domain.split("-")
        self._set_trainable([trunk, norm], False)
        for name, p in embeddings.named_parameters():
            if name.split(".")[0] in domains:
                p.requires_grad = False

    def freeze_encoder_except_specific_embeddings(self, frozen_embedding_domain):
        self._freeze_except(self.encoder, self.encoder_norm, self.encoder_embeddings, frozen_embedding_domain)

    def freeze_decoder_except_specific_embeddings(self, frozen_embedding_domain):
        self._freeze_except(self.decoder, self.decoder_norm, self.decoder_embeddings, frozen_embedding_domain)

    def freeze_shared_params(self):
        self.freeze_encoder(freeze_embeddings=False)
        self.freeze_decoder(freeze_embeddings=False)

    def freeze_params_except_specific_embeddings(self, frozen_embedding_domain):
        self.freeze_encoder_except_specific_embeddings(frozen_embedding_domain)
        self.freeze_decoder_except_specific_embeddings(frozen_embedding_domain)

    def unfreeze_shared_params(self):
        self.unfreeze_encoder(unfreeze_embeddings=False)
        self.unfreeze_decoder(unfreeze_embeddings=False)

    def unfreeze_all(self):
        self.unfreeze_encoder(unfreeze_embeddings=True)
        self.unfreeze_decoder(unfreeze_embeddings=True)


class FM(FourM, PyTorchModelHubMixin):
    """``FM(config)``: build a FourM from the config dict stored with released checkpoints (fm.py:783-831):
    keys ``domains_in``, ``domains_out``, ``image_size``, ``patch_size``, ``norm_bias``, ``act_layer`` plus
    FourM keyword arguments.  Heads are untied (``share_embedding=False``) as upstream does here."""

    def __init__(self, config: dict):
        from fourm.data.modality_info import MODALITY_INFO
        cfg = copy.deepcopy(config)
        domains_in, domains_out = cfg.pop("domains_in"), cfg.pop("domains_out")
        image_size, patch_size = cfg.pop("image_size"), cfg.pop("patch_size")
        info = {m: MODALITY_INFO[m] for m in sorted(set(domains_in) | set(domains_out))}

        def build(mods, key, **extra):
            out = {}
            for m in mods:
                ctor = info[m].get(key)
                if ctor is None:
                    continue
                if info[m]["type"] == "img":
                    out[m] = ctor(patch_size=info[m].get("patch_size", patch_size), image_size=info[m].get("input_size", image_size), **extra)
                else:
                    out[m] = ctor(**extra)
            return out
        cfg["norm_layer"] = partial(LayerNorm, eps=1e-6, bias=cfg.pop("norm_bias"))
        cfg["act_layer"] = getattr(torch.nn, cfg["act_layer"])
        super().__init__(encoder_embeddings=build(domains_in, "encoder_embedding"),
                         decoder_embeddings=build(domains_out, "decoder_embedding", share_embedding=False),
                         modality_info=info, **cfg)


# --------------------------------------------------------------------------------------------------
# named configurations (fm.py:839-1130): (dim, depth, heads)
# --------------------------------------------------------------------------------------------------
_SIZES = {"tiny_6e_6d": (384, 6, 6), "small_8e_8d": (512, 8, 8), "base_12e_12d": (768, 12, 12),
          "large_24e_24d": (1024, 24, 16), "xlarge_24e_24d": (2048, 24, 32)}


def _factory(name, size, **fixed):
    dim, depth, heads = _SIZES[size]

    def build(encoder_embeddings: Dict[str, nn.Module], decoder_embeddings: Dict[str, nn.Module], **kwargs):
        return FourM(encoder_embeddings=encoder_embeddings, decoder_embeddings=decoder_embeddings, encoder_depth=depth,
                     decoder_depth=depth, dim=dim, num_heads=heads, mlp_ratio=4, **fixed, **kwargs)
    build.__name__ = build.__qualname__ = name
    build.__module__ = __name__
    globals()[name] = register_model(build)
    __all__.append(name)


_GELU = dict(qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6))
_SWIGLU = dict(qkv_bias=False, proj_bias=False, mlp_bias=False, norm_layer=partial(LayerNorm, eps=1e-6, bias=False),
               act_layer=nn.SiLU, gated_mlp=True)
for _size in _SIZES:
    _factory(f"fm_{_size}_gelu", _size, **_GELU)
    _factory(f"fm_{_size}_swiglu_nobias", _size, **_SWIGLU)
for _size in ("base_12e_12d", "large_24e_24d", "xlarge_24e_24d"):
    _factory(f"fm_{_size}_swiglu_qknorm_nobias", _size, qk_norm=True, **_SWIGLU)


# names only upstream's same-named module defines resolve lazily (see fourm/_upstream.py)
from fourm import _upstream as _up
__getattr__ = _up.fallthrough(__name__, is_package=False)
