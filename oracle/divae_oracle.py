"""CPU oracle for the diffusion detokenizer (DiVAE decoder): TEST INFRASTRUCTURE, never imported by the product path.

Functional fp32 PyTorch restatement (state dict in, tensors out) of
  * the conditional patched UNet        fourm/vq/models/unet/unet.py:163-322 (ResBlock, AttentionBlock), :345-374 (QKVAttentionLegacy),
                                        :411-690 (UNetModel), :693-744 (PatchedUNetCondCat), nn.py:23-25 (GroupNorm32), :120-140 (timestep_embedding)
  * the DDPM / DDIM schedulers          fourm/vq/scheduling/scheduling_ddim.py:75-120, :151-330; scheduling_ddpm.py:200-330; scheduling_utils.py:19-80
  * the sampling loop                   fourm/vq/scheduling/diffusion_pipeline.py:52-133 (guidance scale 0: one model evaluation per step)
  * DiVAE.decode_quant / decode_tokens  fourm/vq/vqvae.py:640-679

Parity status: PINNED - tests/golden/make_golden_divae.py runs the UNMODIFIED upstream classes (through the inert diffusers stubs of
tests/golden/ref_stubs.py) on the same seeded weights / noise and asserts this file reproduces them before writing tests/golden/divae_small.npz.
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass
class UNetCfg:
    image_size: int = 224
    in_channels: int = 3             # image channels (before the patch projection)
    out_channels: int = 3
    cond_channels: int = 32          # = the tokenizer's latent_dim
    patch_size: int = 4
    model_channels: int = 256
    num_res_blocks: int = 3
    attention_resolutions: Sequence[int] = (4, 8)
    channel_mult: Sequence[int] = (1, 2, 2, 2)
    num_heads: int = 1

    @property
    def in_p(self):
        return self.in_channels * self.patch_size ** 2 + self.cond_channels

    @property
    def out_p(self):
        return self.out_channels * self.patch_size ** 2


def unet_patched_cfg(**kw) -> UNetCfg:
    """``unet_patched`` (unet.py:747-754)."""
    return UNetCfg(patch_size=4, model_channels=256, num_res_blocks=3, attention_resolutions=(4, 8), channel_mult=(1, 2, 2, 2), **kw)


def unet_plan(cfg: UNetCfg):
    """The module tree of UNetModel.__init__ (unet.py:484-640) as a flat plan:
        ("conv3", key, cin, cout) | ("res", key, cin, cout) | ("attn", key, ch) | ("down", key, ch) | ("up", key, ch)
    grouped into input blocks (each pushes a skip), the middle block and output blocks (each pops a skip)."""
    mc, ch = cfg.model_channels, int(cfg.channel_mult[0] * cfg.model_channels)
    inp = [[("conv3", "input_blocks.0.0", cfg.in_p, ch)]]
    chans, ds = [ch], 1
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            i = len(inp)
            blk = [("res", f"input_blocks.{i}.0", ch, int(mult * mc))]
            ch = int(mult * mc)
            if ds in cfg.attention_resolutions:
                blk.append(("attn", f"input_blocks.{i}.1", ch))
            inp.append(blk)
            chans.append(ch)
        if level != len(cfg.channel_mult) - 1:
            inp.append([("down", f"input_blocks.{len(inp)}.0", ch)])
            chans.append(ch)
            ds *= 2
    mid = [("res", "middle_block.0", ch, ch), ("attn", "middle_block.1", ch), ("res", "middle_block.2", ch, ch)]
    out = []
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            ich = chans.pop()
            o = len(out)
            blk = [("res", f"output_blocks.{o}.0", ch + ich, int(mc * mult))]
            ch = int(mc * mult)
            j = 1
            if ds in cfg.attention_resolutions:
                blk.append(("attn", f"output_blocks.{o}.{j}", ch)); j += 1
            if level and i == cfg.num_res_blocks:
                blk.append(("up", f"output_blocks.{o}.{j}", ch))
                ds //= 2
            out.append(blk)
    return inp, mid, out, ch


def unet_param_shapes(cfg: UNetCfg) -> Dict[str, Tuple[int, ...]]:
    """state_dict keys and shapes of upstream's PatchedUNetCondCat for this configuration."""
    mc, te = cfg.model_channels, cfg.model_channels * 4
    sh = {"time_embed.0.weight": (te, mc), "time_embed.0.bias": (te,), "time_embed.2.weight": (te, te), "time_embed.2.bias": (te,)}

    def conv(k, co, ci, ks):
        sh[k + ".weight"] = (co, ci, ks, ks) if ks else (co, ci, 1)
        sh[k + ".bias"] = (co,)

    def gn(k, c):
        sh[k + ".weight"], sh[k + ".bias"] = (c,), (c,)
    inp, mid, out, ch = unet_plan(cfg)
    for blk in inp + [mid] + out:
        for item in blk:
            kind, key = item[0], item[1]
            if kind == "conv3":
                conv(key, item[3], item[2], 3)
            elif kind == "res":
                ci, co = item[2], item[3]
                gn(key + ".in_layers.0", ci); conv(key + ".in_layers.2", co, ci, 3)
                sh[key + ".emb_layers.1.weight"], sh[key + ".emb_layers.1.bias"] = (co, te), (co,)
                gn(key + ".out_layers.0", co); conv(key + ".out_layers.3", co, co, 3)
                if ci != co:
                    conv(key + ".skip_connection", co, ci, 1)
            elif kind == "attn":
                c = item[2]
                gn(key + ".norm", c); conv(key + ".qkv", 3 * c, c, 0); conv(key + ".proj_out", c, c, 0)
            elif kind == "down":
                conv(key + ".op", item[2], item[2], 3)
            elif kind == "up":
                conv(key + ".conv", item[2], item[2], 3)
    gn("out.0", ch); conv("out.2", cfg.out_p, ch, 3)
    return sh


def seeded_unet_state_dict(cfg: UNetCfg, seed: int = 0) -> Dict[str, Tensor]:
    """Deterministic, NON-zero weights for every tensor (upstream zero-initialises the last convolution of every block: a fixture on
    the default init would test nothing behind them)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, s in unet_param_shapes(cfg).items():
        if k.endswith(".bias"):
            sd[k] = torch.randn(s, generator=g) * 0.05
        elif len(s) == 1:
            sd[k] = 1.0 + torch.randn(s, generator=g) * 0.1                 # GroupNorm scale
        else:
            fan_in = int(np.prod(s[1:]))
            sd[k] = torch.randn(s, generator=g) * (0.7 / math.sqrt(fan_in))
    return sd


def timestep_embedding(t: Tensor, dim: int, max_period: float = 10000.0) -> Tensor:
    """nn.py:120-140: [cos | sin] of t * exp(-ln(max_period) * i / half)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _gn(P, k, x):
    return F.group_norm(x.float(), 32, P[k + ".weight"], P[k + ".bias"], 1e-5)


def _res(P, k, x, emb, ci, co):
    """ResBlock._forward without up / down sampling and without scale-shift norm (unet.py:248-272): every ResBlock of unet_patched."""
    h = F.conv2d(F.silu(_gn(P, k + ".in_layers.0", x)), P[k + ".in_layers.2.weight"], P[k + ".in_layers.2.bias"], padding=1)
    e = F.linear(F.silu(emb), P[k + ".emb_layers.1.weight"], P[k + ".emb_layers.1.bias"])
    h = h + e[:, :, None, None]
    h = F.conv2d(F.silu(_gn(P, k + ".out_layers.0", h)), P[k + ".out_layers.3.weight"], P[k + ".out_layers.3.bias"], padding=1)
    if ci != co:
        x = F.conv2d(x, P[k + ".skip_connection.weight"], P[k + ".skip_connection.bias"])
    return x + h


def _attn(P, k, x, heads):
    """AttentionBlock._forward (unet.py:313-319) with QKVAttentionLegacy (:355-370): channels [head][q | k | v][ch], both q and k scaled
    by ch^-1/4, softmax in fp32."""
    b, c, hh, ww = x.shape
    xf = x.reshape(b, c, -1)
    qkv = F.conv1d(_gn(P, k + ".norm", xf), P[k + ".qkv.weight"], P[k + ".qkv.bias"])
    ch = c // heads
    q, kk, v = qkv.reshape(b * heads, ch * 3, -1).split(ch, dim=1)
    scale = 1 / math.sqrt(math.sqrt(ch))
    w = torch.einsum("bct,bcs->bts", q * scale, kk * scale)
    w = torch.softmax(w.float(), dim=-1)
    a = torch.einsum("bts,bcs->bct", w, v).reshape(b, -1, hh * ww)
    h = F.conv1d(a, P[k + ".proj_out.weight"], P[k + ".proj_out.bias"])
    return (xf + h).reshape(b, c, hh, ww)


def _run(P, cfg, blk, h, emb):
    for item in blk:
        kind, key = item[0], item[1]
        if kind == "conv3":
            h = F.conv2d(h, P[key + ".weight"], P[key + ".bias"], padding=1)
        elif kind == "res":
            h = _res(P, key, h, emb, item[2], item[3])
        elif kind == "attn":
            h = _attn(P, key, h, cfg.num_heads)
        elif kind == "down":                                                   # Downsample with conv_resample (unet.py:150-153)
            h = F.conv2d(h, P[key + ".op.weight"], P[key + ".op.bias"], stride=2, padding=1)
        elif kind == "up":                                                     # Upsample: nearest x2, then the convolution (:126-131)
            h = F.conv2d(F.interpolate(h, scale_factor=2, mode="nearest"), P[key + ".conv.weight"], P[key + ".conv.bias"], padding=1)
    return h


def unet_forward(P: Dict[str, Tensor], cfg: UNetCfg, sample: Tensor, timesteps, cond: Tensor, cond_mask: Optional[Tensor] = None) -> Tensor:
    """PatchedUNetCondCat.forward (unet.py:710-744) around UNetModel.forward (:660-690).  sample (B, C, H, W); timesteps: int or (B,) / (1,);
    cond (B, D, Hc, Wc) = the de-quantised latents; cond_mask (B, Hc, Wc) True = conditioning zeroed."""
    B, C, H, W = sample.shape
    p = cfg.patch_size
    nh, nw = H // p, W // p
    x = sample.reshape(B, C, nh, p, nw, p).permute(0, 1, 3, 5, 2, 4).reshape(B, C * p * p, nh, nw)     # b c (nh ph) (nw pw) -> b (c ph pw) nh nw
    if cond_mask is not None:
        cond = torch.where(cond_mask[:, None], torch.zeros((), dtype=cond.dtype), cond)
    x = torch.cat([x, F.interpolate(cond, (nh, nw), mode="nearest")], dim=1)
    t = torch.as_tensor(timesteps)
    t = t.reshape(1) if t.ndim == 0 else t
    emb = timestep_embedding(t, cfg.model_channels)
    emb = F.linear(F.silu(F.linear(emb, P["time_embed.0.weight"], P["time_embed.0.bias"])), P["time_embed.2.weight"], P["time_embed.2.bias"])
    inp, mid, out, _ = unet_plan(cfg)
    hs, h = [], x
    for blk in inp:
        h = _run(P, cfg, blk, h, emb)
        hs.append(h)
    h = _run(P, cfg, mid, h, emb)
    for blk in out:
        h = _run(P, cfg, blk, torch.cat([h, hs.pop()], dim=1), emb)
    y = F.conv2d(F.silu(_gn(P, "out.0", h)), P["out.2.weight"], P["out.2.bias"], padding=1)
    return y.reshape(B, cfg.out_channels, p, p, nh, nw).permute(0, 1, 4, 2, 5, 3).reshape(B, cfg.out_channels, H, W)


# ---- noise schedules and scheduler steps -------------------------------------------------------------------------------------------
@dataclass
class SchedCfg:
    kind: str = "ddpm"                      # "ddpm" | "ddim"
    num_train_timesteps: int = 1000
    beta_schedule: str = "squaredcos_cap_v2"
    prediction_type: str = "v_prediction"
    clip_sample: bool = False
    thresholding: bool = True
    zero_terminal_snr: bool = True
    dynamic_thresholding_ratio: float = 0.995
    sample_max_value: float = 1.0
    clip_sample_range: float = 1.0
    beta_start: float = 0.0001
    beta_end: float = 0.02


def alphas_cumprod(c: SchedCfg) -> Tensor:
    """scheduling_ddim.py:103-124 / scheduling_ddpm.py (same): betas -> (zero terminal SNR) -> cumulative product of 1 - beta."""
    T = c.num_train_timesteps
    if c.beta_schedule.startswith("shifted_cosine:"):                               # scheduling_utils.py:83-101
        shift = float(c.beta_schedule.split(":")[1])
        t = torch.linspace(0, 1, T).to(torch.float64)
        log_snr = (-2 * (torch.tan(torch.pi * t / 2).log() + np.log(shift))).clamp(-15, 15).float()
        ac = log_snr.sigmoid()
        ac[-1] = 0.0
        return ac
    if c.beta_schedule == "linear":
        betas = torch.linspace(c.beta_start, c.beta_end, T, dtype=torch.float32)
    elif c.beta_schedule == "scaled_linear":
        betas = torch.linspace(c.beta_start ** 0.5, c.beta_end ** 0.5, T, dtype=torch.float32) ** 2
    elif c.beta_schedule == "squaredcos_cap_v2":                                    # betas_for_alpha_bar, scheduling_utils.py:52-80
        ab = lambda s: math.cos((s + 0.008) / 1.008 * math.pi / 2) ** 2
        betas = torch.tensor([min(1 - ab((i + 1) / T) / ab(i / T), 0.999) for i in range(T)], dtype=torch.float32)
    else:
        raise NotImplementedError(c.beta_schedule)
    if c.zero_terminal_snr:                                                         # enforce_zero_terminal_snr, scheduling_utils.py:19-49
        abs_ = (1 - betas).cumprod(0).sqrt()
        a0, aT = abs_[0].clone(), abs_[-1].clone()
        abs_ = (abs_ - aT) * (a0 / (a0 - aT))
        ab2 = abs_ ** 2
        alphas = torch.cat([ab2[0:1], ab2[1:] / ab2[:-1]])
        betas = 1 - alphas
    return torch.cumprod(1.0 - betas, dim=0)


def inference_timesteps(c: SchedCfg, n: int, mode: str = "trailing") -> np.ndarray:
    """set_timesteps: scheduling_ddim.py:194-224 (three spacings); scheduling_ddpm.py:168-219 takes ``mode`` into **kwargs and always
    spaces 'leading'."""
    T = c.num_train_timesteps
    ratio = T // n
    if mode == "leading" or c.kind == "ddpm":
        return (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64)
    if mode == "trailing":
        return np.arange(T, 0, -ratio).round().astype(np.int64) - 1
    if mode == "linspace":
        return np.linspace(T, 1, n).round().astype(np.int64) - 1
    raise NotImplementedError(mode)


def threshold_sample(c: SchedCfg, x0: Tensor) -> Tensor:
    """_threshold_sample (scheduling_ddim.py:170-192): s = the 0.995 quantile of |x0| per sample, clamped to [1, sample_max_value]; x0 is
    clamped to [-s, s] and divided by s."""
    B = x0.shape[0]
    flat = x0.reshape(B, -1).float()
    s = torch.quantile(flat.abs(), c.dynamic_thresholding_ratio, dim=1).clamp(min=1, max=c.sample_max_value)[:, None]
    return (torch.clamp(flat, -s, s) / s).reshape(x0.shape)


def _pred_x0_eps(c: SchedCfg, a_t: Tensor, model_output: Tensor, sample: Tensor):
    b_t = 1 - a_t
    if c.prediction_type == "epsilon":
        return (sample - b_t ** 0.5 * model_output) / a_t ** 0.5, model_output
    if c.prediction_type == "sample":
        return model_output, (sample - a_t ** 0.5 * model_output) / b_t ** 0.5
    if c.prediction_type == "v_prediction":
        return (a_t ** 0.5) * sample - (b_t ** 0.5) * model_output, (a_t ** 0.5) * model_output + (b_t ** 0.5) * sample
    raise ValueError(c.prediction_type)


def ddim_step(c: SchedCfg, ac: Tensor, n_inference: int, model_output: Tensor, t: int, sample: Tensor, eta: float = 0.0, noise: Optional[Tensor] = None):
    """DDIMScheduler.step (scheduling_ddim.py:226-330).  Returns (prev_sample, pred_original_sample)."""
    prev_t = t - c.num_train_timesteps // n_inference
    a_t = ac[t]
    a_prev = ac[prev_t] if prev_t >= 0 else torch.tensor(1.0)
    x0, eps = _pred_x0_eps(c, a_t, model_output, sample)
    if c.thresholding:
        x0 = threshold_sample(c, x0)
    elif c.clip_sample:
        x0 = x0.clamp(-c.clip_sample_range, c.clip_sample_range)
    variance = ((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)
    std = eta * variance ** 0.5
    prev = a_prev ** 0.5 * x0 + (1 - a_prev - std ** 2) ** 0.5 * eps
    if eta > 0:
        prev = prev + std * noise
    return prev, x0


def ddpm_step(c: SchedCfg, ac: Tensor, n_inference: int, model_output: Tensor, t: int, sample: Tensor, noise: Optional[Tensor] = None):
    """DDPMScheduler.step (scheduling_ddpm.py:275-345) with variance_type 'fixed_small' (the constructor default DiVAE uses).
    previous_timestep (:433-446, no custom timesteps): t - T // n_inference."""
    prev_t = t - c.num_train_timesteps // n_inference
    a_t = ac[t]
    a_prev = ac[prev_t] if prev_t >= 0 else torch.tensor(1.0)
    b_t, b_prev = 1 - a_t, 1 - a_prev
    cur_a = a_t / a_prev
    cur_b = 1 - cur_a
    x0, _ = _pred_x0_eps(c, a_t, model_output, sample)
    if c.thresholding:
        x0 = threshold_sample(c, x0)
    elif c.clip_sample:
        x0 = x0.clamp(-c.clip_sample_range, c.clip_sample_range)
    prev = (a_prev ** 0.5 * cur_b) / b_t * x0 + cur_a ** 0.5 * b_prev / b_t * sample
    if t > 0:
        var = torch.clamp(b_prev / b_t * cur_b, min=1e-20)
        prev = prev + (var ** 0.5) * noise
    return prev, x0


def sample_loop(P, ucfg: UNetCfg, scfg: SchedCfg, cond: Tensor, noise0: Tensor, n_steps: int, mode: str = "trailing",
                step_noise: Optional[List[Tensor]] = None):
    """PipelineCond.__call__ (diffusion_pipeline.py:52-133) with guidance_scale 0: image = noise0; for t in schedule: out = model(image, t,
    cond); image = scheduler.step(out.float(), t, image).  DDIM runs with eta = 0 (its default).  Returns (image, [model outputs])."""
    ac = alphas_cumprod(scfg)
    ts = inference_timesteps(scfg, n_steps, mode)
    image, outs = noise0, []
    for i, t in enumerate(ts):
        out = unet_forward(P, ucfg, image, int(t), cond)
        outs.append(out)
        if scfg.kind == "ddim":
            image, _ = ddim_step(scfg, ac, n_steps, out.float(), int(t), image)
        else:
            image, _ = ddpm_step(scfg, ac, n_steps, out.float(), int(t), image, None if step_noise is None else step_noise[i])
    return image, outs
