"""Diffusion schedulers and the conditional sampling loop of the DiVAE detokenizers on gfx950.

API of upstream ``fourm/vq/scheduling`` for what inference uses: ``DDPMScheduler`` / ``DDIMScheduler`` (constructor arguments, ``.config``,
``set_timesteps``, ``step`` -> ``.prev_sample`` / ``.pred_original_sample``, ``add_noise`` / ``get_velocity`` / ``get_noise``) and
``PipelineCond.__call__`` (scheduling_ddim.py:75-330, scheduling_ddpm.py:95-345, scheduling_utils.py:19-101, diffusion_pipeline.py:38-133).
The per-step scalars (alpha products, coefficients) are host arithmetic on fp32 torch scalars exactly as upstream; the element-wise step
over the image - x0 from the model output, dynamic thresholding (per-sample 0.995 quantile of |x0| by radix select), the update - runs
on fm_diffusion_x0 / fm_quantile_abs / fm_diffusion_step.  No ``diffusers`` dependency.  ``PNDMScheduler`` falls through to upstream."""
import math
from types import SimpleNamespace
from typing import List, Optional, Tuple, Union

import numpy as np
import torch


class _Config(dict):
    __getattr__ = dict.__getitem__


class SchedulerOutput(SimpleNamespace):
    """prev_sample, pred_original_sample (upstream's DDIMSchedulerOutput / DDPMSchedulerOutput)."""


def enforce_zero_terminal_snr(betas: torch.Tensor) -> torch.Tensor:
    """scheduling_utils.py:19-49 (https://arxiv.org/abs/2305.08891): shift sqrt(alpha_bar) so the last step is 0, rescale the first back."""
    abs_ = (1 - betas).cumprod(0).sqrt()
    a0, aT = abs_[0].clone(), abs_[-1].clone()
    abs_ = (abs_ - aT) * (a0 / (a0 - aT))
    ab = abs_ ** 2
    return 1 - torch.cat([ab[0:1], ab[1:] / ab[:-1]])


def betas_for_alpha_bar(n: int, max_beta: float = 0.999) -> torch.Tensor:
    """scheduling_utils.py:52-80: the cosine schedule (squaredcos_cap_v2)."""
    ab = lambda s: math.cos((s + 0.008) / 1.008 * math.pi / 2) ** 2
    return torch.tensor([min(1 - ab((i + 1) / n) / ab(i / n), max_beta) for i in range(n)], dtype=torch.float32)


def scaled_cosine_alphas(n: int, noise_shift: float = 1.0) -> torch.Tensor:
    """scheduling_utils.py:83-101: cosine schedule shifted in log-SNR space."""
    t = torch.linspace(0, 1, n).to(torch.float64)
    log_snr = (-2 * (torch.tan(torch.pi * t / 2).log() + np.log(noise_shift))).clamp(-15, 15).float()
    ac = log_snr.sigmoid()
    ac[-1] = 0.0
    return ac


class _SchedulerBase:
    order = 1

    # beta tables by name (what scheduling_ddpm.py:119-133 / scheduling_ddim.py:142-156 build): name -> f(T, beta_start, beta_end)
    _BETA_TABLES = {
        "linear": lambda T, b0, b1: torch.linspace(b0, b1, T, dtype=torch.float32),
        "scaled_linear": lambda T, b0, b1: torch.linspace(b0 ** 0.5, b1 ** 0.5, T, dtype=torch.float32) ** 2,
        "squaredcos_cap_v2": lambda T, b0, b1: betas_for_alpha_bar(T),
    }

    def _init_common(self, num_train_timesteps, beta_start, beta_end, beta_schedule, trained_betas, zero_terminal_snr):
        """alphas_cumprod of the training schedule: either the shifted-cosine closed form (no beta table exists then) or the cumulative
        product of 1 - beta over an explicit / named beta table, optionally rescaled to zero terminal SNR."""
        T = num_train_timesteps
        shifted = beta_schedule.partition("shifted_cosine:")
        if shifted[1]:
            self.alphas_cumprod = scaled_cosine_alphas(T, float(beta_schedule.split(":")[1]))
        else:
            if trained_betas is not None:
                betas = torch.tensor(trained_betas, dtype=torch.float32)
            else:
                table = self._BETA_TABLES.get(beta_schedule)
                if table is None:
                    raise NotImplementedError(f"{type(self).__name__}: unknown beta_schedule {beta_schedule!r} (known: {sorted(self._BETA_TABLES)} or 'shifted_cosine:<shift>')")
                betas = table(T, beta_start, beta_end)
            self.betas = enforce_zero_terminal_snr(betas) if zero_terminal_snr else betas
            self.alphas = 1.0 - self.betas
            self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.arange(T - 1, -1, -1, dtype=torch.int64)

    @staticmethod
    def _check_step_count(n, T):
        if n > T:
            raise ValueError(f"{n} inference steps asked of a schedule trained on {T} timesteps")

    def scale_model_input(self, sample, timestep=None):
        return sample

    def __len__(self):
        return self.config.num_train_timesteps

    # ---- element-wise step on the device -------------------------------------------------------------------------------------------
    def _x0_coeffs(self, a_t):
        """x0 = c0 * sample + c1 * model_output; eps = e0 * sample + e1 * model_output (fp32 scalars, as upstream computes them)."""
        b_t = 1 - a_t
        pt = self.config.prediction_type
        if pt == "epsilon":
            return 1 / a_t ** 0.5, -(b_t ** 0.5) / a_t ** 0.5, torch.tensor(0.0), torch.tensor(1.0)
        if pt == "sample":
            return torch.tensor(0.0), torch.tensor(1.0), 1 / b_t ** 0.5, -(a_t ** 0.5) / b_t ** 0.5
        if pt == "v_prediction":
            return a_t ** 0.5, -(b_t ** 0.5), b_t ** 0.5, a_t ** 0.5
        raise ValueError(f"prediction_type given as {pt} must be one of `epsilon`, `sample`, or `v_prediction`")

    def _device_step(self, model_output, sample, c0, c1, k0, k1, k2, k3, noise):
        from fourm.hip import _lib as L, ops
        if not sample.is_cuda:
            raise RuntimeError("scheduler.step runs on the HIP kernels only (tensors on an MI355X)")
        mo = model_output.detach().float().contiguous()
        smp = sample.detach().float().contiguous()
        B = smp.shape[0]
        per = smp.numel() // B
        x0 = torch.empty_like(smp)
        L.check(L.diffusion_x0(ops._p(smp), ops._p(mo), float(c0), float(c1), ops._p(x0), smp.numel(), ops._stream()))
        quant = None
        if self.config.thresholding:
            quant = torch.empty(B, dtype=torch.float32, device=smp.device)
            L.check(L.quantile_abs(ops._p(x0), B, per, float(self.config.dynamic_thresholding_ratio), ops._p(quant), ops._stream()))
        clip = float(self.config.clip_sample_range) if (self.config.clip_sample and not self.config.thresholding) else 0.0
        out, x0c = torch.empty_like(smp), torch.empty_like(smp)
        nz = noise.detach().float().contiguous() if noise is not None else None
        L.check(L.diffusion_step(ops._p(x0), ops._p(quant), float(self.config.sample_max_value), clip, ops._p(smp), ops._p(mo), ops._p(nz),
                                 float(k0), float(k1), float(k2), float(k3), ops._p(out), ops._p(x0c), B, per, ops._stream()))
        return out, x0c

    # ---- training-side helpers (element-wise torch; the DiVAE training loop itself is upstream's) --------------------------------------
    def get_alpha_sigma_sqrts(self, timesteps, device, dtype, shape):
        ac = self.alphas_cumprod.to(device=device, dtype=dtype)
        t = timesteps.to(device)
        sa, ss = (ac[t] ** 0.5).flatten(), ((1 - ac[t]) ** 0.5).flatten()
        while sa.dim() < len(shape):
            sa, ss = sa.unsqueeze(-1), ss.unsqueeze(-1)
        return sa, ss

    def add_noise(self, original_samples, noise, timesteps):
        sa, ss = self.get_alpha_sigma_sqrts(timesteps, original_samples.device, original_samples.dtype, original_samples.shape)
        return sa * original_samples + ss * noise

    def get_velocity(self, sample, noise, timesteps):
        sa, ss = self.get_alpha_sigma_sqrts(timesteps, sample.device, sample.dtype, sample.shape)
        return sa * noise - ss * sample

    def get_noise(self, sample, velocity, timesteps):
        sa, ss = self.get_alpha_sigma_sqrts(timesteps, sample.device, sample.dtype, sample.shape)
        return sa * velocity + ss * sample


def _randn(shape, generator, device, dtype):
    """diffusers.utils.randn_tensor: drawn on the generator's device (CPU generators give the same numbers on every machine), then moved."""
    gdev = generator.device if generator is not None else device
    return torch.randn(tuple(shape), generator=generator, device=gdev, dtype=dtype).to(device)


class DDIMScheduler(_SchedulerBase):
    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.0001, beta_end: float = 0.02, beta_schedule: str = "linear",
                 trained_betas=None, clip_sample: bool = True, set_alpha_to_one: bool = True, steps_offset: int = 0, prediction_type: str = "v_prediction",
                 thresholding: bool = False, dynamic_thresholding_ratio: float = 0.995, clip_sample_range: float = 1.0, sample_max_value: float = 1.0,
                 zero_terminal_snr: bool = True):
        self.config = _Config(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end, beta_schedule=beta_schedule,
                              trained_betas=trained_betas, clip_sample=clip_sample, set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset,
                              prediction_type=prediction_type, thresholding=thresholding, dynamic_thresholding_ratio=dynamic_thresholding_ratio,
                              clip_sample_range=clip_sample_range, sample_max_value=sample_max_value, zero_terminal_snr=zero_terminal_snr)
        self._init_common(num_train_timesteps, beta_start, beta_end, beta_schedule, trained_betas, zero_terminal_snr)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]

    def set_timesteps(self, num_inference_steps: int, device=None, mode: str = "trailing"):
        """scheduling_ddim.py:194-224: 'leading' | 'trailing' | 'linspace' spacing (https://arxiv.org/abs/2305.08891)."""
        T = self.config.num_train_timesteps
        self._check_step_count(num_inference_steps, T)
        self.num_inference_steps = n = num_inference_steps
        stride = T // n
        spacings = {
            "leading": lambda: (np.arange(n) * stride).round()[::-1],
            "trailing": lambda: np.arange(T, 0, -stride).round() - 1,
            "linspace": lambda: np.linspace(T, 1, n).round() - 1,
        }
        if mode not in spacings:
            raise NotImplementedError(f"timestep spacing {mode!r} (known: {sorted(spacings)})")
        ts = np.ascontiguousarray(spacings[mode]()).astype(np.int64)
        self.timesteps = torch.from_numpy(ts).to(device) + self.config.steps_offset

    def step(self, model_output, timestep, sample, eta: float = 0.0, use_clipped_model_output: bool = False, generator=None, variance_noise=None,
             return_dict: bool = True):
        """scheduling_ddim.py:226-330."""
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        if use_clipped_model_output:
            raise NotImplementedError("use_clipped_model_output")
        t = int(timestep)
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        c0, c1, e0, e1 = self._x0_coeffs(a_t)
        variance = ((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)
        std = eta * variance ** 0.5
        dirc = (1 - a_prev - std ** 2) ** 0.5
        noise = None
        if eta > 0:
            if variance_noise is not None and generator is not None:
                raise ValueError("Cannot pass both generator and variance_noise.")
            noise = variance_noise if variance_noise is not None else _randn(model_output.shape, generator, model_output.device, model_output.dtype)
        # prev = sqrt(a_prev) x0' + dirc * eps + std * noise, eps = e0 sample + e1 model_output (from the UNclamped quantities)
        prev, x0 = self._device_step(model_output, sample, c0, c1, a_prev ** 0.5, dirc * e0, dirc * e1, std, noise)
        if not return_dict:
            return (prev,)
        return SchedulerOutput(prev_sample=prev, pred_original_sample=x0)


class DDPMScheduler(_SchedulerBase):
    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.0001, beta_end: float = 0.02, beta_schedule: str = "linear",
                 trained_betas=None, variance_type: str = "fixed_small", clip_sample: bool = True, prediction_type: str = "v_prediction",
                 thresholding: bool = False, dynamic_thresholding_ratio: float = 0.995, clip_sample_range: float = 1.0, sample_max_value: float = 1.0,
                 zero_terminal_snr: bool = True):
        if variance_type != "fixed_small":
            raise NotImplementedError(f"variance_type {variance_type!r}: the DiVAE decoders use 'fixed_small'")
        self.config = _Config(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end, beta_schedule=beta_schedule,
                              trained_betas=trained_betas, variance_type=variance_type, clip_sample=clip_sample, prediction_type=prediction_type,
                              thresholding=thresholding, dynamic_thresholding_ratio=dynamic_thresholding_ratio, clip_sample_range=clip_sample_range,
                              sample_max_value=sample_max_value, zero_terminal_snr=zero_terminal_snr)
        self._init_common(num_train_timesteps, beta_start, beta_end, beta_schedule, trained_betas, zero_terminal_snr)
        self.one = torch.tensor(1.0)
        self.custom_timesteps = False
        self.variance_type = variance_type

    def set_timesteps(self, num_inference_steps: Optional[int] = None, device=None, timesteps: Optional[List[int]] = None, **kwargs):
        """scheduling_ddpm.py:168-219: 'leading' spacing whatever ``mode`` says (it lands in **kwargs upstream too), or custom timesteps."""
        T = self.config.num_train_timesteps
        if timesteps is not None:
            if num_inference_steps is not None:
                raise ValueError("set_timesteps takes a step count or an explicit timestep list, not both")
            ts = np.asarray(timesteps, dtype=np.int64)
            if ts.size > 1 and not np.all(np.diff(ts) < 0):
                raise ValueError("an explicit timestep list must be strictly descending")
            if ts[0] >= T:
                raise ValueError(f"an explicit timestep list must start below the {T} training timesteps")
        else:
            self._check_step_count(num_inference_steps, T)
            self.num_inference_steps = num_inference_steps
            ts = np.ascontiguousarray((np.arange(num_inference_steps) * (T // num_inference_steps)).round()[::-1]).astype(np.int64)
        self.custom_timesteps = timesteps is not None
        self.timesteps = torch.from_numpy(ts).to(device)

    def previous_timestep(self, timestep):
        if self.custom_timesteps:
            index = (self.timesteps == timestep).nonzero(as_tuple=True)[0][0]
            return torch.tensor(-1) if index == self.timesteps.shape[0] - 1 else self.timesteps[index + 1]
        n = self.num_inference_steps if self.num_inference_steps else self.config.num_train_timesteps
        return timestep - self.config.num_train_timesteps // n

    def step(self, model_output, timestep, sample, generator=None, return_dict: bool = True):
        """scheduling_ddpm.py:275-345 (fixed_small variance)."""
        t = int(timestep)
        prev_t = int(self.previous_timestep(t))
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        b_t, b_prev = 1 - a_t, 1 - a_prev
        cur_a = a_t / a_prev
        cur_b = 1 - cur_a
        c0, c1, _, _ = self._x0_coeffs(a_t)
        k0 = (a_prev ** 0.5 * cur_b) / b_t
        k1 = cur_a ** 0.5 * b_prev / b_t
        noise, k3 = None, 0.0
        if t > 0:
            noise = _randn(model_output.shape, generator, model_output.device, model_output.dtype)
            k3 = torch.clamp(b_prev / b_t * cur_b, min=1e-20) ** 0.5
        prev, x0 = self._device_step(model_output, sample, c0, c1, k0, k1, 0.0, k3, noise)
        if not return_dict:
            return (prev,)
        return SchedulerOutput(prev_sample=prev, pred_original_sample=x0)


def rescale_noise_cfg(noise_cfg, noise_pred_conditional, guidance_rescale=0.0):
    """diffusion_pipeline.py:24-35 (https://arxiv.org/abs/2305.08891 §3.4)."""
    std_text = noise_pred_conditional.std(dim=list(range(1, noise_pred_conditional.ndim)), keepdim=True)
    std_cfg = noise_cfg.std(dim=list(range(1, noise_cfg.ndim)), keepdim=True)
    return guidance_rescale * (noise_cfg * (std_text / std_cfg)) + (1 - guidance_rescale) * noise_cfg


class PipelineCond:
    """Conditional sampling loop (diffusion_pipeline.py:38-133): Gaussian start image drawn on the CPU generator like upstream (the same
    seed gives the same image on every machine), one model evaluation per step (two with classifier-free guidance), scheduler step in fp32."""

    def __init__(self, model, scheduler):
        self.model, self.scheduler = model, scheduler

    def register_modules(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    @torch.no_grad()
    def __call__(self, cond: torch.Tensor, generator: Optional[torch.Generator] = None, timesteps: Optional[int] = None, guidance_scale: float = 0.0,
                 guidance_rescale: float = 0.0, image_size=None, verbose: bool = True, scheduler_timesteps_mode: str = "trailing", orig_res=None, **kwargs):
        timesteps = self.scheduler.config.num_train_timesteps if timesteps is None else timesteps
        B = cond.shape[0]
        image_size = self.model.sample_size if image_size is None else image_size
        hw = tuple(image_size) if isinstance(image_size, (tuple, list)) else (image_size, image_size)
        image = torch.randn((B, self.model.in_channels, hw[0], hw[1]), generator=generator).to(self.model.device)
        do_cfg = callable(guidance_scale) or guidance_scale > 1.0
        self.scheduler.set_timesteps(timesteps, mode=scheduler_timesteps_mode)
        it = self.scheduler.timesteps
        if verbose:
            try:
                from tqdm import tqdm
                it = tqdm(it)
            except Exception:      # noqa: BLE001
                pass
        for t in it:
            out = self.model(image, t, cond, orig_res=orig_res, **kwargs)
            if do_cfg:
                unc = self.model(image, t, cond, unconditional=True, **kwargs)
                s = guidance_scale(t / self.scheduler.config.num_train_timesteps) if callable(guidance_scale) else guidance_scale
                cfg = unc + s * (out - unc)
                out = rescale_noise_cfg(cfg, out, guidance_rescale) if guidance_rescale > 0.0 else cfg
            image = self.scheduler.step(out.float(), t, image, generator=generator).prev_sample
        return image


# names only upstream's package defines (PNDMScheduler, ...) resolve lazily (see fourm/_upstream.py)
from fourm import _upstream as _up
__getattr__ = _up.fallthrough(__name__, is_package=True)
_up.extend_path(__name__, __path__)
