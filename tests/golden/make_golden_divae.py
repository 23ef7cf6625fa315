#!/usr/bin/env python3
"""Golden fixture for the diffusion detokenizer from the UNMODIFIED upstream classes (container only):
``PatchedUNetCondCat`` (fourm/vq/models/unet/unet.py), ``DDIMScheduler`` / ``DDPMScheduler`` (fourm/vq/scheduling), ``PipelineCond``
(diffusion_pipeline.py) and ``DiVAE.decode_quant`` (fourm/vq/vqvae.py:640-679), run through the inert diffusers helpers of ref_stubs.py.
The oracle (oracle/divae_oracle.py) must reproduce every recorded tensor before the file is written:
  * one UNet evaluation (per-sample timesteps, with and without a conditioning mask),
  * the noise schedules (alphas_cumprod of the cosine + zero-terminal-SNR and of the linear schedule) and the three timestep spacings,
  * single scheduler steps (DDIM eta = 0 and eta > 0, DDPM with noise; v / epsilon / sample prediction; dynamic thresholding, clipping),
  * a 4-step DDIM and a 3-step DDPM sampling loop of the pipeline from fixed initial noise,
  * the state_dict keys / shapes of upstream's ``unet_patched`` (the 4M RGB detokenizer's decoder).
    python tests/golden/make_golden_divae.py [--check]"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_stubs  # noqa: E402

ref_stubs.install()
import fourm.vq.models.unet.unet as RU  # noqa: E402
from fourm.vq.scheduling import DDIMScheduler, DDPMScheduler, PipelineCond  # noqa: E402

from oracle import divae_oracle as DO  # noqa: E402

SMALL = dict(image_size=32, in_channels=3, out_channels=3, cond_channels=8, patch_size=4, model_channels=64, num_res_blocks=1,
             attention_resolutions=(2,), channel_mult=(1, 2))


def small_cfg():
    return DO.UNetCfg(**SMALL)


def upstream_unet(cfg, sd):
    net = RU.PatchedUNetCondCat(in_channels=cfg.in_channels, out_channels=cfg.out_channels, cond_channels=cfg.cond_channels, patch_size=cfg.patch_size,
                                image_size=cfg.image_size, model_channels=cfg.model_channels, num_res_blocks=cfg.num_res_blocks,
                                attention_resolutions=list(cfg.attention_resolutions), channel_mult=tuple(cfg.channel_mult))
    missing = net.load_state_dict(sd, strict=True)
    return net.eval()


def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    cfg = small_cfg()
    sd = DO.seeded_unet_state_dict(cfg, seed=3)
    net = upstream_unet(cfg, sd)
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == DO.unet_param_shapes(cfg), "oracle plan != upstream module tree"
    fx = {}
    g = torch.Generator().manual_seed(21)
    B = 3
    x = torch.randn(B, 3, 32, 32, generator=g)
    cond = torch.randn(B, 8, 2, 2, generator=g)
    ts = torch.tensor([999, 417, 3])
    mask = torch.tensor([[[False, True], [False, False]], [[True, True], [True, True]], [[False, False], [False, False]]])
    with torch.no_grad():
        y = net(x, ts, cond)
        ym = net(x, ts, cond, cond_mask=mask)
        y1 = net(x, 250, cond)                                              # python int timestep (the pipeline's form)
    for name, want, got in (("unet", y, DO.unet_forward(sd, cfg, x, ts, cond)), ("unet_masked", ym, DO.unet_forward(sd, cfg, x, ts, cond, mask)),
                            ("unet_t250", y1, DO.unet_forward(sd, cfg, x, 250, cond))):
        assert rel(got, want) < 2e-6, (name, rel(got, want))
        fx[name] = want.numpy()
    fx.update(x=x.numpy(), cond=cond.numpy(), ts=ts.numpy(), mask=mask.numpy())
    # schedules and spacings
    for tag, kw in (("cos", dict(beta_schedule="squaredcos_cap_v2", zero_terminal_snr=True)), ("lin", dict(beta_schedule="linear", zero_terminal_snr=False)),
                    ("shift", dict(beta_schedule="shifted_cosine:0.5", zero_terminal_snr=True))):
        s = DDIMScheduler(num_train_timesteps=1000, prediction_type="v_prediction", thresholding=True, clip_sample=False, **kw)
        oc = DO.alphas_cumprod(DO.SchedCfg(kind="ddim", **kw))
        assert float((s.alphas_cumprod - oc).abs().max()) < 1e-7, tag
        fx[f"ac_{tag}"] = s.alphas_cumprod.numpy()
    s = DDIMScheduler(num_train_timesteps=1000)
    for mode in ("trailing", "leading", "linspace"):
        s.set_timesteps(7, mode=mode)
        assert np.array_equal(s.timesteps.numpy(), DO.inference_timesteps(DO.SchedCfg(kind="ddim"), 7, mode)), mode
        fx[f"ts_{mode}"] = s.timesteps.numpy()
    sp = DDPMScheduler(num_train_timesteps=1000)
    sp.set_timesteps(7, mode="trailing")
    assert np.array_equal(sp.timesteps.numpy(), DO.inference_timesteps(DO.SchedCfg(kind="ddpm"), 7, "trailing"))
    fx["ts_ddpm"] = sp.timesteps.numpy()
    # single steps
    mo = torch.randn(B, 3, 32, 32, generator=g) * 1.3
    smp = torch.randn(B, 3, 32, 32, generator=g)
    nz = torch.randn(B, 3, 32, 32, generator=g)
    fx.update(step_model_output=mo.numpy(), step_sample=smp.numpy(), step_noise=nz.numpy())
    cases = []
    for pred in ("v_prediction", "epsilon", "sample"):
        for thr, clip in ((True, False), (False, True), (False, False)):
            cases.append((pred, thr, clip))
    for i, (pred, thr, clip) in enumerate(cases):
        kw = dict(num_train_timesteps=1000, beta_schedule="squaredcos_cap_v2", prediction_type=pred, thresholding=thr, clip_sample=clip, zero_terminal_snr=pred == "v_prediction")
        oc = DO.SchedCfg(kind="ddim", beta_schedule="squaredcos_cap_v2", prediction_type=pred, thresholding=thr, clip_sample=clip, zero_terminal_snr=pred == "v_prediction")
        ac = DO.alphas_cumprod(oc)
        sd_ = DDIMScheduler(**kw); sd_.set_timesteps(10, mode="trailing")
        for t in (899, 99):
            for eta in (0.0, 0.6):
                want = sd_.step(mo, t, smp, eta=eta, variance_noise=nz if eta > 0 else None)
                got, gx0 = DO.ddim_step(oc, ac, 10, mo, t, smp, eta=eta, noise=nz)
                assert rel(got, want.prev_sample) < 3e-6, ("ddim", pred, thr, clip, t, eta)
                fx[f"ddim_{i}_{t}_{int(eta * 10)}"] = (want.prev_sample).numpy()
        oc2 = DO.SchedCfg(kind="ddpm", beta_schedule="squaredcos_cap_v2", prediction_type=pred, thresholding=thr, clip_sample=clip, zero_terminal_snr=pred == "v_prediction")
        sq = DDPMScheduler(**kw); sq.set_timesteps(10)
        for t in (900, 100, 0):
            gen = torch.Generator().manual_seed(77)
            want = sq.step(mo, t, smp, generator=gen)
            wp = want.prev_sample
            noise = torch.randn(mo.shape, generator=torch.Generator().manual_seed(77))
            got, _ = DO.ddpm_step(oc2, ac, 10, mo, t, smp, noise=noise)
            assert rel(got, wp) < 3e-6, ("ddpm", pred, thr, clip, t, rel(got, wp))
            fx[f"ddpm_{i}_{t}"] = wp.numpy()
    fx["step_cases"] = np.array([f"{p}|{int(t)}|{int(c)}" for p, t, c in cases])
    # sampling loops of the pipeline (DiVAE defaults: v-prediction, cosine schedule, zero terminal SNR, dynamic thresholding)
    for kind, Sched, n in (("ddim", DDIMScheduler, 4), ("ddpm", DDPMScheduler, 3)):
        sch = Sched(num_train_timesteps=1000, thresholding=True, clip_sample=False, beta_schedule="squaredcos_cap_v2", prediction_type="v_prediction", zero_terminal_snr=True)
        pipe = PipelineCond(model=net, scheduler=sch)
        gen = torch.Generator().manual_seed(5)
        img = pipe(cond, generator=gen, timesteps=n, verbose=False, scheduler_timesteps_mode="trailing")
        gen = torch.Generator().manual_seed(5)
        noise0 = torch.randn(B, 3, 32, 32, generator=gen)
        step_noise = [torch.randn(B, 3, 32, 32, generator=gen) for _ in range(n)] if kind == "ddpm" else None
        oimg, outs = DO.sample_loop(sd, cfg, DO.SchedCfg(kind=kind), cond, noise0, n, "trailing", step_noise)
        assert rel(oimg, img) < 2e-5, (kind, rel(oimg, img))
        fx[f"loop_{kind}"] = img.numpy()
        fx[f"loop_{kind}_out0"] = outs[0].numpy()
    # the real decoder's parameter layout
    big = RU.unet_patched(in_channels=3, out_channels=3, cond_channels=32, image_size=224)
    shapes = {k: tuple(v.shape) for k, v in big.state_dict().items()}
    assert shapes == DO.unet_param_shapes(DO.unet_patched_cfg(cond_channels=32, image_size=224)), "unet_patched layout"
    fx["unet_patched_keys"] = np.array(sorted(shapes))
    fx["unet_patched_numel"] = np.int64(sum(int(np.prod(s)) for s in shapes.values()))
    path = os.path.join(HERE, "divae_small.npz")
    if a.check:
        old = np.load(path)
        for k, v in fx.items():
            v = np.asarray(v)
            assert (np.array_equal(v, old[k]) if v.dtype.kind in "iUSb" else np.allclose(v, old[k], rtol=0, atol=1e-6)), k
        print("divae_small: fixture reproduced")
        return
    np.savez_compressed(path, **fx)
    print("wrote", path, os.path.getsize(path) // 1024, "KB;", len(fx), "entries; unet_patched", int(fx["unet_patched_numel"]) / 1e6, "M parameters")


if __name__ == "__main__":
    main()
