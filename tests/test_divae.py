"""Diffusion detokenizer (``DiVAE``, fourm/vq/vqvae.py:498-764; conditional UNet fourm/vq/models/unet/unet.py; schedulers and
sampling loop fourm/vq/scheduling).
CPU: the oracle (oracle/divae_oracle.py) against the fixture dumped from the UNMODIFIED upstream classes (tests/golden/make_golden_divae.py),
the parameter tree of fourm.vq.models.unet against upstream's, the schedulers' host arithmetic.
GPU: the HIP UNet / scheduler steps / sampling loop against fixture and oracle."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-4m_amd"))
from oracle import divae_oracle as DO  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "divae_small.npz")
SMALL = dict(image_size=32, in_channels=3, out_channels=3, cond_channels=8, patch_size=4, model_channels=64, num_res_blocks=1,
             attention_resolutions=(2,), channel_mult=(1, 2))        # = tests/golden/make_golden_divae.py SMALL


def fixture():
    return np.load(GOLD)


def rel(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def step_cases(fx):
    return [(p, bool(int(t)), bool(int(c))) for p, t, c in (s.split("|") for s in fx["step_cases"].tolist())]


# ---- CPU -------------------------------------------------------------------------------------------------------------------------------
def test_oracle_unet_reproduces_upstream_fixture():
    fx = fixture()
    cfg = DO.UNetCfg(**SMALL)
    sd = DO.seeded_unet_state_dict(cfg, seed=3)
    x, cond, ts, mask = (torch.from_numpy(fx[k]) for k in ("x", "cond", "ts", "mask"))
    assert rel(DO.unet_forward(sd, cfg, x, ts, cond), fx["unet"]) < 2e-6
    assert rel(DO.unet_forward(sd, cfg, x, ts, cond, mask), fx["unet_masked"]) < 2e-6
    assert rel(DO.unet_forward(sd, cfg, x, 250, cond), fx["unet_t250"]) < 2e-6
    assert rel(fx["unet_masked"], fx["unet"]) > 1e-3                 # the mask changes the result (sample 1 loses all of its conditioning)


def test_oracle_schedules_steps_and_loops_reproduce_upstream_fixture():
    fx = fixture()
    for tag, kw in (("cos", dict(beta_schedule="squaredcos_cap_v2", zero_terminal_snr=True)), ("lin", dict(beta_schedule="linear", zero_terminal_snr=False)),
                    ("shift", dict(beta_schedule="shifted_cosine:0.5"))):
        assert float((DO.alphas_cumprod(DO.SchedCfg(**kw)) - torch.from_numpy(fx[f"ac_{tag}"])).abs().max()) < 1e-7, tag
    for mode in ("trailing", "leading", "linspace"):
        assert np.array_equal(DO.inference_timesteps(DO.SchedCfg(kind="ddim"), 7, mode), fx[f"ts_{mode}"])
    assert np.array_equal(DO.inference_timesteps(DO.SchedCfg(kind="ddpm"), 7, "trailing"), fx["ts_ddpm"])
    mo, smp, nz = (torch.from_numpy(fx[k]) for k in ("step_model_output", "step_sample", "step_noise"))
    for i, (pred, thr, clip) in enumerate(step_cases(fx)):
        oc = DO.SchedCfg(kind="ddim", prediction_type=pred, thresholding=thr, clip_sample=clip, zero_terminal_snr=pred == "v_prediction")
        ac = DO.alphas_cumprod(oc)
        for t in (899, 99):
            for eta in (0.0, 0.6):
                got, _ = DO.ddim_step(oc, ac, 10, mo, t, smp, eta=eta, noise=nz)
                assert rel(got, fx[f"ddim_{i}_{t}_{int(eta * 10)}"]) < 3e-6, (pred, thr, clip, t, eta)
        noise = torch.randn(mo.shape, generator=torch.Generator().manual_seed(77))
        for t in (900, 100, 0):
            got, _ = DO.ddpm_step(DO.SchedCfg(kind="ddpm", prediction_type=pred, thresholding=thr, clip_sample=clip, zero_terminal_snr=pred == "v_prediction"), ac, 10,
                                  mo, t, smp, noise=noise)
            assert rel(got, fx[f"ddpm_{i}_{t}"]) < 3e-6, (pred, thr, clip, t)
    cfg = DO.UNetCfg(**SMALL)
    sd = DO.seeded_unet_state_dict(cfg, seed=3)
    cond = torch.from_numpy(fx["cond"])
    for kind, n in (("ddim", 4), ("ddpm", 3)):
        gen = torch.Generator().manual_seed(5)
        noise0 = torch.randn(3, 3, 32, 32, generator=gen)
        step_noise = [torch.randn(3, 3, 32, 32, generator=gen) for _ in range(n)] if kind == "ddpm" else None
        img, outs = DO.sample_loop(sd, cfg, DO.SchedCfg(kind=kind), cond, noise0, n, "trailing", step_noise)
        assert rel(img, fx[f"loop_{kind}"]) < 2e-5 and rel(outs[0], fx[f"loop_{kind}_out0"]) < 2e-6, kind


def test_unet_parameter_tree_and_schedulers_match_upstream():
    from fourm.vq import DiVAE
    from fourm.vq.models.unet import PatchedUNetCondCat, unet_patched
    from fourm.vq.scheduling import DDIMScheduler, DDPMScheduler
    fx = fixture()
    net = unet_patched(in_channels=3, out_channels=3, cond_channels=32, image_size=224)
    assert sorted(net.state_dict()) == fx["unet_patched_keys"].tolist()
    assert sum(p.numel() for p in net.parameters()) == int(fx["unet_patched_numel"])
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == DO.unet_param_shapes(DO.unet_patched_cfg(cond_channels=32, image_size=224))
    small = PatchedUNetCondCat(**{k: v for k, v in SMALL.items()})
    cfg = DO.UNetCfg(**SMALL)
    small.load_state_dict(DO.seeded_unet_state_dict(cfg, seed=3), strict=True)
    # zero-initialised tails like upstream (zero_module: the last convolution of every block, proj_out, the output convolution)
    assert float(net.out[2].weight.abs().max()) == 0 and float(net.middle_block[1].proj_out.weight.abs().max()) == 0
    for tag, kw in (("cos", dict(beta_schedule="squaredcos_cap_v2", zero_terminal_snr=True)), ("lin", dict(beta_schedule="linear", zero_terminal_snr=False)),
                    ("shift", dict(beta_schedule="shifted_cosine:0.5"))):
        for cls in (DDIMScheduler, DDPMScheduler):
            assert float((cls(**kw).alphas_cumprod - torch.from_numpy(fx[f"ac_{tag}"])).abs().max()) == 0, (tag, cls.__name__)
    s = DDIMScheduler()
    for mode in ("trailing", "leading", "linspace"):
        s.set_timesteps(7, mode=mode)
        assert np.array_equal(s.timesteps.numpy(), fx[f"ts_{mode}"])
    p = DDPMScheduler()
    p.set_timesteps(7, mode="trailing")
    assert np.array_equal(p.timesteps.numpy(), fx["ts_ddpm"]) and int(p.previous_timestep(142)) == 0
    m = DiVAE(image_size=32, n_channels=3, enc_type="vit_s_enc", patch_size=16, codebook_size=64, latent_dim=8, post_mlp=True, scheduler="ddim",
              prediction_type="sample", beta_schedule="linear")
    assert isinstance(m.noise_scheduler, DDIMScheduler) and m.noise_scheduler.config.thresholding and not m.noise_scheduler.config.clip_sample
    assert any(k.startswith("decoder.input_blocks.0.0.") for k in m.state_dict())
    with pytest.raises(RuntimeError):
        small(torch.zeros(1, 3, 32, 32), 10, torch.zeros(1, 8, 2, 2))          # no CPU path
    with pytest.raises(NotImplementedError):
        DiVAE(image_size=32, enc_type="vit_s_enc", patch_size=16, codebook_size=64, latent_dim=8, dec_type="uvit_b_p4_f16")


# ---- GPU -------------------------------------------------------------------------------------------------------------------------------
def _small_net():
    from fourm.vq.models.unet import PatchedUNetCondCat
    cfg = DO.UNetCfg(**SMALL)
    sd = DO.seeded_unet_state_dict(cfg, seed=3)
    net = PatchedUNetCondCat(**SMALL)
    net.load_state_dict(sd, strict=True)
    return net.cuda().eval(), cfg, sd


@pytest.mark.gpu
@pytest.mark.parametrize("B,HW,C,silu,with_add", [(1, 196, 256, True, True), (4, 3136, 256, True, False), (8, 784, 512, True, True), (2, 49, 1024, False, True),
                                                   (8, 196, 128, True, True), (3, 50, 512, False, False)])
def test_groupnorm_nhwc_both_forms(B, HW, C, silu, with_add):
    """fm_groupnorm_nhwc on bf16 NHWC rows (+ a per-sample per-channel addend in front: GroupNorm32(h + emb), unet.py:230-246) against
    F.group_norm in fp32: the three-kernel form (B * groups < 64) and the one-launch form (workgroup = one (sample, group))."""
    from fourm.hip import _lib as L, ops
    torch.manual_seed(B * 1000 + HW + C)
    G = 32
    x = (torch.randn(B * HW, C, device="cuda") * 1.5 + 0.7).bfloat16()
    add = torch.randn(B, C, device="cuda") if with_add else None
    w, b = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.2
    y = torch.zeros(B * HW, C, device="cuda", dtype=torch.bfloat16)
    st = torch.zeros(B * G * ((HW + 31) // 32 + 1) * 2, device="cuda")
    L.check(L.groupnorm_nhwc(ops._p(x), C, ops._p(add), C if with_add else 0, ops._p(w), ops._p(b), ops._p(y), C, ops._p(st), B, HW, C, G, 1e-5, 1 if silu else 0, ops._stream()))
    xin = x.float().view(B, HW, C) + (add[:, None, :] if with_add else 0.0)
    ref = torch.nn.functional.group_norm(xin.permute(0, 2, 1), G, w, b, 1e-5).permute(0, 2, 1)
    if silu:
        ref = torch.nn.functional.silu(ref)
    err = float((y.float().view(B, HW, C) - ref).abs().max())
    assert err <= 2 ** -7 * float(ref.abs().max()), (err, float(ref.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,C,Co,stride,up,f32", [(2, 14, 14, 128, 256, 1, 0, False), (8, 7, 7, 512, 512, 1, 0, False), (3, 28, 28, 256, 256, 2, 0, False),
                                                      (2, 28, 28, 256, 512, 1, 1, False), (2, 56, 56, 256, 48, 1, 0, True), (1, 9, 11, 64, 100, 1, 0, False),
                                                      (8, 56, 56, 256, 256, 1, 0, False)])
def test_implicit_conv_gemm(B, H, W, C, Co, stride, up, f32):
    """fm_gemm_nt with conv_*: the 3 x 3 convolution as an implicit GEMM (the gather inside the LDS-DMA addresses) against fm_unet_im2col + the plain
    launch on the same rows, and against F.conv2d in fp32 on the same bf16 operands (stride 1 | 2, nearest x2 up-sampling in front, fp32 output,
    grids small enough to be split over K and large ones)."""
    from fourm.hip import _lib as L, ops
    torch.manual_seed(H * W + C + Co)
    hi, wi = H >> up, W >> up                                     # physical input grid; (H, W) is the grid the convolution reads
    x = torch.randn(B * hi * wi, C, device="cuda").bfloat16()
    w4 = (torch.randn(Co, C, 3, 3, device="cuda") * (1.0 / (3.0 * C ** 0.5))).bfloat16()
    wk = w4.permute(0, 2, 3, 1).reshape(Co, 9 * C).contiguous()     # taps outermost
    bias = torch.randn(Co, device="cuda")
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    M = B * Ho * Wo
    ldo = (Co + 63) // 64 * 64
    odt = torch.float32 if f32 else torch.bfloat16
    got = torch.full((M, ldo), 3.0, device="cuda", dtype=odt)
    ops.gemm_nt(x, wk, got, epilogue=L.EPI_F32 if f32 else L.EPI_BF16, bias=bias, M=M, N=Co, K=9 * C, conv=dict(C=C, H=H, W=W, Ho=Ho, Wo=Wo, stride=stride, up=up))
    col = torch.zeros(M, 9 * C, device="cuda", dtype=torch.bfloat16)
    L.check(L.unet_im2col(ops._p(x), C, C, None, 0, 0, 0, 0, ops._p(col), 9 * C, 9 * C, B, H, W, 3, stride, up, ops._stream()))
    ref2 = torch.full((M, ldo), 3.0, device="cuda", dtype=odt)
    ops.gemm_nt(col, wk, ref2, epilogue=L.EPI_F32 if f32 else L.EPI_BF16, bias=bias, M=M, N=Co, K=9 * C)
    assert rel(got[:, :Co], ref2[:, :Co]) < 2e-3, rel(got[:, :Co], ref2[:, :Co])
    xin = x.float().view(B, hi, wi, C).permute(0, 3, 1, 2)
    if up:
        xin = torch.nn.functional.interpolate(xin, scale_factor=2, mode="nearest")
    ref = torch.nn.functional.conv2d(xin, w4.float(), (bias if f32 else bias.bfloat16().float()), stride=stride, padding=1).permute(0, 2, 3, 1).reshape(M, Co)
    err = float((got[:, :Co].float() - ref).abs().max())
    assert err <= 2 ** -7 * float(ref.abs().max()) + 1e-3, (err, float(ref.abs().max()))
    assert bool((got[:, (Co + 3) // 4 * 4:] == 3.0).all())            # columns past roundup4(Co) untouched


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,heads,ch", [(8, 196, 1, 512), (8, 49, 1, 512), (2, 30, 2, 64), (4, 200, 2, 128)])
def test_unet_attention(B, T, heads, ch):
    """fm_unet_attention (QKVAttentionLegacy, unet.py:329-358: per head [q | k | v] channel blocks, scale ch^-1/4 on q and k, fp32 softmax) against a
    torch restatement.  (An eight-queries-per-workgroup variant was measured in round 6: 157 us against 49 us per call at batch 8, T = 196, ch = 512 -
    both forms pay one LDS read per multiply-add, and the one-query form has 8 x the workgroups to hide it; dropped.)"""
    from fourm.hip import _lib as L, ops
    torch.manual_seed(T + ch)
    qkv = torch.randn(B * T, heads * 3 * ch, device="cuda").bfloat16()
    out = torch.zeros(B * T, heads * ch, device="cuda", dtype=torch.bfloat16)
    L.check(L.unet_attention(ops._p(qkv), qkv.stride(0), ops._p(out), out.stride(0), B, T, heads, ch, ops._stream()))
    x = qkv.float().view(B, T, heads, 3, ch)
    q, k, v = x[:, :, :, 0], x[:, :, :, 1], x[:, :, :, 2]
    w = torch.softmax(torch.einsum("bthc,bshc->bhts", q, k) / ch ** 0.5, dim=-1)
    ref = torch.einsum("bhts,bshc->bthc", w, v).reshape(B * T, heads * ch)
    err = float((out.float() - ref).abs().max())
    assert err <= 2 ** -7 * float(ref.abs().max()) + 1e-3, (err, float(ref.abs().max()))


@pytest.mark.gpu
def test_unet_evaluation_replayed_from_a_graph_is_bit_identical():
    """In eval mode the third evaluation of a (shape, stream) is captured into a hipGraph and replayed from then on (FOURM_UNET_GRAPH): the same
    bits as the eager launch sequence, for scalar and per-sample timesteps, with and without the conditioning mask, across weight updates."""
    from fourm.vq import DiVAE
    import fourm.vq.models.unet.unet as U
    torch.manual_seed(0)
    m = DiVAE(image_size=64, n_channels=3, enc_type="vit_s_enc", patch_size=16, codebook_size=256, latent_dim=16, post_mlp=True, norm_codes=True,
              scheduler="ddim", prediction_type="sample", beta_schedule="linear", sync_codebook=False)
    for p in m.decoder.parameters():
        if float(p.detach().abs().max()) == 0:
            torch.nn.init.normal_(p, std=0.02)
    m = m.cuda().eval()
    x, q = torch.randn(5, 3, 64, 64, device="cuda"), torch.randn(5, 16, 4, 4, device="cuda")
    ts = torch.tensor([10.0, 500.0, 999.0, 3.0, 250.0], device="cuda")
    mask = torch.rand(5, 4, 4, device="cuda") < 0.3
    saved = U.UNET_GRAPH
    try:
        for args in ((x, 500, q), (x, ts, q), (x, ts, q, mask)):
            U.UNET_GRAPH = False
            want = m.decoder(*args).clone()
            U.UNET_GRAPH = True
            for i in range(5):                                        # two eager warm-ups, the capture, two replays
                xi = x + 0.0 if i != 3 else x * 0.5                   # (a replay with other inputs in between)
                got = m.decoder(xi, *args[1:])
                if i != 3:
                    assert torch.equal(got, want), (len(args), i)
        with torch.no_grad():
            next(m.decoder.parameters()).mul_(1.01)                  # a weight update drops the captured graphs
        U.UNET_GRAPH = False
        want = m.decoder(x, 500, q).clone()
        U.UNET_GRAPH = True
        for i in range(4):
            assert torch.equal(m.decoder(x, 500, q), want), i
    finally:
        U.UNET_GRAPH = saved


@pytest.mark.gpu
def test_decode_token_batches_in_flight_on_two_streams():
    """fourm.vq.decode_token_batches: two sampling loops in flight on their own HIP streams give, bit for bit, the images of one decode after the
    other (same per-batch CPU generators; the UNet engine keeps a scratch set per stream)."""
    from fourm.vq import DiVAE, decode_token_batches
    torch.manual_seed(5)
    m = DiVAE(image_size=64, n_channels=3, enc_type="vit_s_enc", patch_size=16, codebook_size=256, latent_dim=16, post_mlp=True, norm_codes=True,
              scheduler="ddim", prediction_type="sample", beta_schedule="linear", sync_codebook=False)
    for p in m.decoder.parameters():
        if float(p.detach().abs().max()) == 0:
            torch.nn.init.normal_(p, std=0.02)
    m = m.cuda().eval()
    toks = [torch.randint(0, 256, (8 if i % 2 else 5, 4, 4), device="cuda") for i in range(5)]
    want = [m.decode_tokens(t, timesteps=4, generator=torch.Generator().manual_seed(100 + i), verbose=False).clone() for i, t in enumerate(toks)]
    for n in (2, 1):
        got = decode_token_batches(m, toks, n_streams=n, timesteps=4, generator=[torch.Generator().manual_seed(100 + i) for i in range(len(toks))], verbose=False)
        torch.cuda.synchronize()
        assert all(torch.equal(g, w) for g, w in zip(got, want)), n


@pytest.mark.gpu
def test_hip_unet_matches_upstream_fixture():
    """One evaluation of the conditional UNet (per-sample timesteps, conditioning mask, integer timestep): bf16 GEMM operands / fp32
    accumulation against upstream's fp32 run."""
    fx = fixture()
    net, cfg, sd = _small_net()
    x, cond, ts, mask = (torch.from_numpy(fx[k]).cuda() for k in ("x", "cond", "ts", "mask"))
    errs = {}
    for name, got in (("unet", net(x, ts, cond)), ("unet_masked", net(x, ts, cond, cond_mask=mask)), ("unet_t250", net(x, 250, cond))):
        assert got.shape == x.shape and got.dtype == torch.float32
        errs[name] = rel(got, fx[name])
    print("UNet vs upstream fp32:", errs)
    assert max(errs.values()) < 2e-2, errs
    assert torch.equal(net(x, ts, cond), net(x, ts, cond))          # bit-reproducible (no atomics anywhere in the decoder)


@pytest.mark.gpu
def test_hip_scheduler_steps_match_upstream_fixture():
    """DDIM (eta = 0 and > 0) and DDPM steps for the three prediction types with dynamic thresholding / clipping / neither: fp32 element-wise
    kernels + the radix-select quantile against upstream's torch arithmetic."""
    from fourm.vq.scheduling import DDIMScheduler, DDPMScheduler
    fx = fixture()
    mo, smp, nz = (torch.from_numpy(fx[k]).cuda() for k in ("step_model_output", "step_sample", "step_noise"))
    worst = 0.0
    for i, (pred, thr, clip) in enumerate(step_cases(fx)):
        kw = dict(num_train_timesteps=1000, beta_schedule="squaredcos_cap_v2", prediction_type=pred, thresholding=thr, clip_sample=clip, zero_terminal_snr=pred == "v_prediction")
        s = DDIMScheduler(**kw)
        s.set_timesteps(10, mode="trailing")
        for t in (899, 99):
            for eta in (0.0, 0.6):
                got = s.step(mo, t, smp, eta=eta, variance_noise=nz if eta > 0 else None).prev_sample
                e = rel(got, fx[f"ddim_{i}_{t}_{int(eta * 10)}"])
                worst = max(worst, e)
                assert e < 2e-5, ("ddim", pred, thr, clip, t, eta, e)
        p = DDPMScheduler(**kw)
        p.set_timesteps(10)
        for t in (900, 100, 0):
            got = p.step(mo, t, smp, generator=torch.Generator().manual_seed(77)).prev_sample       # CPU generator: upstream's draws
            e = rel(got, fx[f"ddpm_{i}_{t}"])
            worst = max(worst, e)
            assert e < 2e-5, ("ddpm", pred, thr, clip, t, e)
    print("scheduler steps, worst relative error:", worst)
    # the quantile kernel alone against torch.quantile (ties, a constant row, an odd length)
    from fourm.hip import _lib as L, ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(5, 3001, generator=g)
    x[1] = 0.25
    x[2, :1500] = 1.0
    xc = x.cuda().contiguous()
    out = torch.empty(5, device="cuda")
    for q in (0.995, 0.5, 0.0, 1.0):
        L.check(L.quantile_abs(ops._p(xc), 5, 3001, q, ops._p(out), ops._stream()))
        want = torch.quantile(x.abs(), q, dim=1)
        assert float((out.cpu() - want).abs().max()) < 1e-6, (q, out.cpu(), want)


@pytest.mark.gpu
def test_hip_sampling_loop_matches_upstream_fixture():
    """PipelineCond end to end: the start image from the CPU generator, 4 DDIM / 3 DDPM steps of (UNet evaluation, scheduler step)."""
    from fourm.vq.scheduling import DDIMScheduler, DDPMScheduler, PipelineCond
    fx = fixture()
    net, cfg, sd = _small_net()
    cond = torch.from_numpy(fx["cond"]).cuda()
    for kind, cls, n in (("ddim", DDIMScheduler, 4), ("ddpm", DDPMScheduler, 3)):
        sch = cls(num_train_timesteps=1000, thresholding=True, clip_sample=False, beta_schedule="squaredcos_cap_v2", prediction_type="v_prediction", zero_terminal_snr=True)
        img = PipelineCond(model=net, scheduler=sch)(cond, generator=torch.Generator().manual_seed(5), timesteps=n, verbose=False, scheduler_timesteps_mode="trailing")
        want = torch.from_numpy(fx[f"loop_{kind}"])
        e = rel(img, want)
        mse = float(((img.cpu() - want) ** 2).mean())
        psnr = 10 * np.log10(4.0 / max(mse, 1e-20))                  # images live in [-1, 1]
        print(f"{kind}: {n} steps, relative error {e:.2e}, PSNR vs upstream's fp32 decode {psnr:.1f} dB")
        assert e < 4e-2 and psnr > 35, (kind, e, psnr)


@pytest.mark.gpu
def test_divae_full_size_decoder_against_oracle():
    """``fourm.vq.DiVAE`` with the real ``unet_patched`` decoder (196 M parameters, 224 x 224, 56 x 56 patch grid, 14 x 14 conditioning):
    one decoder evaluation through DiVAE.forward and a 2-step DDIM ``decode_tokens`` against the CPU oracle on the same seeded weights."""
    from fourm.vq import DiVAE
    torch.manual_seed(0)
    m = DiVAE(image_size=224, n_channels=3, enc_type="vit_b_enc", patch_size=16, codebook_size=1024, latent_dim=32, post_mlp=True, norm_codes=True,
              scheduler="ddim", prediction_type="sample", beta_schedule="linear", sync_codebook=False)
    ucfg = DO.unet_patched_cfg(cond_channels=32, image_size=224)
    sd = DO.seeded_unet_state_dict(ucfg, seed=1)
    m.decoder.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    g = torch.Generator().manual_seed(2)
    tokens = torch.randint(0, 1024, (1, 14, 14), generator=g)
    quant = m.tokens_to_embedding(tokens.cuda()).float().cpu()
    noised = torch.randn(1, 3, 224, 224, generator=g)
    clean = torch.rand(1, 3, 224, 224, generator=g) * 2 - 1
    # decoder evaluation at a given (quant, noised, t)
    got = m.decoder(noised.cuda(), 601, quant.cuda())
    want = DO.unet_forward(sd, ucfg, noised, 601, quant)
    e1 = rel(got, want)
    dec, code_loss = m(clean.cuda(), noised.cuda(), torch.tensor([601]))
    assert dec.shape == (1, 3, 224, 224) and torch.isfinite(dec).all()
    # two DDIM steps from tokens
    img = m.decode_tokens(tokens.cuda(), timesteps=2, generator=torch.Generator().manual_seed(9), verbose=False)
    noise0 = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(9))
    scfg = DO.SchedCfg(kind="ddim", beta_schedule="linear", prediction_type="sample", thresholding=True, clip_sample=False, zero_terminal_snr=True)
    oimg, _ = DO.sample_loop(sd, ucfg, scfg, quant, noise0, 2, "trailing")
    e2 = rel(img, oimg)
    psnr = 10 * np.log10(4.0 / max(float(((img.cpu() - oimg) ** 2).mean()), 1e-20))
    print(f"unet_patched: model_output {e1:.2e}; 2-step DDIM decode_tokens {e2:.2e}, PSNR {psnr:.1f} dB vs the fp32 oracle")
    assert e1 < 2e-2 and e2 < 4e-2 and psnr > 35, (e1, e2, psnr)
