"""Deterministic test cases shared by the fixture generator (container) and the tests (anywhere).

Every case is fully described by seeds: weights come from ``oracle.fourm_oracle.seeded_state_dict``,
batches from ``synthetic_mod_dict``; the fixtures hold only what the upstream model produced.
"""
from oracle import fourm_oracle as O


def micro_mods():
    """Seven small modalities covering every embedding kind (grid tokens, learned-position grid
    tokens, pixels, sequences, dense-embedding sequences)."""
    return [
        O.ModSpec("cap", "seq", vocab=100, n_pos=12),
        O.ModSpec("det", "seq", vocab=100, n_pos=10),
        O.ModSpec("rgb@32", "patch", n_pos=16, patch=8, in_dec=False),
        O.ModSpec("t5", "seq_emb", n_pos=6, orig_dim=24, in_dec=False),
        O.ModSpec("tok_a@32", "tok", vocab=64, n_pos=16, patch=8),
        O.ModSpec("tok_b@32", "tok", vocab=48, n_pos=16, patch=8),
        O.ModSpec("tok_g", "tok", vocab=40, n_pos=4, patch=16),
    ]


def l_like_mods():
    """4M-L / mod21 shaped modalities at 224 px: pixels, a T5-embedded caption (77 x 4096), grid tokens, a 16-token
    global modality with a learned position table, and two text sequences (BASELINE.json configs[3], depth 1)."""
    return [
        O.ModSpec("caption", "seq", vocab=3000, n_pos=64),
        O.ModSpec("det", "seq", vocab=1000, n_pos=40),
        O.ModSpec("rgb@224", "patch", n_pos=196, patch=16, in_dec=False),
        O.ModSpec("t5_caption", "seq_emb", n_pos=77, orig_dim=4096, in_dec=False),
        O.ModSpec("tok_depth@224", "tok", vocab=8192, n_pos=196, patch=16),
        O.ModSpec("tok_rgb@224", "tok", vocab=16384, n_pos=196, patch=16),
        O.ModSpec("tok_g", "tok", vocab=512, n_pos=16, patch=56),
    ]


def _micro(**kw):
    base = dict(dim=128, enc_depth=2, dec_depth=2, heads=2, mods=micro_mods())
    base.update(kw)
    return O.TrunkCfg(**base)


CASES = {
    # name: (cfg factory, options)
    "micro_swiglu": dict(cfg=lambda: _micro(), B=3, N=20, M=18, bud=(20, 18)),
    "micro_pad": dict(cfg=lambda: _micro(), B=4, N=24, M=24, bud=(17, 15), loss_type="token",
                      no_target=("tok_b@32",), seed=3),
    "micro_gelu": dict(cfg=lambda: _micro(gated=False, act="gelu", qkv_bias=True, proj_bias=True, mlp_bias=True,
                                          registers=2, causal=True),
                       B=3, N=20, M=18, bud=(20, 18), norm_bias=True, share_embedding=False, seed=5),
    "micro_qknorm": dict(cfg=lambda: _micro(qk_norm=True, sep=False), B=2, N=16, M=16, bud=(16, 16), seed=7),
    # 4M-L geometry (D=1024, 16 heads, hidden 2730 -> padded 2752, 256+256 tokens), one block each side
    "l_like": dict(cfg=lambda: O.TrunkCfg(dim=1024, enc_depth=1, dec_depth=1, heads=16, mods=l_like_mods()),
                   B=2, N=256, M=256, bud=(256, 256), seed=11),
    # BASELINE.json configs[0]: 4M-Ti mod7, one masked-modeling step, seq_len 128+128, batch 2
    "ti_mod7": dict(cfg=lambda: O.named_cfg("tiny", O.mod7_specs()), B=2, N=128, M=128, bud=(128, 128), seed=0),
    # BASELINE.json configs[1]: the benched model (4M-B mod7, 12+12 blocks, D=768, hidden 2048), batch 2 of the same
    # 128+128-token batches (the fixture keeps summaries only)
    "b_mod7": dict(cfg=lambda: O.named_cfg("base", O.mod7_specs()), B=2, N=128, M=128, bud=(128, 128), seed=4),
    # the same model at the BENCHED row count: batch 256 = 32768 encoder + 32768 decoder rows (row padding, 256-row head segments,
    # persistent tile walks, the dW tail cut all behave differently here than at batch 2).  Generated from the unmodified upstream
    # model with its blocks wrapped in torch.utils.checkpoint (memory; same arithmetic); summaries only; the oracle is not run
    "b_mod7_256": dict(cfg=lambda: O.named_cfg("base", O.mod7_specs()), B=256, N=128, M=128, bud=(128, 128), seed=8, big=True),
    # BASELINE.json configs[3]: 4M-L mod21 at FULL depth (24 + 24 blocks, D = 1024, hidden 2730, 19 input / 17 target modalities,
    # 256 + 256 tokens), batch 1 (the fixture keeps summaries only; 1.27 G parameters are regenerated from seeds)
    "l_mod21": dict(cfg=lambda: O.named_cfg("large", O.mod21_specs()), B=1, N=256, M=256, bud=(256, 256), seed=6,
                    learned=O.MOD21_LEARNED_POS),
}


def build_case(name: str):
    c = CASES[name]
    cfg = c["cfg"]()
    seed = c.get("seed", 1)
    learned = c.get("learned", ("tok_g",) if any(m.name == "tok_g" for m in cfg.mods) else ())
    share = c.get("share_embedding", True)
    nb = c.get("norm_bias", False)
    sd = O.seeded_state_dict(cfg, seed=seed, share_embedding=share, learned_pos=learned, norm_bias=nb)
    md = O.synthetic_mod_dict(cfg, c["B"], c["bud"][0], c["bud"][1], seed=seed, no_target=c.get("no_target", ()))
    return dict(cfg=cfg, sd=sd, mod_dict=md, N=c["N"], M=c["M"], loss_type=c.get("loss_type", "mod"),
                order_seed=seed, share_embedding=share, norm_bias=nb, learned_pos=learned, big=c.get("big", False))


# VQ tokenizer front end (BASELINE.json configs[4] and a small variant)
VQ_CASES = {
    "vq_small": dict(enc_type="vit_s_enc", image=64, patch=16, codebook=512, post_mlp=True, batch=3, seed=2),
    "vq_rgb224": dict(enc_type="vit_b_enc", image=224, patch=16, codebook=16384, post_mlp=True, batch=2, seed=0),
}

# VQ-VAE (decoder + tokenizer training step)
VQVAE_CASES = {
    "vqvae_small": dict(enc_type="vit_s_enc", dec_type="vit_s_dec", image=64, patch=16, codebook=512, post_mlp=True, batch=3, seed=2, commitment_weight=1.0),
    "vqvae_semseg": dict(enc_type="vit_s_enc", dec_type="vit_s_dec", image=64, patch=16, codebook=256, post_mlp=False, batch=3, seed=7, commitment_weight=0.5,
                         n_labels=12, channels=4, norm_latents=True),
    "vqvae_feat": dict(enc_type="vit_s_enc", dec_type="vit_s_dec", image=96, patch=16, codebook=256, post_mlp=True, batch=3, seed=9, commitment_weight=1.0,
                       channels=40, patch_proj=False),
    "vqvae_b224": dict(enc_type="vit_b_enc", dec_type="vit_b_dec", image=224, patch=16, codebook=16384, post_mlp=False, batch=2, seed=5, commitment_weight=0.25),
}
