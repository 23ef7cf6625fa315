"""Codebook containers in the upstream layout (``fourm/vq/quantizers/quantize_lucid.py``: ``CosineSimCodebook``
:303-428, ``VectorQuantize`` :432-568).  Inference (eval-mode) semantics only: nearest code by cosine
similarity, ``quantize = embed[index]``, zero loss.  The EMA / dead-code training branch is out of scope
(SURVEY §8f item 4)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class CosineSimCodebook(nn.Module):
    def __init__(self, dim, codebook_size, kmeans_init=False, kmeans_iters=10, decay=0.8, eps=1e-5, threshold_ema_dead_code=2,
                 code_replacement_policy="batch_random", use_ddp=False, learnable_codebook=False, sample_codebook_temp=0.):
        super().__init__()
        if learnable_codebook or sample_codebook_temp:
            raise NotImplementedError("learnable / sampled codebooks are not implemented")
        self.decay, self.codebook_size, self.eps = decay, codebook_size, eps
        if kmeans_init:
            embed = torch.zeros(codebook_size, dim)
        else:
            embed = torch.empty(codebook_size, dim)
            nn.init.kaiming_uniform_(embed)
            embed = F.normalize(embed, p=2, dim=-1)
        self.register_buffer("initted", torch.Tensor([not kmeans_init]))
        self.register_buffer("cluster_size", torch.zeros(codebook_size))
        self.register_buffer("embed", embed)


class VectorQuantize(nn.Module):
    def __init__(self, dim, codebook_size, codebook_dim=None, heads=1, decay=0.8, eps=1e-5, kmeans_init=False, kmeans_iters=10,
                 use_cosine_sim=False, threshold_ema_dead_code=0, code_replacement_policy="batch_random", channel_last=False,
                 accept_image_fmap=True, commitment_weight=1., orthogonal_reg_weight=0., orthogonal_reg_active_codes_only=False,
                 orthogonal_reg_max_codes=None, sample_codebook_temp=0., sync_codebook=False, norm_latents=False):
        super().__init__()
        if heads != 1 or (codebook_dim or dim) != dim:
            raise NotImplementedError("multi-head codebooks / codebook projections are not implemented")
        if not use_cosine_sim:
            raise NotImplementedError("only the cosine-similarity codebook (norm_codes=True) has a HIP kernel")
        self.heads, self.codebook_size, self.norm_latents = heads, codebook_size, norm_latents
        self.project_in, self.project_out = nn.Identity(), nn.Identity()
        self._codebook = CosineSimCodebook(dim=dim, codebook_size=codebook_size, kmeans_init=kmeans_init, kmeans_iters=kmeans_iters,
                                           decay=decay, eps=eps, threshold_ema_dead_code=threshold_ema_dead_code,
                                           code_replacement_policy=code_replacement_policy, use_ddp=sync_codebook)

    @property
    def codebook(self):
        return self._codebook.embed

    def indices_to_embedding(self, indices):
        return F.embedding(indices, self.codebook).permute(0, 3, 1, 2)
