"""Codebook containers in the upstream layout (``fourm/vq/quantizers/quantize_lucid.py``: ``CosineSimCodebook``
:303-428, ``VectorQuantize`` :432-568): nearest code by cosine similarity, ``quantize = embed[index]``, and in training mode the
EMA codebook update with dead-code replacement (``CosineSimCodebook.ema_update_``: fm_vq_code_stats + fm_vq_ema_update) and the k-means
codebook initialisation (``init_embed_``, upstream ``kmeans`` :137-167); ``EuclideanCodebook`` (:181-301, ``norm_codes=False``).  The straight-through / commitment gradient of tokenizer training
lives in fourm/vq/engine.py (fm_vq_latent_grad)."""
import torch
import torch.nn as nn
import torch.nn.functional as F



class _InittedOnce:
    """``initted`` is a device buffer (upstream's checkpoint layout) that flips once, on the first training batch of a k-means-initialised
    codebook.  Reading it costs a host synchronisation - per tokenize call, and impossible under hipGraph capture - so the answer is cached on
    the host once it is True; loading a state dict or moving the module forgets the cache."""

    def is_initted(self) -> bool:
        if not self.__dict__.get("_initted_host", False):
            self.__dict__["_initted_host"] = bool(self.initted)
        return self.__dict__["_initted_host"]

    def _mark_initted(self):
        self.initted.fill_(1.0)
        self.__dict__["_initted_host"] = True

    def _load_from_state_dict(self, *args, **kwargs):
        self.__dict__["_initted_host"] = False
        return super()._load_from_state_dict(*args, **kwargs)


class CosineSimCodebook(_InittedOnce, nn.Module):
    def __init__(self, dim, codebook_size, kmeans_init=False, kmeans_iters=10, decay=0.8, eps=1e-5, threshold_ema_dead_code=2,
                 code_replacement_policy="batch_random", use_ddp=False, learnable_codebook=False, sample_codebook_temp=0.):
        super().__init__()
        if learnable_codebook or sample_codebook_temp:
            raise NotImplementedError("learnable / sampled codebooks are not implemented")
        self.decay, self.codebook_size, self.eps, self.kmeans_iters = decay, codebook_size, eps, kmeans_iters
        self.threshold_ema_dead_code, self.code_replacement_policy, self.use_ddp = threshold_ema_dead_code, code_replacement_policy, use_ddp
        self.epoch = 0                       # bumped by every in-place codebook update (derived copies are keyed on it)
        if kmeans_init:
            embed = torch.zeros(codebook_size, dim)
        else:
            embed = torch.empty(codebook_size, dim)
            nn.init.kaiming_uniform_(embed)
            embed = F.normalize(embed, p=2, dim=-1)
        self.register_buffer("initted", torch.Tensor([not kmeans_init]))
        self.register_buffer("cluster_size", torch.zeros(codebook_size))
        self.register_buffer("embed", embed)


    def _assign(self, z, normalized_codes):
        """tokens (R) int64 = argmax_c <l2norm(z[r]), normalized_codes[c]> (first maximum), through fm_vq_assign."""
        from fourm.hip import _lib as L, ops
        K, D = normalized_codes.shape
        R = z.shape[0]
        splits = max(1, min(16, K // 1024))
        wv = torch.empty(R, splits, dtype=torch.float32, device=z.device)
        wi = torch.empty(R, splits, dtype=torch.int32, device=z.device)
        tokens = torch.empty(R, dtype=torch.int64, device=z.device)
        L.check(L.vq_assign(ops._p(z), z.stride(0), ops._p(normalized_codes), ops._p(normalized_codes), K, D, R, 1, 1, ops._p(wv), ops._p(wi), splits,
                            ops._p(tokens), None, ops._stream()))
        return tokens

    @torch.no_grad()
    def init_embed_(self, z, generator=None, init_index=None):
        """k-means initialisation of the codebook from the first batch of latents (upstream ``init_embed_`` :330-341 -> ``kmeans`` :137-167
        with cosine similarity): means = K random l2-normalised latents, then ``kmeans_iters`` rounds of  assign (argmax of the cosine) ->
        per-cluster mean -> l2norm, empty clusters keeping their mean.  Each round is the three kernels of the training step
        (fm_vq_assign, fm_vq_code_stats, fm_vq_ema_update with decay 0).  z: f32 (R, d) raw latents."""
        import torch.distributed as dist
        from fourm.hip import _lib as L, ops
        if self.is_initted():
            return
        multi = self.use_ddp and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        z = z.reshape(-1, z.shape[-1]).float().contiguous()
        K, D = self.embed.shape
        R = z.shape[0]
        if init_index is None:
            if multi:
                raise NotImplementedError("distributed sampling of the initial means (sample_vectors_distributed) is not implemented: pass init_index")
            init_index = torch.randperm(R, device=z.device, generator=generator)[:K] if R >= K else torch.randint(0, R, (K,), device=z.device, generator=generator)
        zn = torch.empty_like(z)
        L.check(L.l2norm_rows(ops._p(z), z.stride(0), ops._p(zn), zn.stride(0), R, D, ops._stream()))
        idx = init_index.to(device=z.device, dtype=torch.int64).contiguous()
        L.check(L.embed_rows_f32(ops._p(zn), ops._p(idx), ops._p(self.embed), self.embed.stride(0), K, D, ops._stream()))
        bins = torch.empty(K, dtype=torch.float32, device=z.device)
        sums = torch.empty(K, D, dtype=torch.float32, device=z.device)
        for _ in range(self.kmeans_iters):
            tokens = self._assign(z, self.embed)                          # (the means are unit vectors from the first round on)
            L.check(L.vq_code_stats(ops._p(z), z.stride(0), ops._p(tokens), R, D, K, ops._p(bins), ops._p(sums), ops._stream()))
            if multi:
                dist.all_reduce(bins)
                dist.all_reduce(sums)
            L.check(L.vq_ema_update(ops._p(bins), ops._p(sums), ops._p(self.embed), ops._p(self.cluster_size), K, D, 0.0, ops._stream()))
        self._mark_initted()
        self.epoch += 1

    @torch.no_grad()
    def ema_update_(self, z, tokens, generator=None):
        """Training-mode branch of upstream ``forward`` after the code assignment (quantize_lucid.py:409-426): per-code counts and
        sums of the L2-normalised latents (all-reduced when the codebook is synchronised), EMA of ``cluster_size`` and ``embed``,
        then ``expire_codes_`` ('batch_random': dead codes take random normalised latents of the batch, :366-383).
        z: f32 (R, d) latents as fed to the quantizer; tokens: int64 (R)."""
        import torch.distributed as dist
        from fourm.hip import _lib as L, ops
        if not self.is_initted():
            raise RuntimeError("the codebook is not initialised: init_embed_ runs in front of the first code assignment")
        z = z.reshape(-1, z.shape[-1])
        if z.dtype != torch.float32 or z.stride(1) != 1:
            z = z.float().contiguous()
        tokens = tokens.reshape(-1).contiguous()
        K, D = self.embed.shape
        R = z.shape[0]
        bins = torch.empty(K, dtype=torch.float32, device=z.device)
        sums = torch.empty(K, D, dtype=torch.float32, device=z.device)
        L.check(L.vq_code_stats(ops._p(z), z.stride(0), ops._p(tokens), R, D, K, ops._p(bins), ops._p(sums), ops._stream()))
        if self.use_ddp and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(bins)
            dist.all_reduce(sums)
        L.check(L.vq_ema_update(ops._p(bins), ops._p(sums), ops._p(self.embed), ops._p(self.cluster_size), K, D, float(self.decay), ops._stream()))
        self.epoch += 1
        if self.threshold_ema_dead_code > 0:
            dead = self.cluster_size < self.threshold_ema_dead_code
            n_dead = int(dead.sum())                                     # (upstream reads mask.sum().item() as well)
            if n_dead and self.code_replacement_policy == "linde_buzo_gray":
                # dead codes restart next to the most used ones (+ noise of 1e-10, re-normalised): quantize_lucid.py:351-356
                most_used = self.cluster_size.argsort(descending=True)[:n_dead]
                codes = self.embed[most_used]
                self.embed[dead] = F.normalize(codes + torch.randn(codes.shape, device=codes.device, generator=generator) * 1e-10, p=2, dim=-1)
            elif n_dead:
                if self.code_replacement_policy != "batch_random":
                    raise ValueError(f"{self.code_replacement_policy} is not a valid dead code replacement strategy.")
                if self.use_ddp and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                    raise NotImplementedError("distributed dead-code sampling (sample_vectors_distributed) is not implemented")
                if R >= n_dead:
                    idx = torch.randperm(R, device=z.device, generator=generator)[:n_dead]
                else:
                    idx = torch.randint(0, R, (n_dead,), device=z.device, generator=generator)
                self.embed[dead] = F.normalize(z[idx], p=2, dim=-1)
        return bins


class EuclideanCodebook(_InittedOnce, nn.Module):
    """Nearest code by Euclidean distance, EMA codebook (upstream ``EuclideanCodebook``, quantize_lucid.py:181-301; ``VectorQuantize(use_cosine_sim=False)``
    = ``VQ(norm_codes=False)``).  Same buffers as upstream (``initted``, ``cluster_size``, ``embed_avg``, ``embed``), so checkpoints load.
    Kernels: fm_vq_assign_bias with the bias -|e|^2 / 2 (arg-max of <z, e> - |e|^2 / 2 = arg-min of |z - e|^2), fm_vq_code_stats_raw,
    fm_vq_ema_update_euclid; dead-code replacement as upstream (which L2-normalises the replacement samples here too, :343-349)."""

    def __init__(self, dim, codebook_size, kmeans_init=False, kmeans_iters=10, decay=0.8, eps=1e-5, threshold_ema_dead_code=2,
                 code_replacement_policy="batch_random", use_ddp=False, learnable_codebook=False, sample_codebook_temp=0.):
        super().__init__()
        if learnable_codebook or sample_codebook_temp:
            raise NotImplementedError("learnable / sampled codebooks are not implemented")
        self.decay, self.codebook_size, self.eps, self.kmeans_iters = decay, codebook_size, eps, kmeans_iters
        self.threshold_ema_dead_code, self.code_replacement_policy, self.use_ddp = threshold_ema_dead_code, code_replacement_policy, use_ddp
        self.epoch = 0
        if kmeans_init:                                      # upstream: zeros until the first batch ran k-means (:196-197, :208); a checkpoint brings initted = 1
            embed = torch.zeros(codebook_size, dim)
        else:
            embed = torch.empty(codebook_size, dim)
            nn.init.kaiming_uniform_(embed)                  # upstream uniform_init (:58-61)
        self.register_buffer("initted", torch.Tensor([not kmeans_init]))
        self.register_buffer("cluster_size", torch.zeros(codebook_size))
        self.register_buffer("embed_avg", embed.clone())
        self.register_buffer("embed", embed)

    euclidean = True

    @torch.no_grad()
    def init_embed_(self, z, generator=None, init_index=None, normalize=False):
        """k-means initialisation from the first batch (upstream ``init_embed_`` :220-231 -> ``kmeans`` :137-167 with Euclidean distances):
        means = K random latents, ``kmeans_iters`` rounds of  nearest mean -> per-cluster mean (empty clusters keep theirs); then
        ``embed = embed_avg = means``, ``cluster_size = the last round's counts``.  Assignment and statistics are the training step's
        kernels (fm_vq_assign_bias, fm_vq_code_stats[_raw]).  z: f32 (R, d) latents (normalize: ``norm_latents`` models quantize l2norm(z))."""
        import torch.distributed as dist
        from fourm.hip import _lib as L, ops
        if self.is_initted():
            return
        multi = self.use_ddp and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        z = z.reshape(-1, z.shape[-1]).float().contiguous()
        K, D = self.embed.shape
        R = z.shape[0]
        if normalize:
            zn = torch.empty_like(z)
            L.check(L.l2norm_rows(ops._p(z), z.stride(0), ops._p(zn), zn.stride(0), R, D, ops._stream()))
            z = zn
        if init_index is None:
            if multi:
                raise NotImplementedError("distributed sampling of the initial means (sample_vectors_distributed) is not implemented: pass init_index")
            init_index = torch.randperm(R, device=z.device, generator=generator)[:K] if R >= K else torch.randint(0, R, (K,), device=z.device, generator=generator)
        idx = init_index.to(device=z.device, dtype=torch.int64).contiguous()
        means = torch.empty(K, D, dtype=torch.float32, device=z.device)
        L.check(L.embed_rows_f32(ops._p(z), ops._p(idx), ops._p(means), means.stride(0), K, D, ops._stream()))
        bins = torch.empty(K, dtype=torch.float32, device=z.device)
        sums = torch.empty(K, D, dtype=torch.float32, device=z.device)
        bias = torch.empty(K, dtype=torch.float32, device=z.device)
        splits = max(1, min(16, K // 1024))
        wv = torch.empty(R, splits, dtype=torch.float32, device=z.device)
        wi = torch.empty(R, splits, dtype=torch.int32, device=z.device)
        tokens = torch.empty(R, dtype=torch.int64, device=z.device)
        for _ in range(self.kmeans_iters):
            L.check(L.vq_code_bias(ops._p(means), K, D, ops._p(bias), ops._stream()))
            L.check(L.vq_assign_bias(ops._p(z), z.stride(0), ops._p(means), ops._p(bias), ops._p(means), K, D, R, 1, 0, ops._p(wv), ops._p(wi), splits,
                                     ops._p(tokens), None, ops._stream()))
            L.check(L.vq_code_stats_raw(ops._p(z), z.stride(0), ops._p(tokens), R, D, K, ops._p(bins), ops._p(sums), ops._stream()))
            if multi:
                dist.all_reduce(bins)
                dist.all_reduce(sums)
            means = torch.where((bins == 0)[:, None], means, sums / bins.clamp(min=1)[:, None])
        self.embed.copy_(means)
        self.embed_avg.copy_(means)
        self.cluster_size.copy_(bins)
        self._mark_initted()
        self.epoch += 1

    @torch.no_grad()
    def code_bias(self):
        """-|embed[k]|^2 / 2, recomputed when the codebook moved."""
        from fourm.hip import _lib as L, ops
        stamp = (self.embed._version, self.embed.data_ptr(), self.epoch)
        c = getattr(self, "_bias", None)
        if c is None or c[0] != stamp or c[1].device != self.embed.device:
            b = torch.empty(self.embed.shape[0], dtype=torch.float32, device=self.embed.device)
            L.check(L.vq_code_bias(ops._p(self.embed), self.embed.shape[0], self.embed.shape[1], ops._p(b), ops._stream()))
            c = self._bias = (stamp, b)
        return c[1]

    @torch.no_grad()
    def ema_update_(self, z, tokens, generator=None, normalize=False):
        """Training branch of upstream ``forward`` (:282-297): counts and sums of the latents per code, EMAs of ``cluster_size``
        and ``embed_avg``, ``embed = embed_avg / Laplace-smoothed cluster size``, then ``expire_codes_``.  The codebook itself never
        normalises; ``normalize=True`` is ``VectorQuantize(norm_latents=True)``, whose forward hands the codebook l2norm(z) (:525-527):
        the sums are then over the normalised rows (fm_vq_code_stats instead of fm_vq_code_stats_raw)."""
        import torch.distributed as dist
        from fourm.hip import _lib as L, ops
        z = z.reshape(-1, z.shape[-1])
        if z.dtype != torch.float32 or z.stride(1) != 1:
            z = z.float().contiguous()
        tokens = tokens.reshape(-1).contiguous()
        K, D = self.embed.shape
        R = z.shape[0]
        bins = torch.empty(K, dtype=torch.float32, device=z.device)
        sums = torch.empty(K, D, dtype=torch.float32, device=z.device)
        total = torch.empty(1, dtype=torch.float32, device=z.device)
        stats = L.vq_code_stats if normalize else L.vq_code_stats_raw
        L.check(stats(ops._p(z), z.stride(0), ops._p(tokens), R, D, K, ops._p(bins), ops._p(sums), ops._stream()))
        multi = self.use_ddp and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if multi:
            dist.all_reduce(bins)
            dist.all_reduce(sums)
        L.check(L.vq_ema_update_euclid(ops._p(bins), ops._p(sums), ops._p(self.embed), ops._p(self.embed_avg), ops._p(self.cluster_size), ops._p(total),
                                       K, D, float(self.decay), float(self.eps), ops._stream()))
        self.epoch += 1
        if self.threshold_ema_dead_code > 0:
            dead = self.cluster_size < self.threshold_ema_dead_code
            n_dead = int(dead.sum())
            if n_dead and self.code_replacement_policy == "linde_buzo_gray":
                most_used = self.cluster_size.argsort(descending=True)[:n_dead]
                codes = self.embed[most_used]
                self.embed[dead] = F.normalize(codes + torch.randn(codes.shape, device=codes.device, generator=generator) * 1e-10, p=2, dim=-1)
            elif n_dead:
                if self.code_replacement_policy != "batch_random":
                    raise ValueError(f"{self.code_replacement_policy} is not a valid dead code replacement strategy.")
                if multi:
                    raise NotImplementedError("distributed dead-code sampling (sample_vectors_distributed) is not implemented")
                idx = torch.randperm(R, device=z.device, generator=generator)[:n_dead] if R >= n_dead else torch.randint(0, R, (n_dead,), device=z.device, generator=generator)
                self.embed[dead] = F.normalize(z[idx], p=2, dim=-1)
            if n_dead:
                self.epoch += 1
        return bins


class VectorQuantize(nn.Module):
    def __init__(self, dim, codebook_size, codebook_dim=None, heads=1, decay=0.8, eps=1e-5, kmeans_init=False, kmeans_iters=10,
                 use_cosine_sim=False, threshold_ema_dead_code=0, code_replacement_policy="batch_random", channel_last=False,
                 accept_image_fmap=True, commitment_weight=1., orthogonal_reg_weight=0., orthogonal_reg_active_codes_only=False,
                 orthogonal_reg_max_codes=None, sample_codebook_temp=0., sync_codebook=False, norm_latents=False):
        super().__init__()
        if heads != 1 or (codebook_dim or dim) != dim:
            raise NotImplementedError("multi-head codebooks / codebook projections are not implemented")
        self.heads, self.codebook_size, self.norm_latents, self.commitment_weight = heads, codebook_size, norm_latents, commitment_weight
        self.project_in, self.project_out = nn.Identity(), nn.Identity()
        codebook_class = CosineSimCodebook if use_cosine_sim else EuclideanCodebook          # (quantize_lucid.py:474)
        self._codebook = codebook_class(dim=dim, codebook_size=codebook_size, kmeans_init=kmeans_init, kmeans_iters=kmeans_iters,
                                        decay=decay, eps=eps, threshold_ema_dead_code=threshold_ema_dead_code,
                                        code_replacement_policy=code_replacement_policy, use_ddp=sync_codebook)

    @property
    def codebook(self):
        return self._codebook.embed

    def indices_to_embedding(self, indices):
        return F.embedding(indices, self.codebook).permute(0, 3, 1, 2)


# names only upstream's same-named module defines resolve lazily (see fourm/_upstream.py)
from fourm import _upstream as _up
__getattr__ = _up.fallthrough(__name__, is_package=False)
