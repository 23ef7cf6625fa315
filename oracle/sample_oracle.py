"""CPU restatement (numpy) of the generation-step sampling kernels csrc/sample.hip.  TEST INFRASTRUCTURE ONLY (see fourm_oracle.py).

What it restates: GenerationSampler.top_k_top_p_filtering / sample_tokens / select_tokens_batched and the scatter updates of
maskgit_step_batched (fourm/models/generate.py:332-420, :650-661), under the determinism contract written at the top of
csrc/sample.hip: fixed-polynomial exp with separately rounded fp32 operations, integer top-p masses, index-order inverse CDF with a
fixed summation tree of 256 contiguous chunks.  Given the same logits and the same uniforms the HIP kernel must return the SAME
token ids, bit for bit (tests/test_generate_gpu.py).

Pinned to the reference by ``check_against_upstream_filter`` (run by tests/test_sample_oracle.py where torch is available): the set
of entries that survive top-k / top-p equals the one upstream's top_k_top_p_filtering leaves finite, on inputs without exact ties
at the cut."""
import numpy as np

F = np.float32
NT = 256


def okey(x):
    u = np.asarray(x, dtype=F).view(np.uint32)
    return np.where(u & np.uint32(0x80000000), ~u, u | np.uint32(0x80000000)).astype(np.uint32)


def exp_det(x):
    """exp(x), x <= 0, as csrc/sample.hip exp_det: every multiply and add rounded to fp32 separately."""
    x = np.asarray(x, dtype=F)
    t = (x * F(1.44269504088896341)).astype(F)
    n = np.rint(t).astype(F)
    r = (x + (-(n * F(0.693359375)).astype(F))).astype(F)
    r = (r + (-(n * F(-2.12194440e-4)).astype(F))).astype(F)
    p = np.full_like(x, F(1.3888888888888889e-3))
    for c in (8.3333333333333333e-3, 4.1666666666666667e-2, 1.6666666666666667e-1, 0.5, 1.0, 1.0):
        p = ((p * r).astype(F) + F(c)).astype(F)
    out = np.ldexp(p, n.astype(np.int32)).astype(F)
    return np.where(x < F(-80.0), F(0.0), out).astype(F)


def survivors(row, top_k, top_p):
    """Boolean mask of the entries of one fp32 logits row that survive top-k then top-p."""
    V = row.shape[0]
    keys = okey(row)
    keep = np.ones(V, dtype=bool)
    if 0 < top_k < V:
        kth = np.sort(keys)[V - top_k]                       # k-th largest key; ties at the threshold all survive
        keep &= keys >= kth
    if 0.0 < top_p < 1.0:
        mx = row.max()
        q = np.where(keep, (exp_det((row + (-mx)).astype(F)) * F(16777216.0)).astype(np.uint32), 0).astype(np.uint64)
        total = int(q.sum())
        thr = int(np.float64(np.float32(top_p)) * np.float64(total))
        order = np.argsort(keys, kind="stable")[::-1]        # descending keys
        ks, qs = keys[order], q[order]
        # mass of strictly larger keys: cumulative mass at the first occurrence of each key value
        cum_excl = np.concatenate([[0], np.cumsum(qs)[:-1]]).astype(np.uint64)
        first = np.concatenate([[True], ks[1:] != ks[:-1]])
        s_gt = np.maximum.accumulate(np.where(first, cum_excl, 0))
        ok = (s_gt <= thr) & (qs > 0)
        cut = ks[ok].min()                                    # the largest entry always survives (mass above it is 0)
        keep &= keys >= cut
    return keep


def sample_tokens(logits, temperature, top_k, top_p, uniforms):
    """logits (R, V) float32 (bf16 logits: convert exactly first) -> (ids int64 (R), probs float32 (R))."""
    logits = np.asarray(logits, dtype=F)
    R, V = logits.shape
    ids, probs = np.zeros(R, dtype=np.int64), np.zeros(R, dtype=F)
    if temperature < 1e-10:
        return logits.argmax(1).astype(np.int64), np.ones(R, dtype=F)
    inv_t = F(1.0) / F(temperature)
    C = (V + NT - 1) // NT
    for r in range(R):
        row = logits[r]
        keep = survivors(row, top_k, top_p)
        mx = row.max()
        p = np.where(keep, exp_det((((row + (-mx)).astype(F)) * inv_t).astype(F)), F(0.0)).astype(F)
        csum = np.zeros(NT, dtype=F)
        for c in range(NT):
            ch = p[c * C:min(V, (c + 1) * C)]
            csum[c] = np.cumsum(np.concatenate([[F(0.0)], ch]), dtype=F)[-1] if ch.size else F(0.0)
        cinc = np.cumsum(csum, dtype=F)
        total = cinc[-1]
        target = F(F(uniforms[r]) * total)
        c = 0
        while c < NT - 1 and not (cinc[c] > target):
            c += 1
        ch = p[c * C:min(V, (c + 1) * C)]
        acc = np.cumsum(np.concatenate([[cinc[c - 1] if c else F(0.0)], ch]), dtype=F)[1:]
        hit = np.nonzero((acc > target) & (ch > 0))[0]
        if hit.size:
            pick = c * C + int(hit[0])
        else:
            live = np.nonzero(ch > 0)[0]
            pick = c * C + int(live[-1]) if live.size else int(np.nonzero(p > 0)[0][-1])
        ids[r], probs[r] = pick, F(p[pick] / total)
    return ids, probs


def maskgit_commit(prob, samples, mod_pos, num_select, tensor, input_mask, target_mask):
    """In-place commit of the num_select most confident samples per batch element; returns top_idx (B, num_select)."""
    B, N = prob.shape
    top = np.zeros((B, num_select), dtype=np.int32)
    for b in range(B):
        order = sorted(range(N), key=lambda i: (-float(prob[b, i]), i))[:num_select]
        top[b] = order
        for i in order:
            pos = int(mod_pos[b, i])
            tensor[b, pos] = samples[b, i]
            input_mask[b, pos] = False
            target_mask[b, pos] = True
    return top


def upstream_filter_survivors(row, top_k, top_p):
    """The entries upstream's top_k_top_p_filtering (generate.py:332-371) leaves finite, restated with torch for the pin test."""
    import torch
    logits = torch.from_numpy(np.asarray(row, dtype=F)).clone()[None]
    if top_k > 0:
        k = min(top_k, logits.shape[-1])
        logits[logits < torch.topk(logits, k)[0][..., -1, None]] = float("-inf")
    if top_p > 0.0:
        sl, si = torch.sort(logits, dim=1, descending=True)
        cum = torch.cumsum(torch.softmax(sl, dim=-1), dim=-1)
        rem = cum > top_p
        rem[..., 1:] = rem[..., :-1].clone()
        rem[..., 0] = 0
        logits[torch.gather(rem, -1, torch.argsort(si, dim=-1))] = float("-inf")
    return torch.isfinite(logits[0]).numpy()


def roar_positions(target_mask, order_noise, num_select):
    """Positions one ROAR step decodes (upstream forward_mask_decoder_roar, generate.py:481-514): per sample the first
    n = min(num_select, #unmasked of sample 0) entries of argsort(target_mask + order_noise * 1e-6) - the still-masked positions with
    the smallest noise - returned in ascending position order (the HIP path decodes them in position order).  fp32 keys as upstream."""
    tm = np.asarray(target_mask, dtype=bool)
    B, L = tm.shape
    n = min(int(num_select), int((~tm[0]).sum()))
    key = tm.astype(np.float32) + np.asarray(order_noise, dtype=np.float32).reshape(1, L) * np.float32(1e-6)
    order = np.argsort(key, axis=1, kind="stable")[:, :n]
    return np.sort(order, axis=1).astype(np.int32)


def cfg_logits(cond, uncond, scale):
    """Classifier-free guidance on fp32 logits (generate.py:684): uncond + (cond - uncond) * scale, one rounding per operation."""
    c, u = np.asarray(cond, dtype=np.float32), np.asarray(uncond, dtype=np.float32)
    return (u + (c - u) * np.float32(scale)).astype(np.float32)
