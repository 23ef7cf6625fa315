"""CPU restatement (numpy / plain Python) of upstream's per-sample token budgets and sequence span masking with every random draw made
an explicit argument.  TEST INFRASTRUCTURE ONLY (see fourm_oracle.py): nothing under ml-4m_amd/ or bench.py's timed region imports it.

What it restates (fourm/data/masking.py of the reference):
  token_budget             UnifiedMasking.input_token_budget :181-205 and target_token_budget :207-234
  max_tokens_remaining     :218-219
  span_masking             simple_span_masking :58-91 and chunk_span_masking :94-127 (one function: a chunk is a run of tokens that
                           share one mask decision)
  sequence_mask            UnifiedMasking.sequence_mask :345-445 after tokenisation, and sequence_token_mask :268-343
  sequence_emb_mask        UnifiedMasking.sequence_emb_mask_span :448-516

Upstream draws its randomness inside these functions (Dirichlet.sample / sample_n, torch.rand, random.uniform, np.random.randint).  Here
the draws are inputs, in the order upstream consumes them, so that the HIP kernels (csrc/masking.hip), this file and upstream are
functions of the same numbers.  Pinned: tests/golden/make_golden_masking.py runs the unmodified UnifiedMasking with its samplers patched
to replay recorded draws and asserts this file reproduces every output before it writes tests/golden/masking.npz."""
import numpy as np

F = np.float32
DECAY = 0.9          # UnifiedMasking.keep_prob_decay_factor (:176)


# ---- token budgets --------------------------------------------------------------------------------------------------------------
def max_tokens_remaining(is_img, max_tokens, min_tokens, input_budget):
    """:218-219: image-like modalities give up the positions already used as inputs; never below min_tokens."""
    is_img, mx, mn, ib = (np.asarray(a) for a in (is_img, max_tokens, min_tokens, input_budget))
    return np.maximum(mn, np.where(is_img, mx - ib, mx)).astype(np.int32)


def token_budget(main_draws, extra_draws, num_tokens, min_tokens, max_tokens):
    """main_draws f32 (T, M): the Dirichlet sample of try t;  extra_draws f32 (T, E, M): the sample_n(diff) draws of try t (the first
    ``diff`` rows are used).  Returns (budget int32 (M), tries used).

        budget = floor(p * n);  diff = n - sum(budget);  budget += bincount(argmax(extra[:diff]));  budget = min(budget, max);
        accept the first try with budget >= min everywhere, else keep the last one (:187-205)."""
    main_draws, extra_draws = np.asarray(main_draws, dtype=F), np.asarray(extra_draws, dtype=F)
    mn, mx = np.asarray(min_tokens, dtype=np.int64), np.asarray(max_tokens, dtype=np.int64)
    T, M = main_draws.shape
    budget = None
    for t in range(T):
        budget = np.floor((main_draws[t] * F(num_tokens)).astype(F)).astype(np.int64)
        diff = int(num_tokens - budget.sum())
        diff = max(0, min(diff, extra_draws.shape[1]))
        if diff:
            budget += np.bincount(np.argmax(extra_draws[t, :diff], axis=-1), minlength=M)
        budget = np.minimum(budget, mx)
        if (budget >= mn).all():
            return budget.astype(np.int32), t + 1
    return budget.astype(np.int32), T + 1          # no try met the minimum: the last one is kept, reported as T + 1 (success on the last try: T)


# ---- span masking ---------------------------------------------------------------------------------------------------------------
def span_masking(tokens, unit_of, noise_row, keep_prob, sentinel_to_id):
    """tokens: list of ids; unit_of[l]: index of the mask decision token l follows (token index for simple_span_masking, chunk index
    for chunk_span_masking); a unit is KEPT iff noise_row[unit] <= float32(keep_prob)  (torch.rand(n) <= keep_prob, :71 / :107)."""
    kp = F(keep_prob)
    inp, tgt = [], []
    prev, count = False, 0                 # (token-level "previous masked" equals upstream's unit-level flag: tokens of one unit share
    for tok, u in zip(tokens, unit_of):    #  the decision, so only the first token of a masked unit can see an unmasked predecessor)
        masked = not (F(noise_row[u]) <= kp)
        if masked:
            if not prev:
                count += 1
                inp.append(sentinel_to_id[count]); tgt.append(sentinel_to_id[count])
            tgt.append(tok)
        else:
            inp.append(tok)
        prev = masked
    tgt.append(sentinel_to_id[count + 1])
    return inp, tgt


def truncate(tokens, unit_of, max_tokens):
    """:365 (tokens) / :376-377 (whole chunks whose cumulative length fits)."""
    n = min(len(tokens), max_tokens)
    if unit_of is None:
        return list(tokens[:n]), list(range(n))
    keep = 0
    for l in range(n):
        if l + 1 == len(tokens) or unit_of[l + 1] != unit_of[l]:
            keep = l + 1
    return list(tokens[:keep]), list(unit_of[:keep])


def _masked_sequences(tokens, unit_of, input_budget, keep_prob, noise, sentinel_to_id):
    """:389-408: the retry loop.  noise f32 (T, units).  Returns (input ids, target ids, tries used)."""
    if input_budget == 0:
        _, tgt = span_masking(tokens, unit_of, noise[0], 0.0, sentinel_to_id)
        return [], tgt, 1
    kp = float(keep_prob)
    t = 0
    inp, tgt = span_masking(tokens, unit_of, noise[0], kp, sentinel_to_id)
    while len(inp) > input_budget:
        kp = kp * DECAY
        t += 1
        if t >= len(noise):                # (the kernels stop after T draws and mask everything: kp -> 0 in the limit)
            inp, tgt = span_masking(tokens, unit_of, np.full(len(noise[0]), 2.0, dtype=F), 0.0, sentinel_to_id)
            break
        inp, tgt = span_masking(tokens, unit_of, noise[t], kp, sentinel_to_id)
    return inp, tgt, t + 1                  # (T + 1 when the draws ran out, T for a fit on the last one)


def sequence_mask(tokens, max_tokens, input_budget, target_budget, keep_prob, noise, r_choice, sentinel_to_id, pad_id,
                  unit_of=None, vocab_offset=0):
    """``tokens``: the ids upstream has after tokenising and appending [EOS] (:363 / :373) - or the raw ids of a seq_token modality, to
    which ``vocab_offset`` is added (:286).  ``keep_prob``: the first keep probability (sample_uniform / 1.0 / random.choice, :395-400).
    ``r_choice``: the integer np.random.randint is replaced by r_choice mod its argument (:425).
    Returns dict(tensor int32, input_mask bool, target_mask bool, decoder_attention_mask int32, tries)."""
    tokens = [int(t) + vocab_offset for t in tokens]
    tokens, units = truncate(tokens, unit_of, max_tokens)
    inp, tgt, tries = _masked_sequences(tokens, units, input_budget, keep_prob, np.asarray(noise, dtype=F), sentinel_to_id)
    L = (max_tokens + 1) * 2
    tensor = np.full(L, pad_id, dtype=np.int32)
    im, tm = np.ones(L, dtype=bool), np.ones(L, dtype=bool)
    dam = np.zeros(L, dtype=np.int32)
    tensor[:len(inp)] = inp
    im[:len(inp)] = False
    if target_budget is not None and len(tgt) > target_budget:
        sent = set(sentinel_to_id.values())
        idxs = [i for i, t in enumerate(tgt) if t in sent]
        chosen = int(r_choice) % max(1, len(idxs) - 1)
        if len(tgt) - idxs[chosen] >= target_budget:
            tgt = tgt[idxs[chosen]:idxs[chosen] + target_budget]
        else:
            for i in idxs:
                if len(tgt) - i <= target_budget:
                    tgt = tgt[i:]
                    break
    tensor[input_budget:input_budget + len(tgt)] = tgt
    tm[input_budget:input_budget + len(tgt)] = False
    dam[input_budget:input_budget + len(tgt)] = 1
    return dict(tensor=tensor, input_mask=im, target_mask=tm, decoder_attention_mask=dam, tries=tries)


def sequence_emb_mask(emb, max_tokens, input_budget, keep_prob, noise, sentinel_to_id):
    """:448-516.  emb f32 (n, D).  Sentinel positions of the input hold zero rows, kept positions their embedding."""
    emb = np.asarray(emb, dtype=F)
    sent = set(sentinel_to_id.values())
    fake, src, idn = [], {}, len(sent)
    while len(fake) < len(emb):
        if idn not in sent:
            src[idn] = len(fake)
            fake.append(idn)
        idn += 1
    fake = fake[:max_tokens]
    inp, _, tries = _masked_sequences(fake, list(range(len(fake))), input_budget, keep_prob, np.asarray(noise, dtype=F), sentinel_to_id)
    tensor = np.zeros((max_tokens, emb.shape[1]), dtype=F)
    for i, t in enumerate(inp):
        if t not in sent:
            tensor[i] = emb[src[t]]
    im = np.ones(max_tokens, dtype=bool)
    im[:len(inp)] = False
    return dict(tensor=tensor, input_mask=im, target_mask=np.ones(max_tokens, dtype=bool),
                decoder_attention_mask=np.zeros(max_tokens, dtype=np.int32), tries=tries)
