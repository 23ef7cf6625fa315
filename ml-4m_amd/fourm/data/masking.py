"""Device-side input / target masks for image-like modalities (first slice).

``image_mask_batched`` is the batched MI355X form of upstream ``UnifiedMasking.image_mask`` (fourm/data/masking.py:237-266), which
the loader's worker processes run per sample on the host: given the per-sample token budgets it ranks one uniform noise vector per
sample (fm_image_mask) and returns ``input_mask`` / ``target_mask`` / ``decoder_attention_mask`` exactly as upstream lays them out.
The token budgets themselves (Dirichlet draws, masking.py:181-235) and the sequence modalities' span masking (text tokenizer) stay
host code upstream; they are not reproduced here."""
from typing import Dict, Optional

import torch


@torch.no_grad()
def image_mask_batched(num_tokens: int, input_budget: torch.Tensor, target_budget: Optional[torch.Tensor] = None,
                       noise: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None) -> Dict[str, torch.Tensor]:
    """input_budget / target_budget: int (B) device tensors (target_budget None = upstream's ``target_budget=None``).
    noise: optional f32 (B, num_tokens) uniform draws (default: torch.rand on the budgets' device)."""
    from fourm.hip import _lib as L, ops
    dev = input_budget.device
    B = input_budget.shape[0]
    if noise is None:
        noise = torch.rand(B, num_tokens, device=dev, generator=generator)
    noise = noise.float().contiguous()
    if tuple(noise.shape) != (B, num_tokens):
        raise ValueError(f"noise must be (B, num_tokens) = {(B, num_tokens)}")
    kin = input_budget.to(torch.int32).contiguous()
    kt = None if target_budget is None else target_budget.to(torch.int32).contiguous()
    im = torch.empty(B, num_tokens, dtype=torch.bool, device=dev)
    tm = torch.empty(B, num_tokens, dtype=torch.bool, device=dev)
    dam = torch.empty(B, num_tokens, dtype=torch.int32, device=dev)
    L.check(L.image_mask(ops._p(noise), ops._p(kin), ops._p(kt), B, num_tokens, ops._p(im), ops._p(tm), ops._p(dam), ops._stream()))
    return {"input_mask": im, "target_mask": tm, "decoder_attention_mask": dam}


# the host-side masking classes (UnifiedMasking, TransferMasking, ...) stay upstream's
from .. import _upstream as _up
_up.merge(__name__, globals())
