"""Reader of upstream's pre-tokenised shard format (README_DATA.md:9-37, save_vq_tokens.py:293-305): per modality a directory of
``shard-XXXXX.tar`` files whose members are ``<sample key>.npy`` = int16 (n_crops, n_tokens) token ids, aligned across modalities by
member name.  ``iter_token_batches`` turns them into RAW batches - ids only, as int16 host arrays, no masks - which is all that has to
cross PCIe once the masking runs on the device (fourm.data.masking.DeviceUnifiedMasking): 196 x 2 bytes per image-like modality and
sample, against the 196 x (8 + 1 + 1 + 4) bytes of upstream's padded int64 ids + two bool masks + int32 attention mask.

Host code (numpy + tarfile), no third-party loader; the device side is ``raw_batch_to_device`` (fm_unpack_ids_u16)."""
import io
import os
import tarfile
from typing import Dict, Iterator, List, Optional, Sequence

import numpy as np
import torch


def shard_path(root: str, modality: str, shard: int) -> str:
    return os.path.join(root, modality, f"shard-{shard:05d}.tar")


def read_token_shard(path: str) -> Dict[str, np.ndarray]:
    """{sample key: int16 (n_crops, n_tokens)} of one modality's shard, in member order."""
    out = {}
    with tarfile.open(path, "r") as tar:
        for m in tar:
            if not m.isfile() or not m.name.endswith(".npy"):
                continue
            arr = np.load(io.BytesIO(tar.extractfile(m).read()), allow_pickle=False)
            if arr.dtype != np.int16 or arr.ndim != 2:
                raise ValueError(f"{path}:{m.name}: expected int16 (n_crops, n_tokens), got {arr.dtype} {arr.shape}")
            out[os.path.splitext(os.path.basename(m.name))[0]] = arr
    return out


def iter_token_batches(root: str, modalities: Sequence[str], shards: Sequence[int], batch_size: int, crop: Optional[int] = None,
                       rng: Optional[np.random.Generator] = None, drop_last: bool = True) -> Iterator[Dict[str, np.ndarray]]:
    """Aligned raw batches {modality: int16 (B, n_tokens)} over the given shards.  A sample is kept when every modality has its key
    (upstream's loader joins the per-modality tar streams by key the same way, unified_datasets.py:131-175).  ``crop``: which of the
    n_crops augmentations to take (None = a random one per sample and batch, the same index for every modality: the crops are aligned)."""
    rng = rng or np.random.default_rng(0)
    pending: Dict[str, List[np.ndarray]] = {m: [] for m in modalities}
    for sh in shards:
        per_mod = {m: read_token_shard(shard_path(root, m, sh)) for m in modalities}
        keys = [k for k in per_mod[modalities[0]] if all(k in per_mod[m] for m in modalities)]
        for k in keys:
            n_crops = min(per_mod[m][k].shape[0] for m in modalities)
            c = int(rng.integers(0, n_crops)) if crop is None else crop % n_crops
            for m in modalities:
                pending[m].append(per_mod[m][k][c])
            if len(pending[modalities[0]]) == batch_size:
                yield {m: np.stack(pending[m]) for m in modalities}
                pending = {m: [] for m in modalities}
    if not drop_last and pending[modalities[0]]:
        yield {m: np.stack(pending[m]) for m in modalities}


def raw_batch_nbytes(batch: Dict[str, np.ndarray]) -> int:
    return sum(a.nbytes for a in batch.values())


@torch.no_grad()
def raw_batch_to_device(batch: Dict[str, np.ndarray], device="cuda", grid: bool = True, pin: bool = False) -> Dict[str, torch.Tensor]:
    """int16 host ids -> int64 device tensors in the shape the embeddings take ((B, side, side) for square token grids when ``grid``):
    2 bytes per token over PCIe, widened on the device by fm_unpack_ids_u16.  The result is what DeviceUnifiedMasking takes for image-like
    modalities."""
    from fourm.hip import _lib as L, ops
    out = {}
    for name, a in batch.items():
        h = torch.from_numpy(np.ascontiguousarray(a))
        if pin:
            h = h.pin_memory()
        d = h.to(device, non_blocking=True)
        t = torch.empty(d.numel(), dtype=torch.int64, device=device)
        L.check(L.unpack_ids_u16(ops._p(d), ops._p(t), d.numel(), ops._stream()))
        B, n = a.shape
        side = int(round(n ** 0.5))
        out[name] = t.view(B, side, side) if grid and side * side == n else t.view(B, n)
    return out
