"""Compact host-to-device batch format: what the loader hands over as padded int64 / bool / fp32 tensors
(fourm/data/unified_datasets.py:488-557, masking.py) crosses PCIe as uint8 pixels, uint16 token ids and bit-packed masks and is
rebuilt into the ``mod_dict`` layout on the device by four small kernels (csrc/masking.hip).  A 4M-B batch of 256 samples is
155 MB of fp32 pixels + 7 MB of ids / masks in upstream's layout and 38.5 MB + 1.2 MB in this one.

``pack_mod_dict`` runs on the host (numpy); ``unpack_mod_dict`` on the device.  Image-like modalities' ``decoder_attention_mask`` is
not shipped when it is the function of ``target_mask`` upstream's ``image_mask`` makes it (masking.py:262-264)."""
from typing import Dict, Optional, Sequence

import numpy as np
import torch


def _pack_bits(mask: torch.Tensor) -> torch.Tensor:
    m = mask.reshape(mask.shape[0], -1).to(torch.bool).cpu().numpy()
    return torch.from_numpy(np.packbits(m, axis=1, bitorder="little"))


def _dam_of_target_mask(tm: torch.Tensor) -> torch.Tensor:
    tm = tm.reshape(tm.shape[0], -1).bool()
    dam = torch.zeros(tm.shape, dtype=torch.int32)
    free = ~tm
    cnt = free.sum(1).to(torch.int32)
    first = torch.where(free.any(1), free.float().argmax(1), torch.zeros(tm.shape[0], dtype=torch.long))
    dam[torch.arange(tm.shape[0]), first] = cnt
    return dam


def pack_mod_dict(mod_dict: Dict[str, Dict[str, torch.Tensor]], images_u8: Optional[Dict[str, torch.Tensor]] = None,
                  pin: bool = False) -> Dict[str, Dict[str, torch.Tensor]]:
    """mod_dict (CPU tensors, loader layout) -> compact dict.  ``images_u8[name]`` = the (B, H, W, C) uint8 pixels of a pixel modality
    (the loader's image BEFORE to_tensor / normalize); a pixel modality without it is shipped as it is."""
    out = {}
    for name, d in mod_dict.items():
        t = d["tensor"]
        e = {"shape": tuple(t.shape), "dtype": t.dtype}
        if images_u8 is not None and name in images_u8:
            img = images_u8[name]
            assert img.dtype == torch.uint8 and img.dim() == 4, "images_u8: (B, H, W, C) uint8"
            e["kind"], e["data"] = "image_u8", img.contiguous()
        elif t.dtype in (torch.int64, torch.int32) and int(t.max()) < 65536 and int(t.min()) >= 0:
            e["kind"], e["data"] = "ids_u16", torch.from_numpy(t.reshape(-1).cpu().numpy().astype(np.uint16).view(np.int16))
        else:
            e["kind"], e["data"] = "raw", t.contiguous()
        for k in ("input_mask", "target_mask"):
            e[k] = _pack_bits(d[k])
            e[k + "_len"] = int(d[k].reshape(d[k].shape[0], -1).shape[1])
            e[k + "_shape"] = tuple(d[k].shape)
        dam = d.get("decoder_attention_mask")
        if dam is not None:
            flat = dam.reshape(dam.shape[0], -1).to(torch.int32)
            e["dam_shape"] = tuple(dam.shape)
            e["dam_dtype"] = dam.dtype
            e["dam"] = None if torch.equal(flat, _dam_of_target_mask(d["target_mask"])) else flat.contiguous()
        if pin:
            for k, v in list(e.items()):
                if torch.is_tensor(v):
                    e[k] = v.pin_memory()
        out[name] = e
    return out


def packed_nbytes(packed) -> int:
    return sum(v.numel() * v.element_size() for e in packed.values() for v in e.values() if torch.is_tensor(v))


@torch.no_grad()
def unpack_mod_dict(packed, device="cuda", image_mean: Sequence[float] = (0.485, 0.456, 0.406),
                    image_std: Sequence[float] = (0.229, 0.224, 0.225)) -> Dict[str, Dict[str, torch.Tensor]]:
    """Compact dict -> ``mod_dict`` on ``device`` (tensors in the loader's dtypes and shapes).  ``image_mean`` / ``image_std``:
    IMAGENET_DEFAULT_MEAN / STD of fourm/utils/data_constants.py, as RGBTransform.rgb_to_tensor applies them."""
    import ctypes as C
    from fourm.hip import _lib as L, ops
    out = {}
    for name, e in packed.items():
        data = e["data"].to(device, non_blocking=True)
        if e["kind"] == "image_u8":
            B, H, W, Cn = data.shape
            t = torch.empty(B, Cn, H, W, dtype=torch.float32, device=device)
            mean, std = (C.c_float * Cn)(*image_mean[:Cn]), (C.c_float * Cn)(*image_std[:Cn])
            L.check(L.unpack_image_u8(ops._p(data), ops._p(t), B, H, W, Cn, mean, std, ops._stream()))
        elif e["kind"] == "ids_u16":
            t64 = torch.empty(data.numel(), dtype=torch.int64, device=device)
            L.check(L.unpack_ids_u16(ops._p(data), ops._p(t64), data.numel(), ops._stream()))
            t = t64.view(e["shape"]) if e["dtype"] == torch.int64 else t64.view(e["shape"]).to(e["dtype"])
        else:
            t = data
        d = {"tensor": t}
        for k in ("input_mask", "target_mask"):
            bits = e[k].to(device, non_blocking=True)
            Bm, Lm = bits.shape[0], e[k + "_len"]
            m = torch.empty(Bm, Lm, dtype=torch.bool, device=device)
            L.check(L.unpack_mask_bits(ops._p(bits), ops._p(m), Bm, Lm, ops._stream()))
            d[k] = m.view(e[k + "_shape"])
        if "dam_shape" in e:
            if e["dam"] is None:
                tm = d["target_mask"].reshape(d["target_mask"].shape[0], -1)
                dam = torch.empty(tm.shape, dtype=torch.int32, device=device)
                L.check(L.decoder_attention_from_target(ops._p(tm), ops._p(dam), tm.shape[0], tm.shape[1], ops._stream()))
            else:
                dam = e["dam"].to(device, non_blocking=True)
            d["decoder_attention_mask"] = dam.view(e["dam_shape"]).to(e["dam_dtype"])
        out[name] = d
    return out
