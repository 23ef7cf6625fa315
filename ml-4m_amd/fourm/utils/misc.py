import hashlib


def generate_uint15_hash(seed_str: str) -> int:
    """15-bit modality id derived from the modality name (upstream fourm/utils/misc.py:39-41): the ids are
    data — they appear in ``mod_mask`` tensors and in the decoder's modality-separation mask."""
    return int(hashlib.sha256(seed_str.encode("utf-8")).hexdigest(), 16) % (1 << 15)


# names only upstream's same-named module defines (see fourm/_upstream.py)
from fourm import _upstream as _up
_up.merge(__name__, globals())
