"""Conditional UNet of the diffusion detokenizers (upstream ``fourm/vq/models/unet``)."""
from .unet import PatchedUNetCondCat, unet_patched  # noqa: F401
