from .quantize_lucid import VectorQuantize as VectorQuantizerLucid
