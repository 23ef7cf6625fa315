"""VQ tokenizer front half (ViT-B/16 encoder + cosine-similarity codebook search) — see vqvae.py."""
