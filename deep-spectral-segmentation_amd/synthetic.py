"""Portable synthetic inputs for the extract hot path (no network, no datasets).

Everything here is generated from ``numpy.random.default_rng`` (PCG64), which is
bit-reproducible across machines, so the GPU box, this container and the golden
fixtures all see identical images / weights / features.

* ``synthetic_image``      - "VOC-shaped" u8 RGB image: grey noise + a few coloured blobs.
* ``synthetic_state_dict`` - DINO-ViT ``state_dict`` (keys of facebookresearch/dino) with
                             trunc-normal(0.02) linears, LN gamma=1 beta=0.
* ``synthetic_features``   - ``[N, D]`` feature matrices ("random" / "blobs") used by the
                             eigen-stage fixtures.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import torch

# name -> (embed_dim, depth, heads, patch)
VIT_CONFIGS: Dict[str, Tuple[int, int, int, int]] = {
    "dino_vits16": (384, 12, 6, 16),
    "dino_vits8": (384, 12, 6, 8),
    "dino_vitb16": (768, 12, 12, 16),
    "dino_vitb8": (768, 12, 12, 8),
}


def synthetic_image(index: int, height: int, width: int, seed: int = 1234) -> np.ndarray:
    """u8 ``[H, W, 3]`` RGB image: background noise (sigma ~10 grey levels) plus 2-5
    axis-aligned rectangles / ellipses of distinct mean colour."""
    rng = np.random.default_rng(seed + index)
    base = rng.integers(60, 200, size=3).astype(np.float32)
    img = np.empty((height, width, 3), np.float32)
    img[:] = base
    yy, xx = np.mgrid[0:height, 0:width]
    n_blobs = int(rng.integers(2, 6))
    for _ in range(n_blobs):
        colour = rng.integers(0, 256, size=3).astype(np.float32)
        cy, cx = rng.uniform(0.15, 0.85) * height, rng.uniform(0.15, 0.85) * width
        ry, rx = rng.uniform(0.08, 0.30) * height, rng.uniform(0.08, 0.30) * width
        if rng.random() < 0.5:
            mask = (np.abs(yy - cy) <= ry) & (np.abs(xx - cx) <= rx)
        else:
            mask = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
        img[mask] = colour
    img += rng.normal(0.0, 10.0, size=img.shape).astype(np.float32)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def _trunc_normal(rng: np.random.Generator, shape, std: float = 0.02) -> torch.Tensor:
    x = rng.normal(0.0, std, size=shape)
    x = np.clip(x, -2.0 * std, 2.0 * std)
    return torch.from_numpy(x.astype(np.float32))


def synthetic_state_dict(model_name: str, seed: int = 0, ln_jitter: float = 0.0) -> Dict[str, torch.Tensor]:
    """Random-init DINO ViT weights with the public ``state_dict`` key set
    (SURVEY.md Appendix A).  ``ln_jitter`` > 0 perturbs LayerNorm gamma/beta and the
    linear biases so tests exercise every parameter (real DINO has non-trivial ones)."""
    dim, depth, heads, patch = VIT_CONFIGS[model_name.lower()]
    rng = np.random.default_rng(seed)
    n0 = (224 // patch) ** 2
    sd: Dict[str, torch.Tensor] = {}
    sd["cls_token"] = _trunc_normal(rng, (1, 1, dim))
    sd["pos_embed"] = _trunc_normal(rng, (1, 1 + n0, dim))
    sd["patch_embed.proj.weight"] = _trunc_normal(rng, (dim, 3, patch, patch))
    sd["patch_embed.proj.bias"] = _trunc_normal(rng, (dim,)) * (1.0 if ln_jitter > 0 else 0.0)

    def ln(prefix: str):
        g = torch.ones(dim)
        b = torch.zeros(dim)
        if ln_jitter > 0:
            g = g + torch.from_numpy(rng.normal(0, ln_jitter, dim).astype(np.float32))
            b = b + torch.from_numpy(rng.normal(0, ln_jitter, dim).astype(np.float32))
        sd[prefix + ".weight"], sd[prefix + ".bias"] = g, b

    def lin(prefix: str, out_f: int, in_f: int):
        sd[prefix + ".weight"] = _trunc_normal(rng, (out_f, in_f))
        bias = torch.zeros(out_f)
        if ln_jitter > 0:
            bias = torch.from_numpy(rng.normal(0, ln_jitter, out_f).astype(np.float32))
        sd[prefix + ".bias"] = bias

    for i in range(depth):
        p = f"blocks.{i}"
        ln(p + ".norm1")
        lin(p + ".attn.qkv", 3 * dim, dim)
        lin(p + ".attn.proj", dim, dim)
        ln(p + ".norm2")
        lin(p + ".mlp.fc1", 4 * dim, dim)
        lin(p + ".mlp.fc2", dim, 4 * dim)
    ln("norm")
    return sd


def dino_like_state_dict(model_name: str, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Random-init weights reshaped to stress what REAL DINO checkpoints stress and plain ``trunc_normal(0.02)`` does not
    (no checkpoint can be downloaded here): three residual-stream channels that carry activations of ~150-300 from
    block 2 on (DINO's "massive activations": large fc2 biases on fixed channels), LayerNorm gains that ignore those
    channels and re-inflate the rest by 20x (what a trained network does about them - without it the outliers crush every
    other channel and all patches get the same feature), sharpened attention logits (q/k rows x3: peaked softmax rows
    next to flat ones) and wide GELU inputs (fc1 x4: the erf tails).  Same key set as ``synthetic_state_dict``."""
    dim, depth, heads, patch = VIT_CONFIGS[model_name.lower()]
    sd = synthetic_state_dict(model_name, seed, ln_jitter=0.05)
    rng = np.random.default_rng(seed + 4242)
    chans = rng.choice(dim, size=3, replace=False)
    bumps = (150.0, -125.0, 300.0)
    for i, blk in enumerate((2, 3, 4)):
        sd[f"blocks.{blk}.mlp.fc2.bias"][chans[i]] += bumps[i]          # residual outliers from here on
    for i in range(3, depth):
        for nm in ("norm1", "norm2"):
            sd[f"blocks.{i}.{nm}.weight"] *= 20.0
            sd[f"blocks.{i}.{nm}.weight"][chans] = 0.0
            sd[f"blocks.{i}.{nm}.bias"][chans] = 0.0
    for i in (1, 5, 9):   # logits x9 at D = 384 (their spread grows with D for random weights: same spread at D = 768):
        sd[f"blocks.{i}.attn.qkv.weight"][: 2 * dim] *= 3.0 * (384.0 / dim) ** 0.5      # peaked attention rows
    for i in (3, 8):
        sd[f"blocks.{i}.mlp.fc1.weight"] *= 4.0                          # GELU inputs out to the erf tails
        sd[f"blocks.{i}.mlp.fc2.weight"] *= 0.25
    return sd


def synthetic_features(kind: str, n: int, d: int, seed: int, hw: Tuple[int, int] | None = None) -> np.ndarray:
    """f32 ``[n, d]`` features.

    ``random``: i.i.d. normal (worst case: clustered spectrum, gaps ~1e-3).
    ``blobs`` : patch grid ``hw`` partitioned into a few regions, each with its own mean
                direction plus noise (structured spectrum, like real DINO keys)."""
    rng = np.random.default_rng(seed)
    if kind == "random":
        return rng.normal(0.0, 1.0, size=(n, d)).astype(np.float32)
    if kind == "blobs":
        if hw is None:
            side = int(math.isqrt(n))
            hw = (side, n // side)
        h, w = hw
        assert h * w == n, (h, w, n)
        n_regions = int(rng.integers(3, 7))
        centres = rng.normal(0.0, 1.0, size=(n_regions, d))
        yy, xx = np.mgrid[0:h, 0:w]
        label = np.zeros((h, w), np.int64)
        for r in range(1, n_regions):
            cy, cx = rng.uniform(0.1, 0.9) * h, rng.uniform(0.1, 0.9) * w
            ry, rx = rng.uniform(0.1, 0.35) * h, rng.uniform(0.1, 0.35) * w
            label[((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0] = r
        feats = centres[label.reshape(-1)] + rng.normal(0.0, 0.8, size=(n, d))
        return feats.astype(np.float32)
    raise ValueError(kind)
