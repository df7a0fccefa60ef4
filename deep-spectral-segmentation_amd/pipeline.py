"""End-to-end hot path for images already resident in HBM: u8 images -> K features -> eigenpairs.

This is ``extract_features`` followed by ``extract_eigs`` without the ``.pth`` round trip through disk
(the CLI in ``extract.py`` keeps the reference's two-stage file layout; ``bench.py`` and ``smoke()`` time
this in-memory form)."""
from __future__ import annotations

from typing import Tuple

import torch

from . import spectral
from .vit import DinoViT


@torch.no_grad()
def features_and_eigs(model: DinoViT, img_u8: torch.Tensor, K: int, which_block: int = -1,
                      normalize: bool = True, threshold_at_zero: bool = True,
                      strict: bool = True) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """``img_u8`` u8 ``[B, H, W, 3]`` on the GPU -> (k ``[B, N, D]``, eigenvalues ``[B, K]``,
    eigenvectors ``[B, K, N]``, info ``[B]``)."""
    k = model.extract_k(img_u8, which_block=which_block)
    ev, vec, info = spectral.laplacian_eigs_from_features(k, K, normalize=normalize,
                                                          threshold_at_zero=threshold_at_zero, strict=strict)
    return k, ev, vec, info
