"""Localization consumer of the eigen files: the eigensegment -> bounding box step of the reference's
``object-localization/object_discovery.py`` (:16-82 ``get_eigenvectors_from_features``, :85-126
``get_bbox_from_patch_mask``, :280-287 ``get_largest_cc_box``) and the IoU its evaluation uses (``datasets.py:269-294``),
as called from ``object-localization/main.py:254-272`` (precomputed eigenvectors, the README recipe) and :355-364
(eigenvectors computed inline from the ViT features).

The boxes are integer bookkeeping on a <= 60 x 60 mask and stay on the host; the inline eigen decomposition runs on the
GPU through the same Lanczos kernel as ``extract_eigs`` (``spectral.laplacian_eigs_from_features``).  LOST's own seed
expansion (``lost``, ``detect_box``), the DINO-attention baseline, dataset classes and visualisations are out of scope.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import spectral


def get_eigenvectors_from_features(feats: torch.Tensor, which_matrix: str = "laplacian", K: int = 2) -> torch.Tensor:
    """``feats``: f32 ``[N, D]`` (or ``[1, N, D]``) on the GPU, NOT normalised (the reference does not normalise here).
    Returns eigenvectors ``[K, N]`` f32 on the GPU: ``'affinity'`` - the K largest-magnitude eigenpairs of ``F F^T`` in
    descending order (object_discovery.py:25-28); ``'laplacian'`` - the K smallest of ``(D - W) v = lambda D v`` with
    ``W = relu(F F^T) / max`` (:31-41).  ARPACK's eigenvector signs are arbitrary there; here extract.py's sign rule is
    applied, and ``get_bbox_from_patch_mask`` flips a majority mask either way.  ``'affinity_torch'`` (the reference's default)
    calls ``torch.eig``, which no longer exists; ``'matting_laplacian'`` raises in the reference too."""
    if which_matrix == "affinity_torch":
        raise NotImplementedError("which_matrix='affinity_torch' needs torch.eig, removed from PyTorch: use 'affinity'")
    if which_matrix not in ("affinity", "laplacian"):
        raise NotImplementedError(which_matrix)
    f = feats.squeeze(0) if feats.dim() == 3 else feats
    f = f.to(torch.float32).contiguous()
    if which_matrix == "affinity":
        _, vec, _ = spectral.laplacian_eigs_from_features(f[None], K, normalize=False, threshold_at_zero=False,
                                                          problem="affinity", strict=False)
        return vec[0]
    _, vec, _ = spectral.laplacian_eigs_from_features(f[None], K, normalize=False, threshold_at_zero=True,
                                                      problem="laplacian", strict=False)
    return vec[0]


def get_largest_cc_box(mask: np.ndarray):
    """``[xmin, ymin, xmax, ymax]`` (exclusive maxima, mask units) of the largest 8-connected component
    (object_discovery.py:280-287; skimage's ``label`` restated with scipy.ndimage, see extract_utils.get_largest_cc)."""
    from .extract_utils import get_largest_cc

    ys, xs = np.where(get_largest_cc(mask))
    return [int(xs.min()), int(ys.min()), int(xs.max()) + 1, int(ys.max()) + 1]


# How a mask of T elements maps onto the padded image (object_discovery.py:88-99), in the reference's order of preference:
# (patch stride of the ViT grid the mask derives from, upsampling factor of the mask over that grid) -> the box scale is
# stride / upsample pixels per mask cell.  Patch 8; patch 16; patch 16 upsampled 2x; patch 32 upsampled 4x.
_MASK_GRIDS = ((8, 1), (16, 1), (16, 2), (32, 4))


def _mask_grid(height: int, width: int, cells: int):
    """(pixels per mask cell, mask rows, mask columns) of the first entry of ``_MASK_GRIDS`` whose grid has ``cells`` cells."""
    for stride, up in _MASK_GRIDS:
        rows, cols = up * (height // stride), up * (width // stride)
        if rows * cols == cells:
            return stride // up, rows, cols
    return None


def get_bbox_from_patch_mask(patch_mask, init_image_size, img_np: Optional[np.ndarray] = None) -> np.ndarray:
    """Boolean patch mask (``eigenvectors[1] > 0``, any shape with ``T`` elements, tensor or array) + the padded image
    size ``(C, H, W)`` -> pixel box ``[xmin, ymin, xmax, ymax]`` (object_discovery.py:85-126): the mask grid is inferred
    from ``T`` (``_MASK_GRIDS``), a mask that covers more than half of the grid - or nothing - is inverted, and the box of
    its largest connected component is scaled to pixels and clipped to the image.  A ``T`` that fits no grid raises the
    reference's ``ValueError`` (same message)."""
    height, width = (int(v) for v in init_image_size[1:])
    mask = patch_mask.cpu().numpy() if isinstance(patch_mask, torch.Tensor) else np.asarray(patch_mask)
    grid = _mask_grid(height, width, mask.size)
    if grid is None:
        raise ValueError(f"{init_image_size=}, {patch_mask.shape=}")
    scale, rows, cols = grid
    mask = mask.reshape(rows, cols)
    covered = float(np.mean(mask))
    if 0.5 < covered < 1.0 or not mask.any():
        mask = (1 - mask).astype(np.uint8)      # reversed segment / nothing detected: box the complement
    box = np.asarray(get_largest_cc_box(mask)) * scale
    return np.minimum(box, [box[0], box[1], width, height])          # (a padded image: the box ends at the image)


def bbox_iou(box1: torch.Tensor, box2: torch.Tensor, eps: float = 1e-7) -> torch.Tensor:
    """IoU of one ``(x1, y1, x2, y2)`` box against ``[n, 4]`` boxes with the reference's epsilons (datasets.py:269-294,
    plain-IoU branch): what ``main.py:391-394`` thresholds at 0.5 for CorLoc."""
    box2 = box2.T
    inter = (torch.min(box1[2], box2[2]) - torch.max(box1[0], box2[0])).clamp(0) * \
            (torch.min(box1[3], box2[3]) - torch.max(box1[1], box2[1])).clamp(0)
    w1, h1 = box1[2] - box1[0], box1[3] - box1[1] + eps
    w2, h2 = box2[2] - box2[0], box2[3] - box2[1] + eps
    return inter / (w1 * h1 + w2 * h2 - inter + eps)
