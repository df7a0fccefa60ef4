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
                      strict: bool = True, affinity_mode: str = "fused"
                      ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """``img_u8`` u8 ``[B, H, W, 3]`` on the GPU -> (k ``[B, N, D]``, eigenvalues ``[B, K]``,
    eigenvectors ``[B, K, N]``, info ``[B]``).  ``affinity_mode``: ``spectral.laplacian_eigs_from_features``'s (default
    here ``"fused"``, see below; ``"split"`` / ``"fp32"`` are the A/B arms)."""
    mode = affinity_mode
    k16 = rn = None
    if mode == "fused" and normalize and threshold_at_zero:
        k, k16, rn = model.extract_k_f16(img_u8, which_block=which_block)
    else:
        k = model.extract_k(img_u8, which_block=which_block)
    # the features just came out of the half-precision ViT (relative error ~1e-3): the fused affinity build, which rounds
    # them to f16 (2^-11) on its way into the MFMAs, costs nothing in accuracy here.  Its own tolerance, for callers with
    # EXACT fp32 features: |dW| <= 5e-5, eigenvalues within ~6e-5 of the fp32 build - looser than the 1e-5 eigenvalue bar
    # `extract_eigs` holds on .pth features, which is why that command uses the 'split' build (affinity_mode="split" here);
    # end to end (f16 ViT + this build) the eigenvalues are within 1e-3, the eigenvectors within 1e-6 in cosine of the
    # all-fp32 CPU path (bench.py `parity`, eigenvalue_tol 1e-3).
    ev, vec, info = spectral.laplacian_eigs_from_features(k, K, normalize=normalize,
                                                          threshold_at_zero=threshold_at_zero, strict=strict,
                                                          affinity_mode=mode, feats16=k16, rnorm=rn)
    return k, ev, vec, info


class OverlappedExtractor:
    """Streams the two stages against each other: the ViT of sub-batch i+1 runs on the caller's stream while the
    spectral stage (normalise -> affinity -> Lanczos) of sub-batch i runs on a side stream.  The eigensolver is
    HBM-bound with one workgroup per image, the ViT's attention is MFMA/VALU-bound: they share the chip well.
    Results are identical to ``features_and_eigs`` (same kernels, same order per image)."""

    def __init__(self, model: DinoViT, K: int, vit_batch: int = 128, which_block: int = -1,
                 normalize: bool = True, threshold_at_zero: bool = True, affinity_mode: str = "fused"):
        self.model, self.K, self.vit_batch, self.which_block = model, K, vit_batch, which_block
        self.affinity_mode = affinity_mode
        self.normalize, self.threshold_at_zero = normalize, threshold_at_zero
        self.side = torch.cuda.Stream(device=model.device)

    @torch.no_grad()
    def __call__(self, img_u8: torch.Tensor, keep_features: bool = False):
        main = torch.cuda.current_stream()
        self.side.wait_stream(main)  # side-stream buffers from the previous call are free to be reused
        outs, feats = [], []
        for s in range(0, img_u8.shape[0], self.vit_batch):
            k = self.model.extract_k(img_u8[s:s + self.vit_batch], which_block=self.which_block)
            ready = torch.cuda.Event()
            ready.record(main)
            with torch.cuda.stream(self.side):
                self.side.wait_event(ready)
                k.record_stream(self.side)
                outs.append(spectral.laplacian_eigs_from_features(
                    k, self.K, normalize=self.normalize, threshold_at_zero=self.threshold_at_zero, strict=False,
                    retry=False, affinity_mode=self.affinity_mode))
            if keep_features:
                feats.append(k)
        main.wait_stream(self.side)
        ev = torch.cat([o[0] for o in outs])
        vec = torch.cat([o[1] for o in outs])
        info = torch.cat([o[2] for o in outs])
        for o in outs:
            for t in o:
                t.record_stream(main)
        return (torch.cat(feats) if keep_features else None), ev, vec, info
