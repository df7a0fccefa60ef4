"""Helpers behind the two hot-path commands; the counterpart of the reference's
``extract/extract_utils.py`` for the symbols the hot path uses:

  ImagesDataset (:17-37), get_model (:40-50), get_transform (:53-59), get_image_sizes (:73-79),
  make_output_dir (:98-104), parallel_process (:138-148), get_diagonal (:207-220),
and for the consumers / options either side of it: get_largest_cc (:107-112),
erode_or_dilate_mask (:115-121), get_border_fraction (:124-135), knn_affinity (:150-189), rw_affinity (:192-204).

Differences that are deliberate and documented in DESIGN.md:
  * images are decoded with PIL (cv2 / torchvision are not part of this stack) and stay uint8 HWC: the
    ToTensor+Normalize transform runs on the GPU (``dss_preprocess_*``);
  * ``get_model`` cannot download from torch.hub (no network): it reads a local DINO checkpoint or, when
    explicitly asked, builds seeded synthetic weights of the same architecture;
  * ``make_output_dir``'s interactive prompt can be pre-answered (``DSS_ASSUME_YES=1``).
"""
from __future__ import annotations

import os
import sys
import time
from pathlib import Path
from typing import Any, Callable, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .synthetic import VIT_CONFIGS, synthetic_state_dict

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)

# file names torch.hub's dino entry points download (hubconf.py of facebookresearch/dino)
_HUB_FILES = {
    "dino_vits16": "dino_deitsmall16_pretrain.pth",
    "dino_vits8": "dino_deitsmall8_pretrain.pth",
    "dino_vitb16": "dino_vitbase16_pretrain.pth",
    "dino_vitb8": "dino_vitbase8_pretrain.pth",
}


class ImagesDataset:
    """Sorted, de-duplicated list of image files; ``__getitem__`` -> (u8 RGB ``[H, W, 3]`` tensor, path,
    index).  Same ordering contract as the reference (it defines ``indices`` and the shard order)."""

    def __init__(self, filenames: Sequence[str], images_root: Optional[str] = None,
                 transform: Optional[Callable] = None, prepare_filenames: bool = True) -> None:
        self.root = None if images_root is None else Path(images_root)
        self.filenames = sorted(set(filenames)) if prepare_filenames else list(filenames)
        self.transform = transform

    def __len__(self) -> int:
        return len(self.filenames)

    def __getitem__(self, index: int) -> Tuple[Any, str, int]:
        from .pthfast import decode_rgb

        path = self.filenames[index]
        full_path = Path(path) if self.root is None else self.root / path
        assert full_path.is_file(), f"Not a file: {full_path}"
        image = torch.from_numpy(decode_rgb(full_path))
        if self.transform is not None:
            image = self.transform(image)
        return image, path, index


def get_transform(name: str) -> Callable[[torch.Tensor], torch.Tensor]:
    """CPU version of the transform (u8 HWC -> normalised f32 CHW).  The product path does not use it -
    the GPU kernel ``dss_preprocess_chw`` is bit-identical - it exists for API parity and tests."""
    if not any(x in name for x in ("dino", "mocov3", "convnext")):
        raise NotImplementedError()
    mean = torch.tensor(IMAGENET_MEAN).view(3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(3, 1, 1)

    def transform(img_u8_hwc):
        x = torch.as_tensor(img_u8_hwc).permute(2, 0, 1).to(torch.float32).div(255)
        return (x - mean) / std

    return transform


def find_weights(name: str) -> Optional[Path]:
    cands: List[Path] = []
    if os.environ.get("DSS_DINO_WEIGHTS"):
        p = Path(os.environ["DSS_DINO_WEIGHTS"])
        cands += [p, p / f"{name}.pth", p / _HUB_FILES.get(name, "")]
    hub = Path(os.environ.get("TORCH_HOME", Path.home() / ".cache" / "torch")) / "hub" / "checkpoints"
    cands.append(hub / _HUB_FILES.get(name, "_"))
    for c in cands:
        if c.is_file():
            return c
    return None


def get_model(name: str, device: Optional[torch.device] = None, dtype: torch.dtype = torch.float16,
              weights: Optional[str] = None, synthetic_seed: Optional[int] = None, gelu: str = "auto"):
    """Returns ``(model, val_transform, patch_size, num_heads)`` like the reference.  ``weights``: path
    to a DINO checkpoint; otherwise ``$DSS_DINO_WEIGHTS`` / the torch.hub cache are searched.
    ``synthetic_seed`` (or ``$DSS_SYNTHETIC_WEIGHTS``) builds random-init weights instead.  ``gelu``: ``DinoViT``'s
    switch - "erf" is DINO's exact GELU in fp32 arithmetic, "erf_f16" the same function as a polynomial form on packed
    f16 in fc1's epilogue (csrc/kres.h; error budget in tests/test_host_logic.py), "auto" (default) = "erf_f16" for the
    D = 384 models and "erf" for D = 768 (vit.py says why)."""
    from .vit import DinoViT, load_dino_state_dict

    name = name.lower()
    if "dino" not in name or name not in VIT_CONFIGS:
        raise ValueError(f"Cannot get model: {name}")
    if synthetic_seed is None and os.environ.get("DSS_SYNTHETIC_WEIGHTS"):
        synthetic_seed = int(os.environ["DSS_SYNTHETIC_WEIGHTS"])
    if weights is not None:
        sd = load_dino_state_dict(weights)
    elif synthetic_seed is not None:
        print(f"[dss] {name}: using SYNTHETIC weights (seed {synthetic_seed}) - no pretrained checkpoint")
        sd = synthetic_state_dict(name, synthetic_seed)
    else:
        found = find_weights(name)
        if found is None:
            raise FileNotFoundError(
                f"No DINO checkpoint for {name}: torch.hub cannot download here.  Pass weights=<path>, set "
                f"DSS_DINO_WEIGHTS, place {_HUB_FILES[name]} in the torch hub cache, or set "
                "DSS_SYNTHETIC_WEIGHTS=<seed> for random-init weights.")
        sd = load_dino_state_dict(str(found))
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    model = DinoViT(name, sd, device, dtype, gelu=gelu)
    # which arithmetic produced the files of this run (a feature directory must be attributable)
    print(f"[dss] {name}: operands {str(dtype)[6:]}, fp32 accumulation / residual stream; GELU '{model.gelu}' = {model.paths()['gelu']}")
    return model, get_transform(name), model.patch_size, model.num_heads


def get_image_sizes(data_dict: dict, downsample_factor: Optional[int] = None):
    p = data_dict["patch_size"] if downsample_factor is None else downsample_factor
    b, c, h, w = data_dict["shape"]
    assert b == 1, "assumption violated :("
    h_patch, w_patch = h // p, w // p
    return (b, c, h, w, p, h_patch, w_patch, h_patch * p, w_patch * p)


def _get_files(p: str):
    if Path(p).is_dir():
        return sorted(Path(p).iterdir())
    if Path(p).is_file():
        return Path(p).read_text().splitlines()
    raise ValueError(p)


def get_paired_input_files(path1: str, path2: str):
    """Pairs the i-th entry of two sorted directories / list files (reference extract_utils.py:82-95)."""
    files1, files2 = _get_files(path1), _get_files(path2)
    assert len(files1) == len(files2)
    return list(enumerate(zip(files1, files2)))


def make_output_dir(output_dir, check_if_empty: bool = True):
    output_dir = Path(output_dir)
    output_dir.mkdir(exist_ok=True, parents=True)
    if check_if_empty and any(output_dir.iterdir()):
        print(f"Output dir: {str(output_dir)}")
        if os.environ.get("DSS_ASSUME_YES", "") not in ("", "0"):
            return
        if input("Output dir already contains files. Continue? (y/n) >> ") != "y":
            sys.exit()


def parallel_process(inputs: Iterable, fn: Callable, multiprocessing: int = 0):
    """Serial driver.  The reference forks ``multiprocessing`` CPU workers around scipy; here the
    parallelism is inside the GPU kernels (one workgroup per image), so the flag is accepted and ignored."""
    start = time.time()
    if multiprocessing:
        print("[dss] multiprocessing flag ignored: images are batched on the GPU")
    for inp in inputs:
        fn(inp)
    print(f"Finished in {time.time() - start:.1f}s")


def get_diagonal(W: torch.Tensor, n: Optional[int] = None, threshold: float = 1e-12) -> torch.Tensor:
    """Degree vector of a DENSE affinity matrix held on the GPU (``[N, N]`` or the ``[ld, ld]`` produced by
    ``hip.affinity_to_dense``): row sums with the reference's clamp.  Diagnostic helper - the eigensolver
    computes the same thing internally from the packed tiles."""
    n = W.shape[0] if n is None else n
    d = W[:n, :n].sum(dim=1)
    d[d < threshold] = 1.0
    return d


# ------------------------------------------------------------------------- consumers of the eigen files
def get_largest_cc(mask: np.ndarray) -> np.ndarray:
    """Largest connected component of a boolean mask (reference extract_utils.py:107-112, ``skimage.measure.label``
    with its default full connectivity - 8-connected in 2-D - restated with ``scipy.ndimage.label``; skimage is not part
    of this stack).  Ties go to the component met first in raster order, as ``argmax(bincount)`` does there."""
    from scipy import ndimage

    mask = np.asarray(mask)
    labels, count = ndimage.label(mask, structure=np.ones((3,) * mask.ndim, dtype=bool))
    if count == 0:   # the reference raises on an empty mask (argmax of an empty sequence)
        raise ValueError("attempt to get argmax of an empty sequence")
    return labels == (np.argmax(np.bincount(labels.ravel())[1:]) + 1)


def erode_or_dilate_mask(x, r: int = 0, erode: bool = True):
    """``r`` rounds of binary erosion (or dilation) that never erase the whole mask (reference extract_utils.py:115-121).
    ``skimage.morphology.binary_erosion / binary_dilation`` with their default footprint (the 4-connected cross) are
    ``scipy.ndimage``'s with ``border_value=True`` for the erosion (pixels outside the image count as set) and ``False``
    for the dilation."""
    from scipy import ndimage

    x = np.asarray(x)
    for _ in range(r):
        x_new = (ndimage.binary_erosion(x, border_value=True) if erode else ndimage.binary_dilation(x))
        if x_new.sum() > 0:  # do not erode the entire mask away
            x = x_new
    return x


def segment_boxes(segmap: np.ndarray, num_erode: int, num_dilate: int, include_background: bool = False):
    """Label map -> ``(labels, boxes)``: for every label present (ascending; 0 only with ``include_background``) its mask is
    eroded ``num_erode`` and dilated ``num_dilate`` times (``erode_or_dilate_mask``: never to nothing) and boxed as
    ``[xmin, ymin, xmax + 1, ymax + 1]`` (reference extract/extract.py:443-458)."""
    labels, boxes = [], []
    for label in np.unique(segmap).tolist():
        if label <= 0 and not include_background:
            continue
        mask = erode_or_dilate_mask(erode_or_dilate_mask(segmap == label, num_erode, erode=True), num_dilate, erode=False)
        cols, rows = np.flatnonzero(mask.any(axis=0)), np.flatnonzero(mask.any(axis=1))
        labels.append(label)
        boxes.append([int(cols[0]), int(rows[0]), int(cols[-1]) + 1, int(rows[-1]) + 1])
    return labels, boxes


def get_border_fraction(segmap: np.ndarray):
    """``(labels, fraction of the 2 (H + W) border pixels carrying each label)`` - corner pixels count twice, labels in
    ascending order (reference extract_utils.py:124-135)."""
    border = np.concatenate([segmap[:, 0], segmap[:, -1], segmap[0, :], segmap[-1, :]])
    indices = np.unique(segmap)
    counts = np.array([(border == i).sum() for i in indices])
    return indices, counts / (2 * (segmap.shape[0] + segmap.shape[1]))


@torch.no_grad()
def knn_affinity(image: torch.Tensor, n_neighbors=(20, 10), distance_weights=(2.0, 0.1)) -> torch.Tensor:
    """KNN colour affinity (reference extract_utils.py:150-189) built ON THE GPU: ``image`` = ``[H, W, 3]`` in [0, 1] on
    the device; returns the DENSE symmetric ``[H W, H W]`` f32 matrix the reference reaches with ``W_lr.todense()``.
    For each (k, distance_weight): every pixel is linked to its k nearest neighbours (itself included, as in pymatting's
    ``knn(f, f, k)``) in the 5-D space ``(r, g, b, dw x, dw y)``, ``x, y`` in [0, 1]; an entry counts how many times the
    pair was linked, from either side, over both scales (``csr_matrix`` sums duplicates; the diagonal is therefore 4).
    pymatting's kd-tree is replaced by exhaustive search over the same float32 points with fp64 distances (what an exact
    kd-tree such as scipy's cKDTree computes).  On 8-bit colours and a regular grid EXACT ties at the k-th place are
    common (5-50 pixels of a 14 x 14 image); a kd-tree resolves them by traversal order, which cannot be reproduced
    without pymatting itself - here the lower pixel index wins (stable sort).  MEASURED: swapping the tied neighbours moves
    the eigenvalues of the fused Laplacian by ~3e-4 - that is the reference's own implementation dependence, not noise of
    this builder (tests/test_consumers.py pins everything else against the reference run with the same tie rule)."""
    h, w = image.shape[:2]
    n = h * w
    dev = image.device
    rgb = image.reshape(n, 3).to(torch.float64)
    x = torch.linspace(0, 1, w, dtype=torch.float64, device=dev).repeat(h)
    y = torch.linspace(0, 1, h, dtype=torch.float64, device=dev).repeat_interleave(w)
    out = torch.zeros((n, n), dtype=torch.float32, device=dev)
    rows = max(1, min(n, (512 << 20) // (8 * n)))   # <= 512 MiB of distances at a time
    for k, dw in zip(n_neighbors, distance_weights):
        if k > n:
            raise ValueError(f"knn_affinity: k={k} neighbours asked of {n} pixels")
        f = torch.cat([rgb, (dw * x)[:, None], (dw * y)[:, None]], dim=1).to(torch.float32).to(torch.float64)
        for s in range(0, n, rows):
            q = f[s:s + rows]
            d2 = torch.zeros((q.shape[0], n), dtype=torch.float64, device=dev)
            for c in range(5):
                d2 += (q[:, c, None] - f[None, :, c]) ** 2
            nb = d2.sort(dim=1, stable=True).indices[:, :k]
            i = torch.arange(s, s + q.shape[0], device=dev)[:, None].expand_as(nb)
            ones = torch.ones(nb.numel(), dtype=torch.float32, device=dev)
            out.index_put_((i.reshape(-1), nb.reshape(-1)), ones, accumulate=True)
            out.index_put_((nb.reshape(-1), i.reshape(-1)), ones, accumulate=True)
    return out


RW_EXPONENT = 900.0   # pymatting's `_rw_laplacian`: `wij = np.exp(-900 * np.linalg.norm(zi - zj) ** 2)` - a constant, not 1 / sigma^2


@torch.no_grad()
def rw_affinity(image: torch.Tensor, sigma: float = 0.033, radius: int = 1) -> torch.Tensor:
    """Random-walk colour affinity (reference extract_utils.py:192-204) on the GPU, dense ``[H W, H W]`` f32.
    The arithmetic is pymatting's ``_rw_laplacian(image, sigma, radius)`` (third party, unpinned in the reference's
    requirements, absent from this image): every pixel is linked to the ``(2 r + 1)^2`` window around it, coordinates
    clamped to the image (so border pixels link to themselves / to a neighbour more than once and the duplicates add up -
    the reference's ``csr_matrix((values, (i, j)))`` sums them), with weight ``exp(-900 |I_i - I_j|^2)``.  As published,
    that function takes ``sigma`` (Grady et al. 2005, eq. 4 would give ``1 / sigma^2 = 918`` at the default 0.033) but
    its body hard-codes the 900: ``sigma`` is accepted and unused, here as there.  Restated from the published source, not
    executed: no pymatting run exists to pin it with."""
    del sigma   # see above: the published implementation ignores it
    h, w = image.shape[:2]
    n = h * w
    dev = image.device
    img = image.to(torch.float64)
    ys, xs = torch.meshgrid(torch.arange(h, device=dev), torch.arange(w, device=dev), indexing="ij")
    i = (xs + ys * w).reshape(-1)
    out = torch.zeros((n, n), dtype=torch.float64, device=dev)
    for dy in range(-radius, radius + 1):
        for dx in range(-radius, radius + 1):
            y2, x2 = (ys + dy).clamp(0, h - 1), (xs + dx).clamp(0, w - 1)
            wij = torch.exp(-RW_EXPONENT * ((img - img[y2, x2]) ** 2).sum(-1)).reshape(-1)
            out.index_put_((i, (x2 + y2 * w).reshape(-1)), wij, accumulate=True)
    return out.to(torch.float32)
