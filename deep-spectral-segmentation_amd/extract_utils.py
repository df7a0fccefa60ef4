"""Helpers behind the two hot-path commands; the counterpart of the reference's
``extract/extract_utils.py`` for the symbols the hot path uses:

  ImagesDataset (:17-37), get_model (:40-50), get_transform (:53-59), get_image_sizes (:73-79),
  make_output_dir (:98-104), parallel_process (:138-148), get_diagonal (:207-220).

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
        from PIL import Image

        path = self.filenames[index]
        full_path = Path(path) if self.root is None else self.root / path
        assert full_path.is_file(), f"Not a file: {full_path}"
        from PIL import ImageOps

        with Image.open(full_path) as im:
            # cv2.imread (the reference's decoder, extract_utils.py:30) applies the EXIF orientation; PIL does not
            # unless asked to
            image = torch.from_numpy(np.array(ImageOps.exif_transpose(im).convert("RGB"), dtype=np.uint8))
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
              weights: Optional[str] = None, synthetic_seed: Optional[int] = None):
    """Returns ``(model, val_transform, patch_size, num_heads)`` like the reference.  ``weights``: path
    to a DINO checkpoint; otherwise ``$DSS_DINO_WEIGHTS`` / the torch.hub cache are searched.
    ``synthetic_seed`` (or ``$DSS_SYNTHETIC_WEIGHTS``) builds random-init weights instead."""
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
    model = DinoViT(name, sd, device, dtype)
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
