"""ORACLE (test infrastructure, never shipped): torch-CPU fp32 restatement of the DINO
VisionTransformer as the reference's feature stage invokes it.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product path (``deep-spectral-segmentation_amd``) never does.

The arithmetic of this stage lives in a third-party dependency that is NOT under
/root/reference: ``torch.hub.load('facebookresearch/dino:main', name)``
(reference call site: extract/extract_utils.py:42; branch ``main``, unpinned).  This file
restates the published architecture (``vision_transformer.py`` / ``hubconf.py`` of that
repository, summarised in SURVEY.md Appendix A).  PARITY PIN: no reference test or golden
vector exists for this stage ("parity unpinned" by the reference itself); it is pinned by
(1) an independent implementation - ``transformers.ViTModel`` with the same weights, see
``oracle/make_golden.py::check_vit_against_hf`` - and (2) golden outputs of the
reference's own ``extract_features`` driver run in this container with this model
injected for ``torch.hub.load`` (tests/golden/features_*.npz).

Interface mirrored from the reference's use of the hub model:
  extract/extract_utils.py:45   ``model.patch_embed.patch_size``      (int)
  extract/extract_utils.py:46   ``model.blocks[0].attn.num_heads``    (int)
  extract/extract.py:53         ``model._modules['blocks'][i]._modules['attn']._modules['qkv']``
  extract/extract.py:94         ``model.get_intermediate_layers(images)``
"""
from __future__ import annotations

import math
from typing import Dict, List

import torch
import torch.nn as nn
import torch.nn.functional as F

CONFIGS = {  # hubconf.py factories: vit_small / vit_base, num_classes=0
    "dino_vits16": dict(dim=384, depth=12, heads=6, patch=16),
    "dino_vits8": dict(dim=384, depth=12, heads=6, patch=8),
    "dino_vitb16": dict(dim=768, depth=12, heads=12, patch=16),
    "dino_vitb8": dict(dim=768, depth=12, heads=12, patch=8),
}


class _Attention(nn.Module):
    def __init__(self, dim: int, heads: int):
        super().__init__()
        self.num_heads = heads
        self.scale = (dim // heads) ** -0.5
        self.qkv = nn.Linear(dim, 3 * dim, bias=True)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        b, t, c = x.shape
        qkv = self.qkv(x).reshape(b, t, 3, self.num_heads, c // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = (q @ k.transpose(-2, -1)) * self.scale
        attn = attn.softmax(dim=-1)
        x = (attn @ v).transpose(1, 2).reshape(b, t, c)
        return self.proj(x)


class _Mlp(nn.Module):
    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()  # erf form
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class _Block(nn.Module):
    def __init__(self, dim: int, heads: int):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, 4 * dim)

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        x = x + self.mlp(self.norm2(x))
        return x


class _PatchEmbed(nn.Module):
    def __init__(self, dim: int, patch: int):
        super().__init__()
        self.patch_size = patch
        self.proj = nn.Conv2d(3, dim, kernel_size=patch, stride=patch)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)  # [B, N, D], row-major over (H_p, W_p)


class RefDinoViT(nn.Module):
    """``state_dict``-compatible with facebookresearch/dino ViTs (num_classes=0)."""

    def __init__(self, dim: int, depth: int, heads: int, patch: int):
        super().__init__()
        self.embed_dim = dim
        self.patch_embed = _PatchEmbed(dim, patch)
        n0 = (224 // patch) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, 1 + n0, dim))
        self.blocks = nn.ModuleList([_Block(dim, heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=1e-6)

    def interpolate_pos_encoding(self, x, w, h):
        npatch = x.shape[1] - 1
        n = self.pos_embed.shape[1] - 1
        if npatch == n and w == h:
            return self.pos_embed
        class_pos = self.pos_embed[:, 0]
        patch_pos = self.pos_embed[:, 1:]
        dim = x.shape[-1]
        side = int(math.sqrt(n))
        w0 = w // self.patch_embed.patch_size + 0.1  # +0.1: the published workaround for
        h0 = h // self.patch_embed.patch_size + 0.1  # floor() in interpolate's size math
        patch_pos = F.interpolate(
            patch_pos.reshape(1, side, side, dim).permute(0, 3, 1, 2),
            scale_factor=(w0 / side, h0 / side), mode="bicubic")
        assert int(w0) == patch_pos.shape[-2] and int(h0) == patch_pos.shape[-1]
        patch_pos = patch_pos.permute(0, 2, 3, 1).reshape(1, -1, dim)
        return torch.cat((class_pos.unsqueeze(0), patch_pos), dim=1)

    def prepare_tokens(self, x):
        b, _, w, h = x.shape  # NB: published code names dim2 "w" and dim3 "h"
        x = self.patch_embed(x)
        x = torch.cat((self.cls_token.expand(b, -1, -1), x), dim=1)
        return x + self.interpolate_pos_encoding(x, w, h)

    def get_intermediate_layers(self, x, n: int = 1) -> List[torch.Tensor]:
        x = self.prepare_tokens(x)
        out = []
        for i, blk in enumerate(self.blocks):
            x = blk(x)
            if len(self.blocks) - i <= n:
                out.append(self.norm(x))
        return out

    def forward(self, x):
        x = self.prepare_tokens(x)
        for blk in self.blocks:
            x = blk(x)
        return self.norm(x)[:, 0]


def build_ref_vit(model_name: str, state_dict: Dict[str, torch.Tensor]) -> RefDinoViT:
    cfg = CONFIGS[model_name.lower()]
    model = RefDinoViT(**cfg)
    model.load_state_dict(state_dict, strict=True)
    return model.eval()


@torch.no_grad()
def ref_preprocess(image_u8_hwc) -> torch.Tensor:
    """extract/extract_utils.py:55-56: ToTensor (u8 HWC -> f32 CHW / 255) + ImageNet Normalize."""
    x = torch.as_tensor(image_u8_hwc).permute(2, 0, 1).to(torch.float32).div(255.0)
    mean = torch.tensor((0.485, 0.456, 0.406)).view(3, 1, 1)
    std = torch.tensor((0.229, 0.224, 0.225)).view(3, 1, 1)
    return (x - mean) / std


@torch.no_grad()
def ref_extract_k(model: RefDinoViT, image_chw: torch.Tensor, which_block: int = -1) -> torch.Tensor:
    """Restates extract/extract.py:82-98 for ONE transformed image ``[3, H, W]``:
    crop to a multiple of P (top-left), forward with a hook on ``blocks[which_block].attn.qkv``,
    keep the K third of the qkv output, drop the CLS token.  Returns ``[1, N, D]`` f32."""
    p = model.patch_embed.patch_size
    heads = model.blocks[0].attn.num_heads
    images = image_chw.unsqueeze(0)
    b, c, h, w = images.shape
    h_patch, w_patch = h // p, w // p
    t = h_patch * w_patch + 1
    images = images[:, :, : h_patch * p, : w_patch * p]
    grabbed = {}
    handle = model.blocks[which_block].attn.qkv.register_forward_hook(
        lambda mod, inp, out: grabbed.__setitem__("qkv", out))
    try:
        model.get_intermediate_layers(images)
    finally:
        handle.remove()
    qkv = grabbed["qkv"].reshape(b, t, 3, heads, -1).permute(2, 0, 3, 1, 4)
    return qkv[1].transpose(1, 2).reshape(b, t, -1)[:, 1:, :]
