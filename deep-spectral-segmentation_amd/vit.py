"""DINO ViT forward for the feature stage, MI355X-native.

What the reference runs (extract/extract.py:51-53,94-98): ``model.get_intermediate_layers(images)``
on the torch.hub DINO ViT with a forward hook on ``blocks[which_block].attn.qkv``; only the K third of
that one Linear's output is kept.  This module computes exactly that tensor and nothing after it:

* blocks ``0 .. which_block-1`` run in full; block ``which_block`` runs ``norm1`` and the K rows of its
  qkv Linear only (its attention/proj/MLP and the final norm never influence the hooked tensor;
  SURVEY.md §0.4) - identical output, ~8 % fewer FLOPs for the default ``which_block=-1``.
* LayerNorm (+ the preceding residual add), attention and - for the D = 384 models - the qkv / proj /
  fc1+GELU Linear layers are the hand-written HIP kernels of ``libdss_hip.so`` (the latter exchange
  activations in the DSS_PLANAR64 layout); the remaining Linear layers (fc2, patch embedding, the last
  block's K projection, everything at D = 768) are PyTorch-ROCm GEMMs (hipBLASLt).  fp16/bf16 operands,
  fp32 accumulation everywhere; the residual stream, LayerNorm statistics, softmax statistics and the
  final K projection stay fp32.
* the image transform + crop + im2col is one HIP kernel, so the patch embedding is a plain GEMM.

``state_dict`` keys are those of facebookresearch/dino, so a real DINO checkpoint
loads unchanged.
"""
from __future__ import annotations

import math
from pathlib import Path
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from . import hip
from .synthetic import VIT_CONFIGS

LN_EPS = 1e-6
_TUNING_FILE = Path(__file__).resolve().parent / "tuning" / "tunableop_gfx950.csv"
_gemm_tuning_ready = False


def setup_gemm_tuning(tune_new_shapes: bool = False, use_table: bool = True) -> None:
    """The Linear layers not covered by the K-resident kernels are hipBLASLt GEMMs issued through PyTorch.  PyTorch's TunableOp picks, per GEMM
    shape, the fastest hipBLASLt/rocBLAS solution; the table measured on MI355X for the bench shapes ships in
    ``tuning/tunableop_gfx950.csv`` (+5 % end to end over the default heuristic).  Shapes not in the table use
    the default heuristic unless ``tune_new_shapes`` asks for on-line tuning (~3 s per new shape, once).
    ``use_table=False`` leaves TunableOp alone altogether (scripts/tune_gemm.sh drives it through PyTorch's own
    ``PYTORCH_TUNABLEOP_*`` variables to PRODUCE the table)."""
    global _gemm_tuning_ready
    if not use_table:
        return
    tun = torch.cuda.tunable
    if not _gemm_tuning_ready:
        tun.enable(True)
        if hasattr(tun, "write_file_on_exit"):
            tun.write_file_on_exit(False)  # never rewrite the shipped table behind the user's back
        tun.set_max_tuning_duration(15)
        if _TUNING_FILE.is_file():
            try:
                tun.read_file(str(_TUNING_FILE))
            except Exception as e:  # a table from another ROCm build is ignored, not fatal
                print(f"[dss] ignoring GEMM tuning table {_TUNING_FILE.name}: {e}")
        _gemm_tuning_ready = True
    tun.tuning_enable(bool(tune_new_shapes))


def interpolate_pos_encoding(pos_embed: torch.Tensor, patch: int, h: int, w: int) -> torch.Tensor:
    """DINO's ``interpolate_pos_encoding`` for an ``h x w`` (pixels, multiples of ``patch``) input:
    bicubic resize of the ``sqrt(N0) x sqrt(N0)`` grid with the published ``+0.1`` scale-factor trick.
    Evaluated once per input shape on the CPU in fp32 (cached by the caller)."""
    pos_embed = pos_embed.detach().float().cpu()
    n0 = pos_embed.shape[1] - 1
    hp, wp = h // patch, w // patch
    if hp * wp == n0 and h == w:
        return pos_embed.clone()
    dim = pos_embed.shape[-1]
    side = int(math.sqrt(n0))
    grid = pos_embed[:, 1:].reshape(1, side, side, dim).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, scale_factor=((hp + 0.1) / side, (wp + 0.1) / side), mode="bicubic")
    if (grid.shape[-2], grid.shape[-1]) != (hp, wp):
        raise RuntimeError(f"pos-embed interpolation produced {tuple(grid.shape[-2:])}, wanted {(hp, wp)}")
    grid = grid.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat((pos_embed[:, :1], grid), dim=1)


class DinoViT:
    """Inference-only DINO ViT holding its weights on one GPU."""

    def __init__(self, model_name: str, state_dict: Dict[str, torch.Tensor], device: torch.device,
                 dtype: torch.dtype = torch.float16, k_proj_fp32: bool = False, gelu: str = "auto",
                 linear_kres: int = 2, fuse_ln: bool = True, gemm_tuning: str = "table", fuse_k: bool = True, fuse_pe: bool = True,
                 fuse_qkv768: bool = True, library_gemm: str = "lt", fc2_into_stream: bool = False):
        name = model_name.lower()
        if name not in VIT_CONFIGS:
            raise ValueError(f"Cannot get model: {model_name}")
        if dtype not in (torch.float16, torch.bfloat16):
            raise ValueError("compute dtype must be torch.float16 or torch.bfloat16")
        self.model_name = name
        self.embed_dim, self.depth, self.num_heads, self.patch_size = VIT_CONFIGS[name]
        self.device, self.dtype = torch.device(device), dtype
        self.k_proj_fp32 = k_proj_fp32
        if gelu not in ("auto", "erf", "erf_f16", "tanh_fused"):
            raise ValueError("gelu must be 'auto', 'erf' (DINO's exact GELU, fp32 arithmetic), 'erf_f16' (the same function evaluated on "
                             "packed f16: f16 operands and the K-resident fc1 kernel only) or 'tanh_fused'")
        # 'auto' (default, round 6): 'erf_f16' for the D = 384 models, 'erf' for D = 768.  The end-to-end gate
        # (tests/test_gpu_e2e.py::test_gelu_f16_form_against_the_exact_form_end_to_end: both forms against the fp32 CPU reference on DINO-like
        # weights) has the packed form 7 % further from that reference than the exact one at D = 384 (5.7e-4 / 5.3e-4 in the features,
        # eigenvectors 6e-7: nothing) - but at D = 768 the f16 path as a whole is close to the 1e-4 eigenvector bar on such weights
        # (exact form: 9.4e-5 in the worst edge cluster of a 224 x 160 image) and the packed form crosses it (1.5e-4): not the default there
        if gelu == "auto":
            gelu = "erf_f16" if self.embed_dim == 384 else "erf"
        if gelu == "erf_f16" and (dtype != torch.float16 or linear_kres < 2 or self.embed_dim not in hip.LINEAR_KRES_WIDTHS):
            gelu = "erf"        # no packed-f16 epilogue on this path: the fp32 form
        # 'tanh_fused': fc1 + bias + GELU in ONE hipBLASLt launch through the library's epilogue, which implements
        # the TANH approximation (measured: 3.6e-7 from tanh-GELU, 4.7e-4 from erf-GELU).  It is NOT DINO's function:
        # opt-in only, never used for the reported numbers.
        # 'erf_f16' (round 5): erf-GELU as a degree-6 polynomial form on v_pk_*_f16 in fc1's epilogue (csrc/kres.h: 5.5
        # instructions per value instead of 17.5, co-issues with the other wave's MFMAs); its error budget against the exact
        # function over every f16 input is a CPU test (tests/test_host_logic.py::test_gelu_f16_poly_error_budget)
        self.gelu = gelu
        self._gelu_code = 2 if gelu == "erf_f16" else 1
        # linear_kres: which Linear layers run on the K-resident kernel (dss_linear_k384 / _k768, planar outputs that the
        # attention and LayerNorm kernels read in place).  0: none (library GEMMs: the A/B arm); 1: qkv + proj of the D = 384
        # models; 2 (default): also fc1 with the erf-GELU fused into its epilogue, for D = 384 and D = 768.
        # fuse_ln (default): the residual add + LayerNorm in front of those Linear layers is their A prologue
        # (dss_lnlinear_*: norm1 -> qkv at D = 384, norm2 -> fc1+GELU at D = 384 / 768) instead of a pass of its own.
        if linear_kres not in (0, 1, 2):
            raise ValueError("linear_kres must be 0, 1 or 2")
        self.linear_k384 = int(linear_kres)
        self.fuse_ln = bool(fuse_ln)
        # fuse_k (default, with fuse_ln): the hooked block's norm1 -> K projection -> CLS drop / f16 copy /
        # inverse norms is ONE kernel (dss_lnlinear_kfeatures) on the pipeline's path (`extract_k_f16`) instead of
        # LayerNorm + library GEMM + dss_kfeatures_finalize
        self.fuse_k = bool(fuse_k)
        # fuse_qkv768 (default; D = 768 models, with fuse_ln): norm1 -> qkv as ONE dss_lnlinear_k768 launch instead of the standalone
        # LayerNorm + the library GEMM.  The one-tile K = 768 kernel is slower than hipBLASLt on the GEMM alone (740-771 vs
        # 888-932 TF/s) but the pair also moves the normalised activations through HBM twice: end to end at C3 the two are equal
        # within the box noise (822 -> 828 and 840 -> 844 images/s, same box each), and no standalone LayerNorm launch is left
        self.fuse_qkv768 = bool(fuse_qkv768)
        # library_gemm: who issues the Linear layers that are not hand-written kernels.  "lt" (default, round 6): dss_linear_lt - hipBLASLt
        # with its Stream-K split of the last round of tiles switched off AND verified off per problem (that split is not
        # reproducible on this stack: profiles/r06_forward_stress.txt).  "torch": F.linear - rounds 1-5, kept as the A/B arm (the
        # package import sets the same Tensile switch for the process, but nothing verifies it on this route)
        if library_gemm not in ("lt", "torch"):
            raise ValueError("library_gemm must be 'lt' (dss_linear_lt) or 'torch' (F.linear)")
        self.library_gemm = library_gemm
        # fc2_into_stream (opt-in, needs library_gemm = "lt"; round 6): `x = x + mlp(...)` inside fc2's own epilogue - the GEMM adds its fp32
        # accumulators (+ fp32 bias) to the fp32 residual stream in place (dss_linear_lt_accumulate: hipBLASLt's beta = 1 with C = D = x).
        # The branch output is then never rounded to the operand type, and the norm1 -> qkv kernel of the NEXT block reads a finished
        # stream (no pending branch output to add, no x write-back: 10 instead of 16 bytes per element through the most HBM-bound kernel
        # of the forward).  Measured (profiles/r06_ab_same_box.txt): the bytes only MOVE - norm1 -> qkv 3.04 -> 2.55 ms per launch, fc2
        # 2.70 -> 3.19 ms (its fp32 read-modify-write epilogue) - +0.1 ... 0.4 % end to end at C2 and C3: not the default; kept as a mode
        # because it is the more accurate form (one rounding fewer per block).
        self.fc2_into_stream = bool(fc2_into_stream) and library_gemm == "lt"
        d = self.embed_dim
        sd = state_dict
        need = ["cls_token", "pos_embed", "patch_embed.proj.weight", "patch_embed.proj.bias"]
        for i in range(self.depth):
            for s in ("norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight",
                      "attn.proj.bias", "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias",
                      "mlp.fc2.weight", "mlp.fc2.bias"):
                need.append(f"blocks.{i}.{s}")
        missing = [k for k in need if k not in sd]
        if missing:
            raise KeyError(f"state_dict is missing {len(missing)} DINO ViT keys, e.g. {missing[:3]}")

        def lp(t):  # GEMM operand
            return t.detach().to(self.device, dtype).contiguous()

        def f32(t):
            return t.detach().to(self.device, torch.float32).contiguous()

        self.cls_token = sd["cls_token"].detach().float().cpu()
        self.pos_embed = sd["pos_embed"].detach().float().cpu()
        self.pe_w = lp(sd["patch_embed.proj.weight"].reshape(d, -1))  # [D, 3*P*P], (c, py, px) inner order
        self.pe_b = lp(sd["patch_embed.proj.bias"])
        # fuse_ln builds at patch size 16: transform + patch embedding + position embedding as one kernel from the u8 image
        self.pe16 = None
        if self.patch_size == 16 and fuse_pe and fuse_ln and linear_kres and d % 64 == 0:
            self.pe16 = hip.patch_embed16_prepare(sd["patch_embed.proj.weight"].to(self.device), sd["patch_embed.proj.bias"].to(self.device), dtype)
        self.blocks = []
        for i in range(self.depth):
            p = f"blocks.{i}."
            self.blocks.append(dict(
                n1w=f32(sd[p + "norm1.weight"]), n1b=f32(sd[p + "norm1.bias"]),
                qkv_w=lp(sd[p + "attn.qkv.weight"]), qkv_b=lp(sd[p + "attn.qkv.bias"]),
                k_w32=f32(sd[p + "attn.qkv.weight"][d:2 * d]), k_b32=f32(sd[p + "attn.qkv.bias"][d:2 * d]),
                k_w=lp(sd[p + "attn.qkv.weight"][d:2 * d]),
                proj_w=lp(sd[p + "attn.proj.weight"]), proj_b=lp(sd[p + "attn.proj.bias"]),
                n2w=f32(sd[p + "norm2.weight"]), n2b=f32(sd[p + "norm2.bias"]),
                fc1_w=lp(sd[p + "mlp.fc1.weight"]), fc1_b=lp(sd[p + "mlp.fc1.bias"]),
                fc2_w=lp(sd[p + "mlp.fc2.weight"]), fc2_b=lp(sd[p + "mlp.fc2.bias"]), fc2_b32=f32(sd[p + "mlp.fc2.bias"]),
            ))
        hip.load_library()  # fail now, not mid-run, if the kernels are missing
        if self.fuse_ln and self.linear_k384 and d in hip.LINEAR_KRES_WIDTHS:
            for i, blk in enumerate(self.blocks):   # LayerNorm affine folded into the Linear behind it, once per layer
                p = f"blocks.{i}."
                if d == 384 or (d == 768 and self.fuse_qkv768):
                    blk["qkv_wg"], blk["qkv_aux"] = hip.lnlinear_prepare(f32(sd[p + "attn.qkv.weight"]), f32(sd[p + "attn.qkv.bias"]),
                                                                         blk["n1w"], blk["n1b"], dtype)
                if self.fuse_k:      # (D = 384 / 768, f16 / bf16: dss_lnlinear_kfeatures; round 4: D = 384 and f16 only)
                    blk["k_wg"], blk["k_aux"] = hip.lnlinear_prepare(blk["k_w32"], blk["k_b32"], blk["n1w"], blk["n1b"], dtype)
                if self.linear_k384 >= 2 and self.gelu in ("erf", "erf_f16"):
                    blk["fc1_wg"], blk["fc1_aux"] = hip.lnlinear_prepare(f32(sd[p + "mlp.fc1.weight"]), f32(sd[p + "mlp.fc1.bias"]),
                                                                         blk["n2w"], blk["n2b"], dtype)
        # final LayerNorm: only the CLS-token path (`forward_cls`, extract_bbox_features) needs it
        self.norm_w = f32(sd["norm.weight"]) if "norm.weight" in sd else None
        self.norm_b = f32(sd["norm.bias"]) if "norm.bias" in sd else None
        self.scale = 64 ** -0.5
        assert d // self.num_heads == 64, "DINO ViTs use 64-dim heads"
        self._pos_cache: Dict[tuple, object] = {}   # (h, w) -> (cls row, pos rows); ("pe16", h, w) -> pos rows + folded bias
        if gemm_tuning not in ("table", "online", "off"):
            raise ValueError("gemm_tuning must be 'table' (shipped TunableOp table), 'online' (also tune new shapes) or 'off'")
        setup_gemm_tuning(tune_new_shapes=gemm_tuning == "online", use_table=gemm_tuning != "off")

    def _linear(self, x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], what: str, out_dtype=None) -> torch.Tensor:
        """A Linear layer that is not a hand-written kernel (see `library_gemm`)."""
        if self.library_gemm == "lt":
            return hip.linear_lt(x, w, b, out_dtype=out_dtype, what=what)
        k = x.shape[-1]
        with hip._timed("library_gemm", m=x.numel() // k, n=w.shape[0], k=k, what=what):
            if out_dtype is not None and out_dtype != x.dtype:
                out = torch.mm(x.reshape(-1, k), w.t(), out_dtype=out_dtype).view(*x.shape[:-1], w.shape[0])
                return out if b is None else out + b.to(out_dtype)
            return F.linear(x, w, b)

    def paths(self) -> Dict[str, str]:
        """Which implementation each layer of a block takes in THIS model (what bench.py reports as `vit_paths`: the
        constructor's switches only apply where a kernel exists for the width / patch size / dtype)."""
        blk, d = self.blocks[0], self.embed_dim
        k384 = bool(self.linear_k384) and d == 384
        kres_fc1 = self.gelu in ("erf", "erf_f16") and self.linear_k384 >= 2 and d in hip.LINEAR_KRES_WIDTHS
        lib = ("library GEMM (hipBLASLt through dss_linear_lt: no Stream-K split, verified)" if self.library_gemm == "lt"
               else "library GEMM (hipBLASLt through F.linear)")
        return {
            "patch_embed": "dss_patch_embed_p16 (transform + GEMM + position rows, one kernel)" if self.pe16 is not None
                           else f"dss_preprocess_patchify + {lib} + add",
            "norm1+qkv": f"dss_lnlinear_k{d}" if ("qkv_wg" in blk and self.linear_k384) else
                         ("dss_layernorm_fwd + dss_linear_k384" if k384 else f"dss_layernorm_fwd + {lib}"),
            "attention": "dss_attention_fwd",
            "proj": "dss_linear_k384" if k384 else lib,
            "norm2+fc1+gelu": f"dss_lnlinear_k{d}" if (kres_fc1 and "fc1_wg" in blk) else
                              (f"dss_layernorm_fwd + dss_linear_k{d}" if kres_fc1 else f"dss_layernorm_fwd + {lib} + GELU pass"),
            "gelu": {"erf": "exact-erf form in fp32 (A&S 7.1.28, |err| <= 3e-7)", "erf_f16": "erf-GELU polynomial form on packed f16 (csrc/kres.h)",
                     "tanh_fused": "hipBLASLt's tanh epilogue (NOT the reference function)"}[self.gelu],
            "fc2": lib + (" adding into the fp32 residual stream (dss_linear_lt_accumulate)" if self.fc2_into_stream else ""),
            "hooked norm1 + K projection + hand-over": "dss_lnlinear_kfeatures" if "k_wg" in self.blocks[-1] else
                                                       f"dss_layernorm_fwd + {lib} + dss_kfeatures_finalize",
        }

    # ------------------------------------------------------------------------------------------
    def _pos(self, h: int, w: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """(cls_token + pos[0]) as ``[D]`` and pos[1:] as ``[N, D]`` on the device, fp32."""
        key = (h, w)
        if key not in self._pos_cache:
            pe = interpolate_pos_encoding(self.pos_embed, self.patch_size, h, w)
            cls_row = (self.cls_token[0, 0] + pe[0, 0]).to(self.device)
            self._pos_cache[key] = (cls_row.contiguous(), pe[0, 1:].to(self.device).contiguous())
        return self._pos_cache[key]

    @torch.no_grad()
    def _run_blocks(self, img_u8: torch.Tensor, nblocks: int):
        """Transform + patch embedding + position encoding, then blocks ``0 .. nblocks-1`` in full.  Returns the fp32
        residual stream ``x [B, T, D]`` and the last branch output not yet added to it (``pending``, fused into the
        next LayerNorm by the caller; ``None`` if ``nblocks == 0``)."""
        b, h, w, _ = img_u8.shape
        p, d, heads = self.patch_size, self.embed_dim, self.num_heads
        hp, wp = h // p, w // p
        t = hp * wp + 1
        cls_row, pos = self._pos(hp * p, wp * p)
        x = torch.empty((b, t, d), dtype=torch.float32, device=self.device)  # fp32 residual stream
        x[:, 0] = cls_row
        if self.pe16 is not None:
            key = ("pe16", hp * p, wp * p)
            if key not in self._pos_cache:   # position embedding + the folded bias, both fp32: the kernel adds them in its epilogue
                self._pos_cache[key] = (pos + self.pe16[1]).contiguous()
            hip.patch_embed16(img_u8.contiguous(), self.pe16[0], None, self._pos_cache[key], x)
        else:
            patches = hip.preprocess_patchify(img_u8.contiguous(), p, self.dtype)  # [B, N, 3PP]
            tok = self._linear(patches, self.pe_w, self.pe_b, "patch_embed")  # [B, N, D]
            torch.add(tok, pos, out=x[:, 1:])

        pending = None  # branch output not yet added to the residual stream (fused into the next LN)
        # K-resident Linear kernel: at D = 384 it beats the library GEMM on qkv, proj and fc1+GELU.  At D = 768 (one
        # row tile per wave: every W fragment feeds one MFMA instead of two) the library wins on qkv / proj (888-932 vs
        # 740-771 TF/s), but fc1 with the erf-GELU fused (dss_linear_k768: 376-390 us against 488 us for the library GEMM
        # + a separate GELU pass at 16 x 3601 tokens) wins end to end too: dino_vitb8 / C3 689-696 -> 704 images/s on
        # the same box.
        k384 = self.linear_k384 and d == 384
        kres_fc1 = self.gelu in ("erf", "erf_f16") and self.linear_k384 >= 2 and d in hip.LINEAR_KRES_WIDTHS
        for i in range(nblocks):
            blk = self.blocks[i]
            qkv_planar = bool(k384)
            if "qkv_wg" in blk and self.linear_k384:      # x += pending; LN1; qkv - one kernel
                qkv = hip.lnlinear(x, pending, blk["qkv_wg"], blk["qkv_aux"], LN_EPS, planar=True)   # [3h, B*T, 64]; pending = fc2's row-major output
                qkv_planar = True
            else:
                hcur = hip.layernorm(x, blk["n1w"], blk["n1b"], LN_EPS, self.dtype, residual=pending)
                if k384:
                    qkv = hip.linear_kres(hcur, blk["qkv_w"], blk["qkv_b"], planar=True)
                else:
                    qkv = self._linear(hcur, blk["qkv_w"], blk["qkv_b"], "qkv")
            o = hip.attention(qkv, heads, self.scale, planar_bt=(b, t)) if qkv_planar else hip.attention(qkv, heads, self.scale)
            if k384:
                pending = hip.linear_kres(o, blk["proj_w"], blk["proj_b"], planar=True)    # [D/64, B*T, 64]
            else:
                pending = self._linear(o, blk["proj_w"], blk["proj_b"], "proj")
            if kres_fc1 and "fc1_wg" in blk:   # x += pending; LN2; fc1; GELU - one kernel (row-major out: fc2 is a library GEMM)
                f1 = hip.lnlinear(x, pending, blk["fc1_wg"], blk["fc1_aux"], LN_EPS, gelu=self._gelu_code, residual_planar=bool(k384))
            else:
                hcur = hip.layernorm(x, blk["n2w"], blk["n2b"], LN_EPS, self.dtype, residual=pending,
                                     residual_planar=bool(k384))
                if kres_fc1:
                    f1 = hip.linear_kres(hcur, blk["fc1_w"], blk["fc1_b"], gelu=self._gelu_code)
                elif self.gelu in ("erf", "erf_f16"):
                    f1 = F.gelu(self._linear(hcur, blk["fc1_w"], blk["fc1_b"], "fc1"))
                else:
                    f1 = torch._addmm_activation(blk["fc1_b"], hcur.view(b * t, d), blk["fc1_w"].t(),
                                                 use_gelu=True).view(b, t, -1)
            if self.fc2_into_stream:
                hip.linear_lt_accumulate(f1, blk["fc2_w"], blk["fc2_b32"], x, what="fc2")     # x += fc2(f1) + b, fp32, in place
                pending = None
            else:
                pending = self._linear(f1, blk["fc2_w"], blk["fc2_b"], "fc2")
        return x, pending

    @torch.no_grad()
    def forward_cls(self, img_u8: torch.Tensor) -> torch.Tensor:
        """DINO's ``model(x)``: all blocks, final LayerNorm, CLS token - ``[B, D]`` fp32.  What the reference's
        ``extract_bbox_features`` (extract/extract.py:500-544) evaluates on every box crop.  ``img_u8``: u8 ``[B, H, W, 3]``
        (cropped to whole patches like every other input)."""
        assert img_u8.dtype == torch.uint8 and img_u8.dim() == 4 and img_u8.shape[-1] == 3
        if self.norm_w is None:
            raise KeyError("state_dict has no final norm.weight / norm.bias: forward_cls needs them")
        if img_u8.shape[1] < self.patch_size or img_u8.shape[2] < self.patch_size:
            raise ValueError(f"image {tuple(img_u8.shape[1:3])} is smaller than one {self.patch_size}x{self.patch_size} patch")
        x, pending = self._run_blocks(img_u8, self.depth)
        cls = x[:, 0] if pending is None else x[:, 0] + pending[:, 0].float()   # (a pending Mlp branch output is row-major [B, T, D] on every path)
        return F.layer_norm(cls, (self.embed_dim,), self.norm_w, self.norm_b, LN_EPS)

    @torch.no_grad()
    def extract_k_f16(self, img_u8: torch.Tensor, which_block: int = -1, out=None):
        """``extract_k`` for a consumer that stays on the GPU (``pipeline.features_and_eigs``): returns
        ``(k [B, N, D] fp32, k16 the same in f16, rnorm [B, N] = 1 / |k16 row|)`` - the hand-over of
        ``hip.kfeatures_finalize`` (bias add, CLS drop, f16 copy and inverse norms in one pass) that the f16-input
        affinity build (``hip.affinity_f16_u16``) starts from."""
        return self.extract_k(img_u8, which_block, _finalize=True, _out=out)

    @torch.no_grad()
    def extract_k(self, img_u8: torch.Tensor, which_block: int = -1, _finalize: bool = False, _out=None) -> torch.Tensor:
        """``img_u8``: u8 ``[B, H, W, 3]`` RGB on the GPU (uncropped).  Returns the hooked K features
        ``[B, N, D]`` fp32, ``N = (H//P)*(W//P)``, rows in row-major patch order, CLS removed."""
        assert img_u8.dtype == torch.uint8 and img_u8.dim() == 4 and img_u8.shape[-1] == 3
        b, h, w, _ = img_u8.shape
        p, d, heads = self.patch_size, self.embed_dim, self.num_heads
        hp, wp = h // p, w // p
        if hp == 0 or wp == 0:
            raise ValueError(f"image {h}x{w} is smaller than one {p}x{p} patch")
        n, t = hp * wp, hp * wp + 1
        wb = which_block if which_block >= 0 else self.depth + which_block
        if not 0 <= wb < self.depth:
            raise IndexError(f"which_block={which_block} out of range for depth {self.depth}")

        x, pending = self._run_blocks(img_u8, wb)
        blk = self.blocks[wb]
        if self.k_proj_fp32:  # all-fp32 K projection (3x slower GEMM; same operand rounding as nowhere else)
            h32 = hip.layernorm(x, blk["n1w"], blk["n1b"], LN_EPS, torch.float32, residual=pending)
            k = F.linear(h32, blk["k_w32"], blk["k_b32"])
        elif "k_wg" in blk and t > 64:
            # residual add + norm1 + K projection + the whole hand-over in one kernel (fp32 features straight from the
            # accumulators); a caller that only wants the features (`extract_features`) drops the other two outputs
            res = hip.lnlinear_kfeatures(x, pending, blk["k_wg"], blk["k_aux"], LN_EPS, out=_out)
            return res if _finalize else res[0]
        else:  # half operands like every other layer, fp32 accumulate AND fp32 output (no rounding of the features)
            hk = hip.layernorm(x, blk["n1w"], blk["n1b"], LN_EPS, self.dtype, residual=pending)
            k = self._linear(hk, blk["k_w"], None, "k_proj", out_dtype=torch.float32)
            if _finalize and n > 0:
                return hip.kfeatures_finalize(k, blk["k_b32"], out=_out)
            k += blk["k_b32"]
        k = k[:, 1:, :].contiguous()
        if _finalize:   # the all-fp32 K projection (or a degenerate grid): same hand-over, no extra bias
            return hip.kfeatures_finalize(torch.cat((k[:, :1], k), dim=1), None, out=_out)
        return k


def wave_filling_batch(tokens: int, target: int = 256, rows_per_workgroup: int = 512,
                       compute_units: Optional[int] = None) -> int:
    """Images per ViT forward near ``target`` such that the token matrix ``[b * tokens, D]`` splits into a whole number
    of waves of 512-row workgroups (one per CU) for ``dss_linear_k384``.  At 480x480 / patch 16 (901 tokens) 256 images
    are 451 workgroups = 1.76 waves on 256 CUs - the second wave runs 76 % full - while 290 images are 511 = 1.996
    waves: same kernel time, 13 % more images (measured: +3.5 % end to end)."""
    if compute_units is None:
        compute_units = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    best, best_eff = target, 0.0
    for b in range(max(1, int(target * 0.85)), int(target * 1.25) + 1):
        tiles = b * tokens / rows_per_workgroup
        waves = math.ceil(math.ceil(tiles) / compute_units)
        eff = tiles / (waves * compute_units)
        if eff > best_eff + 1e-9 or (abs(eff - best_eff) <= 1e-9 and abs(b - target) < abs(best - target)):
            best, best_eff = b, eff
    return best


def load_dino_state_dict(path: str) -> Dict[str, torch.Tensor]:
    """Read a DINO checkpoint (the ``*_pretrain.pth`` files torch.hub would have downloaded, or a
    full-checkpoint dict with a ``teacher``/``student`` entry)."""
    sd = torch.load(path, map_location="cpu", weights_only=True)
    for key in ("teacher", "student", "state_dict", "model"):
        if isinstance(sd, dict) and key in sd and isinstance(sd[key], dict):
            sd = sd[key]
    out = {}
    for k, v in sd.items():
        k = k.replace("module.", "").replace("backbone.", "")
        out[k] = v
    return out
