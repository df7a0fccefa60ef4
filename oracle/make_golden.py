"""ORACLE tooling: generate tests/golden/* by running the REFERENCE ITSELF in this container.

Run (build container only; /root/reference does not exist on the GPU box):
    python oracle/make_golden.py

What it does
------------
The reference (/root/reference/extract/extract.py) cannot be imported unmodified here
(cv2, fire, torchvision, skimage, pymatting are not installed).  None of
those modules carry arithmetic on the north-star path except three tiny symbols, which
are provided by stand-ins that restate their published definitions:
  * ``pymatting.util.util.row_sum(A)``      = ``A.dot(np.ones(A.shape[1], A.dtype))``
  * ``torchvision.transforms.ToTensor/Normalize/Compose`` (u8 HWC -> f32 CHW /255; (x-m)/s)
  * ``cv2.imread`` / ``cv2.cvtColor(BGR2RGB)``  (lossless PNG read through PIL)
Everything else is an inert stub.  ``torch.Tensor.cuda`` is patched to identity, and
``extract.Accelerator`` to a CPU stand-in (``Accelerator(fp16=True)`` raises TypeError under
accelerate 1.14).  ``torch.hub.load`` cannot reach the network; the DINO ViT is third-party
and not under /root/reference, so ``oracle/vit_ref.py`` (restated from the published
architecture) is injected with seeded synthetic weights.

Outputs (data only - seeds, inputs' generators' parameters, expected outputs):
  tests/golden/eigs_<case>.npz     expected (eigenvalues, eigenvectors) of the reference's
                                   ``_extract_eig`` on features from
                                   ``synthetic.synthetic_features(kind, n, d, seed, hw)``
  tests/golden/features.npz        expected feature dicts of the reference's
                                   ``extract_features`` on synthetic PNG images
  tests/golden/index_probe.npz     K-slice / patch-order probe through the reference's lines
"""
from __future__ import annotations

import importlib.util
import os
import sys
import tempfile
import types
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
REFERENCE = Path("/root/reference")
GOLDEN = REPO / "tests" / "golden"
sys.path.insert(0, str(REPO))

from oracle import spectral_ref, vit_ref  # noqa: E402
import dss_amd as dss  # noqa: E402

synthetic = dss.synthetic


# --------------------------------------------------------------------------- stubs
def _install_stubs():
    from PIL import Image

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def imread(path):
        rgb = np.array(Image.open(path).convert("RGB"))
        return rgb[:, :, ::-1].copy()  # cv2 returns BGR

    def cvt_color(img, code):
        assert code == "BGR2RGB"
        return img[:, :, ::-1].copy()

    mod("cv2", imread=imread, cvtColor=cvt_color, COLOR_BGR2RGB="BGR2RGB")
    mod("fire", Fire=lambda *a, **k: None)

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class ToTensor:
        def __call__(self, pic):
            return torch.from_numpy(np.ascontiguousarray(pic)).permute(2, 0, 1).to(torch.float32).div(255)

    class Normalize:
        def __init__(self, mean, std):
            self.mean = torch.tensor(mean).view(-1, 1, 1)
            self.std = torch.tensor(std).view(-1, 1, 1)

        def __call__(self, x):
            return (x - self.mean) / self.std

    tv = mod("torchvision")
    tv.transforms = mod("torchvision.transforms", Compose=Compose, ToTensor=ToTensor, Normalize=Normalize)
    tv.utils = mod("torchvision.utils", draw_bounding_boxes=None)
    sk = mod("skimage")
    sk.morphology = mod("skimage.morphology", binary_dilation=None, binary_erosion=None)
    pm = mod("pymatting")
    pm.util = mod("pymatting.util")
    pm.util.util = mod("pymatting.util.util", row_sum=lambda a: a.dot(np.ones(a.shape[1], a.dtype)))
    torch.Tensor.cuda = lambda self, *a, **k: self


def _import_reference():
    _install_stubs()
    sys.path.insert(0, str(REFERENCE / "extract"))
    spec = importlib.util.spec_from_file_location("ref_extract", REFERENCE / "extract" / "extract.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)

    class CpuAccelerator:  # extract.py:65,67,113,114 use only these members
        device = torch.device("cpu")

        def __init__(self, *a, **k):
            pass

        @staticmethod
        def save(obj, f):
            torch.save(obj, f)

        @staticmethod
        def wait_for_everyone():
            pass

    ref.Accelerator = CpuAccelerator
    return ref


# --------------------------------------------------------------------------- eig cases
EIG_CASES = [
    # name, kind, n, d, seed, hw, K, stored shape (B,C,H,W), patch
    ("g1_random_196", "random", 196, 384, 101, (14, 14), 5, (1, 3, 224, 224), 16),
    ("g1_blobs_196", "blobs", 196, 384, 102, (14, 14), 5, (1, 3, 224, 224), 16),
    ("g2_random_900", "random", 900, 384, 201, (30, 30), 5, (1, 3, 480, 480), 16),
    ("g2_blobs_900", "blobs", 900, 384, 202, (30, 30), 5, (1, 3, 480, 480), 16),
    ("g3_blobs_3600", "blobs", 3600, 768, 301, (60, 60), 15, (1, 3, 480, 480), 8),
    ("g4_blobs_713", "blobs", 713, 384, 401, (23, 31), 5, (1, 3, 375, 500), 16),  # VOC-like non-square
    ("g8_blobs_1600_k20", "blobs", 1600, 768, 801, (40, 40), 20, (1, 3, 320, 320), 8),
]


def make_eig_goldens(ref):
    for name, kind, n, d, seed, hw, K, shape, patch in EIG_CASES:
        feats = synthetic.synthetic_features(kind, n, d, seed, hw)
        with tempfile.TemporaryDirectory() as tmp:
            fdir, odir = Path(tmp) / "f", Path(tmp) / "o"
            fdir.mkdir(), odir.mkdir()
            torch.save({"k": torch.from_numpy(feats)[None], "indices": torch.tensor(0), "file": f"{name}.jpg",
                        "id": name, "model_name": "dino_vits16", "patch_size": patch, "shape": shape},
                       fdir / f"{name}.pth")
            ref._extract_eig((0, str(fdir / f"{name}.pth")), K=K, images_root="", output_dir=str(odir),
                             image_color_lambda=0.0)
            out = torch.load(odir / f"{name}.pth", map_location="cpu", weights_only=False)
            ev, evec = out["eigenvalues"], out["eigenvectors"]
            assert evec.dtype == torch.float32 and tuple(evec.shape) == (K, n), (evec.dtype, evec.shape)
            # the reference's fp32 ARPACK run has heavy-tailed run-to-run noise (oracle/spectral_ref.py,
            # ref_laplacian_eigs_ext): record how far THIS run is from the fp64 solution of the same problem and
            # re-draw a run whose isolated vectors are further than 1e-5 away, so the fixture is a typical output
            lam64, v64 = spectral_ref.dense_f64_eigs(feats, K + 48)
            win = spectral_ref.edge_window_end(lam64, K, 1e-4)
            assert win < K + 47, f"{name}: the 1e-4 window above eigenvalue K-1 holds more than 48 eigenvalues"
            k2 = max(win + 2, K + 3)
            isolated = [lo for lo, hi in spectral_ref.eig_clusters(lam64[:k2], 1e-4) if lo == hi and lo < K]
            for draws in range(1, 6):
                dev = spectral_ref.cos_err(evec.numpy()[isolated], v64[isolated]).max() if isolated else 0.0
                if dev <= 1e-5:
                    break
                print(f"[golden] eigs_{name}: reference draw {draws} is {dev:.1e} from the fp64 solution - re-drawing")
                (odir / f"{name}.pth").unlink()
                ref._extract_eig((0, str(fdir / f"{name}.pth")), K=K, images_root="", output_dir=str(odir),
                                 image_color_lambda=0.0)
                out = torch.load(odir / f"{name}.pth", map_location="cpu", weights_only=False)
                ev, evec = out["eigenvalues"], out["eigenvectors"]
            else:
                raise RuntimeError(f"{name}: 5 reference draws, all further than 1e-5 from the fp64 solution")
        np.savez_compressed(GOLDEN / f"eigs_{name}.npz", kind=kind, n=n, d=d, seed=seed, hw=np.array(hw), K=K,
                            shape=np.array(shape), patch=patch,
                            eigenvalues=ev.numpy().astype(np.float32), eigenvectors=evec.numpy(),
                            eigenvalues_dtype=str(ev.dtype),
                            # fp64 dense solution of the same problem, K + E pairs reaching lambda[K-1] + 1e-4: lets
                            # tests/util.check_eigs decide a cluster of near-equal eigenvalues that straddles K - 1
                            eigenvalues_ext=lam64[:k2], eigenvectors_ext=v64[:k2].astype(np.float32),
                            reference_draws=draws, reference_vs_f64_cos_err=float(dev))
        print(f"[golden] eigs_{name}: lambda={ev.numpy()[:6]} ({draws} draw(s), {dev:.1e} from fp64; "
              f"{k2 - K} extra fp64 pairs, 1e-4 window above K-1 ends at {win})")


# --------------------------------------------------------------------------- feature cases
FEATURE_IMAGES = [  # file name, H, W   (non-multiples of 16 exercise the crop)
    ("img_b.png", 100, 130),
    ("img_a.png", 75, 64),
    ("img_c.png", 224, 224),
    ("img_a.png", 75, 64),  # duplicate line: the dataset de-duplicates and sorts (extract_utils.py:23)
]
FEATURE_MODEL = "dino_vits16"
FEATURE_WEIGHT_SEED = 7
FEATURE_LN_JITTER = 0.05


def make_feature_goldens(ref):
    from PIL import Image

    sd = synthetic.synthetic_state_dict(FEATURE_MODEL, FEATURE_WEIGHT_SEED, FEATURE_LN_JITTER)
    torch.hub.load = lambda repo, name, *a, **k: vit_ref.build_ref_vit(name, sd)
    with tempfile.TemporaryDirectory() as tmp:
        root, odir = Path(tmp) / "images", Path(tmp) / "features"
        root.mkdir()
        for i, (fn, h, w) in enumerate(FEATURE_IMAGES):
            Image.fromarray(synthetic.synthetic_image(hash_name(fn), h, w)).save(root / fn)
        lst = Path(tmp) / "images.txt"
        lst.write_text("\n".join(fn for fn, _, _ in FEATURE_IMAGES) + "\n")
        # DataLoader(num_workers=8) forks; keep it, it is the reference's own loader
        ref.extract_features(images_list=str(lst), images_root=str(root), model_name=FEATURE_MODEL,
                             batch_size=1, output_dir=str(odir))
        out = {}
        for f in sorted(odir.iterdir()):
            dct = torch.load(f, map_location="cpu", weights_only=False)
            stem = f.stem
            assert dct["k"].dtype == torch.float32
            out[f"{stem}__k"] = dct["k"].numpy()
            out[f"{stem}__indices"] = np.array(int(dct["indices"]))
            out[f"{stem}__indices_is_tensor"] = np.array(torch.is_tensor(dct["indices"]))
            out[f"{stem}__file"] = np.array(dct["file"])
            out[f"{stem}__id"] = np.array(dct["id"])
            out[f"{stem}__model_name"] = np.array(dct["model_name"])
            out[f"{stem}__patch_size"] = np.array(dct["patch_size"])
            out[f"{stem}__shape"] = np.array(dct["shape"])
            out[f"{stem}__keys"] = np.array(sorted(dct.keys()))
            print(f"[golden] features {stem}: k{tuple(dct['k'].shape)} shape={dct['shape']} idx={int(dct['indices'])}")
    # 224x224 -> 196x384 f32 = 300 KB: keep every 5th row only, plus a checksum of all rows
    k = out["img_c__k"]
    out["img_c__k_rowsum"] = k.astype(np.float64).sum(-1).astype(np.float32)
    out["img_c__k"] = k[:, ::5, :]
    out["img_c__k_stride"] = np.array(5)
    np.savez_compressed(GOLDEN / "features.npz", model=FEATURE_MODEL, weight_seed=FEATURE_WEIGHT_SEED,
                        ln_jitter=FEATURE_LN_JITTER,
                        files=np.array([fn for fn, _, _ in FEATURE_IMAGES]),
                        sizes=np.array([(h, w) for _, h, w in FEATURE_IMAGES]), **out)


def hash_name(fn: str) -> int:
    """Stable image index from a file name (so duplicates regenerate the same image)."""
    return sum(ord(c) for c in fn) % 1000


# --------------------------------------------------------------------------- index probe
def make_index_probe(ref):
    """Drive the reference's hook + K-slice lines (extract.py:51-53,94-98) with a model whose
    qkv output ENCODES (token, third, head, channel), pinning row-major patch order, CLS
    removal and the [3][h][d_h] column layout bit-exactly."""
    from PIL import Image

    heads, dh, patch = 3, 4, 16
    dim = heads * dh

    class ProbeQKV(torch.nn.Module):
        def forward(self, x):  # x: [B, T, dim]; value = 1000*token + column index of the 3*dim output
            b, t, _ = x.shape
            tok = torch.arange(t, dtype=torch.float32).view(1, t, 1) * 1000.0
            col = torch.arange(3 * dim, dtype=torch.float32).view(1, 1, -1)
            return (tok + col).expand(b, t, 3 * dim).clone()

    class ProbeAttn(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.num_heads = heads
            self.qkv = ProbeQKV()

    class ProbeBlock(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.attn = ProbeAttn()

    class ProbePE(torch.nn.Module):
        patch_size = patch

    class ProbeModel(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.patch_embed = ProbePE()
            self.blocks = torch.nn.ModuleList([ProbeBlock()])

        def get_intermediate_layers(self, images):
            b, _, h, w = images.shape
            t = (h // patch) * (w // patch) + 1
            return [self.blocks[0].attn.qkv(torch.zeros(b, t, dim))]

    torch.hub.load = lambda repo, name, *a, **k: ProbeModel()
    with tempfile.TemporaryDirectory() as tmp:
        root, odir = Path(tmp) / "images", Path(tmp) / "features"
        root.mkdir()
        h, w = 50, 70  # -> 3 x 4 patches, T = 13
        Image.fromarray(synthetic.synthetic_image(5, h, w)).save(root / "probe.png")
        lst = Path(tmp) / "images.txt"
        lst.write_text("probe.png\n")
        ref.extract_features(images_list=str(lst), images_root=str(root), model_name="dino_probe",
                             batch_size=1, output_dir=str(odir))
        dct = torch.load(odir / "probe.pth", map_location="cpu", weights_only=False)
    np.savez_compressed(GOLDEN / "index_probe.npz", heads=heads, dh=dh, patch=patch, h=h, w=w,
                        k=dct["k"].numpy(), shape=np.array(dct["shape"]))
    print(f"[golden] index_probe: k{tuple(dct['k'].shape)} first row {dct['k'][0, 0, :4].tolist()}")


# --------------------------------------------------------------------------- HF cross-check
def check_vit_against_hf(model_name="dino_vits16", seed=3, atol=2e-4):
    """Independent implementation check of oracle/vit_ref.py: ``transformers.ViTModel`` with the
    same weights at the native 224x224 resolution (no positional interpolation)."""
    from transformers import ViTConfig, ViTModel

    cfg = vit_ref.CONFIGS[model_name]
    dim, depth, heads, patch = cfg["dim"], cfg["depth"], cfg["heads"], cfg["patch"]
    sd = synthetic.synthetic_state_dict(model_name, seed, ln_jitter=0.05)
    ours = vit_ref.build_ref_vit(model_name, sd)
    hf = ViTModel(ViTConfig(hidden_size=dim, num_hidden_layers=depth, num_attention_heads=heads,
                            intermediate_size=4 * dim, hidden_act="gelu", layer_norm_eps=1e-6,
                            image_size=224, patch_size=patch, qkv_bias=True,
                            hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0),
                  add_pooling_layer=False).eval()
    m = {"embeddings.cls_token": sd["cls_token"], "embeddings.position_embeddings": sd["pos_embed"],
         "embeddings.patch_embeddings.projection.weight": sd["patch_embed.proj.weight"],
         "embeddings.patch_embeddings.projection.bias": sd["patch_embed.proj.bias"],
         "layernorm.weight": sd["norm.weight"], "layernorm.bias": sd["norm.bias"]}
    for i in range(depth):  # key names of transformers 5.x ViTModel
        s, t = f"blocks.{i}.", f"layers.{i}."
        qw, qb = sd[s + "attn.qkv.weight"], sd[s + "attn.qkv.bias"]
        for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
            m[t + f"attention.{nm}.weight"] = qw[j * dim:(j + 1) * dim]
            m[t + f"attention.{nm}.bias"] = qb[j * dim:(j + 1) * dim]
        m[t + "attention.o_proj.weight"] = sd[s + "attn.proj.weight"]
        m[t + "attention.o_proj.bias"] = sd[s + "attn.proj.bias"]
        m[t + "layernorm_before.weight"], m[t + "layernorm_before.bias"] = sd[s + "norm1.weight"], sd[s + "norm1.bias"]
        m[t + "layernorm_after.weight"], m[t + "layernorm_after.bias"] = sd[s + "norm2.weight"], sd[s + "norm2.bias"]
        m[t + "mlp.fc1.weight"], m[t + "mlp.fc1.bias"] = sd[s + "mlp.fc1.weight"], sd[s + "mlp.fc1.bias"]
        m[t + "mlp.fc2.weight"], m[t + "mlp.fc2.bias"] = sd[s + "mlp.fc2.weight"], sd[s + "mlp.fc2.bias"]
    missing, unexpected = hf.load_state_dict(m, strict=False)
    assert not unexpected and not missing, (len(missing), len(unexpected), missing[:3], unexpected[:3])
    img = vit_ref.ref_preprocess(synthetic.synthetic_image(11, 224, 224))
    grabbed = {}
    hf.layers[-1].attention.k_proj.register_forward_hook(
        lambda mod, inp, out: grabbed.__setitem__("k", out))
    with torch.no_grad():
        hf_out = hf(pixel_values=img[None]).last_hidden_state
        our_out = ours.get_intermediate_layers(img[None])[0]
        our_k = vit_ref.ref_extract_k(ours, img)
    e1 = (hf_out - our_out).abs().max().item()
    e2 = (grabbed["k"][:, 1:] - our_k).abs().max().item()
    print(f"[hf-check] {model_name}: |last_hidden diff|max={e1:.2e}  |K diff|max={e2:.2e}")
    assert e1 < atol and e2 < atol, (e1, e2)
    return e1, e2


MODE_CASES = [  # name, kind, n, d, seed, hw, K, kwargs for the reference's _extract_eig
    ("affinity_blobs_196", "blobs", 196, 384, 102, (14, 14), 5, dict(which_matrix="affinity")),
    ("affinity_blobs_900", "blobs", 900, 384, 202, (30, 30), 5, dict(which_matrix="affinity")),
    ("affinity_nothresh_196", "blobs", 196, 384, 102, (14, 14), 5, dict(which_matrix="affinity", threshold_at_zero=False)),
    ("affinity_svd_blobs_196", "blobs", 196, 384, 102, (14, 14), 5, dict(which_matrix="affinity_svd")),
    ("affinity_svd_random_900", "random", 900, 384, 201, (30, 30), 5, dict(which_matrix="affinity_svd")),
    ("lapnorm_false_blobs_196", "blobs", 196, 384, 102, (14, 14), 5, dict(which_matrix="laplacian", lapnorm=False)),
    ("lapnorm_false_blobs_900", "blobs", 900, 384, 202, (30, 30), 5, dict(which_matrix="laplacian", lapnorm=False)),
    # feature upsampling (extract.py:179-188): P=16 features on an 8-pixel grid -> 2x bilinear, N_lr = 4 N
    ("upsample8_blobs_196", "blobs", 196, 384, 102, (14, 14), 5, dict(which_matrix="laplacian", image_downsample_factor=8)),
    ("upsample8_blobs_23x31", "blobs", 713, 384, 401, (23, 31), 4, dict(which_matrix="laplacian", image_downsample_factor=8)),
    # `W_feat / W_feat.max()` (extract.py:194) matters when the rows are not unit vectors: magnitudes, not just cosines
    ("nonorm_blobs_196", "blobs", 196, 384, 102, (14, 14), 5, dict(which_matrix="laplacian", normalize=False)),
    ("nonorm_lapnorm_false_blobs_196", "blobs", 196, 384, 102, (14, 14), 5,
     dict(which_matrix="laplacian", lapnorm=False, normalize=False)),
    ("upsample8_lapnorm_false_blobs_196", "blobs", 196, 384, 102, (14, 14), 4,
     dict(which_matrix="laplacian", lapnorm=False, image_downsample_factor=8)),
]


def make_mode_goldens(ref):
    """The reference's other _extract_eig branches (extract.py:159-172 affinity / affinity_svd, :230-234 lapnorm=False)."""
    for name, kind, n, d, seed, hw, K, kw in MODE_CASES:
        feats = synthetic.synthetic_features(kind, n, d, seed, hw)
        with tempfile.TemporaryDirectory() as tmp:
            fdir, odir = Path(tmp) / "f", Path(tmp) / "o"
            fdir.mkdir(), odir.mkdir()
            torch.save({"k": torch.from_numpy(feats)[None], "indices": torch.tensor(0), "file": f"{name}.jpg",
                        "id": name, "model_name": "dino_vits16", "patch_size": 16,
                        "shape": (1, 3, hw[0] * 16, hw[1] * 16)}, fdir / f"{name}.pth")
            ref._extract_eig((0, str(fdir / f"{name}.pth")), K=K, images_root="", output_dir=str(odir),
                             image_color_lambda=0.0, **kw)
            out = torch.load(odir / f"{name}.pth", map_location="cpu", weights_only=False)
        ev, evec = out["eigenvalues"], out["eigenvectors"]
        ev_is_numpy = isinstance(ev, np.ndarray)
        ev = np.asarray(ev)
        np.savez_compressed(GOLDEN / f"modes_{name}.npz", kind=kind, n=n, d=d, seed=seed, hw=np.array(hw), K=K,
                            kwargs=np.array(repr(kw)), eigenvalues=ev, eigenvalues_dtype=str(ev.dtype),
                            eigenvalues_is_numpy=ev_is_numpy, eigenvectors=evec.numpy(),
                            eigenvectors_dtype=str(evec.dtype))
        print(f"[golden] modes_{name}: ev({'np' if ev_is_numpy else 'torch'},{ev.dtype})={ev[:5]} vec{tuple(evec.shape)} {evec.dtype}")


def make_single_region_golden(ref):
    """Reference extract_single_region_segmentations (extract.py:383-426) on a synthetic feature/eig pair."""
    from PIL import Image

    rng = np.random.default_rng(5)
    with tempfile.TemporaryDirectory() as tmp:
        fdir, edir, odir = Path(tmp) / "f", Path(tmp) / "e", Path(tmp) / "o"
        fdir.mkdir(), edir.mkdir()
        shape, patch = (1, 3, 75, 100), 16   # 4 x 6 patches
        n = (shape[2] // patch) * (shape[3] // patch)
        vec = rng.normal(size=(3, n)).astype(np.float32)
        torch.save({"k": torch.zeros(1, n, 8), "indices": torch.tensor(0), "file": "seg_x.jpg", "id": "seg_x",
                    "model_name": "dino_vits16", "patch_size": patch, "shape": shape}, fdir / "seg_x.pth")
        torch.save({"eigenvalues": torch.zeros(3), "eigenvectors": torch.from_numpy(vec)}, edir / "seg_x.pth")
        ref.extract_single_region_segmentations(features_dir=str(fdir), eigs_dir=str(edir), output_dir=str(odir))
        png = np.array(Image.open(odir / "seg_x.png"))
    np.savez_compressed(GOLDEN / "single_region.npz", eigenvectors=vec, shape=np.array(shape), patch=patch, png=png)
    print(f"[golden] single_region: png {png.shape} {png.dtype} values {np.unique(png)}")


def main():
    assert REFERENCE.is_dir(), "make_golden.py runs only where /root/reference is mounted"
    GOLDEN.mkdir(parents=True, exist_ok=True)
    torch.set_grad_enabled(False)  # extract.py:838
    os.environ.setdefault("OMP_NUM_THREADS", "8")
    if not sys.argv[1:]:
        check_vit_against_hf()  # before the stubs: transformers probes for a real torchvision
    ref = _import_reference()
    only = set(sys.argv[1:])  # e.g. `python oracle/make_golden.py eigs modes`; nothing = everything
    steps = {"probe": make_index_probe, "features": make_feature_goldens, "eigs": make_eig_goldens,
             "single_region": make_single_region_golden, "modes": make_mode_goldens}
    assert only <= set(steps), f"unknown step(s) {only - set(steps)}; known: {sorted(steps)}"
    for name, fn in steps.items():
        if not only or name in only:
            fn(ref)


if __name__ == "__main__":
    main()
