"""CPU: pin the oracle (oracle/*.py) against the golden vectors produced by the REFERENCE ITSELF
(oracle/make_golden.py, run in the build container) - SURVEY.md §8c."""
import glob

import numpy as np
import pytest
import torch

import dss_amd  # noqa: F401
from dss_amd import synthetic
from oracle import spectral_ref, vit_ref
from tests.util import check_eigs, golden_case, golden_ext, build_w64, d_orthonormality

EIG_FILES = sorted(glob.glob(str(__import__("pathlib").Path(__file__).parent / "golden" / "eigs_*.npz")))


def test_goldens_present():
    assert len(EIG_FILES) >= 6


@pytest.mark.parametrize("path", [p for p in EIG_FILES if "3600" not in p], ids=lambda p: p.split("eigs_")[-1][:-4])
def test_oracle_eigs_match_reference_goldens(path):
    feats, K, ref_lam, ref_vec, g = golden_case(path)
    lam, vec = spectral_ref.ref_laplacian_eigs(torch.from_numpy(feats)[None], K)
    assert vec.dtype == torch.float32 and tuple(vec.shape) == (K, feats.shape[0])
    check_eigs(vec.numpy(), lam.numpy(), ref_vec, ref_lam, what=path, d=build_w64(feats)[1], ext=golden_ext(g))
    # the oracle call the end-to-end parity checks use: validated reference draw + fp64 extra pairs
    lam2, vec2, ext, draws = spectral_ref.ref_laplacian_eigs_ext(torch.from_numpy(feats)[None], K)
    x_lam, x_vec = golden_ext(g)
    assert 1 <= draws <= 4 and ext[1].shape[0] >= K + 3 and ext[1].shape[0] == x_vec.shape[0]
    assert spectral_ref.edge_window_end(ext[0], K, 1e-4) < ext[1].shape[0] - 1
    np.testing.assert_allclose(ext[0], x_lam, rtol=0, atol=1e-9)
    check_eigs(vec2.numpy(), lam2.numpy(), ref_vec, ref_lam, what=path + " (ext)", d=build_w64(feats)[1], ext=ext)
    assert int(g["reference_draws"]) >= 1 and float(g["reference_vs_f64_cos_err"]) <= 1e-5


def test_golden_conventions():
    """Conventions the consumers rely on: ascending eigenvalues, lambda_0 ~ 0 with a
    constant vector, v^T D v = 1, residual small, sign-rule post-condition."""
    feats, K, lam, vec, g = golden_case([p for p in EIG_FILES if "g2_blobs_900" in p][0])
    w, d = build_w64(feats)
    assert abs(lam[0]) < 1e-6 and np.all(np.diff(lam) > 0)
    assert np.std(vec[0]) / abs(np.mean(vec[0])) < 1e-4
    assert d_orthonormality(vec, d=d) < 1e-4
    lap = np.diag(d) - w
    for k in range(K):
        r = lap @ vec[k] - lam[k] * d * vec[k]
        assert np.abs(r).max() < 1e-3
        frac = np.mean(vec[k] > 0)
        assert not (0.5 < frac < 1.0)


def test_sign_rule_cases():
    v = torch.tensor([[1.0, 1.0, 1.0, -1.0],    # 0.75 positive -> flipped
                      [1.0, 1.0, -1.0, -1.0],   # exactly 0.5   -> kept
                      [1.0, 1.0, 1.0, 1.0],     # all positive  -> kept
                      [-1.0, -1.0, -1.0, 1.0],  # 0.25          -> kept
                      [0.0, 1.0, 1.0, 1.0]])    # zero is not > 0: 0.75 -> flipped
    out = spectral_ref.ref_sign_rule(v.clone())
    assert torch.equal(out[0], -v[0]) and torch.equal(out[1], v[1]) and torch.equal(out[2], v[2])
    assert torch.equal(out[3], v[3]) and torch.equal(out[4], -v[4])


def test_oracle_features_match_reference_goldens(golden_dir):
    """oracle/vit_ref.ref_preprocess + ref_extract_k restate extract_features' per-image arithmetic; the
    goldens come from the reference's own driver (dataset order, crop, hook, K-slice, schema)."""
    g = np.load(golden_dir / "features.npz")
    sd = synthetic.synthetic_state_dict(str(g["model"]), int(g["weight_seed"]), float(g["ln_jitter"]))
    model = vit_ref.build_ref_vit(str(g["model"]), sd)
    names = sorted(set(str(f) for f in g["files"]))  # de-duplicated + sorted: extract_utils.py:23
    sizes = {str(f): tuple(s) for f, s in zip(g["files"], g["sizes"])}
    for index, fn in enumerate(names):
        stem = fn[:-4]
        h, w = sizes[fn]
        img = synthetic.synthetic_image(sum(ord(c) for c in fn) % 1000, int(h), int(w))
        k = vit_ref.ref_extract_k(model, vit_ref.ref_preprocess(img)).numpy()
        ref = g[f"{stem}__k"]
        if f"{stem}__k_stride" in g:
            np.testing.assert_allclose(k.astype(np.float64).sum(-1), g[f"{stem}__k_rowsum"], rtol=0, atol=2e-4)
            k = k[:, :: int(g[f"{stem}__k_stride"]), :]
        assert k.shape == ref.shape
        np.testing.assert_allclose(k, ref, rtol=0, atol=2e-5)
        assert int(g[f"{stem}__indices"]) == index and bool(g[f"{stem}__indices_is_tensor"])
        assert tuple(g[f"{stem}__shape"]) == (1, 3, h, w)           # UNCROPPED shape is stored
        assert str(g[f"{stem}__file"]) == fn and str(g[f"{stem}__id"]) == stem
        assert int(g[f"{stem}__patch_size"]) == 16 and str(g[f"{stem}__model_name"]) == "dino_vits16"
        assert list(g[f"{stem}__keys"]) == sorted(["k", "indices", "file", "id", "model_name", "patch_size", "shape"])
        if f"{stem}__k_stride" not in g:
            assert ref.shape[1] == (h // 16) * (w // 16)


def test_index_probe_contract(golden_dir):
    """Bit-exact patch indexing: row n of k <-> token n+1 (CLS dropped), columns = the K third [D:2D] of the
    qkv output, in (head, channel) order - from the reference's own lines 96-98 on an encoding model."""
    g = np.load(golden_dir / "index_probe.npz")
    heads, dh, patch, h, w = (int(g[k]) for k in ("heads", "dh", "patch", "h", "w"))
    dim = heads * dh
    n = (h // patch) * (w // patch)
    k = g["k"]
    assert k.shape == (1, n, dim)
    expect = (np.arange(1, n + 1, dtype=np.float32)[:, None] * 1000.0 + np.arange(dim, 2 * dim, dtype=np.float32)[None, :])
    assert np.array_equal(k[0], expect)


@pytest.mark.timeout(300)
def test_vit_oracle_against_independent_implementation():
    """oracle/vit_ref.py vs transformers.ViTModel with the same weights (architecture pin)."""
    from oracle.make_golden import check_vit_against_hf

    e1, e2 = check_vit_against_hf()
    assert e1 < 2e-4 and e2 < 2e-4


def test_pos_embed_interpolation_matches_oracle():
    from dss_amd.vit import interpolate_pos_encoding

    sd = synthetic.synthetic_state_dict("dino_vits16", 5)
    model = vit_ref.build_ref_vit("dino_vits16", sd)
    for h, w in ((480, 480), (368, 496), (224, 224), (64, 64)):
        x = torch.zeros(1, (h // 16) * (w // 16) + 1, 384)
        ref = model.interpolate_pos_encoding(x, h, w)
        ours = interpolate_pos_encoding(sd["pos_embed"], 16, h, w)
        assert torch.equal(ref, ours)
