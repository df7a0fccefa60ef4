"""the consumers / options either side of the hot path, each pinned to outputs of the REFERENCE ITSELF
(oracle/make_golden.py `consumers`, `localization`, `bbox_features`, `color` -> tests/golden/*.npz).

CPU tests: boxes, eigensegment -> box, the host-side helpers of the segmentation commands (bit-exact).
GPU tests (`-m gpu`): the single / multi-region segmentation commands (device kernels), colour-affinity fusion through the
Lanczos kernel, box-crop CLS features through the ViT kernels, the inline localization eigenvectors."""
import inspect
import json
from pathlib import Path

import numpy as np
import pytest
import torch

import dss_amd  # noqa: F401
from dss_amd import extract, extract_utils, object_discovery, synthetic

REPO = Path(__file__).resolve().parents[1]
GOLDEN = REPO / "tests" / "golden"


def _write_case(tmp: Path, g, name, kind, hw, factor, K):
    """The feature + eigen files oracle/make_golden._consumer_inputs wrote for this case (eigenpairs from the golden)."""
    n = hw[0] * hw[1]
    feats = synthetic.synthetic_features(kind, n, 384, 500 + len(name), tuple(hw))
    (tmp / "f").mkdir(exist_ok=True), (tmp / "e").mkdir(exist_ok=True)
    torch.save({"k": torch.from_numpy(feats)[None], "indices": torch.tensor(0), "file": f"{name}.jpg", "id": name,
                "model_name": "dino_vits16", "patch_size": 16, "shape": (1, 3, hw[0] * 16 + 5, hw[1] * 16 + 9)},
               tmp / "f" / f"{name}.pth")
    torch.save({"eigenvalues": torch.from_numpy(g[f"{name}__eigenvalues"]),
                "eigenvectors": torch.from_numpy(g[f"{name}__eigenvectors"])}, tmp / "e" / f"{name}.pth")


def _cases(g):
    return json.loads(str(g["cases"]))


def _inertia(points, labels):
    return float(sum(((points[labels == c] - points[labels == c].mean(0)) ** 2).sum() for c in np.unique(labels)))


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(6))
def test_multi_region_segmentation_command_against_reference_png(tmp_path, case):
    """extract/extract.py:283-377 through the CLI-level command, which clusters on the device (dss_kmeans_segments, or
    spectral.kmeans_lloyd for the raw-feature baseline).  The reference's KMeans() is unseeded - its PNG (the golden, one
    seeded draw) is one sample of a random partition - so the pins are: same grid, same number of segments, a partition at
    least as tight as the reference's draw to 15 % (inertia over the coordinates the reference clusters), the border rule,
    skip-if-exists, and determinism of a second run.  (From the SAME initial centres the kernel reproduces sklearn's labels:
    test_kmeans_segments_on_device_matches_sklearn_from_the_same_centres.)"""
    from PIL import Image

    g = np.load(GOLDEN / "consumers.npz")
    name, kind, hw, factor, K, seed, kw = _cases(g)[case]
    _write_case(tmp_path, g, name, kind, hw, factor, K)
    extract.extract_multi_region_segmentations(features_dir=str(tmp_path / "f"), eigs_dir=str(tmp_path / "e"),
                                               output_dir=str(tmp_path / "o"), **kw)
    png = np.array(Image.open(tmp_path / "o" / f"{name}.png"))
    want = g[f"{name}__png"]
    assert png.dtype == want.dtype == np.uint8 and png.shape == want.shape
    assert len(np.unique(png)) == len(np.unique(want))
    vec = g[f"{name}__eigenvectors"]
    if kw.get("kmeans_baseline"):
        pts = synthetic.synthetic_features(kind, hw[0] * hw[1], 384, 500 + len(name), tuple(hw)).astype(np.float64)
    else:
        pts = vec[1:1 + min(kw.get("num_eigenvectors", 1_000_000), vec.shape[0] - 1)].T.astype(np.float64)
    assert _inertia(pts, png.reshape(-1)) <= 1.15 * _inertia(pts, want.reshape(-1)) + 1e-9
    if kw.get("infer_bg_index", True):   # the segment owning most of the border is 0
        idx, frac = extract_utils.get_border_fraction(png)
        assert idx[np.argmax(frac)] == 0
    # skip-if-exists, like every command of the reference
    before = (tmp_path / "o" / f"{name}.png").stat().st_mtime_ns
    extract._extract_multi_region_segmentations(
        (0, (str(tmp_path / "f" / f"{name}.pth"), str(tmp_path / "e" / f"{name}.pth"))), adaptive=False,
        non_adaptive_num_segments=2, infer_bg_index=True, kmeans_baseline=False, output_dir=str(tmp_path / "o"),
        num_eigenvectors=10)
    assert (tmp_path / "o" / f"{name}.png").stat().st_mtime_ns == before
    extract.extract_multi_region_segmentations(features_dir=str(tmp_path / "f"), eigs_dir=str(tmp_path / "e"),
                                               output_dir=str(tmp_path / "o2"), **kw)
    assert np.array_equal(np.array(Image.open(tmp_path / "o2" / f"{name}.png")), png)


@pytest.mark.gpu
def test_single_region_command_matches_reference_png(tmp_path):
    """extract/extract.py:383-426 through the command (dss_fiedler_mask): the reference's PNG, bit for bit."""
    from PIL import Image

    g = np.load(GOLDEN / "single_region.npz")
    (tmp_path / "f").mkdir(), (tmp_path / "e").mkdir()
    n = g["eigenvectors"].shape[1]
    torch.save({"k": torch.zeros(1, n, 8), "indices": torch.tensor(0), "file": "img.jpg", "id": "img", "model_name": "dino_vits16",
                "patch_size": int(g["patch"]), "shape": tuple(int(v) for v in g["shape"])}, tmp_path / "f" / "img.pth")
    torch.save({"eigenvalues": torch.zeros(g["eigenvectors"].shape[0]), "eigenvectors": torch.from_numpy(g["eigenvectors"])},
               tmp_path / "e" / "img.pth")
    extract.extract_single_region_segmentations(str(tmp_path / "f"), str(tmp_path / "e"), str(tmp_path / "o"))
    png = np.array(Image.open(tmp_path / "o" / "img.png"))
    assert png.dtype == np.uint8 and np.array_equal(png, g["png"])


def test_segmentation_commands_fall_back_to_the_host_without_a_gpu(tmp_path, monkeypatch):
    """The reference ran `extract_single_region_segmentations` / `extract_multi_region_segmentations` on the CPU; here they use
    the device kernels when a GPU is there and plain tensor code otherwise (`extract._consumer_device`).  Forced onto the host:
    the single-region PNG is the reference's bit for bit (tests/golden/single_region.npz), the multi-region command gives the
    reference's grid and number of segments, a partition at least as tight as the reference's draw to 15 %, the border rule,
    and a deterministic second run."""
    from PIL import Image

    monkeypatch.setattr(extract, "_consumer_device", lambda: torch.device("cpu"))
    g = np.load(GOLDEN / "single_region.npz")
    (tmp_path / "f").mkdir(), (tmp_path / "e").mkdir()
    n = g["eigenvectors"].shape[1]
    torch.save({"k": torch.zeros(1, n, 8), "indices": torch.tensor(0), "file": "img.jpg", "id": "img", "model_name": "dino_vits16",
                "patch_size": int(g["patch"]), "shape": tuple(int(v) for v in g["shape"])}, tmp_path / "f" / "img.pth")
    torch.save({"eigenvalues": torch.zeros(g["eigenvectors"].shape[0]), "eigenvectors": torch.from_numpy(g["eigenvectors"])},
               tmp_path / "e" / "img.pth")
    extract.extract_single_region_segmentations(str(tmp_path / "f"), str(tmp_path / "e"), str(tmp_path / "o"))
    png = np.array(Image.open(tmp_path / "o" / "img.png"))
    assert png.dtype == np.uint8 and np.array_equal(png, g["png"])
    g = np.load(GOLDEN / "consumers.npz")
    for case in (0, 2):
        sub = tmp_path / f"c{case}"
        sub.mkdir()
        name, kind, hw, factor, K, seed, kw = _cases(g)[case]
        _write_case(sub, g, name, kind, hw, factor, K)
        extract.extract_multi_region_segmentations(features_dir=str(sub / "f"), eigs_dir=str(sub / "e"), output_dir=str(sub / "o"), **kw)
        png, want = np.array(Image.open(sub / "o" / f"{name}.png")), g[f"{name}__png"]
        assert png.dtype == np.uint8 and png.shape == want.shape and len(np.unique(png)) == len(np.unique(want))
        vec = g[f"{name}__eigenvectors"]
        if kw.get("kmeans_baseline"):
            pts = synthetic.synthetic_features(kind, hw[0] * hw[1], 384, 500 + len(name), tuple(hw)).astype(np.float64)
        else:
            pts = vec[1:1 + min(kw.get("num_eigenvectors", 1_000_000), vec.shape[0] - 1)].T.astype(np.float64)
        assert _inertia(pts, png.reshape(-1)) <= 1.15 * _inertia(pts, want.reshape(-1)) + 1e-9
        if kw.get("infer_bg_index", True):
            idx, frac = extract_utils.get_border_fraction(png)
            assert idx[np.argmax(frac)] == 0
        extract.extract_multi_region_segmentations(features_dir=str(sub / "f"), eigs_dir=str(sub / "e"), output_dir=str(sub / "o2"), **kw)
        assert np.array_equal(np.array(Image.open(sub / "o2" / f"{name}.png")), png)


def test_kmeans_lloyd_and_border_rule_on_the_host():
    """spectral.kmeans_lloyd / border_owner_to_zero / adaptive_num_segments are plain tensor code (the route for problems
    beyond the K-means kernel's limits): separated blobs are recovered exactly, the run is deterministic in its seed, the
    border rule is the reference's (extract_utils.get_border_fraction, corners twice, ties to the smaller label)."""
    from dss_amd import spectral

    gen = torch.Generator().manual_seed(3)
    pts = torch.cat([torch.randn(100, 5, generator=gen) + 6, torch.randn(80, 5, generator=gen) - 6, torch.randn(60, 5, generator=gen) * 0.5])
    lab = spectral.kmeans_lloyd(pts, 3, seed=1)
    assert sorted(torch.bincount(lab).tolist()) == [60, 80, 100]
    assert len(set(lab[:100].tolist())) == len(set(lab[100:180].tolist())) == len(set(lab[180:].tolist())) == 1
    assert torch.equal(lab, spectral.kmeans_lloyd(pts, 3, seed=1))
    assert spectral.kmeans_lloyd(pts[:2], 5).shape == (2,)                  # more clusters than points
    seg = np.array([[1, 1, 2], [0, 5, 2], [0, 0, 2]])
    idx, frac = extract_utils.get_border_fraction(seg)
    owner = idx[np.argmax(frac)]
    out = spectral.border_owner_to_zero(torch.from_numpy(seg)).numpy()
    assert owner == 2 and np.array_equal(out == 0, seg == 2) and np.array_equal(out == 2, seg == 0)
    ev = torch.tensor([[0.0, 0.10, 0.15, 0.50, 0.55], [0.0, 0.5, 0.51, 0.52, 0.9]])
    order = [np.argsort(np.diff(e.numpy()))[::-1] for e in ev]
    assert spectral.adaptive_num_segments(ev) == [int(o[o != 0][0]) + 1 for o in order] == [3, 4]


@pytest.mark.gpu
def test_multi_region_rejects_a_grid_that_is_neither_1x_nor_2x(tmp_path):
    g = np.load(GOLDEN / "consumers.npz")
    name, kind, hw, factor, K, seed, kw = _cases(g)[0]
    _write_case(tmp_path, g, name, kind, hw, factor, K)
    d = torch.load(tmp_path / "e" / f"{name}.pth")
    d["eigenvectors"] = d["eigenvectors"][:, :100]
    torch.save(d, tmp_path / "e" / f"{name}.pth")
    with pytest.raises(ValueError):
        extract.extract_multi_region_segmentations(str(tmp_path / "f"), str(tmp_path / "e"), str(tmp_path / "o"))


@pytest.mark.gpu
def test_multi_region_beyond_the_kernel_limits_takes_the_tensor_route():
    """More than 8192 points (a 2x upsampled 480 x 480 / patch 8 grid has 14 400): spectral.multi_region_segments still
    answers, through kmeans_lloyd + border_owner_to_zero."""
    from dss_amd import spectral

    rows, cols = 96, 100
    yy, xx = torch.meshgrid(torch.arange(rows), torch.arange(cols), indexing="ij")
    blob = ((yy - 40) ** 2 + (xx - 50) ** 2 < 30 ** 2).float().reshape(-1)
    vec = torch.stack([torch.ones(rows * cols), blob - blob.mean(), torch.randn(rows * cols) * 1e-3])[None].cuda()
    lam = torch.tensor([[0.0, 0.1, 0.9]]).cuda()
    seg = spectral.multi_region_segments(lam, vec, (rows, cols), non_adaptive_num_segments=2)[0].cpu().numpy()
    assert seg.shape == (rows, cols) and set(np.unique(seg).tolist()) == {0, 1}
    assert np.array_equal(seg == 1, blob.reshape(rows, cols).numpy() > 0)     # background (the border's owner) is 0


def _bbox_dirs(tmp_path, g, only=None):
    from PIL import Image

    (tmp_path / "s").mkdir()
    for name, kind, hw, factor, K, seed, kw in _cases(g):
        if only is None or name == only:
            _write_case(tmp_path, g, name, kind, hw, factor, K)
            Image.fromarray(g[f"{name}__png"]).save(tmp_path / "s" / f"{name}.png")


@pytest.mark.parametrize("bcase", range(4))
def test_extract_bboxes_matches_reference(tmp_path, bcase):
    """extract/extract.py:429-495 on the reference's own segmentation PNGs: same list of dicts in one .pth file."""
    g = np.load(GOLDEN / "consumers.npz")
    bname, kw = json.loads(str(g["bbox_cases"]))[bcase]
    _bbox_dirs(tmp_path, g)
    out = tmp_path / "boxes" / "b.pth"
    extract.extract_bboxes(features_dir=str(tmp_path / "f"), segmentations_dir=str(tmp_path / "s"),
                           output_file=str(out), **kw)
    got = torch.load(out, weights_only=False)
    want = json.loads(str(g[f"bboxes__{bname}"]))
    assert len(got) == len(want) == 6
    for a, b in zip(got, want):
        assert set(a) == {"bboxes", "bboxes_original_resolution", "segment_indices", "id", "format"}
        assert a["id"] == b["id"] and a["format"] == b["format"] == "(xmin, ymin, xmax, ymax)"
        assert a["segment_indices"] == b["segment_indices"]
        assert a["bboxes"] == b["bboxes"], (a["id"], a["bboxes"], b["bboxes"])
        assert a["bboxes_original_resolution"] == b["bboxes_original_resolution"]


def test_extract_bboxes_downsample_factor(tmp_path):
    g = np.load(GOLDEN / "consumers.npz")
    _bbox_dirs(tmp_path, g, only="upsampled2x")
    extract.extract_bboxes(str(tmp_path / "f"), str(tmp_path / "s"), str(tmp_path / "b.pth"), downsample_factor=8)
    got = torch.load(tmp_path / "b.pth", weights_only=False)
    want = json.loads(str(g["bboxes__upsampled2x_ds8"]))
    assert [d["bboxes_original_resolution"] for d in got] == [d["bboxes_original_resolution"] for d in want]
    assert all(v % 8 == 0 for d in got for b in d["bboxes_original_resolution"] for v in b)


def test_morphology_and_components_follow_skimage_conventions():
    m = np.zeros((7, 9), bool)
    m[0:3, 0:3] = True            # touches the border: skimage's erosion treats outside as set -> corner survives
    m[4, 5] = True
    er = extract_utils.erode_or_dilate_mask(m, 1, erode=True)
    assert er[0, 0] and er[1, 1] and not er[2, 2] and not er[4, 5] and er.sum() == 4
    assert extract_utils.erode_or_dilate_mask(m, 50, erode=True).sum() > 0      # never erodes everything away
    di = extract_utils.erode_or_dilate_mask(m, 1, erode=False)
    assert di[3, 5] and di[4, 4] and not di[3, 4]                               # cross footprint, not a square
    # 8-connectivity: diagonal neighbours are one component
    c = np.zeros((5, 5), bool)
    c[0, 0] = c[1, 1] = c[2, 2] = True
    c[4, 0] = c[4, 1] = True
    assert extract_utils.get_largest_cc(c).sum() == 3
    with pytest.raises(ValueError):
        extract_utils.get_largest_cc(np.zeros((3, 3), bool))
    seg = np.array([[1, 1, 2], [0, 5, 2], [0, 0, 2]])
    idx, frac = extract_utils.get_border_fraction(seg)
    assert idx.tolist() == [0, 1, 2, 5] and np.allclose(frac * 12, [4, 3, 5, 0])   # corners count twice


def test_bbox_from_patch_mask_matches_reference():
    """object-localization/object_discovery.py:85-126 on 60 seeded masks (patch 8 / 16 / upsampled grids, inverted,
    empty and full masks, sizes where two grids are possible)."""
    g = np.load(GOLDEN / "localization.npz")
    for c in json.loads(str(g["cases"])):
        hl, wl = c["mask_hw"]
        m = np.unpackbits(np.array(c["mask"], np.uint8))[:hl * wl].astype(bool)
        pred = object_discovery.get_bbox_from_patch_mask(torch.from_numpy(m), tuple(c["size"]))
        assert pred.tolist() == c["pred"], (c["size"], c["kind"], pred, c["pred"])
    with pytest.raises(ValueError):
        object_discovery.get_bbox_from_patch_mask(torch.zeros(13, dtype=torch.bool), (3, 224, 224))
    boxes = torch.from_numpy(g["iou_boxes"])
    for i in range(4):
        assert torch.allclose(object_discovery.bbox_iou(boxes[i], boxes), torch.from_numpy(g["ious"][i]), rtol=0, atol=1e-12)


def test_consumer_commands_keep_the_reference_signatures():
    sig = inspect.signature(extract.extract_multi_region_segmentations).parameters
    assert list(sig) == ["features_dir", "eigs_dir", "output_dir", "adaptive", "non_adaptive_num_segments",
                         "infer_bg_index", "kmeans_baseline", "num_eigenvectors", "multiprocessing"]
    assert (sig["adaptive"].default, sig["non_adaptive_num_segments"].default, sig["infer_bg_index"].default,
            sig["kmeans_baseline"].default, sig["num_eigenvectors"].default) == (False, 4, True, False, 1_000_000)
    sig = inspect.signature(extract.extract_bboxes).parameters
    assert list(sig) == ["features_dir", "segmentations_dir", "output_file", "num_erode", "num_dilate", "skip_bg_index",
                         "downsample_factor"]
    assert (sig["num_erode"].default, sig["num_dilate"].default, sig["skip_bg_index"].default) == (2, 3, True)
    assert list(inspect.signature(extract.extract_bbox_features).parameters)[:4] == \
        ["images_root", "bbox_file", "model_name", "output_file"]
    for cmd in ("extract_multi_region_segmentations", "extract_bboxes", "extract_bbox_features"):
        assert cmd in extract.COMMANDS
    fn, kw = extract.parse_cli(["extract_bboxes", "--features_dir", "f", "--segmentations_dir", "s", "--num_erode", "2",
                                "--num_dilate", "5", "--output_file", "o.pth"])
    assert fn is extract.extract_bboxes and kw["num_dilate"] == 5


# ------------------------------------------------------------------------------------------------ GPU
def _np_knn_affinity(image, n_neighbors=(20, 10), distance_weights=(2.0, 0.1)):
    """extract_utils.py:150-189 restated on the host with an exact kd-tree (test-side check of the GPU builder)."""
    import scipy.sparse
    from scipy.spatial import cKDTree

    h, w = image.shape[:2]
    r, g, b = image.reshape(-1, 3).T
    n = w * h
    x = np.tile(np.linspace(0, 1, w), h)
    y = np.repeat(np.linspace(0, 1, h), w)
    i, j = [], []
    for k, dw in zip(n_neighbors, distance_weights):
        f = np.stack([r, g, b, dw * x, dw * y], axis=1, out=np.zeros((n, 5), dtype=np.float32))
        _, nb = cKDTree(f).query(f, k=k)
        i.append(np.repeat(np.arange(n), k))
        j.append(nb.flatten())
    ij, ji = np.concatenate(i + j), np.concatenate(j + i)
    return np.asarray(scipy.sparse.csr_matrix((np.ones(2 * sum(n_neighbors) * n), (ij, ji)), (n, n)).todense())


def _device(name):
    if name == "cuda" and not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device(name)


@pytest.mark.parametrize("hw,dev", [((14, 14), "cpu"), ((18, 22), "cpu"),
                                    pytest.param((14, 14), "cuda", marks=pytest.mark.gpu),
                                    pytest.param((18, 22), "cuda", marks=pytest.mark.gpu),
                                    pytest.param((30, 30), "cuda", marks=pytest.mark.gpu)])
def test_knn_affinity_matches_kdtree(hw, dev):
    """The colour-affinity builder is plain device-agnostic tensor code: the same check on the host (every round's CPU
    suite) and on the GPU."""
    dev = _device(dev)
    img = synthetic.synthetic_image(31, hw[0], hw[1]).astype(np.float64) / 255.0
    got = extract_utils.knn_affinity(torch.from_numpy(img).to(dev)).cpu().numpy()
    want = _np_knn_affinity(img)
    assert got.shape == want.shape and np.array_equal(got, got.T)
    # identical to the kd-tree's graph except in rows where the k-th and (k+1)-th neighbour are at EXACTLY the same
    # distance (8-bit colours on a regular grid): there the kd-tree's traversal order decides, here the lower index
    tied = np.zeros(got.shape[0], bool)
    h, w = hw
    x, y = np.tile(np.linspace(0, 1, w), h), np.repeat(np.linspace(0, 1, h), w)
    for k, dw in ((20, 2.0), (10, 0.1)):
        f = np.concatenate([img.reshape(-1, 3), (dw * x)[:, None], (dw * y)[:, None]], 1).astype(np.float32).astype(np.float64)
        d2 = np.sort(((f[:, None] - f[None]) ** 2).sum(-1), axis=1)
        tied |= d2[:, k - 1] == d2[:, k]
    bad = np.argwhere(got != want)
    assert all(tied[i] or tied[j] for i, j in bad), f"{len(bad)} entries differ outside tied rows"
    assert len(bad) <= 4 * tied.sum()
    assert np.all(np.diag(got) == 4) and got.sum() == 2 * 30 * hw[0] * hw[1]
    with pytest.raises(ValueError):
        extract_utils.knn_affinity(torch.zeros(3, 3, 3, device=dev))      # 9 pixels, 20 neighbours asked


@pytest.mark.parametrize("dev", ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)])
def test_rw_affinity_matches_pixel_loop(dev):
    """Scalar restatement of pymatting's `_rw_laplacian` loop (window r = 1, clamped coordinates, duplicates add, the
    hard-coded exp(-900 |dI|^2) - `sigma` is an unused argument there and here)."""
    dev = _device(dev)
    img = synthetic.synthetic_image(32, 9, 7).astype(np.float64) / 255.0
    h, w = img.shape[:2]
    want = np.zeros((h * w, h * w))
    for y in range(h):
        for x in range(w):
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    y2, x2 = max(0, min(h - 1, y + dy)), max(0, min(w - 1, x + dx))
                    want[x + y * w, x2 + y2 * w] += np.exp(-900 * np.linalg.norm(img[y, x] - img[y2, x2]) ** 2)
    got = extract_utils.rw_affinity(torch.from_numpy(img).to(dev)).cpu().numpy()
    assert np.allclose(got, want, rtol=1e-6, atol=1e-9) and np.allclose(got, got.T)
    assert np.array_equal(got, extract_utils.rw_affinity(torch.from_numpy(img).to(dev), sigma=0.1).cpu().numpy())


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["knn_lambda10", "knn_lambda1_lapnorm_false", "knn_lambda10_upsample8"])
def test_color_affinity_eigs_match_reference(tmp_path, name):
    """extract/extract.py:197-240 with image_color_lambda > 0, through the CLI-level function: W_feat / max + lambda *
    W_knn built on the GPU, packed, solved by the same Lanczos kernel; compared with the reference's own output."""
    import ast
    from tests.util import check_eigs

    g = np.load(GOLDEN / f"color_{name}.npz")
    hw, K, kw = tuple(g["hw"]), int(g["K"]), ast.literal_eval(str(g["kwargs"]))
    feats = synthetic.synthetic_features("blobs", hw[0] * hw[1], 384, int(g["feature_seed"]), hw)
    (tmp_path / "f").mkdir(), (tmp_path / "i").mkdir(), (tmp_path / "o").mkdir()
    (tmp_path / "i" / f"{name}.jpg").write_bytes(g["jpeg"].tobytes())
    torch.save({"k": torch.from_numpy(feats)[None], "indices": torch.tensor(0), "file": f"{name}.jpg", "id": name,
                "model_name": "dino_vits16", "patch_size": 16, "shape": tuple(int(v) for v in g["shape"])},
               tmp_path / "f" / f"{name}.pth")
    extract.extract_eigs(images_root=str(tmp_path / "i"), features_dir=str(tmp_path / "f"),
                         output_dir=str(tmp_path / "o"), which_matrix="laplacian", K=K, **kw)
    out = torch.load(tmp_path / "o" / f"{name}.pth", weights_only=False)
    vec, lam = out["eigenvectors"].numpy(), np.asarray(out["eigenvalues"])
    assert vec.shape == g["eigenvectors"].shape and out["eigenvectors"].dtype == torch.float32
    scale = max(1.0, float(np.abs(g["eigenvalues"]).max()))
    check_eigs(vec, lam / scale, g["eigenvectors"], g["eigenvalues"] / scale, what=name, lam_tol=2e-5)
    ratio = np.linalg.norm(vec, axis=1) / np.linalg.norm(g["eigenvectors"], axis=1)
    assert np.allclose(ratio, 1.0, atol=1e-3), ratio      # ARPACK's normalisation (v^T D v = 1 / unit vectors) kept


@pytest.mark.gpu
def test_bbox_features_match_reference(tmp_path):
    """extract/extract.py:498-544: CLS output (12 blocks + final LayerNorm) of every box crop.  The reference ran in
    fp32; the kernels use f16 operands with fp32 accumulation: compared by cosine and by relative L2."""
    g = np.load(GOLDEN / "bbox_features.npz")
    (tmp_path / "i").mkdir()
    bbox_list = []
    for name in g["names"]:
        (tmp_path / "i" / f"{name}.jpg").write_bytes(g[f"{name}__jpeg"].tobytes())
        boxes = g[f"{name}__boxes"].tolist()
        bbox_list.append({"id": str(name), "bboxes_original_resolution": boxes, "bboxes": [[v // 16 for v in b] for b in boxes]})
    torch.save(bbox_list, tmp_path / "b.pth")
    sd = synthetic.synthetic_state_dict(str(g["model"]), int(g["weight_seed"]), float(g["ln_jitter"]))
    torch.save(sd, tmp_path / "w.pth")
    extract.extract_bbox_features(images_root=str(tmp_path / "i"), bbox_file=str(tmp_path / "b.pth"),
                                  model_name=str(g["model"]), output_file=str(tmp_path / "o.pth"),
                                  weights=str(tmp_path / "w.pth"))
    out = torch.load(tmp_path / "o.pth", weights_only=False)
    assert [d["id"] for d in out] == [str(n) for n in g["names"]]
    for d in out:
        got, want = d["features"].double().numpy(), g[f"{d['id']}__features"].astype(np.float64)
        assert d["features"].dtype == torch.float32 and got.shape == want.shape
        rel = np.linalg.norm(got - want, axis=1) / np.linalg.norm(want, axis=1)
        assert rel.max() < 5e-3, rel
        # the crops must be told apart: differences between boxes are reproduced, not just the common mean
        dw, dg = want - want.mean(0), got - got.mean(0)
        cos = (dw * dg).sum(1) / (np.linalg.norm(dw, axis=1) * np.linalg.norm(dg, axis=1))
        assert cos.min() > 0.99, cos


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["laplacian", "affinity"])
def test_inline_localization_eigenvectors(which):
    """object-localization/object_discovery.py:16-41 (features NOT normalised there) through the GPU solver, against
    the same lines restated with scipy."""
    from scipy.sparse.linalg import eigsh
    from oracle.spectral_ref import cos_err

    feats = synthetic.synthetic_features("blobs", 196, 384, 21, (14, 14)) * np.linspace(0.5, 2.0, 196, dtype=np.float32)[:, None]
    vec = object_discovery.get_eigenvectors_from_features(torch.from_numpy(feats).cuda()[None], which, K=3).cpu().numpy()
    a = (feats @ feats.T).astype(np.float32)
    if which == "affinity":
        _, v = eigsh(a, which="LM", k=3)
        want = v[:, ::-1].T
    else:
        w = a * (a > 0)
        w = w / w.max()
        d = w @ np.ones(w.shape[0])
        d[d < 1e-12] = 1.0
        _, v = eigsh(np.diag(d) - w, k=3, sigma=0, which="LM", M=np.diag(d))
        want = v.T
    assert vec.shape == (3, 196)
    assert cos_err(vec, want).max() < 1e-4, cos_err(vec, want)
    mask = torch.from_numpy(vec[1 if which == "laplacian" else 0] > 0)
    box = object_discovery.get_bbox_from_patch_mask(mask, (3, 224, 224))
    assert box.shape == (4,) and box[2] > box[0] and box[3] > box[1] and box[2] <= 224


@pytest.mark.gpu
@pytest.mark.parametrize("n", [64, 196, 713])
def test_dense_affinity_packing_round_trip_and_solver_agreement(n):
    """affinity_from_dense is the inverse of affinity_to_dense, and the dense-W entry point of the solver gives what the
    feature entry point gives for the same matrix."""
    from dss_amd import hip, spectral
    from oracle.spectral_ref import cos_err

    feats = torch.from_numpy(synthetic.synthetic_features("blobs", n, 384, 5, {64: (8, 8), 196: (14, 14), 713: (23, 31)}[n])).cuda()[None]
    wp = hip.affinity_split(feats, True, True)
    dense = hip.affinity_to_dense(wp, n)
    stored = hip._wsym_index(n)[2].numel()             # the last block of edge mini tiles is padded: never read, never written
    assert stored <= wp.shape[1] < stored + 4096
    assert torch.equal(hip.affinity_from_dense(dense[:, :n, :n].contiguous())[:, :stored], wp[:, :stored])
    w = spectral.feature_affinity_dense(feats)
    assert w.shape == (1, n, n) and float(w.max()) == 1.0
    for problem in ("laplacian", "laplacian_unnormalized"):
        ev1, vec1, info1 = spectral.eigs_from_dense_affinity(w, 4, problem)
        ev2, vec2, info2 = spectral.laplacian_eigs_from_features(feats, 4, problem=problem, w_dtype="f32")
        assert int(info1[0]) > 0 and int(info2[0]) > 0
        assert torch.allclose(ev1, ev2, atol=2e-5 * max(1.0, float(ev2.abs().max())))
        assert cos_err(vec1[0].cpu().numpy(), vec2[0].cpu().numpy()).max() < 1e-5


# ------------------------------------------------------------------------------------------------ on-device segmentation
@pytest.mark.gpu
def test_fiedler_mask_on_device_matches_reference_png():
    """extract.py:383-407 straight from device-resident eigenvectors: the same 0 / 255 mask as the PNG the reference wrote
    (tests/golden/single_region.npz)."""
    from dss_amd import spectral

    g = np.load(GOLDEN / "single_region.npz")
    vec = torch.from_numpy(g["eigenvectors"])[None].cuda()
    mask = spectral.single_region_masks(vec).cpu().numpy()
    hp, wp = int(g["shape"][2]) // int(g["patch"]), int(g["shape"][3]) // int(g["patch"])
    assert mask.dtype == np.uint8 and np.array_equal(mask.reshape(hp, wp), g["png"])
    thr = spectral.single_region_masks(vec, threshold=0.3).cpu().numpy()
    assert np.array_equal(thr[0] > 0, g["eigenvectors"][1] > 0.3)


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(6))
def test_kmeans_segments_on_device_matches_sklearn_from_the_same_centres(case):
    """extract.py:283-352 on the device.  The reference's KMeans is unseeded, so its labels are a random variable; what
    can be pinned is the ALGORITHM: from the same initial centres (sklearn's own k-means++ draw) Lloyd must reach the
    same partition as sklearn.cluster.KMeans - same labels (points equidistant to rounding may flip: >= 99.5 %), same
    inertia - and the border vote must be the reference's (extract_utils.get_border_fraction) on those labels."""
    from sklearn.cluster import KMeans, kmeans_plusplus
    from dss_amd import hip

    g = np.load(GOLDEN / "consumers.npz")
    name, kind, hw, factor, K, seed, kw = _cases(g)[case]
    vec = g[f"{name}__eigenvectors"]
    grid = (hw[0] * factor, hw[1] * factor)
    k = 3 if case % 2 else 4
    dims = min(kw.get("num_eigenvectors", 1_000_000), vec.shape[0] - 1)
    X = np.ascontiguousarray(vec[1:1 + dims].T)
    centres, _ = kmeans_plusplus(X, k, random_state=seed)
    ref = KMeans(n_clusters=k, init=centres, n_init=1).fit(X)
    dv = torch.from_numpy(vec)[None].cuda()
    lab, inertia, iters = hip.kmeans_segments(dv, k, first=1, dims=dims, infer_bg=False,
                                              init=torch.from_numpy(centres.astype(np.float32))[None].cuda())
    lab = lab[0].cpu().numpy()
    assert lab.max() < k and 0 < int(iters[0]) <= 300
    assert (lab == ref.labels_).mean() >= 0.995, (lab != ref.labels_).sum()
    assert abs(float(inertia[0]) - ref.inertia_) <= 1e-3 * ref.inertia_ + 1e-9
    # border vote + label swap, against the reference's own lines applied to the same labels
    voted, _, _ = hip.kmeans_segments(dv, k, first=1, dims=dims, grid=grid, infer_bg=True,
                                      init=torch.from_numpy(centres.astype(np.float32))[None].cuda())
    seg = lab.reshape(grid).astype(np.int64).copy()
    idx, frac = extract_utils.get_border_fraction(seg)
    bg = idx[np.argmax(frac)].item()
    bg_region, zero_region = seg == bg, seg == 0
    seg[bg_region] = 0
    seg[zero_region] = bg
    assert np.array_equal(voted[0].cpu().numpy().reshape(grid), seg)


@pytest.mark.gpu
def test_kmeans_segments_default_seeding_batches_and_adaptive():
    """k-means++ seeding on the device: deterministic, one launch for a batch == per-image launches, the clustering is as
    good as sklearn's (inertia within 15 % of its best of 5 runs), and the `adaptive` rule picks the reference's count."""
    from sklearn.cluster import KMeans
    from dss_amd import hip, spectral

    g = np.load(GOLDEN / "consumers.npz")
    names = [c[0] for c in _cases(g)[:2]]                      # two 14 x 14 cases with K = 6 and K = 8: use 5 vectors each
    vecs = torch.stack([torch.from_numpy(g[f"{n}__eigenvectors"][:6]) for n in names]).cuda()
    a = hip.kmeans_segments(vecs, 4, seed=11)
    b = hip.kmeans_segments(vecs, 4, seed=11)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for i in range(2):
        one = hip.kmeans_segments(vecs[i:i + 1].contiguous(), 4, seed=11)
        # the generator is keyed by (seed, image index in the launch, draw): image 0 of a launch of one == image 0 of the batch
        if i == 0:
            assert torch.equal(one[0][0], a[0][0])
        X = vecs[i, 1:].T.cpu().numpy()
        best = min(KMeans(n_clusters=4, n_init=1, random_state=s).fit(X).inertia_ for s in range(5))
        assert float(a[1][i]) <= 1.15 * best + 1e-9, (float(a[1][i]), best)
        assert set(np.unique(a[0][i].cpu().numpy()).tolist()) <= {0, 1, 2, 3}
    lam = torch.stack([torch.from_numpy(g[f"{n}__eigenvalues"][:6]) for n in names]).cuda()
    seg = spectral.multi_region_segments(lam, vecs, (14, 14), adaptive=True, seed=3)
    assert seg.shape == (2, 14, 14) and seg.dtype == torch.uint8
    for i in range(2):
        ev = lam[i].cpu().numpy()
        order = np.argsort(np.diff(ev))[::-1]
        want_k = int(order[order != 0][0]) + 1
        assert len(np.unique(seg[i].cpu().numpy())) <= want_k
        idx, frac = extract_utils.get_border_fraction(seg[i].cpu().numpy())
        assert idx[np.argmax(frac)] == 0                        # the segment owning most of the border is 0


def test_pthfast_reads_feature_files_without_torch_semantics_lost(tmp_path):
    """The torch-free reader of the extract_eigs loader processes against torch.load on the reference's feature schema
    (extract/extract.py:98-110), including half features, the strided qkv view a CPU run of the reference saves, a
    non-float tensor (reported, not mis-read) and a foreign pickled class (reported)."""
    import mmap
    import os

    import numpy as np
    import torch

    from dss_amd import pthfast

    files = []
    for i, (n, d, dt) in enumerate([(48, 384, torch.float32), (35, 64, torch.float32), (48, 384, torch.float16)]):
        k = torch.randn(1, n, d).to(dt)
        f = tmp_path / f"f{i}.pth"
        torch.save({"k": k, "indices": torch.tensor(7 + i), "file": f"im{i}.jpg", "id": f"im{i}", "model_name": "dino_vits16",
                    "patch_size": 16, "shape": (1, 3, 96, 128)}, f)
        files.append((str(f), k))
    # what the reference saves when it runs on the CPU (extract/extract.py:96-98): k is a strided view into the whole
    # qkv activation [B, T, 3, h, dh] - CLS row dropped, the K third of every token row - and torch.save stores that storage
    qkv = torch.randn(1, 1 + 20, 3 * 6 * 8)
    k_view = qkv.reshape(1, 21, 3, 6, 8).permute(2, 0, 3, 1, 4)[1].transpose(1, 2).reshape(1, 21, -1)[:, 1:, :]
    assert not k_view.is_contiguous() and k_view.shape == (1, 20, 48)
    ref_like = tmp_path / "ref_like.pth"
    torch.save({"k": k_view, "indices": torch.tensor(3), "file": "r.jpg", "id": "r", "model_name": "dino_vits16",
                "patch_size": 16, "shape": (1, 3, 64, 80)}, ref_like)
    files.append((str(ref_like), k_view.contiguous()))
    strided = tmp_path / "strided.pth"
    torch.save({"k": torch.arange(64 * 48, dtype=torch.int32).reshape(1, 64, 48), "file": "s.jpg", "patch_size": 16,
                "shape": (1, 3, 96, 128)}, strided)    # integer "features": not read here, reported
    foreign = tmp_path / "foreign.pth"
    torch.save({"k": torch.randn(1, 4, 4), "extra": np.arange(3), "file": "g.jpg", "patch_size": 16, "shape": (1, 3, 32, 32)}, foreign)
    block, size = tmp_path / "block", 1 << 20
    with open(block, "wb") as fh:
        fh.truncate(size)
    out = pthfast.load_chunk(str(block), size, [f for f, _ in files] + [str(strided), str(foreign)], "k")
    assert len(out) == 6
    with open(block, "r+b") as fh:
        m = mmap.mmap(fh.fileno(), size)
    expect_off = 0
    for (f, k), (meta, off, shape) in zip(files, out[:4]):
        assert off == expect_off and shape == tuple(k.shape[1:])
        got = np.frombuffer(m, dtype=np.float32, count=k.numel(), offset=off).reshape(shape)
        assert np.array_equal(got, k[0].float().numpy())
        ref = torch.load(f, weights_only=True)
        assert meta == {"indices": int(ref["indices"]), "file": ref["file"], "id": ref["id"], "model_name": ref["model_name"],
                        "patch_size": 16, "shape": tuple(ref["shape"])}
        expect_off += 4 * k.numel()
    assert out[4][0] is None and out[4][1] == str(strided) and "dtype" in out[4][2]
    assert out[5][0] is None and out[5][1] == str(foreign) and "numpy" in out[5][2]
    # a block too small for the chunk: the overflow is reported per file, nothing is written past the end
    small = pthfast.load_chunk(str(block), 4 * 48 * 384 + 16, [files[0][0], files[1][0]], "k")
    assert small[0][0] is not None and small[1][0] is None and "block full" in small[1][2]
    assert "torch" not in pthfast.__dict__


def test_worker_process_pump_runs_without_a_gpu(tmp_path, monkeypatch):
    """`extract._pump_chunks` - torch-free loader processes filling /dev/shm blocks, block recycling in chunk order - with
    the stream events it uses on a GPU replaced by no-ops: every file's feature rows come back intact, in order, through
    fewer blocks than there are chunks."""
    import numpy as np
    import torch

    from dss_amd import extract, pthfast

    class _NoEvent:
        def record(self):
            pass

        def synchronize(self):
            pass

    monkeypatch.setattr(torch.cuda, "Event", _NoEvent)
    files, want = [], []
    for i in range(37):
        n = 20 + (i % 3)          # mixed shapes inside a chunk
        k = torch.randn(1, n, 16)
        f = tmp_path / f"{i:03d}.pth"
        torch.save({"k": k, "indices": torch.tensor(i), "file": f"{i:03d}.jpg", "id": f"{i:03d}", "model_name": "m",
                    "patch_size": 16, "shape": (1, 3, 64, 80)}, f)
        files.append(str(f)), want.append(k[0].numpy().copy())
    chunks = [files[s:s + 4] for s in range(0, len(files), 4)]     # 10 chunks through 2 + 2 blocks
    got, blocks_seen = [], set()
    for entries, block, release in extract._pump_chunks(pthfast.load_chunk, chunks, ("k",), 2, 4 * 4096):
        blocks_seen.add(block.data_ptr())
        for meta, off, shape in entries:
            assert meta is not None, (off, shape)
            rows = block[off:off + 4 * shape[0] * shape[1]].view(torch.float32).view(shape).clone()
            got.append((meta["indices"], rows.numpy()))
        release()
    assert [i for i, _ in got] == list(range(37)) and len(blocks_seen) <= 4
    for (i, rows), ref in zip(got, want):
        assert np.array_equal(rows, ref), i
    import os
    assert not [n for n in os.listdir("/dev/shm") if n.startswith(f"dss_{os.getpid()}_")]


def test_pthfast_reads_the_reference_written_feature_file(tmp_path, golden_dir):
    """tests/golden/ref_feature_file.pth is the file the REFERENCE's extract_features wrote on the CPU (made by
    `oracle/make_golden.py feature_file`): `k` is a strided view of the whole qkv activation.  The torch-free reader
    must return exactly what torch.load returns, and the loader entry point what `_load_features` returns."""
    import mmap

    import numpy as np
    import torch

    from dss_amd import extract, pthfast

    path = golden_dir / "ref_feature_file.pth"
    ref = torch.load(path, map_location="cpu", weights_only=True)
    assert not ref["k"].is_contiguous() and ref["k"].shape == (1, 12, 384)      # what this fixture is for
    with pthfast.PthFile(str(path)) as f:
        got = f.read(f.obj["k"])
        assert f.obj["file"] == ref["file"] and f.obj["patch_size"] == 16 and tuple(f.obj["shape"]) == tuple(ref["shape"])
    assert np.array_equal(got, ref["k"].numpy())
    block, size = tmp_path / "block", 1 << 16
    with open(block, "wb") as fh:
        fh.truncate(size)
    (meta, off, shape), = pthfast.load_chunk(str(block), size, [str(path)], "k")
    _, feats = extract._load_features(str(path), "k")
    with open(block, "r+b") as fh:
        m = mmap.mmap(fh.fileno(), size)
    assert shape == (12, 384) and np.array_equal(np.frombuffer(m, dtype=np.float32, count=12 * 384, offset=off).reshape(shape),
                                                 feats.numpy())
    assert meta["id"] == ref["id"] and meta["indices"] == int(ref["indices"])
