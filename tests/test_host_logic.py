"""CPU: host-side logic of the drop-in boundary (no kernels run)."""
import os
import re
from pathlib import Path

import numpy as np
import pytest
import torch

import dss_amd  # noqa: F401
from dss_amd import distributed, extract, extract_utils, hip, spectral, synthetic

REPO = Path(__file__).resolve().parents[1]


def test_abi_library_loads_and_exports_every_declared_symbol():
    header = (REPO / "include" / "dss_hip.h").read_text()
    declared = set(re.findall(r"\b(dss_[a-z0-9_]+)\s*\(", header))
    assert declared == set(hip.SYMBOLS), declared ^ set(hip.SYMBOLS)
    lib = hip.load_library()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.dss_abi_version() == 10 and lib.dss_target_arch() == b"gfx950"
    assert lib.dss_affinity_ld(900) == 960 and lib.dss_affinity_ld(64) == 64
    assert lib.dss_affinity_elems(900) == 106 * 4096 and lib.dss_affinity_elems(64) == 4096 and lib.dss_affinity_elems(960) == 120 * 4096
    assert lib.dss_eigs_workspace_bytes(1, 900, 5, 0) > 0


def test_library_reads_no_environment_variables():
    """include/dss_hip.h: 'no environment variable is read: every choice of kernel variant is an argument'."""
    for f in (REPO / "deep-spectral-segmentation_amd" / "csrc").iterdir():
        assert "getenv" not in f.read_text(), f


def test_no_cpu_fallback():
    with pytest.raises(hip.HipLibraryError):
        hip.normalize_rows(torch.zeros(4, 8))
    with pytest.raises(hip.HipLibraryError):
        hip.load_library(REPO / "does_not_exist.so")


def test_product_never_imports_oracle():
    for f in (REPO / "deep-spectral-segmentation_amd").glob("*.py"):
        assert "oracle" not in f.read_text().replace("oracle/", ""), f


def test_cli_parsing_matches_fire_conventions():
    fn, kw = extract.parse_cli(["extract_eigs", "--images_root", "r", "--features_dir=f", "--output_dir", "o",
                                "--K", "5", "--normalize", "False", "--image_downsample_factor=None", "--lapnorm"])
    assert fn is extract.extract_eigs
    assert kw == dict(images_root="r", features_dir="f", output_dir="o", K=5, normalize=False,
                      image_downsample_factor=None, lapnorm=True)
    with pytest.raises(SystemExit):
        extract.parse_cli(["extract_eigs", "--images_root", "r"])  # missing required flags
    with pytest.raises(SystemExit):
        extract.parse_cli(["extract_features", "--bogus", "1"])
    import inspect
    sig = inspect.signature(extract.extract_eigs).parameters
    assert sig["K"].default == 20 and sig["image_color_lambda"].default == 0.0 and sig["which_matrix"].default == "laplacian"
    assert inspect.signature(extract._extract_eig).parameters["image_color_lambda"].default == 10
    first = list(inspect.signature(extract.extract_features).parameters)[:6]
    assert first == ["images_list", "images_root", "model_name", "batch_size", "output_dir", "which_block"]


def test_unsupported_options_fail_loudly():
    with pytest.raises(NotImplementedError):
        extract._check_eig_options("affinity_torch", True, 0.0, None, 16)
    assert extract._check_eig_options("laplacian", True, 10.0, None, 16) == "laplacian"   # colour fusion is built
    assert extract._color_spec("laplacian", "knn", 10.0, "root") == (10.0, "knn", "root")
    assert extract._color_spec("laplacian", "knn", 0.0, "root") is None
    assert extract._color_spec("affinity", "knn", 10.0, "root") is None      # the affinity branches ignore colour
    with pytest.raises(ValueError):
        extract._color_spec("laplacian", "bogus", 1.0, "root")
    assert extract._lr_grid({"patch_size": 16, "shape": (1, 3, 375, 500)}, 8) == (46, 62)
    assert extract._lr_grid({"patch_size": 16, "shape": (1, 3, 375, 500)}, None) == (23, 31)
    assert extract._check_eig_options("laplacian", True, 0.0, 8, 16) == "laplacian"   # upsampling is built
    dd = {"patch_size": 16, "shape": (1, 3, 375, 500)}
    assert extract._upsample_spec(dd, "laplacian", 8) == ((23, 31), (46, 62))
    assert extract._upsample_spec(dd, "laplacian", 16) is None and extract._upsample_spec(dd, "laplacian", None) is None
    assert extract._upsample_spec(dd, "affinity", 8) is None
    with pytest.raises(ValueError):
        extract._check_eig_options("bogus", True, 0.0, None, 16)
    assert extract._check_eig_options("matting_laplacian", True, 0.0, 16, 16) == "laplacian"
    assert extract._check_eig_options("laplacian", False, 0.0, None, 16) == "laplacian_unnormalized"
    assert extract._check_eig_options("affinity", True, 10.0, None, 16) == "affinity"
    assert extract._check_eig_options("affinity_svd", True, 0.0, None, 16) == "affinity_svd"


def test_dataset_order_and_image_sizes(tmp_path):
    from PIL import Image

    for fn, (h, w) in {"b.png": (20, 30), "a.png": (17, 40)}.items():
        Image.fromarray(synthetic.synthetic_image(1, h, w)).save(tmp_path / fn)
    ds = extract_utils.ImagesDataset(["b.png", "a.png", "b.png"], images_root=str(tmp_path))
    assert ds.filenames == ["a.png", "b.png"] and len(ds) == 2
    img, path, idx = ds[0]
    assert img.dtype == torch.uint8 and tuple(img.shape) == (17, 40, 3) and path == "a.png" and idx == 0
    assert np.array_equal(img.numpy(), synthetic.synthetic_image(1, 17, 40))
    sizes = extract_utils.get_image_sizes({"patch_size": 16, "shape": (1, 3, 375, 500)})
    assert sizes == (1, 3, 375, 500, 16, 23, 31, 368, 496)
    with pytest.raises(AssertionError):
        extract_utils.get_image_sizes({"patch_size": 16, "shape": (2, 3, 32, 32)})


def test_feature_schema_roundtrip(tmp_path):
    d = extract._feature_dict(torch.zeros(1, 6, 8), 3, "x/img_01.jpg", "dino_vits16", 16, (1, 3, 40, 56))
    torch.save(d, tmp_path / "f.pth")
    back = torch.load(tmp_path / "f.pth", map_location="cpu", weights_only=True)
    assert sorted(back) == sorted(["k", "indices", "file", "id", "model_name", "patch_size", "shape"])
    assert back["k"].dtype == torch.float32 and back["indices"].dim() == 0 and back["indices"].dtype == torch.int64
    assert back["id"] == "img_01" and back["file"][:-4] == "x/img_01" and back["shape"] == (1, 3, 40, 56)


def test_cpu_transform_matches_oracle():
    from oracle import vit_ref

    img = synthetic.synthetic_image(3, 33, 47)
    assert torch.equal(extract_utils.get_transform("dino_vits16")(img), vit_ref.ref_preprocess(img))


def test_group_by_shape():
    groups = spectral.group_by_shape([(1, 2), (3, 4), (1, 2), (1, 2), (3, 4)], max_batch=2)
    assert groups == [[0, 2], [1, 4], [3]]


def test_pack_unpack_and_shard():
    K, N, n = 3, 7, 5
    ids = torch.tensor([9, 2, 4, 0, 7])
    val, vec = torch.randn(n, K), torch.randn(n, K, N)
    recs = distributed.unpack_records(*distributed.pack_records(ids, val, vec))
    assert [r[0] for r in recs] == ids.tolist()
    assert all(torch.equal(r[1], val[i]) and torch.equal(r[2], vec[i]) for i, r in enumerate(recs))
    assert distributed.shard_indices(10, 1, 4) == [1, 5, 9]
    allidx = sorted(sum((distributed.shard_indices(10, r, 4) for r in range(4)), []))
    assert allidx == list(range(10))
    meta, payload = distributed.gather_records_to_root(*distributed.pack_records(ids, val, vec))   # world 1: sorts
    assert meta[:, 0].tolist() == [0, 2, 4, 7, 9] and meta.dtype == torch.int64
    assert torch.equal(distributed.unpack_records(meta, payload)[0][2], vec[3])


def test_synthetic_inputs_are_portable():
    a = synthetic.synthetic_image(7, 48, 64)
    assert a.dtype == np.uint8 and a.shape == (48, 64, 3) and int(a.astype(np.int64).sum()) == int(synthetic.synthetic_image(7, 48, 64).astype(np.int64).sum())
    sd = synthetic.synthetic_state_dict("dino_vitb8", 0)
    assert sd["pos_embed"].shape == (1, 785, 768) and sd["blocks.11.attn.qkv.weight"].shape == (2304, 768)
    assert sd["patch_embed.proj.weight"].shape == (768, 3, 8, 8)


def test_bench_chunking_is_balanced_and_keeps_the_copy_pipeline_fed():
    """bench.chunk_counts: forwards of at most vit_batch images; the steady state runs whole vit_batch forwards; any other step
    (a rank's shard) is cut into whole ROUNDS of the Linear kernels' workgroups, at least four forwards while they stay above
    256 images (a one-forward step cannot hide its H2D copy), and a run's FIRST forward is the step's fractional round (its copy
    is exposed, and a partly filled round costs a whole one wherever it is)."""
    import importlib.util
    import math

    spec = importlib.util.spec_from_file_location("bench_mod", Path(__file__).resolve().parents[1] / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rnd = 256 * 512 / 901                                   # dino_vits16 at 480 x 480 on 256 CUs: 145.5 images per round
    assert bench.chunk_counts(1250, 1018, False, 0.0) == [313, 313, 312, 312]          # no round size known: balanced
    assert bench.chunk_counts(1250, 1018, True, rnd) == [88, 436, 436, 290]            # BASELINE configs[3]'s per-rank shard: 0.6 + 3 + 3 + 2 rounds
    assert bench.chunk_counts(1250, 1018, False, rnd) == [436, 290, 290, 234]
    assert bench.chunk_counts(4072, 1018, False, rnd) == [1018] * 4 == bench.chunk_counts(4072, 1018, True, rnd)
    assert bench.chunk_counts(1250, 2473, True, rnd) == [88, 436, 436, 290]            # ... also under the round-5 default forward size
    bench.SHARD_FORWARDS = 3                                                            # --shard-forwards: fewer, longer forwards
    assert bench.chunk_counts(1250, 2473, True, rnd) == [88, 581, 581] and bench.chunk_counts(1250, 2473, False, rnd) == [436, 436, 378]
    bench.SHARD_FORWARDS = 4
    assert bench.chunk_counts(2030, 290, False, rnd) == [290] * 7
    assert bench.chunk_counts(100, 1018, True, rnd) == [100] and bench.chunk_counts(600, 1018, False, 0.0) == [300, 300]
    for cnt, vb, lead in [(1, 5, False), (17, 5, True), (9999, 1018, False), (873, 291, True), (3334, 1018, True), (1018, 1018, True),
                          (1251, 1018, True), (700, 1018, False)]:
        for r in (0.0, rnd):
            c = bench.chunk_counts(cnt, vb, lead, r)
            assert sum(c) == cnt and max(c) <= vb and min(c) > 0, (cnt, vb, lead, r, c)
            if r and cnt >= 2 * r and cnt % vb:
                rounds = sum(math.ceil(x / r - 1e-9) for x in c)
                assert rounds <= math.ceil(cnt / r) + 1, (cnt, c, rounds)       # at most one round more than the work itself
                if lead:                                                            # the lead forward holds the fractional round:
                    assert all(abs(x / r - round(x / r)) < 0.02 for x in c[1:-1])   # every forward between first and last is whole rounds


def test_wave_filling_batch_picks_whole_waves_of_workgroups():
    """vit.wave_filling_batch: images per ViT forward such that ceil(b * tokens / 512) workgroups of the K-resident
    Linear kernel fill whole waves of the CUs (pure arithmetic; no GPU)."""
    import math

    from dss_amd.vit import wave_filling_batch

    b = wave_filling_batch(901, target=256, compute_units=256)          # 480x480 / patch 16 + CLS
    assert b == 290
    tiles = math.ceil(b * 901 / 512)
    assert tiles <= 512 and tiles / 512 > 0.99                           # 511 workgroups = 1.996 waves
    for tokens in (197, 401, 901, 3601):
        for cus in (64, 256, 304):
            b = wave_filling_batch(tokens, target=256, compute_units=cus)
            assert 0.85 * 256 - 1 <= b <= 1.25 * 256 + 1
            t = b * tokens / 512
            eff = t / (math.ceil(math.ceil(t) / cus) * cus)
            base = (256 * tokens / 512) / (math.ceil(math.ceil(256 * tokens / 512) / cus) * cus)
            assert eff >= base - 1e-9                                    # never worse than the plain target


@pytest.mark.parametrize("rnd", ["r02", "r03", "r04", "r05", "r06"])
def test_pmc_traffic_summary_is_reproducible_from_the_committed_counter_csvs(tmp_path, rnd):
    """profiles/rNN_pmc_traffic.json (what bench.py reads for roofline.traffic) must follow from the committed
    rocprofv3 counter summaries: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch."""
    import json
    import subprocess
    import sys

    prof = REPO / "profiles"
    out = tmp_path / "traffic.json"
    subprocess.run([sys.executable, str(REPO / "scripts" / "make_pmc_traffic.py"), str(prof / f"{rnd}_pmc_fetch.csv"),
                    str(prof / f"{rnd}_pmc_write.csv"), str(prof / f"{rnd}_bench_n1.json"), str(out)], check=True,
                   capture_output=True)
    new, old = json.loads(out.read_text()), json.loads((prof / f"{rnd}_pmc_traffic.json").read_text())
    assert new["config"] == old["config"]
    # (round 4's forward has no standalone LayerNorm launch left: its place in the table is the K hand-over kernel's)
    for key in ("attention", "laplacian_eigs", "affinity") + (("lnlinear", "linear_kres", "lnlinear_kfeatures", "patch_embed") if rnd >= "r04" else ("layernorm",)):
        assert abs(new["kernels"][key]["hbm_bytes_per_launch"] - old["kernels"][key]["hbm_bytes_per_launch"]) < 1.0
        k = old["kernels"][key]
        assert abs((2 * k["fetch_size_kb"] + k["write_size_kb"]) * 1024 - k["hbm_bytes_per_launch"]) < 2048


def test_reference_scale_puts_the_division_by_wmax_back():
    """extract.py:194 on the outputs: eigenvectors * sqrt(wmax) for the normalised Laplacian, eigenvalues / wmax for
    lapnorm=False, nothing for the affinity branches; a zero matrix is left alone."""
    import torch
    from dss_amd.spectral import reference_scale

    ev, vec = torch.tensor([[0.0, 2.0], [0.0, 6.0]]), torch.ones(2, 2, 3)
    wmax = torch.tensor([4.0, 0.0])
    e, v = reference_scale("laplacian", wmax, ev, vec)
    assert torch.equal(e, ev) and torch.equal(v[0], 2 * vec[0]) and torch.equal(v[1], vec[1])
    e, v = reference_scale("laplacian_unnormalized", wmax, ev, vec)
    assert torch.equal(e, torch.tensor([[0.0, 0.5], [0.0, 6.0]])) and torch.equal(v, vec)
    e, v = reference_scale("affinity", wmax, ev, vec)
    assert torch.equal(e, ev) and torch.equal(v, vec)


def test_dataset_applies_exif_orientation_like_cv2(tmp_path):
    """cv2.imread (reference extract_utils.py:30) honours the EXIF orientation tag; so must the PIL decode."""
    import numpy as np
    from PIL import Image

    from dss_amd import extract_utils

    arr = (np.arange(20 * 30 * 3) % 251).astype(np.uint8).reshape(20, 30, 3)
    exif = Image.Exif()
    exif[0x0112] = 6   # "rotate 90 CW to display"
    Image.fromarray(arr).save(tmp_path / "rot.png", exif=exif)
    Image.fromarray(arr).save(tmp_path / "plain.png")
    ds = extract_utils.ImagesDataset(["rot.png", "plain.png"], str(tmp_path))
    plain, rot = ds[0][0], ds[1][0]
    assert tuple(plain.shape) == (20, 30, 3) and tuple(rot.shape) == (30, 20, 3)
    assert np.array_equal(rot.numpy(), np.rot90(arr, k=-1))


def test_file_io_worker_processes_round_trip(tmp_path):
    """Large CLI runs hand whole batches to torch-free saver / loader PROCESSES (extract._FastSaver through page-locked
    /dev/shm blocks and the pool's pipe, extract._iter_features): same files as the in-process path, loadable with
    `torch.load(weights_only=True)`, errors surfaced by wait_slot() / close()."""
    k = torch.randn(5, 12, 8)
    items = [(j, 100 + j, f"im_{j}.jpg", "dino_vits16", 16, (1, 3, 48, 64), str(tmp_path / f"im_{j}.pth")) for j in range(5)]
    saver = extract._FastSaver(2)
    blk = saver.block(1, k.numel() * 4)
    blk[:k.numel() * 4].view(torch.float32).view(k.shape).copy_(k)
    saver.submit_features(1, [(j * 12 * 8 * 4, (12, 8), idx, file, m, p, shp, out) for j, idx, file, m, p, shp, out in items], chunk=2)
    ev, vec = torch.randn(5, 3), torch.randn(5, 3, 12)
    (tmp_path / "e").mkdir()
    saver.submit_batch("eigs", (ev.clone(), vec.clone()), [(j, str(tmp_path / "e" / f"im_{j}.pth"), "laplacian") for j in range(5)])
    labels = (torch.arange(5 * 12).reshape(5, 12) % 7).to(torch.uint8)
    saver.submit_batch("png", (labels,), [(j, str(tmp_path / "e" / f"im_{j}.png"), 3, 4) for j in range(5)])
    saver.wait_slot(1)
    saver.close()
    from PIL import Image
    for j in range(5):
        d = torch.load(tmp_path / f"im_{j}.pth", weights_only=True)
        want = extract._feature_dict(k[j:j + 1], 100 + j, f"im_{j}.jpg", "dino_vits16", 16, (1, 3, 48, 64))
        assert set(d) == set(want) and all(type(d[key]) is type(want[key]) for key in want)     # the reference's schema, key for key
        assert torch.equal(d["k"], k[j:j + 1]) and d["k"].dtype == torch.float32 and tuple(d["k"].shape) == (1, 12, 8)
        assert d["indices"].dtype == torch.int64 and d["indices"].dim() == 0 and int(d["indices"]) == 100 + j
        assert d["id"] == f"im_{j}" and d["file"] == f"im_{j}.jpg" and d["shape"] == (1, 3, 48, 64) and d["patch_size"] == 16
        assert d["k"].untyped_storage().nbytes() == 12 * 8 * 4          # one image per file, not the whole batch
        e = torch.load(tmp_path / "e" / f"im_{j}.pth", weights_only=True)
        assert set(e) == {"eigenvalues", "eigenvectors"} and torch.equal(e["eigenvectors"], vec[j]) and torch.equal(e["eigenvalues"], ev[j])
        assert np.array_equal(np.asarray(Image.open(tmp_path / "e" / f"im_{j}.png")), labels[j].reshape(3, 4).numpy())
    assert not list(tmp_path.glob("*.tmp*")) and not [n for n in os.listdir("/dev/shm") if n.startswith(f"dss_{os.getpid()}_")]
    # the small-run path (threads, torch.save) writes the same dicts - including the 'affinity' branch's numpy eigenvalues
    small = extract._AsyncSaver()
    (tmp_path / "s").mkdir()
    small.submit_batch("features", (k,), [(*it[:6], str(tmp_path / "s" / f"im_{it[0]}.pth")) for it in items])
    small.submit_batch("eigs", (ev, vec), [(0, str(tmp_path / "s" / "aff.pth"), "affinity")])
    small.close()
    for j in range(5):
        a, b = (torch.load(tmp_path / sub / f"im_{j}.pth", weights_only=True) for sub in (".", "s"))
        assert all(torch.equal(a[key], b[key]) if torch.is_tensor(a[key]) else a[key] == b[key] for key in a)
    assert isinstance(torch.load(tmp_path / "s" / "aff.pth", weights_only=False)["eigenvalues"], np.ndarray)
    # mixed shapes through the loader processes, every file exactly once (files written by BOTH writers)
    torch.save({"k": torch.randn(1, 7, 8), "file": "odd.jpg", "id": "odd", "patch_size": 16, "shape": (1, 3, 16, 112),
                "indices": torch.tensor(9), "model_name": "m"}, tmp_path / "zz_odd.pth")
    files = sorted(p for p in tmp_path.iterdir() if p.suffix == ".pth")
    got = {d["id"]: f for d, f in extract._iter_features(files, "k", 2, 8)}
    ref = {d["id"]: f for d, f in extract._iter_features(files, "k", 0, 8)}
    assert set(got) == set(ref) == {f"im_{j}" for j in range(5)} | {"odd"}
    assert all(torch.equal(got[i], ref[i]) for i in got) and got["odd"].shape == (7, 8)
    bad = extract._FastSaver(1)
    bad.block(0, 64)
    bad.submit_features(0, [(0, (2, 2), 0, "x.jpg", "m", 16, (1, 3, 4, 4), str(tmp_path / "no_dir" / "x.pth"))])
    with pytest.raises(Exception):
        bad.close()
    # a run that FAILS gives its /dev/shm ring back at once (the blocks used to wait for the 6 h stale sweep) ...
    dead = extract._FastSaver(1)
    dead.block(0, 4096), dead.block(3, 8192)
    mine = lambda: [n for n in os.listdir("/dev/shm") if n.startswith(f"dss_{dead.tag}_")]
    assert len(mine()) == 2
    dead.abort()
    assert not mine()
    # ... and a worker drops its mapping of a ring slot's OLD block when the slot comes back at another size
    from dss_amd import pthfast
    a, b, c = (f"/dev/shm/dss_{os.getpid()}_t_s0_{n}" for n in (4096, 8192, 4096))
    other = f"/dev/shm/dss_{os.getpid()}_t_s1_4096"
    try:
        for path, n in ((a, 4096), (b, 8192), (other, 4096)):
            with open(path, "wb") as f:
                f.truncate(n)
        pthfast._block(a, 4096), pthfast._block(other, 4096)
        assert a in pthfast._BLOCKS
        pthfast._block(b, 8192)
        assert a not in pthfast._BLOCKS and b in pthfast._BLOCKS and other in pthfast._BLOCKS
    finally:
        for path in (a, b, other):
            pthfast._BLOCKS.pop(path, None)
            if os.path.exists(path):
                os.unlink(path)


def test_bucket_batch_is_per_shape_and_inside_the_kernels_row_limits():
    """extract_features sizes every SHAPE BUCKET on its own shape (the first image's size used to set the batch of every
    bucket) and never lets a forward's token matrix leave the 32-bit limits of the kernels (M * 4 D bytes, M * T)."""
    auto = extract.bucket_batch(480, 480, 16, 384, 0, 256, 512)
    assert auto == 581                                       # four rounds of 2 x 256 workgroups of 256 rows at 901 tokens
    assert extract.bucket_batch(480, 480, 16, 384, 128, 256, 512) == 128
    for h, w, p, d in ((32, 32, 16, 384), (48, 640, 8, 768), (640, 640, 8, 768), (2048, 2048, 8, 768), (16, 16, 16, 384)):
        for want in (0, 1, 10 ** 6):
            b = extract.bucket_batch(h, w, p, d, want, 256, 512 if d == 384 else 256)
            t = (h // p) * (w // p) + 1
            assert b >= 1 and (b == 1 or (b * t * 4 * d < 2 ** 32 and b * t * t < 2 ** 32)), (h, w, p, d, want, b)
    assert extract.bucket_batch(32, 32, 16, 384, 0, 256, 512) > auto > extract.bucket_batch(960, 960, 16, 384, 0, 256, 512)


def test_torch_free_writer_matches_torch_save_for_both_schemas(tmp_path):
    """pthfast.write_pth (what the saver processes run instead of `import torch; torch.save`): for the feature and eigen schemas
    the archive loads with `torch.load(weights_only=True)` to exactly what torch.save's own file loads to - same keys, Python
    types, dtypes, shapes, values (0-dim int64 `indices`, a non-ASCII file name, an id above 2^31) - and with the torch-free
    READER of `extract_eigs` as well."""
    from dss_amd import pthfast

    k = torch.randn(1, 30, 16)
    want = extract._feature_dict(k, 2 ** 33 + 5, "sub dir/bild_ä.jpg", "dino_vitb8", 8, (1, 3, 40, 48))
    torch.save(want, tmp_path / "ref.pth")
    pthfast.write_pth(str(tmp_path / "fast.pth"), {
        "k": pthfast.TensorOut(k.numpy()), "indices": pthfast.TensorOut(np.array(2 ** 33 + 5, dtype=np.int64)),
        "file": "sub dir/bild_ä.jpg", "id": "bild_ä", "model_name": "dino_vitb8", "patch_size": 8, "shape": (1, 3, 40, 48)})
    a, b = (torch.load(tmp_path / n, weights_only=True) for n in ("ref.pth", "fast.pth"))
    assert list(a) == list(b)
    for key in a:
        assert type(a[key]) is type(b[key]), key
        if torch.is_tensor(a[key]):
            assert a[key].dtype == b[key].dtype and a[key].shape == b[key].shape and torch.equal(a[key], b[key]), key
        else:
            assert a[key] == b[key], key
    with pthfast.PthFile(str(tmp_path / "fast.pth")) as f:
        assert np.array_equal(f.read(f.obj["k"]), k.numpy()) and int(f.read(f.obj["indices"])) == 2 ** 33 + 5
    ev, vec = torch.randn(4), torch.randn(4, 30)
    pthfast.save_eigs([(ev.numpy(), vec.numpy(), str(tmp_path / "eig.pth"))])
    e = torch.load(tmp_path / "eig.pth", weights_only=True)
    assert list(e) == ["eigenvalues", "eigenvectors"] and torch.equal(e["eigenvalues"], ev) and torch.equal(e["eigenvectors"], vec)
    with pytest.raises(pthfast.Unsupported):
        pthfast.dumps_pth({"x": [1, 2]})


def test_integration_md_binding_stub_matches_the_declared_abi():
    """Every `_lib.<symbol>.argtypes = [...]` line of INTEGRATION.md's reference-side stub names an exported symbol with
    the right number of arguments (the GPU suite executes the stub; this keeps the document honest without one)."""
    text = (REPO / "INTEGRATION.md").read_text()
    block = re.search(r"```python\n(# extract/dss_binding\.py.*?)```", text, re.S).group(1)
    found = re.findall(r"_lib\.(dss_[a-z0-9_]+)\.argtypes = \[([^\]]*)\]", block)
    assert len(found) >= 5
    for name, args in found:
        assert name in hip.SYMBOLS, name
        assert len([a for a in args.split(",") if a.strip()]) == len(hip.SYMBOLS[name][1]), name
    for name in re.findall(r"_lib\.(dss_[a-z0-9_]+)\(", block):
        assert name in hip.SYMBOLS, name


def test_patch_embed16_prepare_folds_transform_and_centring_exactly():
    """hip.patch_embed16_prepare (host side of dss_patch_embed_p16): with the kernel's operand = pixel - 128 in (py, px, c) order,
    Wp . operand + biasp must BE Conv2d(Normalize(ToTensor(image))) - checked in fp64 with the unrounded folded weight - and the
    rounded weight's error must scale with the centred operand, not with the 0..255 level."""
    import torch
    import torch.nn.functional as F
    from dss_amd import hip

    g = torch.Generator().manual_seed(5)
    d = 64
    w = torch.randn(d, 3, 16, 16, generator=g) * 0.03
    b = torch.randn(d, generator=g) * 0.1
    img = torch.randint(0, 256, (3, 32, 48, 3), generator=g, dtype=torch.uint8)
    wp64, bp = hip.patch_embed16_prepare(w, b, torch.float64)
    wp16, bp16 = hip.patch_embed16_prepare(w, b, torch.float16)
    assert bp.dtype == torch.float32 and torch.equal(bp, bp16) and wp16.dtype == torch.float16 and tuple(wp16.shape) == (d, 768)
    mean = torch.tensor(hip.IMAGENET_MEAN, dtype=torch.float64).view(1, 3, 1, 1)
    std = torch.tensor(hip.IMAGENET_STD, dtype=torch.float64).view(1, 3, 1, 1)
    t = (img.permute(0, 3, 1, 2).double() / 255.0 - mean) / std
    ref = F.conv2d(t, w.double(), b.double(), stride=16).flatten(2).transpose(1, 2)            # [B, Np, D]
    # the kernel's operand rows: patch (py, px, c) order, centred
    pat = img.double().reshape(3, 2, 16, 3, 16, 3).permute(0, 1, 3, 2, 4, 5).reshape(3, 6, 768) - 128.0
    got = pat @ wp64.t() + bp.double()
    assert (got - ref).abs().max().item() < 2e-6            # biasp is stored in fp32: that rounding only
    got16 = pat @ wp16.double().t() + bp.double()
    uncentred = (pat + 128.0) @ wp16.double().t() + (b.double() - (w.double() * (mean / std)).sum(dim=(1, 2, 3)))
    assert (got16 - ref).abs().max().item() < (uncentred - ref).abs().max().item()


def test_hot_kernels_have_no_register_spills_and_no_scratch():
    """Code-object notes of the shipped library (scripts/code_object_notes.py): no kernel spills a VGPR or uses scratch memory -
    a reload inside an MFMA / DMA loop shares vmcnt with the loads it sits between and drains them.
    The product kernel file builds exactly one schedule: no lab branch is left in it."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("code_object_notes", REPO / "scripts" / "code_object_notes.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    notes = mod.kernel_notes(hip.LIB_PATH)
    hot = ("linear_kres_kernel", "kfeat_kres_kernel", "patch_embed_kres_kernel", "attn_fwd_kernel", "laplacian_eigs_kernel",
           "gram_f16_dma_kernel", "layernorm_kernel", "preprocess_patchify8", "kfeatures_finalize_kernel")
    seen = {h: 0 for h in hot}
    for name, r in notes.items():
        for h in hot:
            if h in name:
                seen[h] += 1
                assert r["vgpr_spill_count"] == 0 and r["private_segment_fixed_size"] == 0, (name, r)
    assert all(seen.values()), seen
    assert len(notes) > 50
    src = (REPO / "deep-spectral-segmentation_amd" / "csrc" / "linear384.hip").read_text()
    for lab in ("DSS_LIN_LAB", "DSS_LIN_ABL", "DSS_LINEAR_NO_BARRIER", "DSS_LIN_PLAIN_PREFETCH"):
        assert lab not in src, lab


def test_gelu_f16_poly_error_budget():
    """The packed-f16 GELU of fc1's epilogue (`gelu = 2`, csrc/kres.h; tests/util.gelu_f16_poly is the same arithmetic): its
    error against the exact erf-GELU over EVERY finite f16 input, next to what the fp32 form delivers (the correctly rounded
    f16 value).  These are the numbers DESIGN.md quotes as the parity cost of the mode."""
    from scipy.special import erf
    from tests.util import gelu_f16_poly

    allh = np.arange(0, 65536, dtype=np.uint16).view(np.float16)
    xs = allh[np.isfinite(allh)]
    x64 = xs.astype(np.float64)
    ref = 0.5 * x64 * (1.0 + erf(x64 / np.sqrt(2.0)))
    with np.errstate(over="ignore"):
        y = gelu_f16_poly(xs).astype(np.float64)
    rounded = ref.astype(np.float16).astype(np.float64)
    err = np.abs(y - ref)
    with np.errstate(over="ignore"):
        spacing = np.abs(np.spacing(ref.astype(np.float16))).astype(np.float64)
    big = np.abs(x64) > 9                                                  # beyond the polynomial's range the tail term is held at
    assert np.all(y[big & (x64 > 0)] == x64[big & (x64 > 0)])              # -1.5e-4: x itself for x > 0, -1.5e-4 (exact: ~0) for x < 0
    assert err[big & (x64 < 0)].max() <= 1.6e-4
    small = ~big
    assert err[small].max() <= 1.2e-3 and err[small & (x64 < 0)].max() <= 3.5e-4
    near0 = (np.abs(x64) < 0.5) & (np.abs(x64) > 1e-4)
    assert (err[near0] / np.abs(ref[near0])).max() <= 2.2e-3
    for lo, ulps in ((0.25, 2.1), (0.5, 1.6), (1.0, 1.1), (2.0, 0.6)):     # in units of the OUTPUT's own f16 spacing (a correctly rounded
        sel = (x64 > lo) & small                                            # value is within 0.5): at most 2.1 above 1/4, 0.6 above 2
        assert np.all(err[sel] <= ulps * spacing[sel]), lo
    assert np.mean(y[small] == rounded[small]) >= 0.5                      # more than half of the outputs ARE the correctly rounded value

    def nsr(out, sigma):                                                   # noise / signal for pre-activations ~ N(0, sigma^2)
        w = np.exp(-0.5 * (x64[small] / sigma) ** 2) * np.abs(np.spacing(xs[small])).astype(np.float64)
        return float(np.sqrt(np.sum(w * (out[small] - ref[small]) ** 2) / np.sum(w * ref[small] ** 2)))

    for sigma, bar in ((0.7, 1.5), (1.5, 1.15), (3.0, 1.12)):
        assert nsr(y, sigma) <= bar * nsr(rounded, sigma), (sigma, nsr(y, sigma), nsr(rounded, sigma))
    assert nsr(y, 1.5) < 2.3e-4 and nsr(rounded, 1.5) > 1.8e-4


def test_inline_asm_pipelines_are_not_copied_before_their_wait():
    """scripts/check_async_asm.py on the compiler's assembly of the two kernels with inline-asm software pipelines: no asm
    statement issues a register-returning global load (the round-5 race of the fused norm -> Linear kernel: hipcc copied the
    destination register of such a load in front of the asm wait that covered it), and nothing touches the destination of an asm
    LDS read between the read and the counted wait that covers it; round 6: affinity.hip is checked too, and no kernel that issues
    LDS-DMA may have scratch traffic (a spill reload shares vmcnt with the DMA pieces: the rule is exercised on a handwritten
    assembly fragment below, the product's 43 LDS-DMA kernels pass it)."""
    import shutil
    import subprocess
    import sys
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc")
    r = subprocess.run([sys.executable, str(REPO / "scripts" / "check_async_asm.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "linear384.s" in r.stdout and "attention.s" in r.stdout and "affinity.s" in r.stdout and " 0 violation(s)" in r.stdout
    assert "LDS-DMA kernels without scratch" in r.stdout
    # the rules fire on what they are meant to catch
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_async_asm", REPO / "scripts" / "check_async_asm.py")
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    bad = Path(os.environ.get("TMPDIR", "/tmp")) / f"dss_async_{os.getpid()}.s"
    bad.write_text("_Zkernel_a:\n\t;;#ASMSTART\n\tglobal_load_lds_dwordx4 v1, s[2:3]\n\t;;#ASMEND\n\tscratch_load_dword v5, off, off offset:8\n\ts_endpgm\n"
                   "_Zkernel_b:\n\t;;#ASMSTART\n\tds_read_b128 v[4:7], v9 offset:0\n\t;;#ASMEND\n\tv_mov_b32_e32 v20, v5\n\t;;#ASMSTART\n\ts_waitcnt lgkmcnt(0)\n\t;;#ASMEND\n"
                   "_Zkernel_c:\n\t;;#ASMSTART\n\tglobal_load_dword v3, v1, s[2:3]\n\t;;#ASMEND\n\ts_endpgm\n")
    try:
        v, st = chk.check(str(bad))
    finally:
        bad.unlink()
    assert len(v) == 3 and "scratch traffic" in v[0] + v[1] + v[2] and "before its wait" in v[0] + v[1] + v[2] and "register-returning" in v[0] + v[1] + v[2], v


def test_scripts_compile():
    """The measurement / stress scripts under scripts/ are run by hand on the GPU box: at least they must parse."""
    import ast
    files = sorted((REPO / "scripts").glob("*.py")) + sorted((REPO / "scripts" / "debug").glob("*.py"))
    assert len(files) >= 10
    for f in files:
        ast.parse(f.read_text(), filename=str(f))
