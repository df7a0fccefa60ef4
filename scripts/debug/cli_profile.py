import os, sys, time, tempfile, shutil, cProfile, pstats
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
os.environ["DSS_ASSUME_YES"] = "1"
import torch, dss_amd
from dss_amd import synthetic, extract
from PIL import Image
n, size = 1024, 480
tmp = Path(tempfile.mkdtemp()); (tmp / "images").mkdir()
for i in range(32):
    Image.fromarray(synthetic.synthetic_image(i, size, size)).save(tmp / "images" / f"{i:06d}.jpg", quality=95)
for i in range(32, n):
    shutil.copy(tmp / "images" / f"{i % 32:06d}.jpg", tmp / "images" / f"{i:06d}.jpg")
(tmp / "images.txt").write_text("\n".join(f"{i:06d}.jpg" for i in range(n)) + "\n")
(tmp / "warm.txt").write_text("\n".join(f"{i:06d}.jpg" for i in range(128)) + "\n")
torch.set_grad_enabled(False)
common = dict(images_root=str(tmp / "images"), model_name="dino_vits16", batch_size=128, synthetic_weights=0)
extract.extract_features(images_list=str(tmp / "warm.txt"), output_dir=str(tmp / "warm_feat"), **common)
pr = cProfile.Profile(); pr.enable(); t0 = time.time()
extract.extract_features(images_list=str(tmp / "images.txt"), output_dir=str(tmp / "feat"), **common)
dt = time.time() - t0; pr.disable()
print(f"extract_features {n/dt:.0f} img/s")
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
pr = cProfile.Profile(); pr.enable(); t0 = time.time()
extract.extract_eigs(images_root="", features_dir=str(tmp / "feat"), output_dir=str(tmp / "eigs"), K=5, batch_size=256)
dt = time.time() - t0; pr.disable()
print(f"extract_eigs {n/dt:.0f} img/s")
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
shutil.rmtree(tmp)
