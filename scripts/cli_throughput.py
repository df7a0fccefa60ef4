"""CLI-level steady-state throughput (decode + GPU + .pth I/O) of the two commands, startup excluded."""
import os, sys, time, tempfile, shutil
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
os.environ["DSS_ASSUME_YES"] = "1"
import numpy as np, torch
import dss_amd
from dss_amd import synthetic, extract
from PIL import Image


def main():
    n, size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 480
    tmp = Path(tempfile.mkdtemp())
    (tmp / "images").mkdir()
    for i in range(64):
        Image.fromarray(synthetic.synthetic_image(i, size, size)).save(tmp / "images" / f"{i:06d}.jpg", quality=95)
    for i in range(64, n):
        shutil.copy(tmp / "images" / f"{i % 64:06d}.jpg", tmp / "images" / f"{i:06d}.jpg")
    (tmp / "images.txt").write_text("\n".join(f"{i:06d}.jpg" for i in range(n)) + "\n")
    (tmp / "warm.txt").write_text("\n".join(f"{i:06d}.jpg" for i in range(128)) + "\n")
    torch.set_grad_enabled(False)
    common = dict(images_root=str(tmp / "images"), model_name="dino_vits16", batch_size=128, synthetic_weights=0)
    extract.extract_features(images_list=str(tmp / "warm.txt"), output_dir=str(tmp / "warm_feat"), **common)  # warm-up
    t0 = time.time()
    extract.extract_features(images_list=str(tmp / "images.txt"), output_dir=str(tmp / "feat"), **common)
    dt = time.time() - t0
    print(f"extract_features: {n} images in {dt:.1f}s -> {n/dt:.0f} images/s (JPEG decode + ViT + torch.save, model init inside)")
    extract.extract_eigs(images_root=str(tmp / "images"), features_dir=str(tmp / "warm_feat"), output_dir=str(tmp / "warm_eigs"), K=5, batch_size=128)
    t0 = time.time()
    extract.extract_eigs(images_root=str(tmp / "images"), features_dir=str(tmp / "feat"), output_dir=str(tmp / "eigs"), K=5, batch_size=256)
    dt = time.time() - t0
    print(f"extract_eigs: {n} files in {dt:.1f}s -> {n/dt:.0f} images/s (torch.load + spectral + torch.save)")
    shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
