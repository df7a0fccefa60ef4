"""CLI-level throughput (JPEG decode + GPU + per-image .pth I/O) of the two commands on N synthetic 480 x 480 JPEGs:
whole-call rate (worker start-up, model build and final joins included) AND the steady-state rate, read off the growth of the
output directory between 30 % and 95 % of the files (a monitor thread samples its size every 0.2 s).

    python scripts/cli_throughput.py [N=20480] [K=5]"""
import os, sys, time, tempfile, shutil, threading
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
os.environ["DSS_ASSUME_YES"] = "1"
os.environ.setdefault("DSS_CLI_TIMING", "1")
import numpy as np, torch
import dss_amd
from dss_amd import synthetic, extract
from PIL import Image


class Monitor:
    def __init__(self, directory, total):
        self.dir, self.total, self.samples, self.stop = Path(directory), total, [], False
        self.t0 = time.time()
        self.th = threading.Thread(target=self.run, daemon=True)
        self.th.start()

    def run(self):
        while not self.stop:
            try:
                n = sum(1 for _ in os.scandir(self.dir))
            except FileNotFoundError:
                n = 0
            self.samples.append((time.time() - self.t0, n))
            time.sleep(0.2)

    def steady(self):
        self.stop = True
        self.th.join()
        lo = next((s for s in self.samples if s[1] >= 0.30 * self.total), None)
        hi = next((s for s in self.samples if s[1] >= 0.95 * self.total), None)
        first = next((s for s in self.samples if s[1] > 0), None)
        if not lo or not hi or hi[0] <= lo[0]:
            return None, first
        return (hi[1] - lo[1]) / (hi[0] - lo[0]), first


def main():
    n, size = int(sys.argv[1]) if len(sys.argv) > 1 else 20480, 480
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    tmp = Path(tempfile.mkdtemp(dir=os.environ.get("DSS_CLI_TMP")))
    (tmp / "images").mkdir()
    t0 = time.time()
    for i in range(64):
        Image.fromarray(synthetic.synthetic_image(i, size, size)).save(tmp / "images" / f"{i:06d}.jpg", quality=95)
    for i in range(64, n):
        os.link(tmp / "images" / f"{i % 64:06d}.jpg", tmp / "images" / f"{i:06d}.jpg")   # hard links: same bytes, n names
    print(f"[cli] {n} JPEGs ({(tmp / 'images' / '000000.jpg').stat().st_size >> 10} KB each) in {time.time() - t0:.1f} s; "
          f"{os.cpu_count()} host cores; tmp on {tmp}", flush=True)
    (tmp / "images.txt").write_text("\n".join(f"{i:06d}.jpg" for i in range(n)) + "\n")
    (tmp / "warm.txt").write_text("\n".join(f"{i:06d}.jpg" for i in range(128)) + "\n")
    torch.set_grad_enabled(False)
    bsz = int(os.environ.get("DSS_CLI_BATCH", "128"))     # images per ViT forward (0 = the command's own choice: four rounds of workgroups)
    print(f"[cli] extract_features batch_size = {bsz}", flush=True)
    common = dict(images_root=str(tmp / "images"), model_name="dino_vits16", batch_size=bsz, synthetic_weights=0)
    extract.extract_features(images_list=str(tmp / "warm.txt"), output_dir=str(tmp / "warm_feat"), **common)  # warm-up
    mon = Monitor(tmp / "feat", n)
    t0 = time.time()
    extract.extract_features(images_list=str(tmp / "images.txt"), output_dir=str(tmp / "feat"), **common)
    dt = time.time() - t0
    rate, first = mon.steady()
    print(f"extract_features: {n} images in {dt:.1f}s -> {n / dt:.0f} images/s whole call (JPEG decode + ViT + torch.save, model init and "
          f"worker start-up inside); first file after {first[0] if first else float('nan'):.1f} s; steady state (30 % .. 95 % of the files) "
          f"{rate if rate else float('nan'):.0f} images/s", flush=True)
    extract.extract_eigs(images_root=str(tmp / "images"), features_dir=str(tmp / "warm_feat"), output_dir=str(tmp / "warm_eigs"), K=K, batch_size=128)
    mon = Monitor(tmp / "eigs", n)
    t0 = time.time()
    extract.extract_eigs(images_root=str(tmp / "images"), features_dir=str(tmp / "feat"), output_dir=str(tmp / "eigs"), K=K, batch_size=256)
    dt = time.time() - t0
    rate, first = mon.steady()
    print(f"extract_eigs: {n} files in {dt:.1f}s -> {n / dt:.0f} images/s whole call (feature load + spectral + torch.save); first file after "
          f"{first[0] if first else float('nan'):.1f} s; steady state {rate if rate else float('nan'):.0f} images/s", flush=True)
    # the outputs are the reference's schema (spot check of one pair)
    f = torch.load(tmp / "feat" / "000123.pth", weights_only=True)
    e = torch.load(tmp / "eigs" / "000123.pth", weights_only=True)
    assert set(f) == {"k", "indices", "file", "id", "model_name", "patch_size", "shape"} and tuple(f["k"].shape) == (1, 900, 384)
    assert tuple(e["eigenvectors"].shape) == (K, 900) and tuple(e["eigenvalues"].shape) == (K,)
    shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
