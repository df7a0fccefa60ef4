"""Would a K-resident Gram build beat gram_f16_dma_kernel?  Proxy with the product's own kernels, no new code: the plain K-resident
Linear kernel (dss_linear_k384) run as F . W^T with M = B x 900 feature rows against ONE image's 960 feature rows as the weight
(planar output = 64-column planes, the closest existing layout to the 64 x 64 storage tiles) - the FULL square per image, twice the
upper triangle the affinity build needs - beside dss_affinity_f16_u16 on the same number of images.
    python scripts/debug/gram_kres_proxy.py [images=2473]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dss_amd  # noqa
from dss_amd import hip

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2473
N, D = 900, 384
dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(3)
f = torch.randn(B, N, D, device=dev, generator=g).half()
rn = (1.0 / f.float().norm(dim=-1)).contiguous()


def timeit(fn, reps=6):
    ms = []
    for i in range(reps + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(); e1.record(); e1.synchronize()
        if i >= 2:
            ms.append(e0.elapsed_time(e1))
        del out
    return min(ms), sorted(ms)[len(ms) // 2]


for ncols in (960, 448):      # 960: every column chunk (the full square); 448: about the triangle's share of column chunks per row panel
    w = torch.randn(ncols, D, device=dev, generator=g).half() * 0.05
    bias = torch.zeros(ncols, device=dev).half()
    lo, med = timeit(lambda: hip.linear_kres(f.view(B * N, D), w, bias, planar=True))
    print(f"dss_linear_k384 as a Gram proxy: {B} images x 900 rows against {ncols} rows: min {lo:7.3f} ms  median {med:7.3f} ms "
          f"({2.0 * B * N * ncols * D / lo / 1e9:6.1f} TFLOP/s, {B * N * (D + ncols) * 2.0 / lo / 1e9:5.2f} TB/s)")
lo, med = timeit(lambda: hip.affinity_f16_u16(f, rn))
print(f"dss_affinity_f16_u16 (the product's build, upper triangle):            min {lo:7.3f} ms  median {med:7.3f} ms")
