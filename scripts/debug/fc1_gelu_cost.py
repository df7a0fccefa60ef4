"""What the GELU epilogue costs inside the fused norm2 -> fc1 kernel: dss_lnlinear_k384 at the headline shape (M = 2 228 173 rows,
N = 1536) with gelu = 0 (none), 1 (exact erf, fp32), 2 (erf polynomial on packed f16), and the norm1 -> qkv shape (N = 1152)
beside them.  min / median of 12 launches, HIP events, random data.  Record: profiles/r06_hbm_pattern_probe.txt (last section)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dss_amd  # noqa
from dss_amd import hip

M = int(sys.argv[1]) if len(sys.argv) > 1 else 2228173
dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(1)
x0 = torch.randn(M, 384, device=dev, generator=g)
r = (torch.randn(M, 384, device=dev, generator=g) * 0.1).half()
gamma, beta = torch.ones(384, device=dev), torch.zeros(384, device=dev)


def timed(n, gelu, planar):
    w = torch.randn(n, 384, device=dev, generator=g) * 0.05
    b = torch.randn(n, device=dev, generator=g) * 0.1
    wg, aux = hip.lnlinear_prepare(w, b, gamma, beta, torch.float16)
    x = x0.clone()
    ms = []
    for i in range(14):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = hip.lnlinear(x, r, wg, aux, 1e-6, gelu=gelu, planar=planar)
        e1.record()
        e1.synchronize()
        if i >= 2:
            ms.append(e0.elapsed_time(e1))
        del out
    ms.sort()
    return ms[0], ms[len(ms) // 2]


for name, n, gelu, planar in (("norm1 -> qkv (N = 1152, planar out)", 1152, 0, True), ("norm2 -> fc1, no GELU", 1536, 0, False),
                              ("norm2 -> fc1, gelu = 2 (packed f16 erf polynomial)", 1536, 2, False), ("norm2 -> fc1, gelu = 1 (exact erf, fp32)", 1536, 1, False)):
    lo, med = timed(n, gelu, planar)
    byts = M * (3840.0 + 2.0 * n)
    print(f"{name:52s} min {lo:6.3f} ms  median {med:6.3f} ms   {byts / lo / 1e9:5.2f} TB/s at the minimum")
