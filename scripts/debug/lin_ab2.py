"""Same-box A/B of two builds of the library (DSS_HIP_LIBRARY selects the lab build): the K-resident Linear kernel's four
uses at the bench's shapes, kernel-only timing (HIP events over 20 back-to-back launches) - run once per library."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dss_amd
from dss_amd import hip
torch.manual_seed(0)
B = int(os.environ.get("VIT_BATCH", 1018)); M, K = B * 901, 384
dev = "cuda"
x = (torch.randn(M, K, device=dev)).half()
xf = torch.randn(M, K, device=dev) * 2
r = torch.randn(M, K, device=dev).half()
gamma, beta = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(n): fn()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / n * 1e3
out = []
for name, N, gelu, planar in [("qkv", 1152, False, True), ("proj", 384, False, True), ("fc1+gelu", 1536, True, False)]:
    w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev) * 0.1
    wh, bh = w.half(), b.half()
    t = timeit(lambda: hip.linear_kres(x, wh, bh, gelu=gelu, planar=planar))
    out.append(f"{name} {t:.0f}")
    if name != "proj":
        wg, aux = hip.lnlinear_prepare(w, b, gamma, beta, torch.float16)
        t2 = timeit(lambda: hip.lnlinear(xf, r, wg, aux, 1e-6, gelu=gelu, planar=planar))
        out.append(f"res+LN+{name} {t2:.0f}")
print(os.environ.get("DSS_HIP_LIBRARY", "product").split("/")[-1], "| us:", " | ".join(out))
