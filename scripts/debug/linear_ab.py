"""Same-process A/B: K-resident linear kernel vs torch (hipBLASLt) for the D=384 GEMMs, plus correctness."""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dss_amd
from dss_amd import hip
from dss_amd.vit import setup_gemm_tuning
setup_gemm_tuning()
torch.manual_seed(0)
K = int(os.environ.get('K', 384))
M = 256 * 901 if K == 384 else 16 * 3601
x = (torch.randn(M, K, device='cuda') * 1.0).half()
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(n): fn()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / n * 1e3
for name, N, gelu in [("qkv", 3 * K, False), ("proj", K, False), ("fc1+gelu", 4 * K, True), ("fc1", 4 * K, False)][:int(os.environ.get("NCASE", 4))]:
    w = (torch.randn(N, K, device='cuda') * 0.05).half(); b = (torch.randn(N, device='cuda') * 0.1).half()
    mchk = 4096 + 77                      # not a multiple of the 512-row workgroup tile: exercises the ragged tail
    ref = F.linear(x[:mchk].float(), w.float(), b.float())
    if gelu: ref = F.gelu(ref)
    out = hip.linear_kres(x[:mchk], w, b, gelu).float()
    outp = hip.planar_to_rows(hip.linear_kres(x[:mchk], w, b, gelu, planar=True)).float()
    err = max((out - ref).abs().max().item(), (outp - ref).abs().max().item()); scale = ref.abs().max().item()
    t_lib = timeit((lambda: F.gelu(F.linear(x, w, b))) if gelu else (lambda: F.linear(x, w, b)))
    t_own = timeit(lambda: hip.linear_kres(x, w, b, gelu))
    t_pl = timeit(lambda: hip.linear_kres(x, w, b, gelu, planar=True))
    fl = 2.0 * M * N * K
    print(f"{name:9s} N={N:4d}: torch {t_lib:7.1f} us ({fl/t_lib/1e6:5.0f} TF/s)  k384 row-major {t_own:7.1f} us ({fl/t_own/1e6:5.0f})  planar {t_pl:7.1f} us ({fl/t_pl/1e6:5.0f} TF/s)  max err {err:.2e} (|ref|max {scale:.1f})")
