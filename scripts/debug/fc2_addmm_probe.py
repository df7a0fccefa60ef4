"""Probe: can the library fc2 GEMM accumulate straight into the fp32 residual stream (x += h1 . W2^T, f16 operands, f32 C = D,
beta = 1: torch.addmm(..., out_dtype=float32, out=x)) at a cost below the residual prologue it would remove (~115 us)?
The fc2 bias rides on an extra K column (K = 1536 + 64, column 1536 of h1 constant).
Result (round 4, gpurun_out -> DESIGN.md §6): the accumulate-GEMM costs +50 us (hipBLASLt's heuristic solution; TunableOp does
not cover addmm with out_dtype), the prologue it frees saves 54 us: built into DinoViT, measured equal end to end (168.6 vs
167.7 ms per step on the same box), removed again."""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dss_amd
from dss_amd.vit import setup_gemm_tuning
setup_gemm_tuning(tune_new_shapes=os.environ.get("TUNE", "1") == "1")
torch.manual_seed(0)
M, D = 290 * 901, 384
dev = "cuda"
def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(n): fn()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / n * 1e3
x = torch.randn(M, D, device=dev)
for K in (1536, 1600):
    h1 = torch.randn(M, K, device=dev).half()
    w2 = (torch.randn(D, K, device=dev) * 0.03).half()
    b2 = torch.randn(D, device=dev).half()
    w2t = w2.t()
    t_lin = timeit(lambda: F.linear(h1, w2, b2))
    t_mm32 = timeit(lambda: torch.mm(h1, w2t, out_dtype=torch.float32))
    try:
        xa = x.clone()
        ref = xa.double() + h1[:2048].double() @ w2.double().t() if False else None
        t_add = timeit(lambda: torch.addmm(xa, h1, w2t, out_dtype=torch.float32, out=xa))
        xb = x.clone()
        torch.addmm(xb, h1, w2t, out_dtype=torch.float32, out=xb)
        err = (xb[:2048].double() - (x[:2048].double() + h1[:2048].double() @ w2.double().t())).abs().max().item()
        msg = f"addmm f32 in place {t_add:7.1f} us (err {err:.1e})"
    except Exception as e:
        msg = f"addmm f32 in place FAILED: {type(e).__name__}: {str(e)[:200]}"
    print(f"K={K}: F.linear f16 out {t_lin:7.1f} us | mm f32 out {t_mm32:7.1f} us | {msg}")
