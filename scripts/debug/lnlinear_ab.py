"""Same-process A/B: residual + LayerNorm + K-resident Linear as two kernels (dss_layernorm_fwd, dss_linear_k384) vs the one
fused kernel (dss_lnlinear_k384), at the bench's shapes (290 images x 901 tokens; VIT_BATCH=580 etc. for other sizes)."""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dss_amd
from dss_amd import hip
torch.manual_seed(0)
K = int(os.environ.get("K", 384))
B = int(os.environ.get("VIT_BATCH", 290 if K == 384 else 16))
T = 901 if K == 384 else 3601
M = B * T
dev = "cuda"
x0 = torch.randn(M, K, device=dev) * 2 + 0.3
gamma, beta = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(n): fn()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / n * 1e3
def planar(t):
    return t.reshape(M, K // 64, 64).permute(1, 0, 2).contiguous()
cases = [("LN1+qkv", 3 * K, False, True, False), ("LN2+fc1+gelu", 4 * K, True, False, True)]
if K == 768: cases = [("LN2+fc1+gelu", 4 * K, True, False, False)]
for name, N, gelu, out_planar, res_planar in cases:
    w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev) * 0.1
    r = (torch.randn(M, K, device=dev)).half()
    rp = planar(r) if res_planar else r
    wh, bh = (w).half(), b.half()
    wg, aux = hip.lnlinear_prepare(w, b, gamma, beta, torch.float16)
    xa, xb = x0.clone(), x0.clone()
    h = hip.layernorm(xa, gamma, beta, 1e-6, torch.float16, residual=rp, residual_planar=res_planar)
    two = hip.linear_kres(h, wh, bh, gelu=gelu, planar=out_planar)
    one = hip.lnlinear(xb, rp, wg, aux, 1e-6, gelu=gelu, planar=out_planar, residual_planar=res_planar)
    ref = F.linear(F.layer_norm((x0[:4096] + r[:4096].float()).double(), (K,), gamma.double(), beta.double(), 1e-6), w.double(), b.double())
    if gelu: ref = F.gelu(ref)
    def rows(t): return (hip.planar_to_rows(t) if out_planar else t)[:4096].double()
    e1, e2 = (rows(one) - ref).abs().max().item(), (rows(two) - ref).abs().max().item()
    same_x = torch.equal(xa, xb)
    xs = x0.clone()
    t_ln = timeit(lambda: hip.layernorm(xs, gamma, beta, 1e-6, torch.float16, residual=rp, residual_planar=res_planar, out=h))
    t_lin = timeit(lambda: hip.linear_kres(h, wh, bh, gelu=gelu, planar=out_planar))
    t_two = timeit(lambda: hip.linear_kres(hip.layernorm(xs, gamma, beta, 1e-6, torch.float16, residual=rp, residual_planar=res_planar, out=h), wh, bh, gelu=gelu, planar=out_planar))
    t_one = timeit(lambda: hip.lnlinear(xs, rp, wg, aux, 1e-6, gelu=gelu, planar=out_planar, residual_planar=res_planar))
    t_nores = timeit(lambda: hip.lnlinear(xs, None, wg, aux, 1e-6, gelu=gelu, planar=out_planar))
    print(f"{name:13s} M={M} N={N:4d}: LN {t_ln:6.1f} + linear {t_lin:6.1f} = pair {t_two:6.1f} us | fused {t_one:6.1f} us ({100*(1-t_one/t_two):4.1f} % less; no-residual variant {t_nores:6.1f}) "
          f"| err vs fp64: fused {e1:.2e} pair {e2:.2e} (|ref|max {ref.abs().max().item():.1f}) x identical: {same_x}")
