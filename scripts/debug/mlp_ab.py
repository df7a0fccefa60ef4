"""Same-process A/B: fused Mlp kernel (fc1 -> GELU -> fc2 on chip) vs the unfused pair, plus correctness."""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dss_amd
from dss_amd import hip
from dss_amd.vit import setup_gemm_tuning
setup_gemm_tuning()
torch.manual_seed(0)
M = 290 * 901
x = torch.randn(M, 384, device='cuda').half()
w1 = (torch.randn(1536, 384, device='cuda') * 0.05).half(); b1 = (torch.randn(1536, device='cuda') * 0.1).half()
w2 = (torch.randn(384, 1536, device='cuda') * 0.03).half(); b2 = (torch.randn(384, device='cuda') * 0.1).half()
w1p, w2p = hip.mlp_k384_pack(w1, w2)
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(n): fn()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / n * 1e3
mchk = 4096 + 77
h = F.gelu(F.linear(x[:mchk].float(), w1.float(), b1.float())).half().float()   # hidden rounded like both paths round it
ref = F.linear(h, w2.float(), b2.float())
out = hip.mlp_k384(x[:mchk], w1p, b1, w2p, b2).float()
outp = hip.planar_to_rows(hip.mlp_k384(x[:mchk], w1p, b1, w2p, b2, planar=True)).float()
print(f"max err row-major {(out-ref).abs().max().item():.2e}  planar {(outp-ref).abs().max().item():.2e}  (|ref| max {ref.abs().max().item():.2f})")
t_lib = timeit(lambda: F.linear(F.gelu(F.linear(x, w1, b1)), w2, b2))
t_two = timeit(lambda: F.linear(hip.linear_kres(x, w1, b1, gelu=True), w2, b2))
t_f = timeit(lambda: hip.mlp_k384(x, w1p, b1, w2p, b2))
t_fp = timeit(lambda: hip.mlp_k384(x, w1p, b1, w2p, b2, planar=True))
fl = 2.0 * M * 384 * 1536 * 2
print(f"torch fc1+gelu+fc2 {t_lib:7.1f} us | kres fc1gelu + torch fc2 {t_two:7.1f} us | fused {t_f:7.1f} us ({fl/t_f/1e6:5.0f} TF/s) | fused planar {t_fp:7.1f} us ({fl/t_fp/1e6:5.0f} TF/s)")
