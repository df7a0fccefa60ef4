"""Lab: do the two workgroups of a CU run the fused norm -> Linear kernel faster when they are HALF A PERIOD APART (one in its
HBM-bound LayerNorm prologue while the other is in its MFMA chunk loop) than in lockstep?  Needs the lab library
(scripts/build_lablib.sh stagger linear384_r4_lab.hip -DDSS_LIN_LAB_STAGGER; DSS_HIP_LIBRARY=scripts/lablib/libdss_hip_stagger.so):
the first-round workgroups picked by `mode` wait `us` microseconds before their prologue."""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dss_amd
from dss_amd import hip
lib = hip.load_library()
lib.dss_linear_set_stagger.argtypes = [ctypes.c_int, ctypes.c_int]
torch.manual_seed(0)
K, T = 384, 901
dev = "cuda"
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(n): fn()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / n * 1e3
def planar(t, M):
    return t.reshape(M, K // 64, 64).permute(1, 0, 2).contiguous()
for B in [int(b) for b in os.environ.get("BATCHES", "290,1018").split(",")]:
    M = B * T
    x0 = torch.randn(M, K, device=dev) * 2 + 0.3
    gamma, beta = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
    for name, N, gelu, out_planar, res_planar in [("LN1+qkv", 3 * K, False, True, False), ("LN2+fc1+gelu", 4 * K, True, False, True)]:
        w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev) * 0.1
        r = torch.randn(M, K, device=dev).half()
        rp = planar(r, M) if res_planar else r
        wg, aux = hip.lnlinear_prepare(w, b, gamma, beta, torch.float16)
        xs = x0.clone()
        fn = lambda: hip.lnlinear(xs, rp, wg, aux, 1e-6, gelu=gelu, planar=out_planar, residual_planar=res_planar)
        line = []
        for rep in range(2):
            for mode, us in [(0, 0), (1, 30), (1, 60), (1, 100), (1, 150), (2, 60), (2, 100), (3, 60), (3, 100)]:
                assert lib.dss_linear_set_stagger(mode, us * 100) == 0
                line.append((rep, mode, us, timeit(fn)))
        base = min(t for rep, mode, us, t in line if mode == 0)
        print(f"{name:13s} B={B} M={M} N={N}: lockstep {base:7.1f} us | " + "  ".join(
            f"m{mode}/{us}us {min(t for r2, m2, u2, t in line if (m2, u2) == (mode, us)):7.1f}" for mode, us in
            [(1, 30), (1, 60), (1, 100), (1, 150), (2, 60), (2, 100), (3, 60), (3, 100)]))
