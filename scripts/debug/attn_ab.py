"""Same-process A/B of attention kernel variants (DSS_ATTENTION_IMPL is read per call)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dss_amd
from dss_amd import hip
torch.manual_seed(0)
cases = [(128, 901, 6), (8, 3601, 12)]
for (b, t, h) in cases:
    qkv = (torch.randn(b, t, 3 * h * 64, device='cuda') * 1.0).half()
    flops = 4.0 * t * t * h * 64 * b
    res = {}
    for rnd in range(3):
        for impl in sys.argv[1:]:
            os.environ['DSS_ATTENTION_IMPL'] = impl
            out = hip.attention(qkv, h, 0.125)
            torch.cuda.synchronize()
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            for _ in range(20):
                hip.attention(qkv, h, 0.125, out=out)
            en.record(); torch.cuda.synchronize()
            ms = st.elapsed_time(en) / 20
            res.setdefault(impl, []).append(ms)
    ref = None
    for impl, v in res.items():
        os.environ['DSS_ATTENTION_IMPL'] = impl
        o = hip.attention(qkv, h, 0.125).float()
        if ref is None: ref = o
        print(f"B={b} T={t} h={h} impl={impl}: min {min(v)*1e3:.1f} us  median {sorted(v)[1]*1e3:.1f} us  -> {flops/min(v)/1e9:.0f} TF/s   maxdiff vs first {(o-ref).abs().max().item():.2e}")
