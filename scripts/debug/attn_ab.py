"""Timing of the attention kernel on the bench shapes (random data), min / median over interleaved rounds.

    python scripts/debug/attn_ab.py
Prints microseconds and TFLOP/s on the 4*T^2*64*h*B count, plus the error against an fp64 reference on one image."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dss_amd  # noqa
from dss_amd import hip
torch.manual_seed(0)
cases = [(290, 901, 6, True), (290, 901, 6, False), (16, 3601, 12, False), (320, 197, 6, True)]
for (b, t, h, planar) in cases:
    qkv = (torch.randn(b, t, 3 * h * 64, device='cuda') * 1.0).half()
    arg, kw = qkv, {}
    if planar:   # [3h, B*T, 64]
        arg = qkv.reshape(b * t, 3 * h, 64).permute(1, 0, 2).contiguous()
        kw = dict(planar_bt=(b, t))
    flops = 4.0 * t * t * h * 64 * b
    res = []
    for rnd in range(5):
        out = hip.attention(arg, h, 0.125, **kw)
        torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(10):
            hip.attention(arg, h, 0.125, out=out, **kw)
        en.record(); torch.cuda.synchronize()
        res.append(st.elapsed_time(en) / 10)
    q, k, v = qkv[:1].double().reshape(1, t, 3, h, 64).permute(2, 0, 3, 1, 4)
    ref = (((q @ k.transpose(-1, -2)) * 0.125).softmax(-1) @ v).transpose(1, 2).reshape(1, t, h * 64)
    v_ = sorted(res)
    print(f"B={b} T={t} h={h} planar={planar}: min {v_[0]*1e3:7.1f} us  median {v_[len(v_)//2]*1e3:7.1f} us  -> "
          f"{flops/v_[0]/1e9:6.0f} TF/s   max err vs fp64 (image 0) {(out[:1].double()-ref).abs().max().item():.2e}", flush=True)
