"""Same-process interleaved A/B of the attention kernel variants (dss_attention_fwd's `variant` argument).

    python scripts/debug/attn_ab.py            # bench shapes, both variants, random data
Prints per variant: min / median microseconds over the rounds and TFLOP/s on the 4*T^2*D*h*B count."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dss_amd  # noqa
from dss_amd import hip
torch.manual_seed(0)
VARIANTS = {"pingpong8": hip.ATTENTION_PINGPONG, "4wave": hip.ATTENTION_4WAVE}
cases = [(290, 901, 6, True), (290, 901, 6, False), (16, 3601, 12, False), (320, 197, 6, True)]
for (b, t, h, planar) in cases:
    qkv = (torch.randn(b, t, 3 * h * 64, device='cuda') * 1.0).half()
    arg = qkv
    kw = {}
    if planar:   # [3h, B*T, 64]
        arg = qkv.reshape(b * t, 3 * h, 64).permute(1, 0, 2).contiguous()
        kw = dict(planar_bt=(b, t))
    flops = 4.0 * t * t * h * 64 * b
    res, outs = {}, {}
    for rnd in range(5):
        for name, var in VARIANTS.items():
            out = hip.attention(arg, h, 0.125, variant=var, **kw)
            torch.cuda.synchronize()
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            for _ in range(10):
                hip.attention(arg, h, 0.125, out=out, variant=var, **kw)
            en.record(); torch.cuda.synchronize()
            res.setdefault(name, []).append(st.elapsed_time(en) / 10)
            outs[name] = out
    ref = outs["4wave"].float()
    for name, v in res.items():
        v = sorted(v)
        print(f"B={b} T={t} h={h} planar={planar} {name:10s}: min {v[0]*1e3:7.1f} us  median {v[len(v)//2]*1e3:7.1f} us  -> "
              f"{flops/v[0]/1e9:6.0f} TF/s   maxdiff vs 4wave {(outs[name].float()-ref).abs().max().item():.2e}", flush=True)
