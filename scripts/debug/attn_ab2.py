"""Same-process A/B of the attention kernel of two builds of the library (product vs DSS_LAB_LIBRARY), alternating rounds.

    DSS_LAB_LIBRARY=scripts/lablib/libdss_hip_attn_r4.so python scripts/debug/attn_ab2.py
Prints microseconds (min / median over rounds) per library and shape, TFLOP/s on the 4*T^2*64*h*B count, and the error of each
against an fp64 reference on one image."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dss_amd  # noqa
from dss_amd import hip

libs = {"product": hip.load_library()}
for i, p in enumerate(os.environ.get("DSS_LAB_LIBRARY", "").split(":")):
    if p:
        libs[os.path.basename(p).replace("libdss_hip_", "").replace(".so", "")] = hip.load_library(p)
torch.manual_seed(0)
cases = [(1018, 901, 6, True, 1), (290, 901, 6, True, 1), (64, 3601, 12, False, 1), (64, 3601, 12, True, 1), (1280, 197, 6, True, 1), (290, 901, 6, True, 2)]
s = torch.cuda.current_stream().cuda_stream
for (b, t, h, planar, dt) in cases:
    dtype = torch.float16 if dt == 1 else torch.bfloat16
    qkv = (torch.randn(b, t, 3 * h * 64, device='cuda') * 1.0).to(dtype)
    arg = qkv.reshape(b * t, 3 * h, 64).permute(1, 0, 2).contiguous() if planar else qkv
    flops = 4.0 * t * t * h * 64 * b
    outs = {n: torch.empty((b, t, h * 64), dtype=dtype, device='cuda') for n in libs}
    res = {n: [] for n in libs}
    def run(n):
        rc = libs[n].dss_attention_fwd(arg.data_ptr(), 1 if planar else 0, outs[n].data_ptr(), b, t, h, 0.125, dt, s)
        assert rc == 0, rc
    for rnd in range(6):
        for n in libs:
            run(n); torch.cuda.synchronize()
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            for _ in range(8):
                run(n)
            en.record(); torch.cuda.synchronize()
            if rnd:
                res[n].append(st.elapsed_time(en) / 8)
    q, k, v = qkv[:1].double().reshape(1, t, 3, h, 64).permute(2, 0, 3, 1, 4)
    ref = (((q @ k.transpose(-1, -2)) * 0.125).softmax(-1) @ v).transpose(1, 2).reshape(1, t, h * 64)
    line = f"B={b} T={t} h={h} planar={planar} {'f16' if dt == 1 else 'bf16'}:"
    for n in libs:
        v_ = sorted(res[n])
        line += f"  [{n}] min {v_[0]*1e3:7.1f} med {v_[len(v_)//2]*1e3:7.1f} us {flops/v_[0]/1e9:5.0f} TF/s err {(outs[n][:1].double()-ref).abs().max().item():.1e}"
    names = list(libs)
    if len(names) > 1:
        line += f"  | same bits as {names[0]}: " + " ".join(str(bool(torch.equal(outs[names[0]], outs[n]))) for n in names[1:])
    print(line, flush=True)
