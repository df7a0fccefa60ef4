"""The product's Mlp pair timed for the fused-MLP lab (scripts/probes/mlp_fused_lab.hip; profiles/r06_mlp_fused_lab.txt):
fc1 + GELU on the K-resident kernel from a GIVEN f16 operand (`dss_linear_k384`, the footing of the lab: no LayerNorm prologue) and
with the fused residual + LayerNorm prologue (`dss_lnlinear_k384`, what the forward runs), each followed by fc2 through `dss_linear_lt`.

    python scripts/debug/mlp_pair_time.py [images=2473] [tokens=901]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dss_amd  # noqa
from dss_amd import hip

images = int(sys.argv[1]) if len(sys.argv) > 1 else 2473
tokens = int(sys.argv[2]) if len(sys.argv) > 2 else 901
m, d, h = images * tokens, 384, 1536
dev = "cuda"
g = torch.Generator().manual_seed(0)
a = (torch.randn(m, d, generator=g) * 1.0).half().to(dev)
x = torch.randn(m, d, generator=g).to(dev)
r = torch.randn(m, d, generator=g).half().to(dev)
w1, b1 = (torch.randn(h, d, generator=g) * 0.05).to(dev), (torch.randn(h, generator=g) * 0.1).to(dev)
w2, b2 = (torch.randn(d, h, generator=g) * 0.05).half().to(dev), (torch.randn(d, generator=g) * 0.1).half().to(dev)
wg, aux = hip.lnlinear_prepare(w1, b1, torch.ones(d, device=dev), torch.zeros(d, device=dev), torch.float16)
w1h, b1h = w1.half(), b1.half()


def timed(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return min(ts), sum(ts) / len(ts)


flop = 4.0 * m * d * h
f1 = hip.linear_kres(a, w1h, b1h, gelu=2)
rows = [("fc1 + GELU from a given f16 operand (dss_linear_k384, gelu = 2)", lambda: hip.linear_kres(a, w1h, b1h, gelu=2)),
        ("residual + norm2 -> fc1 + GELU (dss_lnlinear_k384, gelu = 2)", lambda: hip.lnlinear(x, r, wg, aux, 1e-6, gelu=2)),
        ("fc2 (dss_linear_lt)", lambda: hip.linear_lt(f1, w2, b2)),
        ("pair: dss_linear_k384 + dss_linear_lt", lambda: hip.linear_lt(hip.linear_kres(a, w1h, b1h, gelu=2), w2, b2)),
        ("pair: dss_lnlinear_k384 + dss_linear_lt (the forward's)", lambda: hip.linear_lt(hip.lnlinear(x, r, wg, aux, 1e-6, gelu=2), w2, b2))]
print(f"product Mlp pair at M = {m} ({images} x {tokens}), D = {d}, hidden = {h}: 2 x 2 M D H = {flop / 1e12:.2f} TFLOP for a pair")
for label, fn in rows:
    mn, mean = timed(fn)
    print(f"  {label:62s} min {mn:7.3f} ms  mean {mean:7.3f} ms" + (f"  = {flop / mn / 1e9:5.0f} TFLOP/s" if label.startswith("pair") else ""))
