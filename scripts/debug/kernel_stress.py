"""Race hunt kernel by kernel, at the shapes of one model: every op of DinoViT._run_blocks / extract_k_f16 repeated on the SAME
inputs, outputs compared bit for bit with the first call's (scripts/debug/forward_stress.py says whether a whole forward is
reproducible; this says which op is not).

    python scripts/debug/kernel_stress.py [model] [size] [batch] [repeats]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dss_amd  # noqa
from dss_amd import hip, synthetic
from dss_amd.vit import DinoViT, LN_EPS
import torch.nn.functional as F

name = sys.argv[1] if len(sys.argv) > 1 else "dino_vitb8"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 480
b = int(sys.argv[3]) if len(sys.argv) > 3 else 24
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 300
only = sys.argv[5] if len(sys.argv) > 5 else ""           # e.g. "library GEMM": only ops whose label contains it get `reps` calls, the others one
dev = torch.device("cuda")
model = DinoViT(name, synthetic.synthetic_state_dict(name, 0), dev, torch.float16)
p, d, heads = model.patch_size, model.embed_dim, model.num_heads
hp = size // p
t = hp * hp + 1
blk = model.blocks[1]
g = torch.Generator().manual_seed(3)
img = torch.randint(0, 256, (b, size, size, 3), dtype=torch.uint8, generator=g).to(dev)
x0 = (torch.randn(b, t, d, generator=g) * 2).to(dev)
pend_rows = torch.randn(b, t, d, generator=g).half().to(dev)
junk = torch.randn(4096, 4096, device=dev)


def stress(label, fn):
    """`reps` calls; a device-side counter of calls whose outputs differ from the first call's (no host sync per call: the
    queue stays full, as in a real forward), a library GEMM in between now and then."""
    global junk
    first = [o.clone() for o in fn()]
    if only and only not in label:
        return first
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    worst = torch.zeros((), dtype=torch.float32, device=dev)
    for i in range(reps):
        if i % 500 == 1:
            junk = junk @ junk * 1e-3
        out = fn()
        diff = torch.zeros((), dtype=torch.bool, device=dev)
        for a, c in zip(out, first):
            ne = a != c
            diff |= ne.any()
            worst = torch.maximum(worst, ((a.float() - c.float()).abs() * ne).max())
        bad += diff
    print(f"{label}: {int(bad)} of {reps} calls differ" + (f" (max |diff| {float(worst):.3g})" if int(bad) else ""), flush=True)
    return first


if model.pe16 is None:
    patches = stress("preprocess_patchify", lambda: [hip.preprocess_patchify(img, p, torch.float16)])[0]
    stress(f"library GEMM patch_embed (M={b * hp * hp}, N={d}, K={patches.shape[-1]})", lambda: [F.linear(patches, model.pe_w, model.pe_b)])
else:
    cls_row, pos = model._pos(hp * p, hp * p)
    posb = (pos + model.pe16[1]).contiguous()
    def pe():
        x = torch.zeros((b, t, d), dtype=torch.float32, device=dev)
        hip.patch_embed16(img, model.pe16[0], None, posb, x)
        return [x]
    stress("patch_embed16", pe)
if "qkv_wg" in blk:
    qkv = stress("lnlinear qkv (residual rows, planar out)", lambda: list((hip.lnlinear(x0.clone(), pend_rows, blk["qkv_wg"], blk["qkv_aux"], LN_EPS, planar=True),)))[0]
    o = stress("attention (planar)", lambda: [hip.attention(qkv, heads, model.scale, planar_bt=(b, t))])[0]
else:
    hcur = stress("layernorm + residual", lambda: [hip.layernorm(x0.clone(), blk["n1w"], blk["n1b"], LN_EPS, torch.float16, residual=pend_rows)])[0]
    qkv = stress(f"library GEMM qkv (M={b * t})", lambda: [F.linear(hcur, blk["qkv_w"], blk["qkv_b"])])[0]
    o = stress("attention (row-major)", lambda: [hip.attention(qkv, heads, model.scale)])[0]
if d == 384 and model.linear_k384:
    pend = stress("linear_kres proj (planar)", lambda: [hip.linear_kres(o, blk["proj_w"], blk["proj_b"], planar=True)])[0]
    planar_res = True
else:
    pend = stress(f"library GEMM proj (M={b * t}, N={d}, K={d})", lambda: [F.linear(o, blk["proj_w"], blk["proj_b"])])[0]
    planar_res = False
if "fc1_wg" in blk:
    def fc1():
        x = x0.clone()
        return [hip.lnlinear(x, pend, blk["fc1_wg"], blk["fc1_aux"], LN_EPS, gelu=model._gelu_code, residual_planar=planar_res), x]
    f1 = stress("lnlinear fc1 + GELU (and the residual stream it writes back)", fc1)[0]
    stress(f"library GEMM fc2 (M={b * t}, N={d}, K={f1.shape[-1]})", lambda: [F.linear(f1, blk["fc2_w"], blk["fc2_b"])])
last = model.blocks[-1]
if "k_wg" in last:
    stress("lnlinear_kfeatures", lambda: list(hip.lnlinear_kfeatures(x0, pend_rows, last["k_wg"], last["k_aux"], LN_EPS)))
