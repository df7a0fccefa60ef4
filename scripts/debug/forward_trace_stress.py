"""Which op of a forward is not reproducible?  Every op DinoViT calls is wrapped to record a checksum of its outputs (and of the
residual stream it updates in place) on the device; the same batch goes through `extract_k_f16` many times - a host synchronisation
in front of every forward (cold start: the case scripts/debug/forward_stress.py caught) - and every forward's checksum list is
compared with the first forward's: the first op whose checksum differs is reported.

    python scripts/debug/forward_trace_stress.py [model] [size] [batch] [forwards]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dss_amd  # noqa
from dss_amd import hip, synthetic
import dss_amd.vit as vit
import torch.nn.functional as F

name = sys.argv[1] if len(sys.argv) > 1 else "dino_vitb8"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 480
b = int(sys.argv[3]) if len(sys.argv) > 3 else 24
n_fwd = int(sys.argv[4]) if len(sys.argv) > 4 else 200
dev = torch.device("cuda")
trace = []


def checksum(t):
    v = t.contiguous().view(torch.int16 if t.element_size() == 2 else (torch.int32 if t.element_size() == 4 else torch.uint8))
    return v.to(torch.int64).sum()


def wrap(mod, fname, inplace_arg=None):
    orig = getattr(mod, fname)
    def f(*a, **k):
        out = orig(*a, **k)
        outs = out if isinstance(out, (tuple, list)) else (out,)
        cs = [checksum(o) for o in outs if isinstance(o, torch.Tensor)]
        if inplace_arg is not None and isinstance(a[inplace_arg], torch.Tensor):
            cs.append(checksum(a[inplace_arg]))
        trace.append((fname, torch.stack(cs).sum()))
        return out
    setattr(mod, fname, f)


for fn, ia in (("lnlinear", 0), ("attention", None), ("linear_kres", None), ("layernorm", 0), ("preprocess_patchify", None), ("lnlinear_kfeatures", None)):
    wrap(hip, fn, ia)
_pe = hip.patch_embed16
def pe16(*a, **k):
    r = _pe(*a, **k)
    trace.append(("patch_embed16", checksum(a[4])))
    return r
hip.patch_embed16 = pe16
wrap(F, "linear")
_add = torch.add
def add(*a, **k):
    r = _add(*a, **k)
    if "out" in k:
        trace.append(("torch.add", checksum(k["out"])))
    return r
torch.add = add

model = vit.DinoViT(name, synthetic.synthetic_state_dict(name, 0), dev, torch.float16)
g = torch.Generator().manual_seed(7)
img = torch.randint(0, 256, (b, size, size, 3), dtype=torch.uint8, generator=g).to(dev)
runs = []
junk = torch.randn(4096, 4096, device=dev)
for i in range(n_fwd):
    trace = []
    if i % 2 == 0:
        torch.cuda.synchronize()                      # cold start (idle GPU) every other forward, a full queue otherwise
    elif i % 10 == 1:
        junk = junk @ junk * 1e-3                     # an unrelated library GEMM in front now and then
    out = model.extract_k_f16(img)
    runs.append((list(trace), checksum(out[0])))
torch.cuda.synchronize()
ref_names = [n for n, _ in runs[0][0]]
ref = torch.stack([c for _, c in runs[0][0]]).cpu()
bad = 0
for i, (tr, fin) in enumerate(runs[1:], 1):
    cs = torch.stack([c for _, c in tr]).cpu()
    if not torch.equal(cs, ref):
        bad += 1
        j = int((cs != ref).nonzero()[0])
        print(f"forward {i}: first differing op is #{j} of {len(ref_names)}: {ref_names[j]} (ops before it: {ref_names[max(0, j - 3):j]}); {int((cs != ref).sum())} later checksums differ", flush=True)
print(f"{name} {size}x{size} batch {b}: {bad} of {n_fwd - 1} forwards differ from the first; ops per forward: {len(ref_names)}")
