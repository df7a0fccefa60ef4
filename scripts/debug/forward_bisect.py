"""Locate a rare irreproducibility of a ViT forward (round 6: the D = 768 event of DESIGN history r5 item 10).

    python scripts/debug/forward_bisect.py <forwards> [model] [batch] [size] [arm] [DinoViT switches k=v,...]

The same batch goes through `DinoViT.extract_k_f16` again and again; the three outputs are compared bit for bit with the first
forward's.  Arms:
  capture   every op's output tensor of the forward is KEPT (a reference held, no extra kernel, no clone) until the forward's outputs
            have been compared; on a mismatch every kept tensor is compared with the first forward's: the first differing op and the
            row / column footprint of each differing op are printed
  plain     outputs only (what scripts/debug/forward_stress.py does), with the footprint of the final features on a mismatch
  rocblas   plain, with torch's GEMMs on rocBLAS instead of hipBLASLt (torch.backends.cuda.preferred_blas_library("cublas"))
Environment arms (AMD_SERIALIZE_KERNEL=3, ROCBLAS_USE_HIPBLASLT=0, ...) are set by the caller."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dss_amd  # noqa
from dss_amd import hip, synthetic
from dss_amd.vit import DinoViT
import torch.nn.functional as F

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
name = sys.argv[2] if len(sys.argv) > 2 else "dino_vitb8"
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 24
size = int(sys.argv[4]) if len(sys.argv) > 4 else 480
arm = sys.argv[5] if len(sys.argv) > 5 else "plain"
opts = {}
for kv in (sys.argv[6].split(",") if len(sys.argv) > 6 and sys.argv[6] else []):
    k_, v_ = kv.split("=")
    opts[k_] = v_ if k_ in ("gelu", "gemm_tuning", "library_gemm") else (int(v_) if k_ == "linear_kres" else bool(int(v_)))
if arm == "rocblas":
    torch.backends.cuda.preferred_blas_library("cublas")
dev = torch.device("cuda")
model = DinoViT(name, synthetic.synthetic_state_dict(name, 0), dev, torch.float16, **opts)
g = torch.Generator().manual_seed(7)
img = torch.randint(0, 256, (min(batch, 64), size, size, 3), dtype=torch.uint8, generator=g).to(dev)
img = img.repeat((batch + img.shape[0] - 1) // img.shape[0], 1, 1, 1)[:batch].contiguous()

kept = []          # (label, [tensors]) of the running forward


def wrap(mod, fn_name, label):
    orig = getattr(mod, fn_name)

    def f(*a, **kw):
        out = orig(*a, **kw)
        outs = [o for o in (out if isinstance(out, (tuple, list)) else (out,)) if torch.is_tensor(o)]
        kept.append((f"{len(kept):02d} {label} {kw.get('what', '')}", outs))
        return out
    setattr(mod, fn_name, f)


if arm == "capture":
    for fn in ("preprocess_patchify", "patch_embed16", "layernorm", "attention", "linear_kres", "lnlinear", "lnlinear_kfeatures", "kfeatures_finalize"):
        wrap(hip, fn, "hip." + fn)
    wrap(F, "linear", "F.linear (library GEMM)")


def footprint(a, b):
    ne = (a != b)
    n = int(ne.sum())
    if not n:
        return None
    idx = ne.nonzero()
    dims = []
    for dmn in range(idx.shape[1]):
        col = idx[:, dmn]
        dims.append(f"dim{dmn}[{int(col.min())}..{int(col.max())}; {int(col.unique().numel())} distinct]")
    return f"{n} of {a.numel()} values differ, max |diff| {float((a.float() - b.float()).abs().max()):.3g}; shape {tuple(a.shape)}: " + " ".join(dims)


def forward():
    kept.clear()
    out = model.extract_k_f16(img)
    return out, list(kept)


first, first_kept = forward()
first = [t.clone() for t in first]
print(f"{name} {size}x{size} batch {batch} arm {arm} {opts or ''}: {len(first_kept)} ops captured per forward; paths {model.paths()}", flush=True)
junk = torch.randn(4096, 4096, device=dev)
bad, t0 = 0, time.time()
for i in range(reps):
    if i % 3 == 1:
        junk = junk @ junk * 1e-3
    elif i % 3 == 2:
        torch.cuda.synchronize()
    out, ops = forward()
    same = [torch.equal(a, b) for a, b in zip(out, first)]
    if not all(same):
        bad += 1
        if bad <= 12:
            d = (out[0] != first[0]).nonzero()
            print(f"  forward {i}: outputs equal {same}; features: {footprint(out[0], first[0])}; images {sorted(set(d[:, 0].tolist()))}", flush=True)
            nshown = 0
            for (lab, ts), (_, rs) in zip(ops, first_kept):
                for j, (a, b) in enumerate(zip(ts, rs)):
                    fp = footprint(a, b)
                    if fp is not None and nshown < 14:
                        nshown += 1
                        print(f"      op {lab} out{j}: {fp}", flush=True)
    del out, ops
print(f"RESULT {name} {size}x{size} batch {batch} arm {arm} {opts or ''} env {{{', '.join(k + '=' + os.environ[k] for k in ('AMD_SERIALIZE_KERNEL', 'ROCBLAS_USE_HIPBLASLT', 'HIP_LAUNCH_BLOCKING') if k in os.environ)}}}: "
      f"{bad} of {reps} forwards differ from the first ({time.time() - t0:.0f} s)", flush=True)
