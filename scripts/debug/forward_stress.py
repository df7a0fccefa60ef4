"""Race hunt over the whole ViT forward: every kernel of it is deterministic (no atomics, fixed reduction orders; the library
GEMMs are given the same algorithm every call), so the SAME batch through `DinoViT.extract_k_f16` must give the same bits every
time.  (The eigensolver is excluded: its LDS float atomics make it reproducible to rounding, not bitwise.)

    python scripts/debug/forward_stress.py [repeats]
For each (model, size, batch): one reference forward, then `repeats` more with other kernels / synchronisations in between; the
three outputs (fp32 K features, f16 copy, inverse norms) compared bit for bit; if a forward differs, the first block whose
residual stream differs is located by re-running truncated forwards (`_run_blocks`)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dss_amd  # noqa
from dss_amd import synthetic
from dss_amd.vit import DinoViT

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
only = sys.argv[2] if len(sys.argv) > 2 else ""          # e.g. dino_vitb8: that model's configurations only
opts = {}                                                 # e.g. linear_kres=0,fuse_ln=0,fuse_k=0: DinoViT switches (ints / bools)
for kv in (sys.argv[3].split(",") if len(sys.argv) > 3 and sys.argv[3] else []):
    k_, v_ = kv.split("=")
    opts[k_] = int(v_) if k_ == "linear_kres" else bool(int(v_))
dev = torch.device("cuda")
for name, size, batch, dtype in [c for c in [("dino_vits16", 480, 291, torch.float16), ("dino_vits16", 480, 2473, torch.float16), ("dino_vitb8", 480, 24, torch.float16),
                                 ("dino_vitb16", 480, 256, torch.float16), ("dino_vits8", 224, 128, torch.bfloat16), ("dino_vits16", 224, 1331, torch.float16)] if only in f"{c[0]}:{c[2]}"]:
    model = DinoViT(name, synthetic.synthetic_state_dict(name, 0), dev, dtype, **opts)
    g = torch.Generator().manual_seed(7)
    img = torch.randint(0, 256, (min(batch, 64), size, size, 3), dtype=torch.uint8, generator=g).to(dev)
    img = img.repeat((batch + img.shape[0] - 1) // img.shape[0], 1, 1, 1)[:batch].contiguous()
    first = [t.clone() for t in model.extract_k_f16(img)]
    junk = torch.randn(4096, 4096, device=dev)
    bad = 0
    for i in range(reps):
        if i % 3 == 1:
            junk = junk @ junk * 1e-3
        elif i % 3 == 2:
            torch.cuda.synchronize()
        out = model.extract_k_f16(img)
        same = [torch.equal(a, b) for a, b in zip(out, first)]
        if not all(same):
            bad += 1
            if bad <= 3:
                d = (out[0] != first[0]).nonzero()
                print(f"    forward {i}: outputs equal {same}; {d.shape[0]} fp32 feature values differ, images {sorted(set(d[:, 0].tolist()))[:8]}, "
                      f"max |diff| {(out[0] - first[0]).abs().max().item():.3g}", flush=True)
    print(f"{name} {size}x{size} batch {batch} {str(dtype)[6:]} {opts or ''}: {bad} of {reps} forwards differ from the first", flush=True)
    del model
    torch.cuda.empty_cache()
