"""Eigensolver lab: the bench's own W (random-init dino_vits16 on synthetic 480x480 images -> f16 features -> u16 W),
then `dss_laplacian_eigs_u16` timed with HIP events.  DSS_HIP_LIBRARY selects a lab build of the library
(scripts/gpu_r3.sh eigs_lab); a -DDSS_EIGS_TIMELINE build also prints where a workgroup's cycles go.

    python scripts/debug/eigs_lab.py [--images 2030] [--reps 5] [--tag name] [--save ref.npz | --ref ref.npz]"""
import argparse, ctypes, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dss_amd
from dss_amd import hip, synthetic
from dss_amd.vit import DinoViT, setup_gemm_tuning

p = argparse.ArgumentParser()
p.add_argument("--images", type=int, default=2030); p.add_argument("--reps", type=int, default=5)
p.add_argument("--K", type=int, default=5); p.add_argument("--size", type=int, default=480)
p.add_argument("--model", default="dino_vits16"); p.add_argument("--vit-batch", type=int, default=290)
p.add_argument("--max-restarts", type=int, default=0)
p.add_argument("--tag", default="product"); p.add_argument("--save"); p.add_argument("--ref")
a = p.parse_args()
dev = torch.device("cuda:0")
setup_gemm_tuning()
model = DinoViT(a.model, synthetic.synthetic_state_dict(a.model, 0), dev, torch.float16)
n_distinct = min(a.images, 290)
from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(32) as ex:
    imgs = torch.from_numpy(np.stack(list(ex.map(lambda i: synthetic.synthetic_image(i, a.size, a.size), range(n_distinct))))).to(dev)
parts = []
for s in range(0, a.images, a.vit_batch):
    idx = torch.arange(s, min(s + a.vit_batch, a.images), device=dev) % n_distinct
    parts.append(model.extract_k_f16(imgs[idx]))
k16 = torch.cat([q[1] for q in parts]); rn = torch.cat([q[2] for q in parts])
n = k16.shape[1]
w = hip.affinity_f16_u16(k16, rn)
del parts, model
ws = None
ev, vec, info = hip.laplacian_eigs(w, n, a.K, max_restarts=a.max_restarts); torch.cuda.synchronize()
lib = hip.load_library()
has_tl = hasattr(lib, "dss_eigs_timeline")
if has_tl:
    lib.dss_eigs_timeline.restype = ctypes.c_int
    lib.dss_eigs_timeline.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    lib.dss_eigs_timeline(None, 1)
st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
st.record()
for _ in range(a.reps):
    ev, vec, info = hip.laplacian_eigs(w, n, a.K, max_restarts=a.max_restarts)
en.record(); torch.cuda.synchronize()
ms = st.elapsed_time(en) / a.reps
passes = info.abs().float().mean().item()
out = {"tag": a.tag, "lib": os.environ.get("DSS_HIP_LIBRARY", "product"), "images": a.images, "n": n, "ms": round(ms, 3),
       "passes": round(passes, 2), "unconverged": int((info < 0).sum()),
       "us_per_pass_per_512_images": round(ms * 1e3 / passes / (a.images / 512), 2), "TBps_alg": round(passes * n * (n + 1) * a.images / ms / 1e9, 3)}
if has_tl:
    buf = (ctypes.c_ulonglong * 16)()
    lib.dss_eigs_timeline(buf, 0)
    t = np.array(list(buf), dtype=np.float64)
    names = ["degree+start", "matvec", "gram-schmidt", "rr mid-cycle", "rr end-of-cycle", "restart", "ritz vectors"]
    tot = t[:7].sum()
    out["timeline_frac"] = {nm: round(t[i] / tot, 4) for i, nm in enumerate(names)}
    out["cycles_per_image"] = round(tot / (a.images * a.reps))
    out["rr_calls_per_image"] = round(t[8] / (a.images * a.reps), 2)
    out["jacobi_sweeps_per_rr"] = round(t[9] / max(t[8], 1), 2)
    per = a.images * a.reps
    out["kcycles_per_image"] = {"tile loop, no check": round(t[10] / per / 1e3), "tile loop beside a check": round(t[12] / per / 1e3),
                                "check (its wave)": round(t[11] / per / 1e3), "matvec to barrier, no check": round(t[13] / per / 1e3),
                                "matvec to barrier, with check": round(t[14] / per / 1e3)}
if hasattr(lib, "dss_eigs_rho_buffer"):   # one more solve with the per-check residual ratios recorded
    rho = torch.zeros((a.images, 64), dtype=torch.float32, device=dev)
    lib.dss_eigs_rho_buffer.argtypes = [ctypes.c_void_p]
    lib.dss_eigs_rho_buffer(rho.data_ptr())
    hip.laplacian_eigs(w, n, a.K, max_restarts=a.max_restarts); torch.cuda.synchronize()
    lib.dss_eigs_rho_buffer(None)
    np.save("gpurun_out/eigs_rho.npy", rho.cpu().numpy())
lam = ev.double().cpu().numpy(); v = vec.double().cpu().numpy()
if a.save:
    np.savez(a.save, lam=lam, vec=v)
if a.ref:
    r = np.load(a.ref)
    out["max_dlam_vs_ref"] = float(np.abs(lam - r["lam"]).max())
    # |cos| between matching eigenvectors (D-inner product not needed for a same-W comparison of near-identical vectors)
    c = np.abs((v * r["vec"]).sum(-1)) / (np.linalg.norm(v, axis=-1) * np.linalg.norm(r["vec"], axis=-1))
    out["min_cos_vs_ref"] = float(c.min()); out["n_cos_below_1m1e-4"] = int((c < 1 - 1e-4).sum())
print(json.dumps(out))
