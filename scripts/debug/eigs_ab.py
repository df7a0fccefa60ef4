"""Same-process A/B of the spectral stage (affinity build + eigensolver) of two builds of the library: product vs
DSS_LAB_LIBRARY (e.g. the build before the edge strip of the packed W: scripts/lablib/libdss_hip_prestrip.so), alternating rounds.

    DSS_LAB_LIBRARY=scripts/lablib/libdss_hip_prestrip.so python scripts/debug/eigs_ab.py
Each library builds ITS packed W (dss_affinity_f16_u16, its own dss_affinity_elems) from the same f16 features and solves it
(dss_laplacian_eigs_u16).  Prints milliseconds per launch (min / median), passes per image, GB/s on the algorithmic bytes
(2 N (N + 1) / 2 per pass for the solver) and the largest eigenvalue / |cos| difference between the libraries."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dss_amd  # noqa
from dss_amd import hip

libs = {"product": hip.load_library()}
for p in os.environ.get("DSS_LAB_LIBRARY", "").split(":"):
    if p:
        libs[os.path.basename(p).replace("libdss_hip_", "").replace(".so", "")] = hip.load_library(p)
torch.manual_seed(0)
s = torch.cuda.current_stream().cuda_stream
cases = [(2036, 900, 384, 5), (4072, 900, 384, 5), (256, 3600, 768, 15), (2036, 196, 384, 5)]
if os.environ.get("EIGS_AB_CASES"):        # "B,N,D,K;B,N,D,K;..."
    cases = [tuple(int(v) for v in c.split(",")) for c in os.environ["EIGS_AB_CASES"].split(";") if c]
for (b, n, d, k) in cases:
    side = int(n ** 0.5)
    base = torch.randn(b, n, d, device="cuda")
    yy, xx = torch.meshgrid(torch.arange(side, device="cuda"), torch.arange(side, device="cuda"), indexing="ij")
    blob = ((yy.reshape(-1) // (side // 3)) * 3 + xx.reshape(-1) // (side // 3)).clamp(max=8)        # 9 coarse regions
    feats = (base * 0.7 + torch.randn(b, 9, d, device="cuda")[:, blob]).half()
    rn = 1.0 / feats.float().norm(dim=-1).clamp_min(1e-12)
    out, res = {}, {name: {"aff": [], "eig": []} for name in libs}
    def run(name, timed):
        lib = libs[name]
        w = torch.empty((b, int(lib.dss_affinity_elems(n))), dtype=torch.int16, device="cuda")
        need = int(lib.dss_eigs_workspace_bytes(b, n, k, 0))
        ws = torch.empty(need, dtype=torch.uint8, device="cuda")
        ev = torch.empty((b, k), device="cuda"); vec = torch.empty((b, k, n), device="cuda")
        info = torch.zeros(b, dtype=torch.int32, device="cuda")
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        assert lib.dss_affinity_f16_u16(feats.data_ptr(), rn.data_ptr(), w.data_ptr(), b, n, d, s) == 0
        e[1].record()
        assert lib.dss_laplacian_eigs_u16(w.data_ptr(), b, n, k, ev.data_ptr(), vec.data_ptr(), info.data_ptr(), 0, 0.0, 0,
                                          ws.data_ptr(), need, s) == 0
        e[2].record()
        torch.cuda.synchronize()
        if timed:
            res[name]["aff"].append(e[0].elapsed_time(e[1])); res[name]["eig"].append(e[1].elapsed_time(e[2]))
        out[name] = (ev, vec, info, w.shape[1])
    for rnd in range(7):
        for name in libs:
            run(name, rnd > 0)
    line = f"B={b} N={n} D={d} K={k}:"
    for name in libs:
        ev, vec, info, elems = out[name]
        passes = info.abs().float().mean().item()
        a, g = sorted(res[name]["aff"]), sorted(res[name]["eig"])
        gbs = passes * b * n * (n + 1) / (g[0] * 1e-3) / 1e9
        line += (f"\n   [{name}] W elems/image {elems} ({elems / (n * (n + 1) / 2):.3f}x the triangle)  affinity min {a[0]:.3f} med {a[len(a)//2]:.3f} ms"
                 f"  eigs min {g[0]:.3f} med {g[len(g)//2]:.3f} ms  passes {passes:.2f}  {gbs:.0f} GB/s  converged {(info > 0).float().mean().item():.3f}")
    names = list(libs)
    for other in names[1:]:
        e0, v0 = out[names[0]][:2]; e1, v1 = out[other][:2]
        cos = (v0 * v1).sum(-1).abs() / (v0.norm(dim=-1) * v1.norm(dim=-1))
        line += f"\n   {names[0]} vs {other}: max |d lambda| {(e0 - e1).abs().max().item():.2e}  max 1-|cos| {(1 - cos).max().item():.2e}"
    print(line, flush=True)
