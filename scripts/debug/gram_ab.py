"""dss_affinity_kres_u16 against dss_affinity_f16_u16: the same inputs, min / median of 8 launches (HIP events), bits compared.
    python scripts/debug/gram_ab.py ["B,N" ...]      default: 2473,900  14838,900  7000,196  291,3600  96,6400"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dss_amd  # noqa
from dss_amd import hip

cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(2473, 900), (14838, 900), (7000, 196), (291, 3600), (96, 6400)]
dev = torch.device("cuda")


def timeit(fn, reps=8):
    ms = []
    for i in range(reps + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(); e1.record(); e1.synchronize()
        if i >= 2:
            ms.append(e0.elapsed_time(e1))
        del out
    ms.sort()
    return ms[0], ms[len(ms) // 2]


for b, n in cases:
    g = torch.Generator(device="cuda").manual_seed(n)
    f = torch.randn(b, n, 384, device=dev, generator=g).half()
    rn = (1.0 / f.float().norm(dim=-1)).contiguous()
    wa = torch.zeros((b, hip.affinity_elems(n)), dtype=torch.int16, device=dev)
    wb = torch.zeros_like(wa)
    lib, st = hip.load_library(), torch.cuda.current_stream().cuda_stream
    hip._check(lib.dss_affinity_f16_u16(f.data_ptr(), rn.data_ptr(), wa.data_ptr(), b, n, 384, st), "f16")
    hip._check(lib.dss_affinity_kres_u16(f.data_ptr(), rn.data_ptr(), wb.data_ptr(), b, n, 384, st), "kres")
    same = bool(torch.equal(wa, wb))
    byts = b * (2.0 * n * 384 + 4.0 * n + n * (n + 1.0))
    ta, tb = timeit(lambda: hip.affinity_f16_u16(f, rn)), timeit(lambda: hip.affinity_kres_u16(f, rn))
    print(f"B={b:6d} N={n:5d}: dma build {ta[0]:8.3f} / {ta[1]:8.3f} ms ({byts / ta[0] / 1e9:5.2f} TB/s)   K-resident {tb[0]:8.3f} / {tb[1]:8.3f} ms "
          f"({byts / tb[0] / 1e9:5.2f} TB/s)   {100 * (tb[0] / ta[0] - 1):+6.1f} %   same bits: {same}")
    del f, rn, wa, wb
