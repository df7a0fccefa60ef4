#!/usr/bin/env python
"""profiles/rNN_pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE rocprofv3 PMC passes of `scripts/gpu_r2.sh pmc`
(one counter per pass; the per-kernel CSVs are written by scripts/rocpd_pmc_multi.py).

    python scripts/make_pmc_traffic.py profiles/r02_pmc_fetch.csv profiles/r02_pmc_write.csv \
        gpurun_out/pmc_fetch/bench.json profiles/r02_pmc_traffic.json

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KB, and FETCH_SIZE is doubled per
MI355X_MICROARCH.md section HBM (gfx950 counts 128-byte requests as 64 B; checked in the same run on a plain
elementwise torch kernel whose traffic is known)."""
import csv
import json
import re
import sys

def _kres_lnm(name):
    """LNM of a linear_kres_kernel instance (None for other kernels): the template arguments after the element type are
    <GELU, KS, RT, NW, LNM[, PIPE]> - mangled ...linear_kres_kernelIDF16_Lb1ELi24ELi2ELi4ELi<LNM>E[Li0E]EEvPK...; 0 = plain
    Linear, 1 / 2 = LayerNorm prologue."""
    m = re.search(r"linear_kres_kernelI\w+?_?L[bi]\d+ELi\d+ELi\d+ELi\d+ELi(\d+)E", name)   # (GELU: a bool until round 4, an int since)
    if not m:   # demangled: dss::linear_kres_kernel<_Float16, true, 24, 2, 4, 2, 0>
        m = re.search(r"linear_kres_kernel<[^,>]+,\s*\w+,\s*\d+,\s*\d+,\s*\d+,\s*(\d+)", name)
    return int(m.group(1)) if m else None


KEYS = {  # bench.py kernel key -> substring of the rocprof kernel name (or a predicate on it)
    # one template, two bench keys
    "linear_kres": lambda n: _kres_lnm(n) == 0,
    "lnlinear": lambda n: _kres_lnm(n) in (1, 2),
    "lnlinear_kfeatures": "kfeat_kres_kernel",
    "patch_embed": "patch_embed_kres_kernel",
    "attention": "attn_fwd",
    "laplacian_eigs": "laplacian_eigs_kernel",
    "affinity": "gram_",
    "layernorm": "layernorm_kernel",
    "kfeatures_finalize": "kfeatures_finalize_kernel",
    "normalize_rows_split": "normalize_rows_split_kernel",
    "torch_gelu (FETCH_SIZE calibration)": "GeluCUDAKernelImpl",
}


def load(path):
    """kernel -> (dispatches, average counter value, average us); the counter is the last column."""
    rows = {}
    with open(path) as f:
        rd = csv.DictReader(l for l in f if not l.startswith("#"))
        counter = rd.fieldnames[-1]
        for r in rd:
            rows[r["kernel"]] = (int(r["dispatches"]), float(r[counter]), float(r["avg_us"]))
    return rows


def main():
    fetch, write, bench_json, out = sys.argv[1:5]
    fr, wr = load(fetch), load(write)
    cfg = json.loads(open(bench_json).read().strip().splitlines()[-1])["config"]
    kernels = {}
    for key, sub in KEYS.items():
        names = [n for n in fr if (sub(n) if callable(sub) else sub in n) and n in wr]
        if not names:
            continue
        # several template instantiations (e.g. GELU / no GELU) share a key: dispatch-weighted average
        nd = sum(fr[n][0] for n in names)
        f = sum(fr[n][0] * fr[n][1] for n in names) / nd
        w = sum(wr[n][0] * wr[n][1] for n in names) / sum(wr[n][0] for n in names)
        us = sum(fr[n][0] * fr[n][2] for n in names) / nd
        kernels[key] = {"rocprof_name": names[0][:120], "instantiations": len(names), "dispatches": nd,
                        "fetch_size_kb": round(f, 1), "write_size_kb": round(w, 1),
                        "hbm_bytes_per_launch": round((2 * f + w) * 1024, 1), "avg_us_under_pmc": round(us, 2)}
    doc = {"_comment": __doc__.split("\n\n")[-1].replace("\n", " "),
           "config": {"model": cfg["workload"].split()[0], "size": int(cfg["workload"].split()[1].split("x")[0]),
                      "K": int(cfg["workload"].split("K=")[1].split(",")[0]), "batch": cfg["images_per_step"],
                      "vit_batch": cfg["vit_batch"]},
           "kernels": kernels}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc["config"]), list(kernels))


if __name__ == "__main__":
    main()
