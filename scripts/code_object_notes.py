#!/usr/bin/env python
"""Per-kernel code-object metadata (registers, spills, scratch, LDS) of every gfx950 kernel in libdss_hip.so.

    python scripts/code_object_notes.py [path/to/libdss_hip.so]      -> one line per kernel, sorted by name

The shared library carries one clang offload bundle per translation unit in its .hip_fatbin section; this script finds the
bundles in the file (magic __CLANG_OFFLOAD_BUNDLE__, uncompressed), cuts out the hipv4-amdgcn-amd-amdhsa--gfx950 code objects and
reads their amdhsa.kernels notes with llvm-readelf.  tests/test_host_logic.py asserts on the result (no VGPR spills, no scratch
in the hot kernels); profiles/r05_code_object_notes.txt is this script's output for the shipped library."""
from __future__ import annotations

import re
import shutil
import struct
import subprocess
import sys
import tempfile
from pathlib import Path

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
READELF = shutil.which("llvm-readelf") or "/opt/rocm/lib/llvm/bin/llvm-readelf"
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
          "group_segment_fixed_size", "max_flat_workgroup_size")


def code_objects(blob: bytes, arch: str = "gfx950"):
    """Yield the device code objects for `arch` found in the offload bundles of `blob`."""
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        (n,) = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tlen].decode()
            q += 24 + tlen
            if triple.startswith("hip") and triple.endswith(arch) and size:
                yield blob[pos + off:pos + off + size]
        pos = q


def kernel_notes(lib: Path):
    """{demangled-ish kernel name: {field: int}} for every kernel of the library."""
    out = {}
    blob = Path(lib).read_bytes()
    with tempfile.TemporaryDirectory() as td:
        for i, co in enumerate(code_objects(blob)):
            f = Path(td) / f"co{i}.elf"
            f.write_bytes(co)
            txt = subprocess.run([READELF, "--notes", str(f)], capture_output=True, text=True, check=True).stdout
            # the amdhsa.kernels list: entries start at "  - .agpr_count" (first key, alphabetical)
            for ent in re.split(r"\n\s*- (?=\.agpr_count:|\.args:)", txt):
                m = re.search(r"\.name:\s+(\S+)", ent)
                if not m or ".vgpr_count" not in ent:
                    continue
                rec = {}
                for k in FIELDS:
                    mk = re.search(rf"\.{k}:\s+(\d+)", ent)
                    rec[k] = int(mk.group(1)) if mk else 0
                out[m.group(1)] = rec
    return out


def demangle(names):
    filt = shutil.which("llvm-cxxfilt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    try:
        r = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        return dict(zip(names, r))
    except Exception:
        return {n: n for n in names}


def main():
    lib = Path(sys.argv[1]) if len(sys.argv) > 1 else Path(__file__).resolve().parent.parent / "deep-spectral-segmentation_amd" / "lib" / "libdss_hip.so"
    notes = kernel_notes(lib)
    dm = demangle(list(notes))
    print(f"# {lib.name}: {len(notes)} kernels (vgpr agpr sgpr | vgpr_spill sgpr_spill scratch_B | lds_B threads)")
    for name in sorted(notes, key=lambda n: dm[n]):
        r = notes[name]
        short = re.sub(r"\(.*$", "", dm[name])
        print(f"{short:110s} {r['vgpr_count']:4d} {r['agpr_count']:4d} {r['sgpr_count']:4d} | {r['vgpr_spill_count']:3d} {r['sgpr_spill_count']:3d} "
              f"{r['private_segment_fixed_size']:5d} | {r['group_segment_fixed_size']:6d} {r['max_flat_workgroup_size']:4d}")


if __name__ == "__main__":
    main()
