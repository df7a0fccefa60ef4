"""Build-time check of the inline-asm software pipelines (csrc/linear384.hip, csrc/attention.hip).

An asm statement that ISSUES a load into a register (`ds_read_b128 %0, ...`, `global_load_dword %0, ...`) hands the compiler
a value that does not exist yet: it arrives when a LATER asm statement (`s_waitcnt lgkmcnt(N)` / `vmcnt(N)`, the register tied to it
as an operand) has waited for it.  Nothing stops the register allocator from COPYING that register in between - a `v_mov` in
front of the wait reads whatever the register held before.  Round 5 found exactly that in the fused norm -> Linear kernel (the
correction-term load of the last chunk: DESIGN.md §0, scripts/debug/lnlinear_stress.py); that load is now a plain C++ load, and
this script checks what is left, on the compiler's own assembly:

  * no asm statement issues a register-returning GLOBAL load (LDS-DMA `global_load_lds_*` has no destination register);
  * between an asm `ds_read*` and the wait that covers it (LDS returns in order: `lgkmcnt(N)` covers every asm read older than the
    N youngest), no instruction touches the destination registers;
  * (round 6) a kernel that issues LDS-DMA (`global_load_lds_*`) has no scratch traffic at all: a spill reload shares vmcnt with the
    DMA pieces, and the compiler's own `s_waitcnt vmcnt(0)` behind it then waits for the stage that was just issued (seen in the
    pipelined attention lab kernel of round 6: twice as slow, results still right - nothing else would have flagged it).

    python scripts/check_async_asm.py [file.s ...]      # default: compiles linear384.hip, attention.hip and affinity.hip with -S (minutes)
Exit status 1 and one line per violation if any."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "deep-spectral-segmentation_amd", "csrc")
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def check(path):
    violations, stats = [], {"asm_ds_reads": 0, "functions": 0, "dma_kernels": 0}
    func, in_asm, pending = None, False, []          # pending: [(regs, line number)] of asm LDS reads not yet waited for
    dma_at, scratch_at = None, None                  # first LDS-DMA / first scratch access of the running function

    def close_function():
        if dma_at is not None:
            stats["dma_kernels"] += 1
            if scratch_at is not None:
                violations.append(f"{path}:{scratch_at}: {func}: scratch traffic in a kernel that issues LDS-DMA (first DMA at line {dma_at}): "
                                  "spill reloads share vmcnt with the DMA pieces")
    for ln, line in enumerate(open(path), 1):
        s = line.strip()
        if line[:1] not in " \t.;#" and s.endswith(":") or (line[:1] not in " \t.;#" and ": " in s and s.split(":")[0].startswith("_Z")):
            close_function()
            func, pending, in_asm, dma_at, scratch_at = s.split(":")[0], [], False, None, None
            stats["functions"] += 1
            continue
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        ins = s.split(";")[0].strip()
        op = ins.split()[0]
        if in_asm and op.startswith("global_load_lds") and dma_at is None:
            dma_at = ln
        if op.startswith("scratch_") and scratch_at is None:
            scratch_at = ln
        m = re.match(r"s_waitcnt\b.*lgkmcnt\((\d+)\)", ins)
        if m:
            n = int(m.group(1))
            pending = pending[len(pending) - n:] if n else []
            continue
        if op == "s_waitcnt" and "lgkmcnt" not in ins and "vmcnt" in ins:
            continue
        if in_asm and re.match(r"(global|flat|buffer|scratch)_load_", op) and "_lds_" not in op and "lds" not in ins.split(",")[-1]:
            violations.append(f"{path}:{ln}: {func}: asm issues a register-returning global load: {ins}")
            continue
        touched = regs_of(ins)
        for regs, at in pending:
            if touched & regs:
                violations.append(f"{path}:{ln}: {func}: `{ins}` touches v{sorted(touched & regs)} of the asm LDS read at line {at} before its wait")
        if in_asm and op.startswith("ds_read"):
            dst = regs_of(ins.split(",")[0])
            pending.append((dst, ln))
            stats["asm_ds_reads"] += 1
    close_function()
    return violations, stats


def main():
    files = sys.argv[1:]
    tmp = None
    if not files:
        tmp = tempfile.mkdtemp()
        # every source with an asm-pinned pipeline: the K-resident Linear family (all instantiations: K = 384 and K = 768, the fused
        # LayerNorm prologues, the hand-over and patch-embedding modes), attention (K-fragment pipeline, LDS-DMA stages) and the
        # affinity build (gram_f16_dma_kernel's LDS-DMA panels behind counted vmcnt waits)
        for src, extra in (("linear384.hip", []), ("attention.hip", ["-fno-honor-nans", "-mno-amdgpu-ieee"]), ("affinity.hip", [])):
            out = os.path.join(tmp, src.replace(".hip", ".s"))
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", CSRC, "--cuda-device-only", "-S",
                            *extra, os.path.join(CSRC, src), "-o", out], check=True, stderr=subprocess.DEVNULL)
            files.append(out)
    bad = 0
    for f in files:
        v, st = check(f)
        print(f"{os.path.basename(f)}: {st['functions']} functions, {st['asm_ds_reads']} asm LDS reads checked, {st['dma_kernels']} LDS-DMA kernels without scratch, "
              f"{len(v)} violation(s)")
        for line in v[:40]:
            print("  " + line)
        bad += len(v)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
