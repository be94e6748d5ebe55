#!/usr/bin/env python3
"""Build-time check of the kernels whose global loads are inline asm retired by hand-counted s_waitcnt vmcnt(N)
(head_loss.hip: hl_fetch / hl_wait).  The compiler does not know those loads are pending, so nothing but the
source's own discipline keeps it from touching their destination registers early (a copy, an accumulator move under
register pressure, a spill).  On the device assembly of the source (hipcc -S --cuda-device-only, the library's flags), for every kernel whose name matches PATTERN:

  * no scratch_ instruction at all (no spills, no stack arrays);
  * a linear scan in program order: every vector-memory load's destination registers are "pending" until an
    s_waitcnt vmcnt(N) that leaves at most N younger vector-memory operations outstanding (vmcnt retires loads, stores
    and LDS-DMA in issue order); any instruction that names a pending register before that is reported.

    python scripts/check_asm_hazards.py fea_amd/csrc/head_loss.hip head_loss_.*_mfma_kernel
    python scripts/check_asm_hazards.py fea_amd/csrc/upgemm.hip upg_kernel --scratch-only -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form
(upgemm.hip: its MFMAs and LDS reads are inline asm and its vmcnt waits count LDS-DMA pieces and stores by hand -- a scratch
spill there is a vector-memory operation the counts do not know, so "no scratch" is the property checked.)
"""
import re
import subprocess
import sys

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def check(src, pattern, extra=(), scratch_only=False):
    # device-only assembly of the source with the library's flags
    import os
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "dev.s")
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only",
                            "-Wno-unused-value", *extra, "-o", out, src], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(out):
            print(f"check_asm_hazards: cannot compile {src}: {r.stderr.strip()[-400:]}", file=sys.stderr)
            return 2
        dis = open(out).read()
    return scan(dis, pattern, src, scratch_only)


def scan(dis, pattern, obj="<asm>", scratch_only=False):
    pat = re.compile(pattern)
    bad = 0
    name, pending, nkern = None, [], 0          # pending: list of (is_load, dst register set, text) in issue order
    for line in dis.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1) if pat.search(m.group(1)) else None
            pending = []
            nkern += name is not None
            continue
        if name is None:
            continue
        ins = line.strip()
        if not ins or ins.startswith((";", ".")) or ins.endswith(":"):
            continue
        if ins.startswith("s_endpgm"):
            name = None
            continue
        ins = ins.split(";")[0].strip()
        op = ins.split()[0] if ins.split() else ""
        if op.startswith("scratch_"):
            print(f"{name}: scratch access `{ins}`")
            bad += 1
        if scratch_only:            # (the linear pending-register scan does not follow branches: kernels with compiler-managed loads)
            continue
        m = re.match(r"s_waitcnt\b(.*)", ins)
        if m:
            v = re.search(r"vmcnt\((\d+)\)", ins)
            if v:
                n = int(v.group(1))
                pending = pending[len(pending) - n:] if n else []
            continue
        is_vmem = op.startswith(("global_", "buffer_", "flat_")) and not op.startswith(("buffer_inv", "buffer_wbl2", "buffer_gl"))
        used = regs_of(ins.split(None, 1)[1]) if len(ins.split(None, 1)) > 1 else set()
        for is_load, dst, text in pending:
            if is_load and dst & used:
                # a load's own address registers may overlap nothing pending; anything else is a hazard
                print(f"{name}: `{ins}` touches v{sorted(dst & used)} while `{text}` is still pending")
                bad += 1
                break
        if is_vmem:
            is_load = "load" in op and "lds" not in ins.split()[-1:] and not op.endswith("_lds")
            operands = ins.split(None, 1)[1].split(",") if len(ins.split(None, 1)) > 1 else []
            dst = regs_of(operands[0]) if (is_load and operands) else set()
            pending.append((is_load, dst, ins))
    if nkern == 0:
        print(f"check_asm_hazards: no kernel matches {pattern} in {obj}", file=sys.stderr)
        return 2
    print(f"check_asm_hazards: {nkern} kernels checked, {bad} finding(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    # further arguments: extra compiler flags (the Makefile's per-file CXXFLAGS)
    rest = sys.argv[3:]
    sys.exit(check(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "head_loss_.*_mfma_kernel",
                   tuple(a for a in rest if a != "--scratch-only"), "--scratch-only" in rest))
