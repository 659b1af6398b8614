#!/usr/bin/env python3
"""tools/check_leaf_asm.py [sph.s] — the SPH walk-only kernels fetch a waiting leaf's records with inline-asm scalar loads that the
compiler knows nothing about (csrc/sph.hip, LEAF_ASM): between those loads and the `s_waitcnt lgkmcnt(0)` of process_pending nothing may
copy or spill the destination registers.  This reads the ISA (`hipcc -S --cuda-device-only` of sph.hip, made here when no file is given)
and fails when, in a kernel that holds such a load block,
  * a destination register of the block is spilled to a vector lane (`v_writelane_b32 …, sN`) anywhere inside the kernel's loops, or
  * the instructions right behind the block (up to the branch that ends it) read one of its destination registers.
Runs on the CPU (hipcc cross-compiles); tests/test_leaf_asm_isa_cpu.py calls it."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_isa():
    out = os.path.join(tempfile.mkdtemp(prefix="leafasm_"), "sph.s")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out,
           os.path.join(ROOT, "shenqi_amd", "csrc", "sph.hip")]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def check(path):
    txt = open(path).read()
    funcs = re.split(r"\n(?=_Z[^\n]*:\s*; @)", txt)
    report, bad = [], []
    for f in funcs:
        blocks = list(re.finditer(r";;#ASMSTART\n((?:\ts_load_[^\n]*\n)+)\t;;#ASMEND\n", f))
        if not blocks:
            continue
        name = f.split(":", 1)[0]
        regs = set()
        for b in blocks:
            for m in re.finditer(r"s_load_dword(?:x\d+)? s(?:\[(\d+):(\d+)\]|(\d+)),", b.group(1)):
                if m.group(1):
                    regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
                else:
                    regs.add(int(m.group(3)))
        # spills inside loops: the kernel's prologue (before the first loop header comment) may park anything
        first_loop = f.find("=>This")
        body = f[first_loop:] if first_loop >= 0 else f
        spills = [m.group(0) for m in re.finditer(r"v_writelane_b32 v\d+, s(\d+),", body) if int(m.group(1)) in regs]
        reads = []
        for b in blocks:
            tail = f[b.end():].split("\n")
            for line in tail[:12]:
                ins = line.strip()
                if not ins or ins.startswith(";") or ins.startswith("."):
                    continue
                if ins.startswith("s_branch") or ins.startswith("s_cbranch"):
                    break
                ops = ins.split(None, 1)[1] if " " in ins else ""
                srcs = ops.split(",", 1)[1] if "," in ops else ""
                used = set()
                for m in re.finditer(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b", srcs):
                    used.update(range(int(m.group(1)), int(m.group(2)) + 1) if m.group(1) else [int(m.group(3))])
                if used & regs:
                    reads.append(ins)
        # every other read of those registers must come (in the listing's order) behind an asm wait, not behind a load block
        lines = body.split("\n")
        state = "init"
        for k, line in enumerate(lines):
            ins = line.strip()
            if ins == ";;#ASMSTART" and k + 1 < len(lines):
                nxt = lines[k + 1].strip()
                state = "wait" if nxt.startswith("s_waitcnt lgkmcnt(0)") else ("load" if nxt.startswith("s_load_") else state)
                continue
            if state != "load" or not ins or ins[0] in ";." or ins.startswith("s_load_"):
                continue
            ops = ins.split(None, 1)[1] if " " in ins else ""
            srcs = ops.split(",", 1)[1] if "," in ops else ""
            used = set()
            for m in re.finditer(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b", srcs):
                used.update(range(int(m.group(1)), int(m.group(2)) + 1) if m.group(1) else [int(m.group(3))])
            if used & regs and ins not in reads:
                reads.append(ins)
        report.append((name, len(regs), spills, reads))
        if spills or reads:
            bad.append(name)
    return report, bad


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else make_isa()
    report, bad = check(path)
    for name, nregs, spills, reads in report:
        print("%-100s regs %2d  spills %d  reads behind the block %d" % (name[:100], nregs, len(spills), len(reads)))
        for s_ in spills[:4] + reads[:4]:
            print("      ", s_)
    if not report:
        print("no kernel holds an inline-asm scalar load block (SPH_LEAF_ASM off?)")
        sys.exit(2)
    sys.exit(1 if bad else 0)
