"""The SPH walk-only kernels load a waiting leaf's records with inline-asm scalar loads the compiler cannot see (csrc/sph.hip, LEAF_ASM).
That is only sound while nothing copies or spills the destination registers between the loads and the wait in process_pending; this test
compiles sph.hip to ISA for gfx950 (hipcc cross-compiles on the CPU) and lets tools/check_leaf_asm.py look for exactly that."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")) is None, reason="hipcc not found")
def test_no_copy_or_spill_of_the_registers_of_the_inline_asm_leaf_loads():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_leaf_asm.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    # the six walk-only kernels (density and hydro, three SPH kernels each) hold the block, and nothing else does
    held = [line for line in r.stdout.splitlines() if " regs " in line]
    assert len(held) == 6 and all("ELi1ELb0E" in line for line in held), r.stdout
