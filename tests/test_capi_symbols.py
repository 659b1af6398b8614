"""The C-ABI library must load and export every symbol include/*.h declares (no compute here)."""
import ctypes as C
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = open(h).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"\b(shq_[a-z0-9_]+)\s*\(", text):
            names.append(m.group(1))
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(os.path.join(ROOT, "shenqi_amd", "lib", "libshenqi_hip.so"))
    names = declared_functions()
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_pod_layouts_match_header():
    from shenqi_amd import capi
    assert C.sizeof(capi.Node) == 120                      # struct NODE, forcetree.h:38-66
    assert capi.PARTICLE_DTYPE.itemsize == 160             # struct particle_data
    assert capi.SPH_DTYPE.itemsize == 176                  # struct sph_particle_data
    assert C.sizeof(capi.GravParams) == 8 * 8 + 8 + 2 * 4 * 512 + 8
    assert capi.hip.shq_version().startswith(b"shenqi_hip")


def test_product_does_not_import_oracle():
    """The product path must never route through the oracle (test infrastructure only)."""
    bad = ("liboracle", "import orc", "from oracle", "oracle/", "orc_")
    for path in glob.glob(os.path.join(ROOT, "shenqi_amd", "**", "*"), recursive=True):
        if os.path.isfile(path) and path.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
            text = open(path).read()
            for b in bad:
                assert b not in text, (path, b)
