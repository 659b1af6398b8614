"""The reference's own struct layouts against the ones the C-ABI views and the Python dtypes assume.

oracle/_ref/layout_probe is oracle/ref_layout_probe.cpp compiled against /root/reference/libgadget/partmanager.h and
slotsmanager.h as they lie (no stand-ins; `make -C oracle ref`).  It evaluates the view initialisers INTEGRATION.md shows for the
shenqi side and reports where the IsGarbage / Swallowed bits sit."""
import json
import os
import subprocess

import numpy as np
import pytest

from shenqi_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "oracle", "_ref", "layout_probe")


@pytest.fixture(scope="module")
def layout():
    if not os.path.exists(PROBE):
        if os.path.isdir("/root/reference/libgadget"):
            pytest.fail("oracle/_ref/layout_probe missing although /root/reference is present: run `make -C oracle ref`")
        pytest.skip("no reference and no prebuilt probe")
    return json.loads(subprocess.check_output([PROBE]).decode())


def test_particle_data_layout(layout):
    d = capi.PARTICLE_DTYPE
    f = {k: d.fields[k][1] for k in d.names}
    assert layout["sizeof_particle_data"] == d.itemsize == 160
    v = layout["part_view"]
    assert v["elsize"] == 160
    want = {"off_pos": "Pos", "off_mass": "Mass", "off_type": "Type", "off_flags": "Flags", "off_pi": "PI", "off_vel": "Vel",
            "off_treeacc": "FullTreeGravAccel", "off_gravpm": "GravPM", "off_potential": "Potential", "off_hsml": "Hsml",
            "off_dthsml": "DtHsml", "off_timebin_hydro": "TimeBinHydro", "off_timebin_gravity": "TimeBinGravity"}
    for key, name in want.items():
        assert v[key] == f[name], (key, v[key], f[name])
    for name, off in layout["particle_data"].items():
        assert f[name] == off, name
    # flag byte: bit 0 IsGarbage, bit 1 Swallowed (what the library's packer reads at off_flags)
    assert layout["bit_IsGarbage"] == 8 * f["Flags"] + 0
    assert layout["bit_Swallowed"] == 8 * f["Flags"] + 1
    # Generation: the upper 4 bits of the same byte (shq_spawn_layout::generation_shift = 4), wrapping at 16
    assert layout["bit_Generation"] == 8 * f["Flags"] + 4 and layout["generation_after_15"] == 0
    # field types the packer reads: f32 mass in an f64 record, 1-byte type and bins
    assert d.fields["Mass"][0] == np.dtype("<f4") and d.fields["Type"][0] == np.dtype("u1") and d.fields["Pos"][0].base == np.dtype("<f8")


def test_part_view_of_the_host_mirror_matches(layout):
    """the view the library's own host mirror hands to the C-ABI is the one the reference-side shim would build"""
    import shenqi_amd as sq
    pv = sq.PartManager(4, 1.0).view()
    v = layout["part_view"]
    for key in v:
        if key != "elsize":
            assert getattr(pv, key) == v[key], key
    assert pv.elsize == v["elsize"]


def test_sph_and_bh_slot_layout(layout):
    d = capi.SPH_DTYPE
    f = {k: d.fields[k][1] for k in d.names}
    assert layout["sizeof_sph_particle_data"] == d.itemsize == 176
    v = layout["sph_view"]
    want = {"off_density": "Density", "off_egywtdensity": "EgyWtDensity", "off_entropy": "Entropy", "off_dtentropy": "DtEntropy",
            "off_maxsignalvel": "MaxSignalVel", "off_hydroaccel": "HydroAccel", "off_dhsmlegydensityfactor": "DhsmlEgyDensityFactor",
            "off_divvel": "DivVel", "off_curlvel": "CurlVel", "off_delaytime": "DelayTime"}
    for key, name in want.items():
        assert v[key] == f[name], key
    for name, off in layout["sph_particle_data"].items():
        assert f[name] == off, name
    sv = capi.sph_view(np.zeros(2, dtype=d))
    for key in want:
        assert getattr(sv, key) == v[key]
    # black-hole slots: the two fields density() writes (densitytree2.hpp:117-173)
    assert layout["bh_view"] == {"elsize": 248, "off_density": 24, "off_divvel": 32}
    # ... and the fields of the resident step (repositioning, dynamic friction / drag kicks, time-step limiter)
    b = capi.BH_DTYPE
    fb = {k: b.fields[k][1] for k in b.names}
    assert layout["sizeof_bh_particle_data"] == b.itemsize == 248
    dv = layout["bh_dyn_view"]
    wantb = {"off_mintimebin": "minTimeBin", "off_timebindynfric": "TimeBinDynFric", "off_jumptominpot": "JumpToMinPot", "off_dfaccel": "DFAccel",
             "off_df_surroundingvel": "DF_SurroundingVel", "off_dragaccel": "DragAccel", "off_minpotpos": "MinPotPos", "off_minpotvel": "MinPotVel"}
    for key, name in wantb.items():
        assert dv[key] == fb[name], key
    for name, off in layout["bh_particle_data"].items():
        assert fb[name] == off, name
    st = capi.STAR_DTYPE
    assert layout["sizeof_star_particle_data"] == st.itemsize == 72
    for name, off in layout["star_particle_data"].items():
        assert st.fields[name][1] == off, name
    bv = capi.bh_dyn_view(np.zeros(2, dtype=b))
    for key in wantb:
        assert getattr(bv, key) == dv[key]
    assert bv.elsize == dv["elsize"]
