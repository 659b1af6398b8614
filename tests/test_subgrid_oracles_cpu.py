"""The restatements of the wind and metal-return walks (oracle/winds.py, oracle/metal_return.py) on the CPU: properties that do not
need the device — the wind outcome does not depend on the order of the stars, the nearest star kicks, returned mass and metals are
conserved.  (No reference fixture exists for either walk: parity unpinned, oracle/README.md.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import winds as ow  # noqa: E402
import metal_return as omr  # noqa: E402
import common as cm  # noqa: E402
import test_gpu_winds as tw  # noqa: E402
import test_gpu_metal_return as tm  # noqa: E402


def test_winds_do_not_depend_on_the_order_of_the_stars():
    _, prm = tw.params()
    pman, S, ST, rnd, new = tw.setup(1, ngrid=10, nstar=120, nnew=40)
    P = pman.Base
    ids = np.ascontiguousarray(P["ID"])
    P1, S1 = P.copy(), S.copy()
    tw1, k1, a1 = ow.winds_and_feedback(P1, S1, ST, ids, new, prm, rnd)
    P2, S2 = P.copy(), S.copy()
    tw2, k2, a2 = ow.winds_and_feedback(P2, S2, ST, ids, new[::-1].copy(), prm, rnd)
    assert k1 == k2 and a1 == a2 > 5 and np.array_equal(tw1, tw2)
    assert np.array_equal(P1["Vel"], P2["Vel"]) and np.array_equal(S1["Entropy"], S2["Entropy"]) and np.array_equal(S1["DelayTime"], S2["DelayTime"])
    # the star that kicks a contested particle is the nearest candidate; a kicked particle was not a wind particle before
    first = {}
    for part, dist, sid, v, th in k1:
        if part in first:
            assert dist >= first[part][0]
        else:
            first[part] = (dist, sid)
    kicked = np.flatnonzero(np.any(P1["Vel"] != P["Vel"], axis=1))
    assert sorted(first) == kicked.tolist() and (S["DelayTime"][P["PI"][kicked]] == 0).all()
    # TotalWeight counts exactly the non-wind gas inside Hsml
    i = int(new[3])
    d = ow.nearest(P["Pos"][i] - P["Pos"], cm.BOX)
    r = np.sqrt((d * d).sum(axis=1))
    inside = (r <= P["Hsml"][i]) & (P["Type"] == 0) & ((P["Flags"] & 1) == 0)
    inside[inside] &= S["DelayTime"][P["PI"][inside]] == 0
    assert np.isclose(tw1[P["PI"][i]], P["Mass"][inside].astype(np.float64).sum(), rtol=1e-12)


def test_metal_return_conserves_mass_and_metals():
    pman, S, queue, starvol, massgen, metalgen, species = tm.setup(3, ngrid=10, nstar=80, nq=50)
    P = pman.Base
    P0, S0 = P.copy(), S.copy()
    mret = omr.metal_return(P, S, queue, starvol, massgen, metalgen, species, 4.0, 1, 1, cm.BOX)
    gas = (P["Type"] == 0) & ((P["Flags"] & 1) == 0)
    pi = P["PI"][gas]
    gained = (P["Mass"][gas].astype(np.float64) - P0["Mass"][gas].astype(np.float64)).sum()
    assert abs(gained - mret.sum()) < 2e-6 * P0["Mass"][gas].sum() and mret.sum() > 0.1 and (mret >= 0).all()
    assert (mret[massgen == 0] == 0).all()
    z = lambda PP, SS: (SS["Metallicity"][pi] * PP["Mass"][gas]).sum()      # noqa: E731
    assert z(P, S) > z(P0, S0)
    # the returned metal mass equals the stars' share: sum over pairs of returnfraction * MetalGenerated, bounded by the total generated
    assert z(P, S) - z(P0, S0) <= metalgen.sum() * (mret.sum() / max(massgen.sum(), 1e-300)) * 4 + 1e-9
    assert np.abs((P["Mass"][gas] / S["Density"][pi]) / (P0["Mass"][gas] / S0["Density"][pi]) - 1).max() < 5e-7
    assert (P["Mass"][gas] <= 4.0).all() and np.array_equal(P["Mass"][~gas], P0["Mass"][~gas])
