"""GPU tests of the SPH walks driven in phases (shq_density_open ... shq_density_close, shq_hydro_*): the
reference's distributed walk — primary over the local tree with pseudo nodes, export table, secondary walks of the
exported queries on the owners' tree, reduce, postprocess — assembled through the C-ABI must reproduce the
one-shot walk over the undivided tree (which test_gpu_sph.py pins to the oracle)."""
import ctypes as C

import numpy as np
import pytest

import shenqi_amd as sq
from shenqi_amd import capi
import orc
import common as cm

pytestmark = pytest.mark.gpu


def _gas(n1=16, seed=5, kernel=1):
    n = n1**3
    pos = cm.random_positions(orc.boost_mt19937_uniform(seed, 3 * n), n)
    pos = pos[sq.hilbert_order(pos, cm.BOX)]
    pman, SphP, BhP = cm.make_gas(pos, np.full(n, 1.2 * cm.BOX / n1))
    BhP = np.zeros(2, dtype=sq.BH_SLOT_DTYPE)
    rng = np.random.default_rng(seed)
    pman.Base["Vel"] = rng.normal(size=(n, 3))
    SphP["Entropy"] = rng.uniform(0.5, 2.0, size=n)
    sq.set_densitypar(DensityResolutionEta=1.0, MaxNumNgbDeviation=0.5, DensityKernelType=kernel, BlackHoleNgbFactor=2.0,
                      MinGasHsml=0.006)
    return pman, SphP, BhP, rng


class TwoSides:
    """One device context per side of the exchange: `loc` holds the tree rank `me` would have (remote top leaves
    are pseudo nodes), `rem` the undivided tree standing in for every owner of a remote leaf."""

    def __init__(self, pman, me=1, ntask=3, depth=2):
        self.pman = pman
        self.full = sq.force_tree_rebuild_mask(pman, sq.GASMASK)
        cm.make_domain(self.full, ntask=ntask, me=me, depth=depth, pseudo=False)
        self.dom = sq.force_tree_rebuild_mask(pman, sq.GASMASK)
        self.tl = cm.make_domain(self.dom, ntask=ntask, me=me, depth=depth)
        self.loc, self.rem = sq.Context(0), sq.Context(0)

    def close(self):
        self.loc.close()
        self.rem.close()


def _exchange(loc, rem, params, qdtype, rdtype, secondary, reduce_):
    """exports of the open walk on `loc` -> queries -> secondary on `rem` -> reduce on `loc`; returns the export count."""
    nexp = C.c_int64()
    capi.check(capi.hip.shq_sph_exports(loc.h, None, None, 0, C.byref(nexp)))
    table = np.zeros(nexp.value, dtype=capi.DATA_INDEX_DTYPE)
    capi.check(capi.hip.shq_sph_exports(loc.h, None, capi.ptr(table), len(table), C.byref(nexp)))
    q = np.zeros(len(table), dtype=qdtype)
    capi.check(capi.hip.shq_sph_fill_queries(loc.h, capi.ptr(table), len(table), capi.ptr(q)))
    assert np.array_equal(q["NodeList"], table["NodeList"])
    res = np.zeros(len(q), dtype=rdtype)
    nint = C.c_int64()
    capi.check(secondary(rem.h, C.byref(params), capi.ptr(q), len(q), capi.ptr(res), C.byref(nint)))
    place = np.ascontiguousarray(table["Index"])
    capi.check(reduce_(loc.h, capi.ptr(place), capi.ptr(res), len(place)))
    return len(table), nint.value


@pytest.mark.parametrize("update_hsml,DoEgy", [(0, 1), (1, 0), (1, 1)])
def test_distributed_density_equals_single_domain(ctx, update_hsml, DoEgy):
    pman, SphP, BhP, rng = _gas()
    P = pman.Base
    n = len(P)
    two = TwoSides(pman)
    try:
        dp = cm.density_params(update_hsml=update_hsml, DoEgyDensity=DoEgy)
        pv = pman.view()
        sv, bv = capi.sph_view(SphP), capi.bh_view(BhP)
        # reference result: the one-shot walk over the undivided tree
        P0, S0 = P.copy(), SphP.copy()
        tvf = two.full.view()
        evp0 = np.zeros(len(SphP)); g0 = np.zeros(len(SphP))
        st0 = capi.SphStats()
        capi.check(capi.hip.shq_density(ctx.h, C.byref(tvf), None, C.byref(pv), C.byref(sv), C.byref(bv), None, 0, C.byref(dp),
                                        capi.ptr(evp0), capi.ptr(g0), C.byref(st0)))
        Pref, Sref = P.copy(), SphP.copy()
        P[:], SphP[:] = P0, S0
        # the owners' side: the same particles with the undivided tree, walk opened so that the SPH state is resident
        nq = C.c_int64()
        capi.check(capi.hip.shq_density_open(two.rem.h, C.byref(tvf), C.byref(pv), C.byref(sv), C.byref(bv), None, 0, C.byref(dp), 1,
                                             C.byref(nq)))
        # the local side
        tvd = two.dom.view()
        capi.check(capi.hip.shq_density_open(two.loc.h, C.byref(tvd), C.byref(pv), C.byref(sv), C.byref(bv), None, 0, C.byref(dp), 1,
                                             C.byref(nq)))
        assert nq.value == n
        sq.toptree_upload(two.loc, two.dom, two.tl)
        niter, nexports = 0, 0
        while True:
            capi.check(capi.hip.shq_density_ev_primary(two.loc.h))
            ne, _ = _exchange(two.loc, two.rem, dp, capi.DENSITY_QUERY_DTYPE, capi.DENSITY_RESULT_DTYPE,
                              capi.hip.shq_density_ev_secondary, capi.hip.shq_density_ev_reduce)
            nexports += ne
            nredo = C.c_int64()
            capi.check(capi.hip.shq_density_ev_postprocess(two.loc.h, C.byref(nredo)))
            niter += 1
            if nredo.value == 0:
                break
        evp = np.zeros(len(SphP)); g = np.zeros(len(SphP))
        st = capi.SphStats()
        capi.check(capi.hip.shq_density_close(two.loc.h, None, C.byref(pv), C.byref(sv), C.byref(bv), capi.ptr(evp), capi.ptr(g),
                                              C.byref(st)))
        assert nexports > n // 10 and niter == st0.niterations == st.niterations
        assert np.abs(P["Hsml"] / Pref["Hsml"] - 1).max() < 1e-12
        for name in ("Density", "EgyWtDensity", "DivVel", "CurlVel"):
            assert np.abs(SphP[name] - Sref[name]).max() < 1e-11 * np.abs(Sref[name]).max(), name
        ok = np.abs(Sref["DhsmlEgyDensityFactor"]) < 100
        assert np.all(np.abs(SphP["DhsmlEgyDensityFactor"][ok] - Sref["DhsmlEgyDensityFactor"][ok]) <
                      1e-10 * (1 + np.abs(Sref["DhsmlEgyDensityFactor"][ok])) ** 2)
        assert np.abs(P["DtHsml"] - Pref["DtHsml"]).max() < 1e-10 * np.abs(Pref["DtHsml"]).max()
        assert np.abs(g - g0).max() < 1e-10 * np.abs(g0).max() and np.array_equal(evp, evp0)
    finally:
        two.close()


def test_distributed_hydro_equals_single_domain(ctx):
    pman, SphP, BhP, rng = _gas(seed=7)
    P = pman.Base
    n = len(P)
    # a converged density state first (single domain), then hmax on the tree for the symmetric search
    tree = sq.force_tree_rebuild_mask(pman, sq.GASMASK)
    evp, _ = sq.density(ctx, None, 1, 1, 0, None, tree, pman, SphP, BhP)
    two = TwoSides(pman)
    try:
        for t in (two.full, two.dom):
            sq.force_tree_update_hmax(t, pman)
        hp = cm.hydro_params()
        pv, sv = pman.view(), capi.sph_view(SphP)
        S0 = SphP.copy()
        tvf, tvd = two.full.view(), two.dom.view()
        st0 = capi.SphStats()
        capi.check(capi.hip.shq_hydro_force(ctx.h, C.byref(tvf), C.byref(pv), C.byref(sv), None, 0, C.byref(hp), capi.ptr(evp), C.byref(st0)))
        Sref = SphP.copy()
        SphP[:] = S0
        nq = C.c_int64()
        capi.check(capi.hip.shq_hydro_open(two.rem.h, C.byref(tvf), C.byref(pv), C.byref(sv), None, 0, C.byref(hp), capi.ptr(evp), C.byref(nq)))
        capi.check(capi.hip.shq_hydro_open(two.loc.h, C.byref(tvd), C.byref(pv), C.byref(sv), None, 0, C.byref(hp), capi.ptr(evp), C.byref(nq)))
        sq.toptree_upload(two.loc, two.dom, two.tl)
        capi.check(capi.hip.shq_hydro_ev_primary(two.loc.h))
        ne, nint2 = _exchange(two.loc, two.rem, hp, capi.HYDRO_QUERY_DTYPE, capi.HYDRO_RESULT_DTYPE,
                              capi.hip.shq_hydro_ev_secondary, capi.hip.shq_hydro_ev_reduce)
        capi.check(capi.hip.shq_hydro_ev_postprocess(two.loc.h))
        st = capi.SphStats()
        capi.check(capi.hip.shq_hydro_close(two.loc.h, C.byref(pv), C.byref(sv), C.byref(st)))
        assert ne > n // 10 and nint2 > 0
        assert st.ninteractions + nint2 == st0.ninteractions      # candidates met: local + remote = undivided
        scale = np.abs(Sref["HydroAccel"]).max()
        assert scale > 0 and np.abs(SphP["HydroAccel"] - Sref["HydroAccel"]).max() < 1e-11 * scale
        assert np.abs(SphP["DtEntropy"] - Sref["DtEntropy"]).max() < 1e-11 * np.abs(Sref["DtEntropy"]).max()
        assert np.array_equal(SphP["MaxSignalVel"], Sref["MaxSignalVel"])
    finally:
        two.close()


def test_sph_phase_state_errors(ctx):
    pman, SphP, BhP, rng = _gas(n1=8)
    dp = cm.density_params(update_hsml=0)
    tree = sq.force_tree_rebuild_mask(pman, sq.GASMASK)
    pv, tv, sv, bv = pman.view(), tree.view(), capi.sph_view(SphP), capi.bh_view(BhP)
    nredo = C.c_int64()
    assert capi.hip.shq_density_ev_primary(ctx.h) != 0                       # nothing open
    assert capi.hip.shq_sph_exports(ctx.h, None, None, 0, C.byref(nredo)) != 0
    capi.check(capi.hip.shq_density_open(ctx.h, C.byref(tv), C.byref(pv), C.byref(sv), C.byref(bv), None, 0, C.byref(dp), 0, None))
    assert capi.hip.shq_density_open(ctx.h, C.byref(tv), C.byref(pv), C.byref(sv), C.byref(bv), None, 0, C.byref(dp), 0, None) != 0
    assert capi.hip.shq_hydro_ev_primary(ctx.h) != 0                         # the open walk is a density walk
    capi.check(capi.hip.shq_density_ev_primary(ctx.h))
    capi.check(capi.hip.shq_density_ev_postprocess(ctx.h, C.byref(nredo)))
    assert nredo.value == 0
    capi.check(capi.hip.shq_density_close(ctx.h, None, C.byref(pv), C.byref(sv), C.byref(bv), None, None, None))
    assert np.all(SphP["Density"] > 0)
    # a one-shot call after an abandoned open walk starts afresh
    capi.check(capi.hip.shq_density_open(ctx.h, C.byref(tv), C.byref(pv), C.byref(sv), C.byref(bv), None, 0, C.byref(dp), 0, None))
    evp, st = sq.density(ctx, None, 0, 0, 0, None, tree, pman, SphP, BhP)
    assert st.ntargets == len(SphP)


def test_hydro_kick_bit_exact(ctx):
    """shq_kick_hydro = do_hydro_kick for gas (timestep.cpp:970-1003) on the state a hydro run leaves resident: plain IEEE
    operations in the reference's order, so velocities and entropies equal a numpy restatement to the bit."""
    pman, SphP, BhP, rng = _gas(n1=12, seed=11)
    P = pman.Base
    n = len(P)
    P["TimeBinHydro"] = rng.integers(20, 24, size=n).astype(np.uint8)
    P["Flags"] = (rng.random(n) < 0.03).astype(np.uint8)            # some garbage
    P["Vel"] *= 40.0                                                # a few beyond the velocity limit below
    tree = sq.force_tree_rebuild_mask(pman, sq.GASMASK)
    evp, _ = sq.density(ctx, None, 1, 1, 0, None, tree, pman, SphP, BhP)
    sq.force_tree_update_hmax(tree, pman)
    sq.set_hydropar(1, 100.0, 0.75)
    sq.hydro_force(ctx, None, 0.1, 0.1, evp, None, tree, pman, SphP)
    hk = np.zeros(capi.TIMEBINS + 1); de = np.zeros(capi.TIMEBINS + 1)
    hk[20:24] = [1e-3, 2e-3, 4e-3, 8e-3]
    de[20:24] = [3e-4, 6e-4, 1.2e-3, 2.4e-3]
    atime, vmax = 0.1, 900.0
    act = np.sort(rng.choice(n, size=n // 2, replace=False)).astype(np.int32)
    # numpy restatement
    vel, ent = P["Vel"].copy(), SphP["Entropy"].copy()
    nlim = 0
    for i in act:
        if P["Flags"][i] & 3 or P["Type"][i] != 0:
            continue
        b = P["TimeBinHydro"][i]
        v = vel[i] + SphP["HydroAccel"][P["PI"][i]] * hk[b]
        vv = np.sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])
        if vv > 0 and vv / atime > vmax:
            v = v * (vmax * atime / vv)
            nlim += 1
        vel[i] = v
        ent[P["PI"][i]] = ent[P["PI"][i]] + SphP["DtEntropy"][P["PI"][i]] * de[b]
    sq.dynamics_upload(ctx, pman)          # makes the velocities downloadable again (same values: the SPH calls do not move them)
    got_lim = sq.kick_hydro(ctx, hk, de, atime, vmax, act, from_hydro_output=True)
    sq.dynamics_download(ctx, pman)
    ent_dev = sq.entropy_download(ctx, n)
    assert got_lim == nlim and nlim > 0
    gas = P["Type"] == 0
    assert np.array_equal(ent_dev[gas], ent[P["PI"][gas]])
    assert np.array_equal(P["Vel"], vel)
