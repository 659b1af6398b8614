"""GPU parity tests of the SPH edge cases the reference handles explicitly: wind-decoupled gas
(DelayTime > 0: hydratree2.hpp:266-267, 141-147; densitytree2.hpp:373-374), the predictors with non-zero kick factors and mixed
time bins (density2.h:89-128), and the EntVarPred == NULL convention (densitytree2.hpp:36-52, 146-149, 397-400)."""
import ctypes as C

import numpy as np
import pytest

import shenqi_amd as sq
from shenqi_amd import capi
import orc
import common as cm

pytestmark = pytest.mark.gpu


def kick_tables(rng, scale=1.0):
    """KickFactorData with every table non-zero and different per bin (density2.h:52-87)"""
    kf = sq.KickFactors()
    kf.FgravkickB = 0.013 * scale
    for b in range(47):
        kf.gravkicks[b] = scale * 0.02 * (1 + 0.1 * b) * (1 if b % 2 else -1)
        kf.hydrokicks[b] = scale * 0.015 * (1 + 0.07 * b)
        kf.dloga_kick[b] = scale * 0.004 * (b % 5 - 1.5)
        kf.dloga_for_bin[b] = 0.002 * (b + 1)
    return kf


def copy_kf(dst, src):
    C.memmove(C.addressof(dst), C.addressof(src), C.sizeof(sq.KickFactors))


def gas_setup(n1=16, seed=5, kernel=1, nbh=0):
    n = n1**3
    pos = cm.random_positions(orc.boost_mt19937_uniform(seed, 3 * n), n)
    rng = np.random.default_rng(seed)
    ngas = n - nbh
    pman = sq.PartManager(n, cm.BOX)
    P = pman.Base
    P["Pos"] = pos
    P["Mass"] = rng.uniform(0.5, 1.5, size=n).astype(np.float32)
    P["ID"] = np.arange(1, n + 1)
    P["Type"][:ngas] = 0
    P["PI"][:ngas] = np.arange(ngas)
    if nbh:
        P["Type"][ngas:] = 5
        P["PI"][ngas:] = np.arange(nbh)
    P["Hsml"] = cm.BOX / n1 * rng.uniform(1.0, 2.2, size=n)
    P["Vel"] = rng.normal(size=(n, 3)) * 2.0
    P["FullTreeGravAccel"] = rng.normal(size=(n, 3)) * 30.0
    P["GravPM"] = rng.normal(size=(n, 3)) * 10.0
    P["TimeBinHydro"] = rng.integers(1, 6, size=n)
    P["TimeBinGravity"] = P["TimeBinHydro"] + rng.integers(0, 3, size=n)
    SphP = np.zeros(ngas, dtype=sq.SPH_DTYPE)
    SphP["Entropy"] = rng.uniform(0.5, 2.0, size=ngas)
    SphP["DtEntropy"] = rng.normal(size=ngas) * 150.0         # large enough for the 0.05 Entropy floor of SPH_EntVarPred to bite sometimes
    SphP["HydroAccel"] = rng.normal(size=(ngas, 3)) * 40.0
    SphP["Density"] = 1
    BhP = np.zeros(max(nbh, 2), dtype=sq.BH_SLOT_DTYPE)
    sq.set_densitypar(DensityResolutionEta=1.0, MaxNumNgbDeviation=0.5, DensityKernelType=kernel, BlackHoleNgbFactor=2.0, MinGasHsml=0.006)
    return pman, SphP, BhP, rng


def run_density(ctx, pman, SphP, BhP, tree, dp, want_evp=True):
    """the C-ABI call (the host mirror fixes WindsDecouple = 0 and always asks for EntVarPred)"""
    pv, tv, sv, bv = pman.view(), tree.view(), capi.sph_view(SphP), capi.bh_view(BhP)
    evp = np.zeros(len(SphP)) if want_evp else None
    st = capi.SphStats()
    capi.check(capi.hip.shq_density(ctx.h, C.byref(tv), capi.ptr(tree.Nodes_base), C.byref(pv), C.byref(sv), C.byref(bv), None, 0, C.byref(dp),
                                    capi.ptr(evp), None, C.byref(st)))
    return evp, st


DENS_FIELDS = (("Density", "density"), ("EgyWtDensity", "egywtdensity"), ("DivVel", "divvel"), ("CurlVel", "curlvel"))


@pytest.mark.parametrize("kernel", [1, 2])
def test_density_with_kick_factors_and_mixed_bins(ctx, kernel):
    """SPH_VelPred / SPH_EntVarPred with non-zero gravkicks, hydrokicks, dloga_kick, FgravkickB and per-particle time bins:
    DivVel, CurlVel and the entropy-weighted density depend on every one of them"""
    pman, SphP, BhP, rng = gas_setup(kernel=kernel)
    P = pman.Base
    tree = sq.force_tree_rebuild_mask(pman, sq.GASMASK)
    kf = kick_tables(rng)
    dp = cm.density_params(kernel=kernel, update_hsml=0, DoEgyDensity=1)
    copy_kf(dp.kf, kf)
    st = orc.SphState(P, SphP, BhP)
    rc, oevp, _, _, onint = orc.density(tree.Nodes_base.copy(), tree.firstnode, None, st, dp)
    assert rc == 0
    # the predictors really differ from the unkicked ones
    dp0 = cm.density_params(kernel=kernel, update_hsml=0, DoEgyDensity=1)
    st0 = orc.SphState(P, SphP, BhP)
    orc.density(tree.Nodes_base.copy(), tree.firstnode, None, st0, dp0)
    assert np.abs(st.divvel - st0.divvel).max() > 1e-3 * np.abs(st0.divvel).max()
    assert np.abs(st.egywtdensity - st0.egywtdensity).max() > 1e-3 * np.abs(st0.egywtdensity).max()
    evp, gs = run_density(ctx, pman, SphP, BhP, tree, dp)
    assert gs.ninteractions == onint
    for name, oname in DENS_FIELDS:
        ref = getattr(st, oname)
        assert np.abs(SphP[name] - ref).max() < 1e-10 * np.abs(ref).max(), name
    assert np.abs(evp - oevp).max() < 1e-13 * np.abs(oevp).max()
    # the entropy floor (0.05 Entropy) was exercised
    floor = SphP["Entropy"] + SphP["DtEntropy"] * np.array([kf.dloga_kick[b] for b in P["TimeBinHydro"]]) < 0.05 * SphP["Entropy"]
    assert floor.any() and not floor.all()
    # EntVarPred == NULL: the reference then predicts per neighbour (densitytree2.hpp:397-400); same numbers
    S2 = SphP.copy()
    S2["Density"] = 0
    S2["EgyWtDensity"] = 0
    evp2, gs2 = run_density(ctx, pman, S2, BhP, tree, dp, want_evp=False)
    assert evp2 is None
    for name, _ in DENS_FIELDS + (("DhsmlEgyDensityFactor", ""),):
        assert np.array_equal(S2[name], SphP[name]), name
    ost = orc.SphState(P, SphP, BhP)
    orc.density(tree.Nodes_base.copy(), tree.firstnode, None, ost, dp, want_entvarpred=False)
    assert np.abs(ost.egywtdensity - st.egywtdensity).max() < 1e-13 * np.abs(st.egywtdensity).max()


@pytest.mark.parametrize("kernel,disph", [(1, 1), (2, 0)])
def test_hydro_with_kick_factors_mixed_bins_and_no_entvarpred(ctx, kernel, disph):
    """hydro_force with predicted velocities / entropies from non-zero kick tables and mixed bins, once with the EntVarPred
    cache density() made and once without (HydroPriv ctor, hydratree2.hpp:100-122: pressures from SPH_EntVarPred per particle)"""
    pman, SphP, BhP, rng = gas_setup(seed=8, kernel=kernel)
    P = pman.Base
    tree = sq.force_tree_rebuild_mask(pman, sq.GASMASK)
    kf = kick_tables(rng, scale=0.3)
    dp = cm.density_params(kernel=kernel, update_hsml=0, DoEgyDensity=disph)
    copy_kf(dp.kf, kf)
    evp, _ = run_density(ctx, pman, SphP, BhP, tree, dp)
    sq.force_tree_update_hmax(tree, pman)
    hp = cm.hydro_params(kernel=kernel, DensityIndependentSphOn=disph)
    copy_kf(hp.kf, kf)
    sq.set_hydropar(DensityIndependentSphOn=disph, DensityContrastLimit=100.0, ArtBulkViscConst=0.75)
    results = []
    for use_evp in (True, False):
        S = SphP.copy()
        st = orc.SphState(P, S, BhP)
        onint = orc.hydro(tree.Nodes_base, tree.firstnode, st, hp, evp if use_evp else None)
        gs = sq.hydro_force(ctx, None, 0.1, cm.HUBBLE, evp if use_evp else None, kf, tree, pman, S)
        assert gs.ninteractions == onint
        a, oa = S["HydroAccel"], st.hydroaccel
        assert np.abs(a - oa).max() < 1e-10 * np.abs(oa).max()
        assert cm.force_err(a, oa).max() < 1e-5                       # runtests.cpp:536
        assert np.abs(S["DtEntropy"] - st.dtentropy).max() < 1e-10 * np.abs(st.dtentropy).max()
        assert np.abs(S["MaxSignalVel"] / st.maxsignalvel - 1).max() < 1e-12
        results.append(a.copy())
    assert np.abs(results[0] - results[1]).max() < 1e-12 * np.abs(results[0]).max()
    # and the kicks matter: without them the accelerations differ
    S = SphP.copy()
    sq.hydro_force(ctx, None, 0.1, cm.HUBBLE, evp, None, tree, pman, S)
    assert np.abs(S["HydroAccel"] - results[0]).max() > 1e-4 * np.abs(results[0]).max()


def test_hydro_wind_decoupled_gas(ctx):
    """gas with DelayTime > 0: invisible as a hydro neighbour (hydratree2.hpp:266-267); as a target its acceleration and
    DtEntropy are zeroed and MaxSignalVel set by winds_decoupled_hydro (hydratree2.hpp:141-147, winds.h:60-68)"""
    pman, SphP, BhP, rng = gas_setup(seed=12)
    P = pman.Base
    n = len(SphP)
    wind = rng.random(n) < 0.1
    SphP["DelayTime"][wind] = rng.uniform(0.1, 1.0, size=wind.sum())
    tree = sq.force_tree_rebuild_mask(pman, sq.GASMASK)
    dp = cm.density_params(update_hsml=0, DoEgyDensity=1)
    evp, _ = run_density(ctx, pman, SphP, BhP, tree, dp)
    sq.force_tree_update_hmax(tree, pman)
    hp = cm.hydro_params()
    hp.WindSpeed, hp.WindFreeTravelDensThresh = 3.5, 0.2
    st = orc.SphState(P, SphP, BhP)
    onint = orc.hydro(tree.Nodes_base, tree.firstnode, st, hp, evp)
    pv, tv, sv = pman.view(), tree.view(), capi.sph_view(SphP)
    gs = capi.SphStats()
    capi.check(capi.hip.shq_hydro_force(ctx.h, C.byref(tv), C.byref(pv), C.byref(sv), None, 0, C.byref(hp), capi.ptr(evp), C.byref(gs)))
    assert gs.ninteractions == onint
    a, oa = SphP["HydroAccel"], st.hydroaccel
    assert np.all(a[wind] == 0) and np.all(SphP["DtEntropy"][wind] == 0)
    assert np.abs(a - oa).max() < 1e-10 * np.abs(oa).max()
    assert np.abs(SphP["DtEntropy"] - st.dtentropy).max() < 1e-10 * np.abs(st.dtentropy).max()
    assert np.abs(SphP["MaxSignalVel"] / st.maxsignalvel - 1).max() < 1e-12
    # the wind particles' MaxSignalVel is the wind formula, not the pairwise maximum
    fac_mu = 0.1 ** (3 * (5.0 / 3 - 1) / 2) / 0.1
    hsml_c = np.cbrt(0.2 / SphP["Density"][wind]) * 0.1
    assert np.all(SphP["MaxSignalVel"][wind] >= hsml_c * 2 * 3.5 * 0.1 * fac_mu * (1 - 1e-12))
    # invisible as neighbours: the same run with the wind particles' pressure changed gives the others the same force
    S2 = SphP.copy()
    S2["Entropy"][wind] *= 50.0
    evp2 = evp.copy()
    evp2[wind] *= 50.0 ** (3.0 / 5.0)
    sv2 = capi.sph_view(S2)
    capi.check(capi.hip.shq_hydro_force(ctx.h, C.byref(tv), C.byref(pv), C.byref(sv2), None, 0, C.byref(hp), capi.ptr(evp2), None))
    assert np.array_equal(S2["HydroAccel"][~wind], a[~wind])


def test_density_bh_targets_skip_wind_gas(ctx):
    """black-hole targets with WindsDecouple: wind-decoupled gas is left out of their density (densitytree2.hpp:373-374) but
    not out of the gas particles' own; BH results land in the BH slots"""
    nbh = 40
    pman, SphP, BhP, rng = gas_setup(seed=15, nbh=nbh)
    P = pman.Base
    ngas = len(SphP)
    wind = rng.random(ngas) < 0.25
    SphP["DelayTime"][wind] = 0.5
    tree = sq.force_tree_rebuild_mask(pman, sq.GASMASK)
    for wd in (1, 0):
        dp = cm.density_params(update_hsml=0, DoEgyDensity=1, BlackHoleOn=1)
        dp.WindsDecouple = wd
        S, B = SphP.copy(), BhP.copy()
        st = orc.SphState(P, S, B)
        rc, oevp, _, _, onint = orc.density(tree.Nodes_base.copy(), tree.firstnode, None, st, dp)
        assert rc == 0
        evp, gs = run_density(ctx, pman, S, B, tree, dp)
        assert gs.ninteractions == onint and gs.ntargets == ngas + nbh
        assert np.abs(S["Density"] - st.density).max() < 1e-10 * st.density.max()
        assert np.abs(B["Density"][:nbh] - st.bh_density[:nbh]).max() < 1e-10 * st.bh_density[:nbh].max()
        assert np.abs(B["DivVel"][:nbh] - st.bh_divvel[:nbh]).max() < 1e-10 * np.abs(st.bh_divvel[:nbh]).max()
        if wd:
            bh_wd, gas_wd = B["Density"][:nbh].copy(), S["Density"].copy()
        else:
            assert np.all(B["Density"][:nbh] >= bh_wd) and (B["Density"][:nbh] > bh_wd * (1 + 1e-6)).any()
            assert np.array_equal(S["Density"], gas_wd)         # gas targets never skip wind neighbours


@pytest.mark.parametrize("kernel", [1, 2])
def test_heavy_targets_get_a_wave_or_a_workgroup(ctx, kernel):
    """targets whose neighbour list does not fit a lane's list (NL_CAP 256) leave the group walk and are walked by a whole wave
    (lanes on candidates); the heaviest of those (> 16384 candidates) by a whole workgroup.  Same candidate counts as the
    oracle, sums to rounding (their order differs), for density and hydro."""
    n1 = 32
    n = n1**3
    pos = cm.random_positions(orc.boost_mt19937_uniform(31, 3 * n), n)
    rng = np.random.default_rng(kernel)
    hsml = cm.BOX / n1 * rng.uniform(1.0, 2.0, size=n)
    wave_tier = rng.choice(n, size=300, replace=False)
    hsml[wave_tier] *= 4.0                                  # ~64 x the neighbours: thousands
    block_tier = wave_tier[:3]
    hsml[block_tier] = 0.49 * cm.BOX                        # half the box: tens of thousands of candidates
    pman, SphP, BhP = cm.make_gas(pos, hsml)
    BhP = np.zeros(2, dtype=sq.BH_SLOT_DTYPE)
    sq.set_densitypar(DensityResolutionEta=1.0, MaxNumNgbDeviation=0.5, DensityKernelType=kernel, BlackHoleNgbFactor=2.0, MinGasHsml=0.006)
    P = pman.Base
    P["Vel"] = rng.normal(size=(n, 3))
    SphP["Entropy"] = rng.uniform(0.5, 2.0, size=n)
    tree = sq.force_tree_rebuild_mask(pman, sq.GASMASK)
    dp = cm.density_params(kernel=kernel, update_hsml=0, DoEgyDensity=1)
    st = orc.SphState(P, SphP, BhP)
    rc, oevp, ogr, _, onint = orc.density(tree.Nodes_base.copy(), tree.firstnode, None, st, dp, want_gradrho=True)
    assert rc == 0
    gmag = np.zeros(n)
    evp, gs = sq.density(ctx, None, 0, 1, 0, None, tree, pman, SphP, BhP, GradRho_mag=gmag)
    assert gs.ninteractions == onint
    for name, oname in (("Density", "density"), ("EgyWtDensity", "egywtdensity"), ("DivVel", "divvel"), ("CurlVel", "curlvel")):
        ref = getattr(st, oname)
        assert np.abs(SphP[name] - ref).max() < 1e-10 * np.abs(ref).max(), name
        assert np.abs(SphP[name][block_tier] / ref[block_tier] - 1).max() < 1e-10, name
    assert np.abs(gmag - np.linalg.norm(ogr, axis=1)).max() < 1e-10 * np.linalg.norm(ogr, axis=1).max()
    assert np.abs(P["DtHsml"] - st.dthsml).max() < 1e-10 * np.abs(st.dthsml).max()
    # hydro: the three giants are neighbours of everybody within half a box (r < h_j), and targets with huge lists themselves
    sq.force_tree_update_hmax(tree, pman)
    sq.set_hydropar(DensityIndependentSphOn=1, DensityContrastLimit=100.0, ArtBulkViscConst=0.75)
    hp = cm.hydro_params(kernel=kernel)
    st = orc.SphState(P, SphP, BhP)
    onint = orc.hydro(tree.Nodes_base, tree.firstnode, st, hp, evp)
    hs = sq.hydro_force(ctx, None, 0.1, cm.HUBBLE, evp, None, tree, pman, SphP)
    assert hs.ninteractions == onint
    a, oa = SphP["HydroAccel"], st.hydroaccel
    assert np.abs(a - oa).max() < 1e-10 * np.abs(oa).max()
    assert np.abs(a[block_tier] - oa[block_tier]).max() < 1e-10 * np.abs(oa[block_tier]).max()
    assert np.abs(SphP["DtEntropy"] - st.dtentropy).max() < 1e-10 * np.abs(st.dtentropy).max()
    assert np.abs(SphP["MaxSignalVel"] / st.maxsignalvel - 1).max() < 1e-12
    # run-to-run reproducibility of the heavy tiers (fixed assignment of candidates to threads)
    S2 = SphP.copy()
    sq.hydro_force(ctx, None, 0.1, cm.HUBBLE, evp, None, tree, pman, S2)
    assert np.array_equal(S2["HydroAccel"], SphP["HydroAccel"])
