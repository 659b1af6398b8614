"""GPU parity tests of SPH density (incl. the Hsml loop) and hydro force through the C-ABI."""
import ctypes as C

import numpy as np
import pytest

import shenqi_amd as sq
from shenqi_amd import capi
import orc
import common as cm

pytestmark = pytest.mark.gpu


def _setup(pos, hsml, lastisbh=False, kernel=1):
    pman, SphP, BhP = cm.make_gas(pos, hsml, lastisbh=lastisbh)
    BhP = np.zeros(2, dtype=sq.BH_SLOT_DTYPE)
    sq.set_densitypar(DensityResolutionEta=1.0, MaxNumNgbDeviation=0.5, DensityKernelType=kernel, BlackHoleNgbFactor=2.0,
                      MinGasHsml=0.006)
    return pman, SphP, BhP


def _do_density_test(ctx, pos, hsml, expected, err, lastisbh=False):
    """do_density_test, tests/test_density.cpp:134-206, through the host mirror of the reference API."""
    pman, SphP, BhP = _setup(pos, hsml, lastisbh)
    P = pman.Base
    tree = sq.force_tree_rebuild_mask(pman, sq.GASMASK + sq.BHMASK)
    sq.set_init_hsml(tree, pman.BoxSize, pman)
    tree = sq.force_tree_rebuild_mask(pman, sq.GASMASK)
    evp, st = sq.density(ctx, None, 1, 0, 0, None, tree, pman, SphP, BhP)
    # check_densities, tests/test_density.cpp:73-90
    assert np.all(np.isfinite(P["Hsml"])) and np.all(np.isfinite(SphP["Density"])) and np.all(SphP["Density"] > 0)
    assert P["Hsml"].min() >= 0.006 and P["Hsml"].max() <= pman.BoxSize
    assert abs(P["Hsml"].mean() - expected) < err, P["Hsml"].mean()
    h0 = P["Hsml"].copy()
    evp, st2 = sq.density(ctx, None, 1, 0, 0, None, tree, pman, SphP, BhP)
    assert np.all(np.abs(h0 / P["Hsml"] - 1) < 0.5 / sq.GetNumNgb())          # :203
    return pman, SphP, tree, st


def test_reference_gate_density_flat_gpu(ctx):
    pos = cm.grid_positions(32)
    pman, SphP, tree, st = _do_density_test(ctx, pos, np.full(len(pos), 1.5 * cm.BOX / 32), 0.5, 5e-4)
    assert abs(pman.Base["Hsml"].mean() - 0.500387) < 2e-6        # the reference run
    assert abs(SphP["Density"].mean() - 64.002) < 2e-3
    print("density flat: %d iterations, %.1f candidates/target, %.2f ms" % (st.niterations, st.ninteractions / st.ntargets, st.kernel_ms))


def test_reference_gate_density_close_gpu(ctx):
    pos, hsml = cm.density_close_positions()
    _do_density_test(ctx, pos, hsml, 0.131726, 1e-4, lastisbh=True)


def test_reference_gate_density_random_gpu(ctx):
    n = 32**3
    pos = cm.random_positions(orc.boost_mt19937_uniform(0, 3 * n), n)
    pman, SphP, tree, st = _do_density_test(ctx, pos, np.full(n, cm.BOX / 32), 0.187515, 1e-3)
    # the loop starts from Hsml = BOX / 32, well above where most particles end: the stats report the largest radius any walk of the
    # loop searched with (what a sharded caller's halo must cover), not the largest final one
    assert st.hsml_max_tried >= cm.BOX / 32 and st.hsml_max_tried >= pman.Base["Hsml"].max()
    assert pman.Base["Hsml"].mean() < 0.7 * st.hsml_max_tried


@pytest.mark.parametrize("kernel", [1, 2, 4])
@pytest.mark.parametrize("DoEgy", [0, 1])
def test_density_parity_fixed_hsml(ctx, kernel, DoEgy):
    """L1 ladder: identical Hsml in => identical candidate counts (integers) and all density outputs to
    rounding, for all three kernels, against the oracle."""
    n = 16**3
    pos = cm.random_positions(orc.boost_mt19937_uniform(3, 3 * n), n)
    rng = np.random.default_rng(kernel)
    hsml = cm.BOX / 16 * rng.uniform(1.0, 2.5, size=n)
    pman, SphP, BhP = _setup(pos, hsml, kernel=kernel)
    P = pman.Base
    P["Vel"] = rng.normal(size=(n, 3))
    SphP["Entropy"] = rng.uniform(0.5, 2.0, size=n)
    tree = sq.force_tree_rebuild_mask(pman, sq.GASMASK)
    dp = cm.density_params(kernel=kernel, update_hsml=0, DoEgyDensity=DoEgy)
    st = orc.SphState(P, SphP, BhP)
    rc, oevp, ogr, _, onint = orc.density(tree.Nodes_base.copy(), tree.firstnode, None, st, dp, want_gradrho=True)
    assert rc == 0
    gmag = np.zeros(n)
    evp, gs = sq.density(ctx, None, 0, DoEgy, 0, None, tree, pman, SphP, BhP, GradRho_mag=gmag)
    assert gs.ninteractions == onint
    for name, ref in (("Density", st.density), ("EgyWtDensity", st.egywtdensity), ("DhsmlEgyDensityFactor", st.dhsmlegydensityfactor),
                      ("DivVel", st.divvel), ("CurlVel", st.curlvel)):
        got = SphP[name]
        if name == "DhsmlEgyDensityFactor":
            # 1/(1 + dlnrho/dlnh * ...) is singular where the denominator crosses zero: compare where it is
            # well conditioned, with the error amplification (1 + |f|)^2 of that map
            ok = np.isfinite(ref) & (np.abs(ref) < 100)
            assert ok.mean() > 0.75
            assert np.all(np.abs(got[ok] - ref[ok]) < 1e-11 * (1 + np.abs(ref[ok])) ** 2), name
        else:
            assert np.abs(got - ref).max() < 1e-10 * np.abs(ref).max(), name
    assert np.abs(P["DtHsml"] - st.dthsml).max() < 1e-10 * np.abs(st.dthsml).max()
    assert np.abs(evp - oevp).max() < 1e-13
    assert np.abs(gmag - np.linalg.norm(ogr, axis=1)).max() < 1e-10 * np.linalg.norm(ogr, axis=1).max()
    assert np.array_equal(P["Hsml"], hsml)       # update_hsml = 0 leaves Hsml alone


def test_density_hsml_loop_parity(ctx):
    """Whole Hsml iteration vs the oracle: same iteration count, Hsml equal to rounding, hmax of the leaves
    raised as update_tree_hmax_father does; active subset leaves the others untouched."""
    n = 16**3
    pos = cm.random_positions(orc.boost_mt19937_uniform(9, 3 * n), n)
    pman, SphP, BhP = _setup(pos, np.full(n, cm.BOX / 16))
    P = pman.Base
    tree = sq.force_tree_rebuild_mask(pman, sq.GASMASK)
    sq.set_init_hsml(tree, pman.BoxSize, pman)
    tree = sq.force_tree_rebuild_mask(pman, sq.GASMASK)
    act = np.arange(0, n, 3, dtype=np.int32)
    dp = cm.density_params(DoEgyDensity=1)
    st = orc.SphState(P, SphP, BhP)
    onodes = tree.Nodes_base.copy()
    fp = capi.host.shqh_tree_father(tree._h)
    father = np.frombuffer((C.c_char * (4 * n)).from_address(fp), dtype=np.int32).copy()
    rc, oevp, _, oniter, _ = orc.density(onodes, tree.firstnode, father, st, dp, active=act)
    h_before = P["Hsml"].copy()
    evp, gs = sq.density(ctx, act, 1, 1, 0, None, tree, pman, SphP, BhP)
    assert gs.niterations == oniter
    assert np.abs(P["Hsml"][act] - st.hsml[act]).max() < 1e-12
    mask = np.ones(n, dtype=bool)
    mask[act] = False
    assert np.array_equal(P["Hsml"][mask], h_before[mask])
    assert np.abs(SphP["Density"][act] - st.density[act]).max() < 1e-10 * st.density.max()
    assert np.abs(tree.Nodes_base["hmax"] - onodes["hmax"]).max() < 1e-12


@pytest.mark.parametrize("kernel,disph", [(1, 1), (2, 1), (4, 0)])
def test_hydro_parity(ctx, kernel, disph):
    """hydro_force vs the oracle (runtests.cpp:507-539 bar: max < 1e-5 on |a| and MaxSignalVel; here rounding level)."""
    n = 16**3
    pos = cm.random_positions(orc.boost_mt19937_uniform(21, 3 * n), n)
    rng = np.random.default_rng(4)
    pman, SphP, BhP = _setup(pos, np.full(n, cm.BOX / 16), kernel=kernel)
    P = pman.Base
    P["Vel"] = rng.normal(size=(n, 3)) * 3.0
    SphP["Entropy"] = rng.uniform(0.5, 2.0, size=n)
    tree = sq.force_tree_rebuild_mask(pman, sq.GASMASK)
    sq.set_init_hsml(tree, pman.BoxSize, pman)
    tree = sq.force_tree_rebuild_mask(pman, sq.GASMASK)
    evp, _ = sq.density(ctx, None, 1, disph, 0, None, tree, pman, SphP, BhP)
    sq.force_tree_update_hmax(tree, pman)            # force_tree_calc_moments between density and hydro (run.cpp:493)
    sq.set_hydropar(DensityIndependentSphOn=disph, DensityContrastLimit=100.0, ArtBulkViscConst=0.75)
    st = orc.SphState(P, SphP, BhP)
    hp = cm.hydro_params(kernel=kernel, DensityIndependentSphOn=disph)
    kf = sq.KickFactors()
    for b in range(47):
        kf.dloga_for_bin[b] = 0.01           # exercises the viscosity limiter branch
        hp.kf.dloga_for_bin[b] = 0.01
    onint = orc.hydro(tree.Nodes_base, tree.firstnode, st, hp, evp)
    gs = sq.hydro_force(ctx, None, 0.1, cm.HUBBLE, evp, kf, tree, pman, SphP)
    assert gs.ninteractions == onint
    a, oa = SphP["HydroAccel"], st.hydroaccel
    assert np.abs(a - oa).max() < 1e-10 * np.abs(oa).max()
    assert cm.force_err(a, oa).max() < 1e-5
    assert np.abs(SphP["DtEntropy"] - st.dtentropy).max() < 1e-10 * np.abs(st.dtentropy).max()
    assert np.abs(SphP["MaxSignalVel"] / st.maxsignalvel - 1).max() < 1e-12
    assert np.abs((P["Mass"][:, None] * a).sum(axis=0)).max() < 1e-9 * np.abs(a).sum()      # pair antisymmetry


def test_sph_error_behaviour(ctx):
    pos = cm.grid_positions(8)
    pman, SphP, BhP = _setup(pos, np.full(len(pos), 1.5))
    tree = sq.force_tree_rebuild_mask(pman, sq.GASMASK)
    with pytest.raises(sq.ShqError):
        sq.density(ctx, None, 1, 0, 0, None, tree, pman, SphP, BhP, UseGPU=False)
    with pytest.raises(sq.ShqError):
        sq.hydro_force(ctx, None, 0.1, cm.HUBBLE, None, None, tree, pman, SphP, UseGPU=False)
    dm_tree = sq.force_tree_rebuild_mask(pman, sq.DMMASK)
    with pytest.raises(sq.ShqError):            # tree without gas
        sq.density(ctx, None, 1, 0, 0, None, dm_tree, pman, SphP, BhP)


@pytest.mark.parametrize("kernel,weighting", [(1, 0), (2, 1), (4, 1)])
def test_stellar_density_matches_oracle(ctx, kernel, weighting):
    """shq_stellar_density (stellar_density2.cpp) against the oracle's restatement: same number of Hsml iterations, same
    candidates met, final star Hsml and StarVolumeSPH to rounding (the trial radii go through pow() on both sides)."""
    import test_oracle_cpu as toc
    pman, SphP, ng, nstar = toc._stars_in_gas(n1=16, nstar=500, seed=kernel)
    P = pman.Base
    tree = sq.force_tree_rebuild_mask(pman, sq.GASMASK)
    support = {1: 4, 2: 6, 4: 5}[kernel]
    des = 4.0 / 3 * np.pi * (support / 2.0) ** 3
    queue = np.arange(ng, ng + nstar, dtype=np.int32)[::-1].copy()      # any order
    st = orc.SphState(P, SphP)
    rc, ovol, oniter, onint = orc.stellar_density(tree.Nodes_base, tree.firstnode, st, queue, cm.BOX, des, 2.0, weighting, kernel)
    assert rc == 0
    sp = capi.StellarParams(cm.BOX, des, 2.0, weighting, kernel)
    vol = np.zeros(nstar)
    stats = capi.SphStats()
    pv, tv, sv = pman.view(), tree.view(), capi.sph_view(SphP)
    hs0 = P["Hsml"].copy()
    capi.check(capi.hip.shq_stellar_density(ctx.h, C.byref(tv), C.byref(pv), C.byref(sv), capi.ptr(queue), len(queue), C.byref(sp),
                                            capi.ptr(vol), C.byref(stats)))
    assert stats.niterations == oniter
    assert np.array_equal(P["Hsml"][:ng], hs0[:ng])                      # gas untouched
    assert np.abs(P["Hsml"][ng:] / st.hsml[ng:] - 1).max() < 1e-9
    assert np.abs(vol - ovol[ng:]).max() < 1e-8 * np.abs(ovol[ng:]).max()
    # the device walk keeps the largest trial radius throughout, the reference shrinks it: it meets at least as many candidates
    assert stats.ninteractions >= onint
    # edge cases: empty queue, a non-star in the queue, Hsml == 0
    capi.check(capi.hip.shq_stellar_density(ctx.h, C.byref(tv), C.byref(pv), C.byref(sv), None, 0, C.byref(sp), capi.ptr(vol), None))
    bad = np.array([0], dtype=np.int32)
    assert capi.hip.shq_stellar_density(ctx.h, C.byref(tv), C.byref(pv), C.byref(sv), capi.ptr(bad), 1, C.byref(sp), capi.ptr(vol), None) != 0
    P["Hsml"][ng] = 0
    assert capi.hip.shq_stellar_density(ctx.h, C.byref(tv), C.byref(pv), C.byref(sv), capi.ptr(queue), len(queue), C.byref(sp), capi.ptr(vol),
                                        None) != 0


def test_bh_veldisp_matches_oracle(ctx):
    """shq_bh_veldisp (veldisp2.cpp:20-199) against the oracle: neighbour counts as integers, moments and VDisp to rounding,
    outputs placed by black-hole slot, an active list that skips some holes, swallowed holes ignored."""
    import test_oracle_cpu as toc
    pman, kf, nd, nbh = toc._bhs_in_dm(n1=16, nbh=150, seed=9)
    P = pman.Base
    P["Flags"][nd + 3] = 2                                   # a swallowed black hole has no work
    tree = sq.force_tree_rebuild_mask(pman, sq.DMMASK)
    st = orc.SphState(P, np.zeros(1, dtype=sq.SPH_DTYPE))
    act = np.arange(nd - 10, nd + nbh - 20, dtype=np.int32)  # some DM (no work), most of the holes
    work = np.array([i for i in act if P["Type"][i] == 5 and not (P["Flags"][i] & 3)], dtype=np.int32)
    out, vd = orc.bh_veldisp(tree.Nodes_base, tree.firstnode, st, work, cm.BOX, kf)
    num = np.full(nbh, -1.0); v1 = np.full((nbh, 3), np.nan); v2 = np.full(nbh, np.nan); vdisp = np.full(nbh, -7.0)
    pv, tv = pman.view(), tree.view()
    capi.check(capi.hip.shq_bh_veldisp(ctx.h, C.byref(tv), C.byref(pv), capi.ptr(act), len(act), C.byref(kf), capi.ptr(num), capi.ptr(v1),
                                       capi.ptr(v2), capi.ptr(vdisp)))
    slots = P["PI"][work]
    assert np.array_equal(num[slots], out[:, 0]) and out[:, 0].min() > 0
    assert np.abs(v1[slots] - out[:, 1:4]).max() < 1e-10 * np.abs(out[:, 1:4]).max()
    assert np.abs(v2[slots] / out[:, 4] - 1).max() < 1e-12
    assert np.abs(vdisp[slots] / vd - 1).max() < 1e-10
    untouched = np.setdiff1d(np.arange(nbh), slots)
    assert len(untouched) > 0 and np.all(num[untouched] == -1.0) and np.all(vdisp[untouched] == -7.0)


def test_wind_veldisp_matches_oracle(ctx):
    """shq_wind_veldisp (veldisp2.cpp:203-528) against the oracle: same number of iterations of the five-radius loop, VDisp
    to rounding (neighbour counts are integers, so the converged sets are the same), written by gas slot."""
    import test_oracle_cpu as toc
    pman, kf, nd, ngas = toc._gas_in_dm(n1=16, ngas=400, seed=21)
    P = pman.Base
    tree = sq.force_tree_rebuild_mask(pman, sq.DMMASK)
    st = orc.SphState(P, np.zeros(ngas, dtype=sq.SPH_DTYPE))
    queue = np.arange(nd, nd + ngas, dtype=np.int32)[::3].copy()
    Time, hubble = 0.25, 3.0
    rc, ovd, odm, oniter = orc.wind_veldisp(tree.Nodes_base, tree.firstnode, st, queue, cm.BOX, kf, Time, hubble)
    assert rc == 0
    vdisp = np.full(ngas, -5.0)
    stats = capi.SphStats()
    pv, tv = pman.view(), tree.view()
    capi.check(capi.hip.shq_wind_veldisp(ctx.h, C.byref(tv), C.byref(pv), capi.ptr(queue), len(queue), C.byref(kf), Time, hubble,
                                         capi.ptr(vdisp), C.byref(stats)))
    assert stats.niterations == oniter
    slots = P["PI"][queue]
    assert np.abs(vdisp[slots] / ovd - 1).max() < 1e-9
    untouched = np.setdiff1d(np.arange(ngas), slots)
    assert np.all(vdisp[untouched] == -5.0)
    # a non-gas particle in the queue is refused
    bad = np.array([0], dtype=np.int32)
    assert capi.hip.shq_wind_veldisp(ctx.h, C.byref(tv), C.byref(pv), capi.ptr(bad), 1, C.byref(kf), Time, hubble, capi.ptr(vdisp), None) != 0


@pytest.mark.parametrize("method", [0, 1, 2])
def test_bh_dynfric_matches_oracle(ctx, method):
    """shq_bh_dynfric (bhdynfric.cpp:44-295): the potential minimum (same particle as the oracle, depth-first tie-breaking),
    its reduce into the caller's MinPot arrays, and the post-processed friction sums."""
    import test_oracle_cpu as toc
    pman, kf, nd, nbh = toc._bhs_in_stars_and_dm(n1=16, nbh=120, seed=30 + method)
    P = pman.Base
    mask = {0: sq.ALLMASK, 1: sq.STARMASK + sq.BHMASK, 2: sq.STARMASK + sq.BHMASK + sq.DMMASK}[method]
    tree = sq.force_tree_rebuild_mask(pman, mask)
    st = orc.SphState(P, np.zeros(1, dtype=sq.SPH_DTYPE))
    queue = np.arange(nd, nd + nbh, dtype=np.int32)[::2].copy()
    raw = orc.bh_dynfric(tree.Nodes_base, tree.firstnode, st, P["Potential"], queue, cm.BOX, kf, method, 2, mask)
    slots = P["PI"][queue]
    # blackhole_init_potential (:296-310): start from the hole's own potential and position; make half of them unbeatable
    minpot = np.full(nbh, 1e29); minpos = np.zeros((nbh, 3)); minvel = np.zeros((nbh, 3)); upd = np.zeros(nbh, dtype=np.int32)
    minpot[slots] = P["Potential"][queue]
    minpos[slots] = P["Pos"][queue]
    minpot[slots[::2]] = -1e30
    want_pot, want_pos = minpot.copy(), minpos.copy()
    better = raw[:, 0] < minpot[slots]
    want_pot[slots[better]] = raw[better, 0]
    want_pos[slots[better]] = raw[better, 1:4]
    dens = np.full(nbh, np.nan); dvel = np.full((nbh, 3), np.nan); drms = np.full(nbh, np.nan)
    out = capi.BhDynFricOut(capi.ptr(minpot), capi.ptr(minpos), capi.ptr(minvel), capi.ptr(upd), capi.ptr(dens), capi.ptr(dvel), capi.ptr(drms))
    pv, tv = pman.view(), tree.view()
    capi.check(capi.hip.shq_bh_dynfric(ctx.h, C.byref(tv), C.byref(pv), capi.ptr(queue), len(queue), C.byref(kf), method, 2, mask, C.byref(out)))
    assert np.array_equal(minpot, want_pot) and np.array_equal(minpos, want_pos)
    assert np.array_equal(np.nonzero(upd)[0], np.sort(slots[better])) and better.any() and (~better).any()
    assert np.array_equal(minvel[slots[better]], raw[better, 4:7])
    if method > 0:
        d = raw[:, 7]
        assert (d > 0).any()
        assert np.abs(dens[slots] - d).max() < 1e-11 * d.max()
        pos = d > 0
        assert np.abs(dvel[slots][pos] - raw[pos, 8:11] / d[pos, None]).max() < 1e-9 * np.abs(raw[pos, 8:11] / d[pos, None]).max()
        assert np.abs(drms[slots][pos] / np.sqrt(raw[pos, 11] / d[pos]) - 1).max() < 1e-11
    else:
        assert np.all(np.isnan(dens))
