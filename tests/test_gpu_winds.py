"""GPU parity tests of the wind walks for new stars (shq_winds_and_feedback, csrc/sph.hip) against the restatement of
libgadget/winds.cpp:227-565 in oracle/winds.py.  No reference fixture exists (parity unpinned); the outcome is order-independent
by construction (nearest star, then smaller ID), so the comparison is exact."""
import ctypes as C
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest

import shenqi_amd as sq
from shenqi_amd import capi
import common as cm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import winds as ow  # noqa: E402

pytestmark = pytest.mark.gpu

PARAMS = dict(BoxSize=cm.BOX, Time=0.2, WindFreeTravelLength=20.0, MaxWindFreeTravelTime=0.6, WindEfficiency=2.0, WindSpeed=350.0, WindSigma0=353.0,
              WindSpeedFactor=3.7, MinWindVelocity=100.0, WindThermalFactor=0.0, WindModel=ow.WIND_USE_HALO + ow.WIND_DECOUPLE_SPH)


def params(**kw):
    d = dict(PARAMS)
    d.update(kw)
    p = capi.WindParams()
    for k, v in d.items():
        setattr(p, k, v)
    return p, SimpleNamespace(**d)


def setup(seed, ngrid=16, nstar=400, nnew=60, hscale=1.0):
    rng = np.random.default_rng(seed)
    ngas = ngrid**3
    sp = cm.BOX / ngrid
    gas = np.mod(cm.grid_positions(ngrid) + rng.normal(size=(ngas, 3)) * 0.3 * sp, cm.BOX)
    stars = rng.random((nstar, 3)) * cm.BOX
    stars[1] = stars[0] + 0.3 * sp                          # neighbouring new stars compete for the same gas
    stars[3] = stars[2]                                     # two stars at the same place: the tie goes to the smaller ID
    pos = np.concatenate([gas, stars, rng.random((200, 3)) * cm.BOX])
    n = len(pos)
    types = np.concatenate([np.zeros(ngas, np.uint8), np.full(nstar, 4, np.uint8), np.ones(200, np.uint8)])
    perm = rng.permutation(n)
    inv = np.empty(n, dtype=np.int64)
    inv[perm] = np.arange(n)
    pos, types = pos[perm], types[perm]
    pman = sq.PartManager(n, cm.BOX)
    P = pman.Base
    P["Pos"], P["Type"] = pos, types
    P["Mass"] = rng.uniform(0.8, 1.2, n).astype(np.float32)
    P["ID"] = rng.permutation(n).astype(np.uint64) + 7
    P["Vel"] = rng.normal(size=(n, 3)) * 30
    P["Hsml"] = sp * rng.uniform(1.0, 2.2, n) * hscale
    isgas, isstar = types == 0, types == 4
    P["PI"][isgas] = rng.permutation(ngas)
    P["PI"][isstar] = rng.permutation(nstar)
    gi = np.flatnonzero(isgas)
    P["Flags"][gi[:7]] |= 1
    S = np.zeros(ngas, dtype=capi.SPH_DTYPE)
    S["Entropy"] = rng.uniform(50, 150, ngas)
    S["Density"] = rng.uniform(0.5, 2.0, ngas) * ngas / cm.BOX**3
    S["DelayTime"] = np.where(rng.random(ngas) < 0.2, 0.4, 0.0)
    ST = np.zeros(nstar, dtype=capi.STAR_DTYPE)
    ST["VDisp"] = rng.uniform(5, 60, nstar).astype(np.float32)
    ST["VDisp"][:5] = 0                                     # no dispersion found: no wind from these
    new = inv[ngas + np.arange(nnew)].astype(np.int32)      # stars 0 .. nnew-1 are the new ones, in shuffled particle order
    P["Hsml"][new[3]] = P["Hsml"][new[2]]
    far = new[-1]
    P["Hsml"][far] = 1e-3 * sp                              # a star with no gas inside its Hsml: TotalWeight 0
    rnd = rng.random(2053)
    return pman, S, ST, rnd, np.ascontiguousarray(rng.permutation(new))


def run_both(ctx, seed, hscale=1.0, size=None, **kw):
    cp, prm = params(**kw)
    pman, S, ST, rnd, new = setup(seed, hscale=hscale, **(size or {}))
    P = pman.Base
    ids = np.ascontiguousarray(P["ID"])
    tree = sq.force_tree_rebuild_mask(pman, sq.GASMASK)
    oP, oS = P.copy(), S.copy()
    otw, okicks, oapplied = ow.winds_and_feedback(oP, oS, ST, ids, new, prm, rnd)
    tw = np.full(len(ST), -1.0)
    kicks = np.zeros(max(4 * len(okicks), 16), dtype=capi.WIND_KICK_DTYPE)
    nk, na = C.c_int64(), C.c_int64()
    pv, tv, sv = pman.view(), tree.view(), capi.sph_view(S)
    stv = capi.StarView(ST.ctypes.data, ST.dtype.itemsize, len(ST), ST.dtype.fields["VDisp"][1])
    P0, S0 = P.copy(), S.copy()
    capi.check(capi.hip.shq_winds_and_feedback(ctx.h, C.byref(tv), C.byref(pv), C.byref(sv), C.byref(stv), capi.ptr(ids), capi.ptr(new), len(new), C.byref(cp),
                                               capi.ptr(rnd), len(rnd), capi.ptr(tw), capi.ptr(kicks), len(kicks), C.byref(nk), C.byref(na)))
    slots = P["PI"][new]
    assert np.array_equal(tw[slots], otw[slots]) and (np.delete(tw, slots) == -1).all()
    assert nk.value == len(okicks) and na.value == oapplied
    got = [(int(k["part_index"]), float(k["StarDistance"]), int(k["StarID"]), float(k["StarKickVelocity"]), float(k["StarTherm"])) for k in kicks[:nk.value]]
    assert got == okicks
    # the kick itself runs acos / sin / cos / pow of two libms (glibc here, numpy's in the restatement): last-bit differences
    assert np.abs(P["Vel"] - oP["Vel"]).max() < 1e-12 and np.abs(S["Entropy"] / oS["Entropy"] - 1).max() < 1e-14 and np.array_equal(S["DelayTime"], oS["DelayTime"])
    assert np.array_equal(np.any(P["Vel"] != P0["Vel"], axis=1), np.any(oP["Vel"] != P0["Vel"], axis=1))
    return SimpleNamespace(P=P, S=S, P0=P0, S0=S0, tw=tw, kicks=kicks[:nk.value], applied=na.value, new=new, prm=prm, ST=ST)


def test_halo_winds_with_decoupling(ctx):
    r = run_both(ctx, 1)
    assert len(r.kicks) > r.applied > 20                       # several stars reach for the same particle: the nearest wins
    kicked = np.flatnonzero(np.any(r.P["Vel"] != r.P0["Vel"], axis=1))
    assert len(kicked) == r.applied and (r.P["Type"][kicked] == 0).all()
    pi = r.P["PI"][kicked]
    assert (r.S0["DelayTime"][pi] == 0).all() and (r.S["DelayTime"][pi] > 0).all() and (r.S["DelayTime"][pi] <= 0.6).all()
    assert (r.tw[r.P["PI"][r.new]] == 0).any() and (r.tw[r.P["PI"][r.new]] > 10).any()
    # the winner of every contested particle is the nearest star
    for p in np.unique(r.kicks["part_index"]):
        mine = r.kicks[r.kicks["part_index"] == p]
        assert mine["StarDistance"][0] == mine["StarDistance"].min()


def test_fixed_efficiency_winds_with_thermal_energy(ctx):
    r = run_both(ctx, 2, WindModel=ow.WIND_FIXED_EFFICIENCY, WindThermalFactor=0.5)
    kicked = np.flatnonzero(np.any(r.P["Vel"] != r.P0["Vel"], axis=1))
    pi = r.P["PI"][kicked]
    assert r.applied > 20 and (r.S["Entropy"][pi] > r.S0["Entropy"][pi]).all() and np.array_equal(r.S["DelayTime"], r.S0["DelayTime"])
    dv = np.sqrt(((r.P["Vel"][kicked] - r.P0["Vel"][kicked]) ** 2).sum(axis=1))
    assert np.allclose(dv, 350.0 * 0.2, rtol=1e-12)


def test_large_kernels_overflow_the_lane_lists(ctx):
    """stars with up to ~700 gas neighbours: the per-lane lists are flushed in the middle of both walks"""
    r = run_both(ctx, 4, hscale=2.6, WindEfficiency=30.0, WindModel=ow.WIND_FIXED_EFFICIENCY + ow.WIND_DECOUPLE_SPH)
    assert r.tw[r.P["PI"][r.new]].max() > 300 and r.applied > 20


def test_many_new_stars_span_several_workgroups(ctx):
    """900 new stars: 15 waves in 4 workgroups of the walk kernels"""
    r = run_both(ctx, 7, size=dict(ngrid=24, nstar=1200, nnew=900))
    assert r.applied > 300 and len(r.kicks) > r.applied


def test_subgrid_model_and_errors(ctx):
    cp, prm = params(WindModel=ow.WIND_SUBGRID + ow.WIND_USE_HALO)
    pman, S, ST, rnd, new = setup(3, ngrid=8, nstar=50, nnew=10)
    P = pman.Base
    ids = np.ascontiguousarray(P["ID"])
    tree = sq.force_tree_rebuild_mask(pman, sq.GASMASK)
    pv, tv, sv = pman.view(), tree.view(), capi.sph_view(S)
    stv = capi.StarView(ST.ctypes.data, ST.dtype.itemsize, len(ST), ST.dtype.fields["VDisp"][1])
    nk, na = C.c_int64(5), C.c_int64(5)
    P0 = P.copy()

    def call(cp, new):
        return capi.hip.shq_winds_and_feedback(ctx.h, C.byref(tv), C.byref(pv), C.byref(sv), C.byref(stv), capi.ptr(ids), capi.ptr(new), len(new), C.byref(cp),
                                               capi.ptr(rnd), len(rnd), None, None, 0, C.byref(nk), C.byref(na))
    capi.check(call(cp, new))                              # "The subgrid model does nothing here"
    assert nk.value == 0 and na.value == 0 and np.array_equal(P["Vel"], P0["Vel"])
    cp2, _ = params(WindModel=ow.WIND_DECOUPLE_SPH)
    with pytest.raises(sq.ShqError):                       # "WindModel is strange"
        capi.check(call(cp2, new))
    cp3, _ = params()
    gas = np.flatnonzero(P["Type"] == 0)[:2].astype(np.int32)
    with pytest.raises(sq.ShqError):                       # "not a star"
        capi.check(call(cp3, gas))
    capi.check(call(cp3, new[:0].copy()))
    assert nk.value == 0


def test_winds_evolve_and_subgrid_winds(ctx):
    """the wind model's two particle loops: DelayTime ageing / recoupling (winds_evolve) and the subgrid kicks of freshly star-forming
    gas (winds_subgrid, winds_make_after_sf) against their restatements"""
    cp, prm = params(WindModel=ow.WIND_SUBGRID + ow.WIND_USE_HALO + ow.WIND_DECOUPLE_SPH)
    pman, S, ST, rnd, new = setup(6, ngrid=12, nstar=50, nnew=10)
    P = pman.Base
    rng = np.random.default_rng(9)
    ids = np.ascontiguousarray(P["ID"])
    gas = np.flatnonzero((P["Type"] == 0) & ((P["Flags"] & 1) == 0)).astype(np.int32)
    S["VDisp"] = rng.uniform(5, 60, len(S))
    S["DelayTime"] = np.where(rng.random(len(S)) < 0.5, rng.uniform(0.0, 1.0, len(S)), 0.0)       # some above MaxWindFreeTravelTime 0.6
    P["TimeBinHydro"] = rng.integers(16, 22, len(P))
    kf = sq.KickFactors()
    for b in range(47):
        kf.dloga_for_bin[b] = 1e-3 * 2.0 ** (b - 16) if b > 0 else 0.0
    pv, sv = pman.view(), capi.sph_view(S)
    # ---- winds_evolve on a list and on everything
    lst = np.ascontiguousarray(rng.permutation(gas)[:len(gas) // 2])
    for which in (lst, None):
        oP, oS = P.copy(), S.copy()
        thresh = float(np.median(S["Density"])) * 8.0
        ow.winds_evolve(oP, oS, range(len(P)) if which is None else which, 8.0, 0.3, thresh, 0.6, kf)
        capi.check(capi.hip.shq_winds_evolve(ctx.h, C.byref(pv), C.byref(sv), None if which is None else capi.ptr(which), len(P) if which is None else len(which),
                                             8.0, 0.3, thresh, 0.6, C.byref(kf)))
        assert np.array_equal(S["DelayTime"], oS["DelayTime"])
        if which is None:
            assert (S["DelayTime"][P["PI"][gas]] <= 0.6).all()
    assert (S["DelayTime"] == 0).sum() > (len(S) // 2) and (S["DelayTime"] > 0).any()
    # ---- subgrid winds for a list of star-forming gas
    maybe = np.ascontiguousarray(rng.permutation(gas)[:400])
    sm = rng.uniform(0.0, 0.2, len(maybe))
    oP, oS = P.copy(), S.copy()
    P0 = P.copy()
    on = ow.winds_subgrid(oP, oS, ids, maybe, sm, prm, rnd)
    nk = C.c_int64()
    off_vdisp = capi.SPH_DTYPE.fields["VDisp"][1]
    capi.check(capi.hip.shq_winds_subgrid(ctx.h, C.byref(pv), C.byref(sv), off_vdisp, capi.ptr(ids), capi.ptr(maybe), len(maybe), capi.ptr(sm), C.byref(cp),
                                          capi.ptr(rnd), len(rnd), C.byref(nk)))
    assert nk.value == on and 10 < on < len(maybe)
    assert np.array_equal(np.any(P["Vel"] != P0["Vel"], axis=1), np.any(oP["Vel"] != P0["Vel"], axis=1))
    assert np.abs(P["Vel"] - oP["Vel"]).max() < 1e-11 and np.abs(S["Entropy"] / oS["Entropy"] - 1).max() < 1e-14 and np.array_equal(S["DelayTime"], oS["DelayTime"])
    # without the subgrid bit the call does nothing
    cp2, _ = params()
    before = P["Vel"].copy()
    capi.check(capi.hip.shq_winds_subgrid(ctx.h, C.byref(pv), C.byref(sv), off_vdisp, capi.ptr(ids), capi.ptr(maybe), len(maybe), capi.ptr(sm), C.byref(cp2),
                                          capi.ptr(rnd), len(rnd), C.byref(nk)))
    assert nk.value == 0 and np.array_equal(P["Vel"], before)
    stars = np.flatnonzero(P["Type"] == 4)[:2].astype(np.int32)
    with pytest.raises(sq.ShqError):
        capi.check(capi.hip.shq_winds_subgrid(ctx.h, C.byref(pv), C.byref(sv), off_vdisp, capi.ptr(ids), capi.ptr(stars), 2, capi.ptr(sm), C.byref(cp),
                                              capi.ptr(rnd), len(rnd), C.byref(nk)))
