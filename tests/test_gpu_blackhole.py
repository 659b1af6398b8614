"""GPU parity tests of the black-hole accretion and feedback walks (shq_bh_accretion / shq_bh_feedback, csrc/sph.hip) against the
restatement of libgadget/blackhole.cpp:373-1003 in oracle/blackhole.py (brute-force neighbours).  The reference's tests hold no
fixture for this module (parity unpinned): besides the restatement the tests check what the walks conserve."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import shenqi_amd as sq
from shenqi_amd import capi
import common as cm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import blackhole as obh  # noqa: E402

pytestmark = pytest.mark.gpu

from blackhole_fixtures import params, setup, make_work  # noqa: E402


def close(a, b, rtol=1e-11):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-300) if b.size else 1.0
    return a.shape == b.shape and np.abs(a - b).max(initial=0.0) <= rtol * scale


def run_both(ctx, seed, queue_stride=1, bh_hsml=1.0, size=None, **kw):
    cp, prm = params(**kw)
    pman, S, B, kf, rnd, bi = setup(seed, bh_hsml=bh_hsml, **(size or {}))
    P = pman.Base
    n, ngas, nbh = len(P), len(S), len(B)
    ids = np.ascontiguousarray(P["ID"])
    queue = np.ascontiguousarray(bi[::queue_stride].astype(np.int32))
    Ti = 1 << 18                                               # bins 16..18 active, 19 and 20 not
    tree = sq.force_tree_rebuild_mask(pman, sq.GASMASK + sq.BHMASK)
    sq.force_tree_update_hmax(tree, pman)
    # ---- oracle on copies
    oP, oS, oB = P.copy(), S.copy(), B.copy()
    ow, _ = make_work(ngas, nbh)
    obh.accretion(oP, oS, oB, ids, queue, kf, prm, Ti, rnd, ow)
    # ---- device
    w, cw = make_work(ngas, nbh)
    pv, tv, sv, bv = pman.view(), tree.view(), capi.sph_view(S), capi.bh_slot_view(B)
    capi.check(capi.hip.shq_bh_accretion(ctx.h, C.byref(tv), C.byref(pv), C.byref(sv), C.byref(bv), capi.ptr(ids), capi.ptr(queue), len(queue), C.byref(kf),
                                         C.byref(cp), Ti, capi.ptr(rnd), len(rnd), C.byref(cw)))
    acc = dict(P=P.copy(), S=S.copy(), B=B.copy(), w={k: v.copy() for k, v in w.items()}, oP=oP.copy(), oS=oS.copy(), oB=oB.copy(),
               ow={k: v.copy() for k, v in ow.items()}, queue=queue, prm=prm)
    # ---- feedback
    eeqos = (np.arange(n) % 3 == 0).astype(np.uint8)
    onsph, onbh = obh.feedback(oP, oS, oB, ids, queue, kf, prm, len(P) + 50, rnd, eeqos, ow)
    ns, nb = C.c_int64(), C.c_int64()
    capi.check(capi.hip.shq_bh_feedback(ctx.h, C.byref(tv), C.byref(pv), C.byref(sv), C.byref(bv), capi.ptr(ids), capi.ptr(queue), len(queue), C.byref(kf),
                                        C.byref(cp), len(P) + 50, capi.ptr(rnd), len(rnd), capi.ptr(eeqos), C.byref(cw), C.byref(ns), C.byref(nb)))
    fb = dict(P=P, S=S, B=B, w=w, oP=oP, oS=oS, oB=oB, ow=ow, counts=(ns.value, nb.value), ocounts=(onsph, onbh))
    return acc, fb


def check_accretion(a):
    w, ow, B, oB = a["w"], a["ow"], a["B"], a["oB"]
    assert np.array_equal(w["SPH_SwallowID"], ow["SPH_SwallowID"]) and np.array_equal(w["BH_SwallowID"], ow["BH_SwallowID"])
    assert np.array_equal(B["encounter"], oB["encounter"])
    for k in ("BH_FeedbackWeightSum", "BH_Entropy", "BH_SurroundingGasVel", "MgasEnc"):
        assert close(w[k], ow[k]), k
    assert np.array_equal(w["KEflag"], ow["KEflag"])
    for k in ("Mdot", "Mass", "DragAccel", "KineticFdbkEnergy"):
        assert close(B[k], oB[k], 1e-12), k
    for k in ("Density", "Mtrack", "CountProgs", "SwallowID", "minTimeBin", "DFAccel"):                       # untouched by the accretion walk
        assert np.array_equal(B[k], oB[k]), k
    assert np.array_equal(a["P"]["Flags"], a["oP"]["Flags"]) and np.array_equal(a["S"]["Entropy"], a["oS"]["Entropy"])


def check_feedback(f):
    P, oP, S, oS, B, oB, w, ow = f["P"], f["oP"], f["S"], f["oS"], f["B"], f["oB"], f["w"], f["ow"]
    assert f["counts"] == f["ocounts"]
    assert np.array_equal(P["Flags"], oP["Flags"]) and np.array_equal(P["Type"], oP["Type"])
    assert np.array_equal(S["ReverseLink"], oS["ReverseLink"])
    assert close(S["Entropy"], oS["Entropy"], 1e-12) and close(P["Vel"], oP["Vel"], 1e-12) and close(P["Mass"], oP["Mass"], 1e-7)
    for k in ("BH_accreted_Mass", "BH_accreted_BHMass", "BH_accreted_momentum"):
        assert close(w[k], ow[k]), k
    for k in ("SwallowID", "encounter", "CountProgs", "minTimeBin"):
        assert np.array_equal(B[k], oB[k]), k
    for k in ("Mass", "Mtrack", "SwallowTime", "KineticFdbkEnergy"):
        assert close(B[k], oB[k], 1e-12), k


def test_accretion_and_thermal_feedback_with_repositioning(ctx):
    """repositioning on: every close pair merges (larger ID swallows, or the active one an inactive one); stochastic gas swallowing;
    thermal energy into the kernel with some particles reaching the temperature cap; BHHeated on the flagged particles"""
    a, f = run_both(ctx, 3)
    check_accretion(a)
    ow = a["ow"]
    assert (ow["BH_SwallowID"] != 0).sum() >= 8 and 5 < (ow["SPH_SwallowID"] != 0).sum() < 600
    assert (a["oB"]["encounter"] == 1).sum() >= 16 and (a["oB"]["Mdot"] > 0).sum() > 20
    # the Eddington cap binds for some holes and not for others
    medd = a["prm"].EddingtonConst * a["B"]["Mass"] * a["prm"].UnitTime_in_s / a["prm"].HubbleParam
    capped = np.isclose(a["oB"]["Mdot"], a["prm"].BlackHoleEddingtonFactor * medd, rtol=1e-3)
    assert capped.any() and (~capped & (a["oB"]["Mdot"] > 0)).any()
    check_feedback(f)
    P, oP, S = f["P"], f["oP"], f["S"]
    assert f["counts"][0] > 5 and f["counts"][1] >= 5
    assert ((P["Flags"] & 8) != 0).sum() > 10 and (np.isclose(S["Entropy"] * (S["Density"] * 64.0) ** (2 / 3) / (2 / 3), 1.5e4, rtol=1e-9)).any()
    # what the walk conserves: the dynamical mass of the swallowed gas and holes went to their swallowers
    gone_gas = ((P["Flags"] & 1) != 0) & (P["Type"] == 0) & ((a["P"]["Flags"] & 1) == 0)
    gone_bh = ((P["Flags"] & 2) != 0) & (P["Type"] == 5)
    assert np.isclose(f["w"]["BH_accreted_Mass"].sum(), a["P"]["Mass"][gone_gas].astype(np.float64).sum() + a["P"]["Mass"][gone_bh].astype(np.float64).sum(), rtol=1e-12)
    assert np.isclose(f["w"]["BH_accreted_BHMass"].sum(), a["B"]["Mass"][P["PI"][gone_bh]].sum(), rtol=1e-12)
    assert (f["B"]["SwallowTime"][P["PI"][gone_bh]] == 0.25).all()


def test_accretion_bound_mergers_seed_mass_drag_and_winds(ctx):
    """no repositioning: check_grav_bound decides the mergers; SeedBHDynMass regime (Mtrack instead of the particle mass, the three
    branches of the postprocess); drag acceleration; wind particles ignored; only every second hole active"""
    a, f = run_both(ctx, 4, queue_stride=2, RepositionEnabled=0, MergeGravBound=1, SeedBHDynMass=2.0, BH_DRAG=1, WindsDecoupleSph=1, DensityKernelType=4)
    check_accretion(a)
    enc, marked = (a["oB"]["encounter"] == 1).sum(), (a["ow"]["BH_SwallowID"] != 0).sum()
    assert enc >= 4 and np.abs(a["oB"]["DragAccel"]).max() > 0
    check_feedback(f)
    a2, _ = run_both(ctx, 4, queue_stride=2, RepositionEnabled=0, MergeGravBound=0, SeedBHDynMass=2.0, BH_DRAG=2, WindsDecoupleSph=1, DensityKernelType=4)
    check_accretion(a2)
    assert (a2["ow"]["BH_SwallowID"] != 0).sum() > marked            # unbound pairs merge once the test is off
    B, oB = f["B"], f["oB"]
    grew = B["Mtrack"] != a["B"]["Mtrack"]
    assert grew.any() and (B["Mtrack"][grew] <= 2.0).all()


def test_kinetic_feedback(ctx):
    """BlackHoleKineticOn: low-Eddington holes accumulate KineticFdbkEnergy, those above their threshold release it as kicks in the
    directions get_random_dir draws (no thermal energy from them in that step), and are reset"""
    a, f = run_both(ctx, 5, BlackHoleKineticOn=1, DensityKernelType=2)
    check_accretion(a)
    ke = a["ow"]["KEflag"]
    assert (ke == 1).any() and (ke == 2).any()
    check_feedback(f)
    kicked = np.any(f["P"]["Vel"] != a["P"]["Vel"], axis=1) & (f["P"]["Type"] == 0)
    assert kicked.sum() > 20
    rel = f["B"]["KineticFdbkEnergy"][(ke == 2) & (a["ow"]["BH_SwallowID"] == 0)]
    assert len(rel) > 0 and (rel == 0).all()


def test_large_kernels_overflow_the_lane_lists(ctx):
    """black holes with ~400-700 neighbours: the per-lane neighbour lists (256 entries) are flushed in the middle of the walk"""
    a, f = run_both(ctx, 8, bh_hsml=2.0, BH_DRAG=1)
    check_accretion(a)
    check_feedback(f)
    assert (a["ow"]["MgasEnc"] == 0).all() and (a["ow"]["SPH_SwallowID"] != 0).sum() > 50


def test_many_black_holes_span_several_workgroups(ctx):
    """700 black holes in 27 000 gas particles: 11 waves in 3 workgroups of the walk kernels (the other cases fit one wave)"""
    a, f = run_both(ctx, 12, size=dict(ngrid=30, nbh=700, ndm=500), BH_DRAG=2, SeedBHDynMass=1.5)
    check_accretion(a)
    check_feedback(f)
    assert (a["ow"]["SPH_SwallowID"] != 0).sum() > 500 and f["counts"][0] > 400


def test_bh_walk_argument_errors(ctx):
    cp, prm = params()
    pman, S, B, kf, rnd, bi = setup(6, ngrid=6, nbh=20, ndm=10)
    P = pman.Base
    ids = np.ascontiguousarray(P["ID"])
    tree = sq.force_tree_rebuild_mask(pman, sq.GASMASK + sq.BHMASK)
    sq.force_tree_update_hmax(tree, pman)
    w, cw = make_work(len(S), len(B))
    pv, tv, sv, bv = pman.view(), tree.view(), capi.sph_view(S), capi.bh_slot_view(B)
    gas = np.flatnonzero(P["Type"] == 0)[:1].astype(np.int32)
    with pytest.raises(sq.ShqError):                     # not a black hole
        capi.check(capi.hip.shq_bh_accretion(ctx.h, C.byref(tv), C.byref(pv), C.byref(sv), C.byref(bv), capi.ptr(ids), capi.ptr(gas), 1, C.byref(kf), C.byref(cp), 0,
                                             capi.ptr(rnd), len(rnd), C.byref(cw)))
    q = bi[:3].astype(np.int32)
    with pytest.raises(sq.ShqError):                     # empty random table
        capi.check(capi.hip.shq_bh_accretion(ctx.h, C.byref(tv), C.byref(pv), C.byref(sv), C.byref(bv), capi.ptr(ids), capi.ptr(q), 3, C.byref(kf), C.byref(cp), 0,
                                             capi.ptr(rnd), 0, C.byref(cw)))
    # an empty queue zeroes the marks and changes nothing else
    capi.check(capi.hip.shq_bh_accretion(ctx.h, C.byref(tv), C.byref(pv), C.byref(sv), C.byref(bv), capi.ptr(ids), None, 0, C.byref(kf), C.byref(cp), 0,
                                         capi.ptr(rnd), len(rnd), C.byref(cw)))
    assert (w["SPH_SwallowID"] == 0).all() and (w["BH_SwallowID"] == 0).all() and (B["encounter"] == 7).all()


def _sequence(ctx, current):
    """accretion -> feedback -> accretion on the same views; with `current` the second and third call take particles, SPH state, tree and
    IDs from the context (shq_set_inputs_current) instead of uploading the views again"""
    cp, prm = params()
    pman, S, B, kf, rnd, bi = setup(3)
    P = pman.Base
    n, ngas, nbh = len(P), len(S), len(B)
    ids = np.ascontiguousarray(P["ID"])
    queue = np.ascontiguousarray(bi.astype(np.int32))
    Ti = 1 << 18
    tree = sq.force_tree_rebuild_mask(pman, sq.GASMASK + sq.BHMASK)
    sq.force_tree_update_hmax(tree, pman)
    w, cw = make_work(ngas, nbh)
    pv, tv, sv, bv = pman.view(), tree.view(), capi.sph_view(S), capi.bh_slot_view(B)
    eeqos = (np.arange(n) % 3 == 0).astype(np.uint8)
    ns, nb = C.c_int64(), C.c_int64()
    snaps = []

    def snap():
        snaps.append(dict(P=P.copy(), S=S.copy(), B=B.copy(), w={k: v.copy() for k, v in w.items()}, counts=(ns.value, nb.value)))

    def accretion():
        capi.check(capi.hip.shq_bh_accretion(ctx.h, C.byref(tv), C.byref(pv), C.byref(sv), C.byref(bv), capi.ptr(ids), capi.ptr(queue), len(queue), C.byref(kf),
                                             C.byref(cp), Ti, capi.ptr(rnd), len(rnd), C.byref(cw)))
        snap()

    capi.check(capi.hip.shq_set_inputs_current(ctx.h, 0))
    accretion()
    if current:
        capi.check(capi.hip.shq_set_inputs_current(ctx.h, capi.CURRENT_PARTICLES | capi.CURRENT_SPH | capi.CURRENT_TREE | capi.CURRENT_IDS))
    capi.check(capi.hip.shq_bh_feedback(ctx.h, C.byref(tv), C.byref(pv), C.byref(sv), C.byref(bv), capi.ptr(ids), capi.ptr(queue), len(queue), C.byref(kf),
                                        C.byref(cp), len(P) + 50, capi.ptr(rnd), len(rnd), capi.ptr(eeqos), C.byref(cw), C.byref(ns), C.byref(nb)))
    snap()
    # the holes that were swallowed are no targets any more (blackhole_haswork)
    alive = np.ascontiguousarray(queue[(P["Flags"][queue] & 3) == 0])
    queue = alive
    accretion()
    capi.check(capi.hip.shq_set_inputs_current(ctx.h, 0))
    return snaps


def test_inputs_current_gives_the_uploading_calls_bits(ctx):
    """shq_set_inputs_current: a feedback walk and a second accretion walk that take their inputs from the context — the particles as the
    first call uploaded them and the feedback walk changed them in place — end in the bits of the calls that upload the views every time"""
    a, b = _sequence(ctx, False), _sequence(ctx, True)
    assert len(a) == len(b) == 3 and a[1]["counts"] == b[1]["counts"] and a[1]["counts"][0] > 5
    for x, y in zip(a, b):
        for rec in ("P", "S", "B"):
            for name in x[rec].dtype.names:                  # field by field: the records' padding is not initialised
                assert np.array_equal(x[rec][name], y[rec][name], equal_nan=True), (rec, name)
        for k in x["w"]:
            assert np.array_equal(x["w"][k], y["w"][k]), k
    assert capi.hip.shq_set_inputs_current(ctx.h, 16) != 0       # not a combination of SHQ_CURRENT_*
