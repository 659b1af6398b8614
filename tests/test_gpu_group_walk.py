"""GPU parity tests of the source-parallel walk (walk_mode SHQ_WALK_GROUP), through the C-ABI.

Same bar as the exact walk (test_gpu_gravity.py): every target's interaction count equals the oracle's per-target
reference walk as an integer (same opening decisions, same interaction set) and the forces agree to rounding —
only the order of a target's sum differs."""
import ctypes as C

import numpy as np
import pytest

import shenqi_amd as sq
from shenqi_amd import capi
import orc
import common as cm

pytestmark = pytest.mark.gpu


def gpu_walk(ctx, pman, tree, gp, oldacc_from, mode, active=None, update_potential=True, upload=True):
    P = pman.Base
    if upload:
        P["FullTreeGravAccel"] = oldacc_from[0]
        P["GravPM"] = oldacc_from[1]
        pv, tv = pman.view(), tree.view()
        capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
        capi.check(capi.hip.shq_tree_upload(ctx.h, C.byref(tv)))
        capi.check(capi.hip.shq_grav_refresh_oldacc(ctx.h, gp.G))
    a = None if active is None else np.ascontiguousarray(active, dtype=np.int32)
    capi.check(capi.hip.shq_grav_short_run(ctx.h, C.byref(gp), capi.ptr(a), 0 if a is None else len(a), int(update_potential), mode))
    n = pman.NumPart
    acc = np.zeros((n, 3)); pot = np.zeros(n); nint = np.zeros(n, dtype=np.int64)
    st = sq.WalkStats()
    capi.check(capi.hip.shq_grav_short_download(ctx.h, capi.ptr(acc), capi.ptr(pot), capi.ptr(nint), C.byref(st)))
    return acc, pot, nint, st


def setup16(kind, usebh, n=16**3, order=True):
    pos = {"grid": lambda: cm.grid_positions(round(n ** (1 / 3))), "close": lambda: cm.close_positions(round(n ** (1 / 3))),
           "random": lambda: cm.random_positions(orc.boost_mt19937_uniform(0, 3 * n), n)}[kind]()
    if order:
        pos = pos[sq.hilbert_order(pos, cm.BOX)]
    pman = cm.make_partmanager(pos)
    tree = sq.force_tree_full(pman)
    cm.reference_treepar(ErrTolForceAcc=0.002, MaxBHOpeningAngle=0.9, TreeUseBH=usebh)
    sq.gravshort_set_softenings(cm.BOX / np.cbrt(n))
    gp = sq.make_grav_params(cm.BOX, 1.5, 48, cm.G, cm.RHO0)
    rng = np.random.default_rng(5)
    told = rng.normal(size=(n, 3)) * 500.0
    pold = rng.normal(size=(n, 3)) * 50.0
    return pos, pman, tree, gp, told, pold


@pytest.mark.parametrize("kind", ["grid", "close", "random"])
@pytest.mark.parametrize("usebh", [1, 0])
@pytest.mark.parametrize("order", [True, False])
def test_group_walk_parity_16(ctx, kind, usebh, order):
    """identical opening decisions => identical interaction counts (integers); forces to rounding.  order=False leaves the
    particles in generation order, so the groups of 8 are spread over the box (sparse masks, wide boxes, per-pair wraps)."""
    pos, pman, tree, gp, told, pold = setup16(kind, usebh, order=order)
    n = len(pos)
    acc, pot, nint, st = gpu_walk(ctx, pman, tree, gp, (told, pold), sq.WALK_GROUP)
    oldacc = np.linalg.norm(told + pold, axis=1) / cm.G
    mass = pman.Base["Mass"]
    oacc, opot, onint = orc.grav_walk(tree.Nodes_base, tree.firstnode, pos, mass, oldacc, gp)
    orc.grav_postprocess(mass, gp, oacc, opot, True)
    assert np.array_equal(nint, onint)
    assert st.ninteractions == onint.sum() and st.min_interactions == onint.min() and st.max_interactions == onint.max()
    scale = np.abs(oacc).max()
    if kind == "grid":
        # on the lattice the force cancels to ~1e-13 of its terms and the two sums add the same terms in a different order:
        # the bound is rounding of nint terms of size G m / spacing^2
        assert np.abs(acc - oacc).max() < nint.max() * 2.3e-16 * cm.G / (cm.BOX / 16) ** 2
    else:
        assert np.abs(acc - oacc).max() < 1e-11 * scale
    if kind != "grid":
        assert cm.force_err(acc, oacc).max() < 1e-5          # runtests.cpp:441-443
    assert np.allclose(pot, opot, rtol=1e-10, atol=1e-10 * np.abs(opot).max())


def test_group_walk_active_ragged_empty_nopot(ctx):
    """ragged target counts (not a multiple of 8 or 64), an unsorted active list, an empty list, a walk without potential;
    rows of inactive particles stay untouched"""
    pos, pman, tree, gp, told, pold = setup16("random", 0)
    n = len(pos)
    oldacc = np.linalg.norm(told + pold, axis=1) / cm.G
    mass = pman.Base["Mass"]
    rng = np.random.default_rng(11)
    acc_all, pot_all, nint_all, _ = gpu_walk(ctx, pman, tree, gp, (told, pold), sq.WALK_GROUP)
    for m in (1, 7, 63, 65, 1001):
        act = rng.choice(n, size=m, replace=False).astype(np.int32)
        if m != 1001:
            act = np.sort(act)
        acc, pot, nint, st = gpu_walk(ctx, pman, tree, gp, (told, pold), sq.WALK_GROUP, active=act, upload=False)
        oacc, opot, onint = orc.grav_walk(tree.Nodes_base, tree.firstnode, pos, mass, oldacc, gp, targets=act)
        orc.grav_postprocess(mass[act], gp, oacc, opot, True)
        assert np.array_equal(nint[act], onint)
        assert st.ntargets == m and st.ninteractions == onint.sum()
        assert np.abs(acc[act] - oacc).max() < 1e-11 * np.abs(oacc).max()
        assert np.allclose(pot[act], opot, rtol=1e-10, atol=1e-10 * np.abs(opot).max())
        rest = np.setdiff1d(np.arange(n), act)
        assert np.array_equal(acc[rest], acc_all[rest])      # untouched rows keep the previous walk's values
        acc_all, pot_all = acc, pot
    capi.check(capi.hip.shq_grav_short_run(ctx.h, C.byref(gp), capi.ptr(np.zeros(1, dtype=np.int32)), 0, 1, sq.WALK_GROUP))
    acc1, _, nint1, _ = gpu_walk(ctx, pman, tree, gp, (told, pold), sq.WALK_GROUP, upload=False)
    acc0, _, nint0, _ = gpu_walk(ctx, pman, tree, gp, (told, pold), sq.WALK_GROUP, update_potential=False, upload=False)
    assert np.array_equal(nint0, nint1)
    assert np.abs(acc0 - acc1).max() <= 1e-14 * np.abs(acc1).max()


def test_group_walk_equals_exact_walk_64_cluster(ctx):
    """S-cluster 64^3, Nmesh 192, ErrTolForceAcc 0.005 (north-star setting): the oracle's counts, and the same forces as the
    exact walk of the same library to rounding; rms against the CPU reference walk far below the north-star 1e-3."""
    n = 64**3
    L = 1.0
    pos = sq.synth_positions("cluster", n, L=L)
    pos = pos[sq.hilbert_order(pos, L)]
    pman = cm.make_partmanager(pos, box=L)
    tree = sq.force_tree_full(pman)
    mass = pman.Base["Mass"]
    z = np.zeros((n, 3))
    sq.set_gravshort_treepar(ErrTolForceAcc=0.005, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=1, Rcut=6.0)
    sq.gravshort_set_softenings(L / 64)
    gp_bh = sq.make_grav_params(L, 1.5, 192, cm.G, cm.RHO0)
    acc1, _, nint1, st1 = gpu_walk(ctx, pman, tree, gp_bh, (z, z), sq.WALK_GROUP)            # Barnes-Hut seeding pass
    o1, op1, on1 = orc.grav_walk(tree.Nodes_base, tree.firstnode, pos, mass, np.zeros(n), gp_bh)
    orc.grav_postprocess(mass, gp_bh, o1, op1, True)
    assert np.array_equal(nint1, on1)
    assert np.abs(acc1 - o1).max() < 1e-11 * np.abs(o1).max()
    sq.set_gravshort_treepar(ErrTolForceAcc=0.005, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=0, Rcut=6.0)
    gp = sq.make_grav_params(L, 1.5, 192, cm.G, cm.RHO0)
    acc2, pot2, nint2, st2 = gpu_walk(ctx, pman, tree, gp, (o1, z), sq.WALK_GROUP)
    o2, op2, on2 = orc.grav_walk(tree.Nodes_base, tree.firstnode, pos, mass, np.linalg.norm(o1, axis=1) / cm.G, gp)
    orc.grav_postprocess(mass, gp, o2, op2, True)
    assert np.array_equal(nint2, on2)
    assert cm.force_err(acc2, o2).max() < 1e-5
    rms = np.sqrt(np.mean(np.sum((acc2 - o2) ** 2, axis=1) / np.sum(o2 ** 2, axis=1)))
    assert rms < 1e-12
    acc3, pot3, nint3, st3 = gpu_walk(ctx, pman, tree, gp, (o1, z), sq.WALK_EXACT, upload=False)
    assert np.array_equal(nint3, nint2)
    assert np.abs(acc3 - acc2).max() < 1e-12 * np.abs(acc3).max()
    assert np.allclose(pot3, pot2, rtol=1e-11, atol=1e-11 * np.abs(pot3).max())
    print("64^3 cluster: interactions/target %.1f, node tests per target %.1f, source-parallel walk %.2f ms, exact walk %.2f ms, rms vs CPU %.1e"
          % (on2.mean(), st2.nnodes_visited / n, st2.kernel_ms, st3.kernel_ms, rms))
