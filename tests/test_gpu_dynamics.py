"""GPU parity tests of the resident drift / kick (csrc/dynamics.hip) against a numpy restatement of
libgadget/drift.cpp:16-99 and libgadget/timestep.cpp:838-872, 937-968 (plain IEEE multiplies and adds in
the reference's order: results must agree to the bit)."""
import ctypes as C

import numpy as np
import pytest

import shenqi_amd as sq
from shenqi_amd import capi
import common as cm

pytestmark = pytest.mark.gpu


def ref_drift(P, ddrift, box, shift):
    """real_drift_particle, drift.cpp:16-86 (no black-hole repositioning)."""
    pos, vel = P["Pos"].copy(), P["Vel"]
    hsml = P["Hsml"].copy()
    dead = (P["Flags"] & 3) != 0
    gas = (P["Type"] == 0) & ~dead
    hsml[gas] = hsml[gas] + P["DtHsml"][gas] * ddrift
    hsml[gas] = np.minimum(hsml[gas], box / 2.0)
    step = vel * ddrift + shift
    step[dead] = shift
    pos = pos + step
    for _ in range(8):
        pos = np.where(pos > box, pos - box, pos)
        pos = np.where(pos <= 0, pos + box, pos)
    return pos, hsml


def _setup(n=20000, seed=4):
    rng = np.random.default_rng(seed)
    pos = rng.random((n, 3)) * cm.BOX
    pman = cm.make_partmanager(pos)
    P = pman.Base
    P["Type"] = rng.choice([0, 1, 4, 5], size=n).astype(np.uint8)
    P["Vel"] = rng.normal(size=(n, 3)) * 300.0
    P["Hsml"] = 0.01 * cm.BOX * (1 + rng.random(n))
    P["DtHsml"] = rng.normal(size=n) * 1e-4 * cm.BOX
    P["TimeBinGravity"] = rng.integers(20, 30, size=n).astype(np.uint8)
    fl = np.zeros(n, dtype=np.uint8)
    fl[rng.random(n) < 0.03] |= 1
    fl[rng.random(n) < 0.03] |= 2
    P["Flags"] = fl
    P["FullTreeGravAccel"] = rng.normal(size=(n, 3)) * 1e3
    P["GravPM"] = rng.normal(size=(n, 3)) * 1e2
    return pman, rng


def test_drift_bit_exact(ctx):
    pman, rng = _setup()
    P = pman.Base
    pv = pman.view()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    sq.dynamics_upload(ctx, pman)
    ddrift, shift = 3.7e-4, np.array([0.31, -0.27, 0.05]) * cm.BOX  # fast particles cross the box edge
    rpos, rhsml = ref_drift(P, ddrift, cm.BOX, shift)
    sq.drift(ctx, ddrift, cm.BOX, shift)
    sq.dynamics_download(ctx, pman)
    assert np.array_equal(P["Pos"], rpos)
    assert np.array_equal(P["Hsml"], rhsml)
    assert P["Pos"].min() > 0 and P["Pos"].max() <= cm.BOX
    # a drift makes the tree stale: a walk without a rebuild must be refused
    cm.reference_treepar()
    sq.gravshort_set_softenings(cm.BOX / 27.0)
    gp = sq.make_grav_params(cm.BOX, 1.5, 48, cm.G, cm.RHO0)
    assert capi.hip.shq_grav_short_run(ctx.h, C.byref(gp), None, 0, 1, sq.WALK_EXACT) != 0


def test_drift_rejects_bad_hsml(ctx):
    pman, _ = _setup(n=1000)
    P = pman.Base
    P["Type"] = 0
    P["Flags"] = 0
    P["DtHsml"] = -1e6
    pv = pman.view()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    sq.dynamics_upload(ctx, pman)
    with pytest.raises(sq.ShqError):
        sq.drift(ctx, 1.0, cm.BOX)


def test_kicks_bit_exact(ctx):
    pman, rng = _setup()
    P = pman.Base
    n = len(P)
    pv = pman.view()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    sq.dynamics_upload(ctx, pman)
    gravkick = np.zeros(capi.TIMEBINS + 1)
    gravkick[22:28] = rng.random(6) * 1e-3  # bins outside are inactive: factor 0
    active = np.sort(rng.choice(n, size=n // 3, replace=False)).astype(np.int32)
    alive = (P["Flags"] & 3) == 0
    vel = P["Vel"].copy()
    sel = active[alive[active]]
    vel[sel] = vel[sel] + P["FullTreeGravAccel"][sel] * gravkick[P["TimeBinGravity"][sel]][:, None]
    sq.kick_short(ctx, gravkick, active)
    Fpm = 2.5e-4
    vel[alive] = vel[alive] + P["GravPM"][alive] * Fpm
    sq.kick_pm(ctx, Fpm)
    sq.dynamics_download(ctx, pman)
    assert np.array_equal(P["Vel"], vel)


def test_resident_step_equals_reupload(ctx):
    """drift -> device tree build -> walk on resident data gives the forces of a fresh upload of the
    drifted particles with a host-built tree, bit for bit."""
    n = 24**3
    L = cm.BOX
    pos = sq.synth_positions("cluster", n, L=L)
    pman = cm.make_partmanager(pos)
    P = pman.Base
    P["Vel"] = np.random.default_rng(9).normal(size=(n, 3)) * 50.0
    cm.reference_treepar(ErrTolForceAcc=0.005, MaxBHOpeningAngle=0.9, Rcut=6.0, TreeUseBH=1)
    sq.gravshort_set_softenings(L / 24)
    gp = sq.make_grav_params(L, 1.5, 72, cm.G, cm.RHO0)
    pv = pman.view()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    sq.dynamics_upload(ctx, pman)
    sq.drift(ctx, 2e-3, L)
    sq.tree_build_device(ctx, L)
    capi.check(capi.hip.shq_grav_short_run(ctx.h, C.byref(gp), None, 0, 1, sq.WALK_EXACT))
    acc = np.zeros((n, 3)); nint = np.zeros(n, dtype=np.int64)
    capi.check(capi.hip.shq_grav_short_download(ctx.h, capi.ptr(acc), None, capi.ptr(nint), None))
    sq.dynamics_download(ctx, pman)       # host copy now holds the drifted positions
    host = sq.force_tree_full(pman)
    pv, tv = pman.view(), host.view()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    capi.check(capi.hip.shq_tree_upload(ctx.h, C.byref(tv)))
    capi.check(capi.hip.shq_grav_short_run(ctx.h, C.byref(gp), None, 0, 1, sq.WALK_EXACT))
    acc2 = np.zeros((n, 3)); nint2 = np.zeros(n, dtype=np.int64)
    capi.check(capi.hip.shq_grav_short_download(ctx.h, capi.ptr(acc2), None, capi.ptr(nint2), None))
    assert np.array_equal(nint, nint2) and np.array_equal(acc, acc2)


def test_tree_order_targets_same_forces(ctx):
    """SHQ_WALK_TREE_ORDER only changes which wave handles which target: per-particle results are
    bit-identical to the index-order walk."""
    n = 24**3
    L = cm.BOX
    pos = sq.synth_positions("cluster", n, L=L)     # deliberately NOT sorted along a space-filling curve
    pman = cm.make_partmanager(pos)
    pman.Base["Flags"][::97] = 1                     # some garbage particles: not in the tree, not targets
    cm.reference_treepar(ErrTolForceAcc=0.005, MaxBHOpeningAngle=0.9, Rcut=6.0, TreeUseBH=1)
    sq.gravshort_set_softenings(L / 24)
    gp = sq.make_grav_params(L, 1.5, 72, cm.G, cm.RHO0)
    pv = pman.view()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    sq.tree_build_device(ctx, L)
    out = []
    for mode in (sq.WALK_EXACT, sq.WALK_EXACT | sq.WALK_TREE_ORDER):
        capi.check(capi.hip.shq_grav_short_run(ctx.h, C.byref(gp), None, 0, 1, mode))
        acc = np.zeros((n, 3)); pot = np.zeros(n); nint = np.zeros(n, dtype=np.int64)
        st = sq.WalkStats()
        capi.check(capi.hip.shq_grav_short_download(ctx.h, capi.ptr(acc), capi.ptr(pot), capi.ptr(nint), C.byref(st)))
        out.append((acc, pot, nint, st.ntargets))
    alive = (pman.Base["Flags"] & 3) == 0
    assert out[1][3] == alive.sum()
    for k in range(3):
        assert np.array_equal(out[0][k][alive], out[1][k][alive])


def test_dynamics_empty(ctx):
    """No particles: every resident call is a no-op that succeeds."""
    pman = cm.make_partmanager(np.zeros((0, 3)))
    pv = pman.view()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    sq.dynamics_upload(ctx, pman)
    sq.drift(ctx, 0.1, cm.BOX, [0.1, 0.2, 0.3])
    sq.kick_pm(ctx, 1.0)
    sq.kick_short(ctx, np.zeros(capi.TIMEBINS + 1))
    sq.dynamics_download(ctx, pman)
    st = sq.tree_build_device(ctx, cm.BOX)
    assert st.nparticles == 0


def ref_timebin_active(bins, Ti):
    """is_timebin_active, timestep.cpp:132-139."""
    bins = bins.astype(np.int64)
    return (bins <= 0) | (Ti <= 0) | (Ti % (np.int64(1) << np.maximum(bins, 0)) == 0)


def ref_active(P, Ti, pm_step):
    """build_active_particles, timestep.cpp:1286-1349: (list or None, NumActiveGravity, NumActiveHydro, TimeBinCountType)."""
    n = len(P)
    dead = (P["Flags"] & 3) != 0
    hydro = (P["Type"] == 0) | (P["Type"] == 5)
    ga = ref_timebin_active(P["TimeBinGravity"], Ti)
    counts = np.zeros((6, capi.TIMEBINS + 1), dtype=np.int64)
    bins = np.where(hydro, P["TimeBinHydro"], P["TimeBinGravity"])
    if pm_step:
        keep = ~dead
        np.add.at(counts, (P["Type"][keep], bins[keep]), 1)
        return None, n, int(hydro.sum()), counts.ravel()
    on = ~dead & (ga | (hydro & ref_timebin_active(P["TimeBinHydro"], Ti)))
    act = np.nonzero(on)[0].astype(np.int32)
    np.add.at(counts, (P["Type"][act], bins[act]), 1)
    return act, int(ga[act].sum()), int(hydro[act].sum()), counts.ravel()


@pytest.mark.parametrize("Ti", [0, 1 << 22, 3 << 20, (1 << 26) + (1 << 21), 12345])
def test_active_lists_equal_reference(ctx, Ti):
    """ActivePredicate / SubActivePredicate (timestep.cpp:1265-1282, 1354-1371) on the device: same lists in the
    same (index) order, same tallies."""
    pman, rng = _setup(n=50001)
    P = pman.Base
    P["TimeBinGravity"] = rng.integers(18, 28, size=len(P)).astype(np.uint8)
    P["TimeBinHydro"] = rng.integers(16, 26, size=len(P)).astype(np.uint8)
    P["TimeBinGravity"][:7] = 0    # bin 0 is always active
    pv = pman.view()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    sq.dynamics_upload(ctx, pman)
    for pm_step in (False, True):
        ract, rgrav, rhydro, rcounts = ref_active(P, Ti, pm_step)
        info = sq.build_active_particles(ctx, Ti, pm_step)
        got = sq.active_download(ctx)
        want = np.arange(len(P), dtype=np.int32) if ract is None else ract
        assert info.NumActiveParticle == len(want) and np.array_equal(got, want)
        assert info.NumActiveGravity == rgrav and info.NumActiveHydro == rhydro
        assert np.array_equal(np.array(info.TimeBinCountType[:]), rcounts)
        for maxbin in (19, 22, 46):
            dead = (P["Flags"][want] & 3) != 0
            bg = P["TimeBinGravity"][want]
            rsub = want[~dead & (bg <= maxbin) & ref_timebin_active(bg, Ti)]
            assert sq.build_active_sublist(ctx, maxbin, Ti) == len(rsub)
            assert np.array_equal(sq.active_download(ctx, sublist=True), rsub)


def test_resident_active_list_drives_tree_walk_and_kick(ctx):
    """SHQ_ACTIVE_RESIDENT / SHQ_SUBLIST_RESIDENT as the `active` argument give the same tree, forces and
    velocities as the same list passed from the host."""
    n = 24**3
    L = cm.BOX
    pos = sq.synth_positions("cluster", n, L=L)
    pman = cm.make_partmanager(pos)
    rng = np.random.default_rng(5)
    P = pman.Base
    P["TimeBinGravity"] = rng.integers(20, 24, size=n).astype(np.uint8)
    P["Vel"] = rng.normal(size=(n, 3))
    Ti = 1 << 22
    cm.reference_treepar(ErrTolForceAcc=0.005, MaxBHOpeningAngle=0.9, Rcut=6.0, TreeUseBH=1)
    sq.gravshort_set_softenings(L / 24 / 30)
    gp = sq.make_grav_params(L, 1.5, 72, cm.G, cm.RHO0)
    gk = np.zeros(capi.TIMEBINS + 1)
    gk[20:24] = [1e-3, 2e-3, 4e-3, 8e-3]
    pv = pman.view()
    out = []
    vel0 = P["Vel"].copy()
    for resident in (False, True):
        P["Vel"] = vel0
        capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
        sq.dynamics_upload(ctx, pman)
        sq.build_active_particles(ctx, Ti)
        nsub = sq.build_active_sublist(ctx, 21, Ti)
        sub = sq.active_download(ctx, sublist=True)
        assert 0 < nsub < sq.active_download(ctx).size < n
        if resident:
            act, nact = capi.SUBLIST_RESIDENT, 0
            sq.tree_build_device(ctx, L, sq.ALLMASK, sq.RESIDENT)
        else:
            act, nact = capi.ptr(sub), len(sub)
            sq.tree_build_device(ctx, L, sq.ALLMASK, sq.active_download(ctx))
        capi.check(capi.hip.shq_grav_short_run(ctx.h, C.byref(gp), act, nact, 1, sq.WALK_EXACT))
        acc = np.zeros((n, 3)); pot = np.zeros(n); nint = np.zeros(n, dtype=np.int64)
        capi.check(capi.hip.shq_grav_short_download(ctx.h, capi.ptr(acc), capi.ptr(pot), capi.ptr(nint), None))
        sq.kick_short(ctx, gk, sq.RESIDENT_SUB if resident else sub)
        sq.dynamics_download(ctx, pman)
        out.append((acc[sub].copy(), pot[sub].copy(), nint[sub].copy(), P["Vel"].copy()))
    assert np.abs(out[0][0]).max() > 0 and not np.array_equal(out[0][3], vel0)
    for x, y in zip(out[0], out[1]):
        assert np.array_equal(x, y)


def test_resident_active_list_requires_build(ctx):
    pman, _ = _setup(n=100)
    pv = pman.view()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    sq.dynamics_upload(ctx, pman)
    with pytest.raises(sq.ShqError):
        sq.tree_build_device(ctx, cm.BOX, sq.ALLMASK, sq.RESIDENT)
    with pytest.raises(sq.ShqError):
        sq.active_download(ctx)
