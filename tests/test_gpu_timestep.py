"""GPU parity tests of the integer time line on the device (csrc/timestep.hip) and of its host mirror
(integration/reference_side/timestep.cpp: find_timesteps, find_hydro_timesteps, hierarchical_gravity_and_timesteps) against the
restatement of libgadget/timestep.cpp in oracle/timeline.py.  Time bins and tallies are integers: they must be equal."""
import ctypes as C
import math
import os
import sys

import numpy as np
import pytest

import shenqi_amd as sq
from shenqi_amd import capi
import common as cm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import timeline as tl  # noqa: E402

pytestmark = pytest.mark.gpu

OUTS = [0.1, 0.2, 0.8, 1.0]
TB = tl.TIMEBASE
TP = dict(ErrTolIntAccuracy=0.02, ForceEqualTimesteps=0, MinSizeTimestep=1e-9, MaxSizeTimestep=0.1, MaxRMSDisplacementFac=0.2, MaxGasVel=3e5,
          CourantFac=0.15)
COSMO = dict(OmegaBaryon=0.045, OmegaCDM=0.255, OmegaNu1=0.001, RhoCrit=27.7455, Omega0=0.3, Hubble=0.1, GravInternal=43.0071)


class Parts:
    """the fields oracle/timeline.py reads, as views into the PARTICLE_DTYPE array"""

    def __init__(self, P):
        self.P = P
        self.f = {"IsGarbage": (P["Flags"] & 1).astype(bool), "Swallowed": ((P["Flags"] >> 1) & 1).astype(bool)}

    def __len__(self):
        return len(self.P)

    def __getitem__(self, k):
        return self.f[k] if k in self.f else self.P[k]


def _setup(n=20000, seed=8, nbh=40):
    rng = np.random.default_rng(seed)
    pos = rng.random((n, 3)) * cm.BOX
    pman = cm.make_partmanager(pos)
    P = pman.Base
    P["Type"] = rng.choice([0, 1, 4], size=n, p=[0.4, 0.5, 0.1]).astype(np.uint8)
    bh = np.sort(rng.choice(n, size=nbh, replace=False))
    P["Type"][bh] = 5
    P["PI"][bh] = rng.permutation(nbh)          # slots in another order than the particles
    P["Mass"] = rng.choice([1.0, 0.2, 0.05], size=n)
    P["Vel"] = rng.normal(size=(n, 3)) * 300.0
    P["Hsml"] = 0.01 * cm.BOX * (1 + rng.random(n))
    P["DtHsml"] = rng.normal(size=n) * 1e-2 * cm.BOX
    fast = rng.random(n) < 0.2                      # collapsing / expanding fast: the Hsml criterion wins over the Courant one
    P["DtHsml"][fast] = rng.normal(size=fast.sum()) * 1e5 * cm.BOX
    fl = np.zeros(n, dtype=np.uint8)
    fl[rng.random(n) < 0.03] |= 1
    fl[rng.random(n) < 0.03] |= 2
    P["Flags"] = fl
    # accelerations over five decades: the bins spread over ~8 levels
    P["FullTreeGravAccel"] = rng.normal(size=(n, 3)) * (10 ** rng.uniform(0, 5, size=n))[:, None]
    P["GravPM"] = rng.normal(size=(n, 3)) * 30.0
    zero = rng.choice(n, size=5, replace=False)     # ac2 == 0 branch
    P["FullTreeGravAccel"][zero] = 0
    P["GravPM"][zero] = 0
    msv = 50.0 * (10 ** rng.uniform(0, 3, size=n))
    BhP = np.zeros(nbh, dtype=capi.BH_DTYPE)
    BhP["minTimeBin"] = rng.integers(0, 40, size=nbh)
    BhP["minTimeBin"][:3] = [0, 45, 46]           # limiter off / just outside its range
    BhP["TimeBinDynFric"] = rng.integers(0, 30, size=nbh)
    BhP["DF_SurroundingVel"] = rng.normal(size=(nbh, 3)) * 200.0
    BhP["DFAccel"] = rng.normal(size=(nbh, 3)) * 50.0
    BhP["DragAccel"] = rng.normal(size=(nbh, 3)) * 20.0
    return pman, msv, BhP, bh, rng


def _upload(ctx, pman, msv, BhP):
    pv = pman.view()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    sq.dynamics_upload(ctx, pman)
    capi.check(capi.hip.shq_maxsignalvel_upload(ctx.h, capi.ptr(np.ascontiguousarray(msv))))
    bv = capi.bh_dyn_view(BhP)
    capi.check(capi.hip.shq_bh_dynamics_upload(ctx.h, C.byref(pv), C.byref(bv)))
    return pv, bv


def _params(tbm_host, Ti, atime, hubble, dti_max, soft, first=0, equal=0, mintimebin=0, mingrav=0):
    p = capi.TimestepParams()
    p.ErrTolIntAccuracy, p.CourantFac, p.MinSizeTimestep = TP["ErrTolIntAccuracy"], TP["CourantFac"], TP["MinSizeTimestep"]
    p.ForceSoftening, p.atime, p.hubble = soft, atime, hubble
    p.fac3 = math.pow(atime, 3 * (1 - tl.GAMMA) / 2.0)
    p.dti_max, p.ForceEqualTimesteps, p.isFirstTimeStep, p.mintimebin, p.mingravtimebin = dti_max, equal, first, mintimebin, mingrav
    capi.host.shqh_tbm_timeline_at(tbm_host, Ti, C.byref(p.tl))
    return p


@pytest.fixture(scope="module")
def tbm_host():
    la = (C.c_double * 4)(*[math.log(a) for a in OUTS])
    h = capi.host.shqh_timebinmgr_create(la, 4)
    yield h
    capi.host.shqh_timebinmgr_destroy(h)


def _bins(ctx, n):
    bg, bh = np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.uint8)
    capi.check(capi.hip.shq_timebins_download(ctx.h, capi.ptr(bg), capi.ptr(bh)))
    return bg, bh


def _bh_maps(P, BhP, bhidx):
    mint = {int(i): int(BhP["minTimeBin"][P["PI"][i]]) for i in bhidx}
    dyn = {int(i): int(BhP["TimeBinDynFric"][P["PI"][i]]) for i in bhidx}
    surr = {int(i): BhP["DF_SurroundingVel"][P["PI"][i]].copy() for i in bhidx}
    return mint, dyn, surr


# Ti inside the second segment (non-zero: only some bins are active), just before a sync point (steps cross into the next,
# narrower segment), and 0 (every bin active)
@pytest.mark.parametrize("Ti", [TB + (3 << 36), 2 * TB - (1 << 30), 0])
@pytest.mark.parametrize("listed", [False, True])
def test_find_timesteps_loop(ctx, tbm_host, Ti, listed):
    pman, msv, BhP, bhidx, rng = _setup()
    P = pman.Base
    n = len(P)
    # old bins: active ones mostly (as an active list has), a few inactive
    P["TimeBinHydro"] = rng.integers(28, 38, size=n).astype(np.uint8)
    P["TimeBinGravity"] = P["TimeBinHydro"]
    _upload(ctx, pman, msv, BhP)
    t = tl.TimeBinMgr(OUTS)
    atime = math.exp(t.loga_from_ti(Ti))
    hubble, soft, dti_max = 0.1 * atime ** -1.5, cm.BOX / 30.0, 1 << 40
    act = np.sort(rng.choice(n, size=n // 2, replace=False)).astype(np.int32) if listed else None
    mint, dyn, surr = _bh_maps(P, BhP, bhidx)
    want = tl.find_timesteps_loop(Parts(P), act, msv, mint, Ti, t, dti_max, atime, hubble, TP, soft)
    p = _params(tbm_host, Ti, atime, hubble, dti_max, soft)
    r = capi.TimestepResult()
    capi.check(capi.hip.shq_find_timesteps(ctx.h, C.byref(p), capi.ptr(act), 0 if act is None else len(act), 0, -1, C.byref(r)))
    bg, bh = _bins(ctx, n)
    assert np.array_equal(bh, P["TimeBinHydro"]) and np.array_equal(bg, P["TimeBinGravity"])
    assert (r.badstepsizecount, r.mTimeBin, r.maxTimeBin) == (want["badstepsizecount"], want["mTimeBin"], want["maxTimeBin"])
    assert [r.ntiaccel, r.nticourant, r.ntiaccrete, r.ntineighbour, r.ntihsml] == want["counts"]
    assert len(np.unique(bh)) >= 6 and want["counts"][tl.TI_COURANT] > 0 and want["counts"][tl.TI_HSML] > 0 and want["counts"][tl.TI_NEIGH] > 0


def test_find_timesteps_equal_steps_and_first_step(ctx, tbm_host):
    """ForceEqualTimesteps (find_global_timestep, then one step for all) and set_bh_first_timestep"""
    pman, msv, BhP, bhidx, rng = _setup(seed=9)
    P = pman.Base
    n = len(P)
    _upload(ctx, pman, msv, BhP)
    t = tl.TimeBinMgr(OUTS)
    Ti = 0
    atime = OUTS[0]
    hubble, soft, dti_max = 0.1 * atime ** -1.5, cm.BOX / 30.0, 1 << 42
    mint, dyn, surr = _bh_maps(P, BhP, bhidx)
    tp = dict(TP, ForceEqualTimesteps=1)
    dmin = tl.find_global_timestep(Parts(P), msv, mint, Ti, t, dti_max, atime, hubble, tp, soft)
    want = tl.find_timesteps_loop(Parts(P), None, msv, mint, Ti, t, dti_max, atime, hubble, tp, soft, dti_min_global=dmin)
    P["TimeBinHydro"][P["Type"] == 5] = want["mTimeBin"]      # set_bh_first_timestep, timestep.cpp:567-578
    p = _params(tbm_host, Ti, atime, hubble, dti_max, soft, first=1, equal=1)
    r = capi.TimestepResult()
    capi.check(capi.hip.shq_find_global_timestep(ctx.h, C.byref(p), C.byref(r)))
    assert r.dti_min == dmin and 1 < dmin < dti_max
    capi.check(capi.hip.shq_find_timesteps(ctx.h, C.byref(p), None, 0, r.dti_min, -1, C.byref(r)))
    bg, bh = _bins(ctx, n)
    assert np.array_equal(bh, P["TimeBinHydro"]) and np.array_equal(bg, P["TimeBinGravity"])
    assert r.mTimeBin == r.maxTimeBin == want["mTimeBin"] == tl.get_timestep_bin(tl.round_down_power_of_two(dmin))


@pytest.mark.parametrize("Ti", [TB + (3 << 36), 0])
def test_find_hydro_timesteps_loop(ctx, tbm_host, Ti):
    pman, msv, BhP, bhidx, rng = _setup(seed=10)
    P = pman.Base
    n = len(P)
    P["TimeBinGravity"] = rng.integers(30, 38, size=n).astype(np.uint8)
    P["TimeBinHydro"] = np.minimum(P["TimeBinGravity"], rng.integers(28, 38, size=n)).astype(np.uint8)
    pv, bv = _upload(ctx, pman, msv, BhP)
    t = tl.TimeBinMgr(OUTS)
    atime = math.exp(t.loga_from_ti(Ti))
    hubble, soft, dti_max = 0.1 * atime ** -1.5, cm.BOX / 30.0, 1 << 40
    act = np.sort(rng.choice(n, size=(2 * n) // 3, replace=False)).astype(np.int32)
    mint, dyn, surr = _bh_maps(P, BhP, bhidx)
    want = tl.find_hydro_timesteps_loop(Parts(P), act, msv, mint, dyn, surr, Ti, t, dti_max, atime, hubble, TP)
    mfix, mintb = tl.hydro_mintimebin_fixups(want["mTimeBin"], 29, 31, Ti)
    p = _params(tbm_host, Ti, atime, hubble, dti_max, soft, mintimebin=29, mingrav=31)
    r = capi.TimestepResult()
    capi.check(capi.hip.shq_find_hydro_timesteps(ctx.h, C.byref(p), capi.ptr(act), len(act), C.byref(r)))
    bg, bh = _bins(ctx, n)
    assert np.array_equal(bh, P["TimeBinHydro"]) and np.array_equal(bg, P["TimeBinGravity"])
    assert (r.badstepsizecount, r.mTimeBin, r.mintimebin) == (want["badstepsizecount"], mfix, mintb)
    assert [r.ntiaccel, r.nticourant, r.ntiaccrete, r.ntineighbour, r.ntihsml] == want["counts"]
    assert (r.nbh, r.dynratio, r.maxdyndiff) == (want["nbh"], want["dynratio"], want["maxdyndiff"]) and r.nbh > 0
    capi.check(capi.hip.shq_bh_dynamics_download(ctx.h, C.byref(pv), C.byref(bv)))
    for i in bhidx:
        assert BhP["TimeBinDynFric"][P["PI"][i]] == dyn[int(i)]


def test_gas_without_state_is_refused(ctx, tbm_host):
    """a gas particle on the list and no MaxSignalVel on the device: an error, not a silent default"""
    pman, msv, BhP, bhidx, rng = _setup(n=2000, nbh=4)
    pv = pman.view()
    c2 = sq.Context(0)
    try:
        capi.check(capi.hip.shq_particles_upload(c2.h, C.byref(pv)))
        sq.dynamics_upload(c2, pman)
        p = _params(tbm_host, 0, 0.1, 3.0, 1 << 40, cm.BOX / 30.0)
        r = capi.TimestepResult()
        assert capi.hip.shq_find_timesteps(c2.h, C.byref(p), None, 0, 0, -1, C.byref(r)) != 0
        assert b"MaxSignalVel" in capi.hip.shq_last_error()
    finally:
        c2.close()


def test_hierarchical_loops_and_moments(ctx, tbm_host):
    """the three particle loops of hierarchical_gravity_and_timesteps and the long-range moments"""
    pman, msv, BhP, bhidx, rng = _setup(seed=12)
    P = pman.Base
    n = len(P)
    P["TimeBinGravity"] = rng.integers(30, 38, size=n).astype(np.uint8)
    P["TimeBinHydro"] = P["TimeBinGravity"]
    _upload(ctx, pman, msv, BhP)
    t = tl.TimeBinMgr(OUTS)
    Ti = TB + (1 << 38)
    atime = math.exp(t.loga_from_ti(Ti))
    hubble, soft, dti_max = 0.1 * atime ** -1.5, cm.BOX / 30.0, 1 << 38
    p = _params(tbm_host, Ti, atime, hubble, dti_max, soft)
    r = capi.TimestepResult()
    act = np.sort(rng.choice(n, size=n // 2, replace=False)).astype(np.int32)
    counts = tl.hier_gravity_bins(Parts(P), act, P["FullTreeGravAccel"], Ti, t, dti_max, atime, hubble, TP, soft, 36)
    capi.check(capi.hip.shq_hier_gravity_bins(ctx.h, C.byref(p), capi.ptr(act), len(act), 0, 36, C.byref(r)))
    assert list(r.timebincounts) == counts and sum(counts) > 0 and counts[36] > 0
    assert np.array_equal(_bins(ctx, n)[0], P["TimeBinGravity"])
    P["TimeBinGravity"][act] = np.minimum(P["TimeBinGravity"][act], 33)        # timestep.cpp:407-414
    capi.check(capi.hip.shq_hier_push_down(ctx.h, capi.ptr(act), len(act), 33))
    assert np.array_equal(_bins(ctx, n)[0], P["TimeBinGravity"])
    for ti in (33, 1):
        sub = np.ascontiguousarray(act[::3])
        bad = tl.hier_refine(Parts(P), sub, P["FullTreeGravAccel"], Ti, t, dti_max, atime, hubble, TP, soft, ti)
        capi.check(capi.hip.shq_hier_refine(ctx.h, C.byref(p), capi.ptr(sub), len(sub), 0, ti, C.byref(r)))
        assert r.badstepsizecount == bad
        assert np.array_equal(_bins(ctx, n)[0], P["TimeBinGravity"])
    v, mim, cnt = tl.long_range_moments(Parts(P))
    gv, gm, gc = np.zeros(6), np.zeros(6), np.zeros(6, dtype=np.int64)
    capi.check(capi.hip.shq_velocity_moments(ctx.h, capi.ptr(gv), capi.ptr(gm), capi.ptr(gc)))
    assert np.array_equal(gc, cnt) and np.array_equal(gm, mim) and np.array_equal(gv, v)


def test_bh_kick_and_reposition(ctx):
    """black-hole half of do_hydro_kick (timestep.cpp:973-979) and the jump to the potential minimum (drift.cpp:32-53)"""
    pman, msv, BhP, bhidx, rng = _setup(seed=13)
    P = pman.Base
    n = len(P)
    P["TimeBinHydro"] = rng.integers(20, 30, size=n).astype(np.uint8)
    P["DtHsml"] = 0
    P["Flags"][bhidx[:2]] = [1, 2]                       # a garbage and a swallowed BH
    BhP["JumpToMinPot"] = rng.integers(0, 2, size=len(BhP))
    BhP["MinPotPos"] = P["Pos"][bhidx][np.argsort(P["PI"][bhidx])] + rng.normal(size=(len(BhP), 3)) * 0.01 * cm.BOX
    BhP["MinPotPos"] = np.mod(BhP["MinPotPos"], cm.BOX)
    BhP["MinPotVel"] = rng.normal(size=(len(BhP), 3)) * 100.0
    pv, bv = _upload(ctx, pman, msv, BhP)
    gravkick = np.zeros(capi.TIMEBINS + 1)
    gravkick[20:30] = rng.random(10) * 1e-3
    act = np.sort(np.concatenate([bhidx[::2], rng.choice(n, size=500, replace=False)])).astype(np.int32)
    act = np.unique(act)
    vel = P["Vel"].copy()
    for i in act:
        if P["Type"][i] == 5 and not (P["Flags"][i] & 3):
            F = gravkick[P["TimeBinHydro"][i]]
            s = P["PI"][i]
            vel[i] = vel[i] + BhP["DFAccel"][s] * F
            vel[i] = vel[i] + BhP["DragAccel"][s] * F
    capi.check(capi.hip.shq_kick_bh(ctx.h, capi.ptr(gravkick), capi.ptr(act), len(act)))
    sq.dynamics_download(ctx, pman)
    assert np.array_equal(P["Vel"], vel)
    # drift with repositioning
    ddrift, shift = 2.0e-4, np.array([0.01, -0.02, 0.03]) * cm.BOX
    pos, velr = P["Pos"].copy(), P["Vel"].copy()
    jumped = 0
    for i in bhidx:
        if P["Flags"][i] & 3:
            continue
        s = P["PI"][i]
        if BhP["JumpToMinPot"][s]:
            pos[i] = BhP["MinPotPos"][s]
            velr[i] = BhP["MinPotVel"][s]
            jumped += 1
    dead = (P["Flags"] & 3) != 0
    step = velr * ddrift + shift
    step[dead] = shift
    pos = pos + step
    for _ in range(4):
        pos = np.where(pos > cm.BOX, pos - cm.BOX, pos)
        pos = np.where(pos <= 0, pos + cm.BOX, pos)
    capi.check(capi.hip.shq_set_bh_reposition(ctx.h, 1))
    try:
        sq.drift(ctx, ddrift, cm.BOX, shift)
    finally:
        capi.check(capi.hip.shq_set_bh_reposition(ctx.h, 0))
    sq.dynamics_download(ctx, pman)
    assert jumped > 5
    assert np.array_equal(P["Pos"], pos) and np.array_equal(P["Vel"], velr)
    capi.check(capi.hip.shq_bh_dynamics_download(ctx.h, C.byref(pv), C.byref(bv)))
    live_slots = P["PI"][[i for i in bhidx if not (P["Flags"][i] & 3)]]
    assert not BhP["JumpToMinPot"][live_slots].any()        # cleared for every live BH (drift.cpp:52)
    # a jump further than 0.1 BoxSize ends the run in the reference: an error here (the reference tests the signed
    # Pos - MinPotPos only, drift.cpp:38-40, and so does the kernel)
    BhP["JumpToMinPot"] = 1
    BhP["MinPotPos"] = np.mod(BhP["MinPotPos"] - 0.3 * cm.BOX, cm.BOX)
    capi.check(capi.hip.shq_bh_dynamics_upload(ctx.h, C.byref(pv), C.byref(bv)))
    capi.check(capi.hip.shq_set_bh_reposition(ctx.h, 1))
    try:
        with pytest.raises(sq.ShqError):
            sq.drift(ctx, ddrift, cm.BOX, shift)
    finally:
        capi.check(capi.hip.shq_set_bh_reposition(ctx.h, 0))


def _cosmo(hubble_now):
    c = capi.HostCosmo()
    c.OmegaBaryon, c.OmegaCDM, c.OmegaNu1, c.RhoCrit = COSMO["OmegaBaryon"], COSMO["OmegaCDM"], COSMO["OmegaNu1"], COSMO["RhoCrit"]
    c.Omega0, c.Hubble, c.GravInternal, c.hubble_now = COSMO["Omega0"], COSMO["Hubble"], COSMO["GravInternal"], hubble_now
    return c


def test_host_find_timesteps_pm_step(ctx, tbm_host):
    """find_timesteps of the host mirror on a PM step: long-range criterion from the device moments, PM_length / PM_start,
    the particle loop, PM_length capped by the longest occupied bin (timestep.cpp:705-822)"""
    pman, msv, BhP, bhidx, rng = _setup(seed=14)
    P = pman.Base
    n = len(P)
    P["Vel"] *= 3.0
    _upload(ctx, pman, msv, BhP)
    t = tl.TimeBinMgr(OUTS)
    Ti = TB + (1 << 40)
    atime = math.exp(t.loga_from_ti(Ti))
    hubble, soft, asmth = 0.1 * atime ** -1.5, cm.BOX / 30.0, 1.5 * cm.BOX / 48
    cm.reference_treepar()
    sq.gravshort_set_softenings(cm.BOX)
    soft = capi.host.shqh_FORCE_SOFTENING()                 # what the host mirror passes as ForceSoftening
    capi.host.shqh_set_timestep_params(TP["ErrTolIntAccuracy"], 0, TP["MinSizeTimestep"], TP["MaxSizeTimestep"], TP["MaxRMSDisplacementFac"],
                                       TP["MaxGasVel"], TP["CourantFac"])
    times = capi.DriftKickTimes()
    times.Ti_Current, times.PM_start, times.PM_length, times.PM_kick = Ti, Ti - (1 << 38), 1 << 38, Ti
    # oracle
    v, mim, cnt = tl.long_range_moments(Parts(P))
    dloga = tl.long_range_dloga(v, mim, cnt, atime, hubble, COSMO, TP, 2, asmth)
    dti = tl.round_down_power_of_two(t.dti_from_dloga(dloga, Ti))
    dti_max = min(dti, t.find_next_ti_sync(Ti) - Ti)
    mint, dyn, surr = _bh_maps(P, BhP, bhidx)
    want = tl.find_timesteps_loop(Parts(P), None, msv, mint, Ti, t, dti_max, atime, hubble, TP, soft)
    pm_length = min(dti_max, tl.dti_from_timebin(want["maxTimeBin"]))
    bad = C.c_int(-1)
    cosmo = _cosmo(hubble)
    rc = capi.host.shqh_find_timesteps(ctx.h, 0, n, C.byref(times), tbm_host, atime, 2, C.byref(cosmo), asmth, 0, C.byref(bad))
    assert rc == 0, capi.host.shqh_last_error()
    bg, bh = _bins(ctx, n)
    assert np.array_equal(bh, P["TimeBinHydro"]) and np.array_equal(bg, P["TimeBinGravity"])
    assert (times.PM_length, times.PM_start) == (pm_length, Ti)
    assert (times.mintimebin, times.maxtimebin, bad.value) == (want["mTimeBin"], want["maxTimeBin"], want["badstepsizecount"])
    assert 1 < dti_max < TB


def _gravkick(t):
    """a stand-in for get_exact_gravkick_factor (the cosmology integral is the caller's): any function of the two integer
    times does, both sides use the same one"""
    return lambda ti0, ti1, user=None: 0.37 * (t.loga_from_ti(int(ti1)) - t.loga_from_ti(int(ti0)))


@pytest.mark.parametrize("pm_step", [True, False])
def test_hierarchical_gravity_and_timesteps(ctx, tbm_host, pm_step):
    """hierarchical_gravity_and_timesteps of the host mirror (timestep.cpp:305-480) on the resident set: bins from the stored
    acceleration, push-down on a PM step, then per level sub-list -> tree of the sub-list -> walk -> refinement -> kick.
    The restatement below runs the same levels with host lists on a second context, the particle loops from the oracle and the
    kicks in numpy: time bins, velocities and DriftKickTimes must come out equal."""
    n = 16 ** 3
    rng = np.random.default_rng(21)
    pos = sq.synth_positions("cluster", n, L=cm.BOX)
    pos = pos[sq.hilbert_order(pos, cm.BOX)]
    pman = cm.make_partmanager(pos)
    P = pman.Base
    P["Vel"] = rng.normal(size=(n, 3)) * 100.0
    t = tl.TimeBinMgr(OUTS)
    Ti = TB + (1 << 40) if pm_step else TB + (1 << 40) + (1 << 33)
    atime = math.exp(t.loga_from_ti(Ti))
    hubble = 0.1 * atime ** -1.5
    Nmesh = 48
    cm.reference_treepar(ErrTolForceAcc=0.005, MaxBHOpeningAngle=0.9, Rcut=6.0, TreeUseBH=1)
    sq.gravshort_set_softenings(cm.BOX / 16)
    soft = capi.host.shqh_FORCE_SOFTENING()
    tp = dict(TP, ErrTolIntAccuracy=0.02)
    capi.host.shqh_set_timestep_params(tp["ErrTolIntAccuracy"], 0, tp["MinSizeTimestep"], tp["MaxSizeTimestep"], tp["MaxRMSDisplacementFac"],
                                       tp["MaxGasVel"], tp["CourantFac"])
    gp_bh = sq.make_grav_params(cm.BOX, 1.5, Nmesh, cm.G, cm.RHO0)
    cm.reference_treepar(ErrTolForceAcc=0.005, MaxBHOpeningAngle=0.9, Rcut=6.0, TreeUseBH=0)
    gp = sq.make_grav_params(cm.BOX, 1.5, Nmesh, cm.G, cm.RHO0)
    pmp = sq.PMParams(Nmesh, 0, cm.BOX, 1.5, cm.G)
    # old bins: the PM step ends a step of every bin <= 40; the short step is one of bin 33
    P["TimeBinGravity"] = rng.integers(31, 34, size=n).astype(np.uint8) if not pm_step else rng.integers(33, 41, size=n).astype(np.uint8)
    P["TimeBinHydro"] = P["TimeBinGravity"]
    pv = pman.view()

    def forces(c):
        capi.check(capi.hip.shq_particles_upload(c.h, C.byref(pv)))
        sq.dynamics_upload(c, pman)
        sq.tree_build_device(c, cm.BOX)
        capi.check(capi.hip.shq_pm_run(c.h, C.byref(pmp)))
        capi.check(capi.hip.shq_grav_short_run(c.h, C.byref(gp_bh), None, 0, 1, sq.WALK_EXACT))
        capi.check(capi.hip.shq_grav_refresh_oldacc(c.h, cm.G))
        capi.check(capi.hip.shq_grav_short_run(c.h, C.byref(gp), None, 0, 1, sq.WALK_EXACT))

    forces(ctx)
    acc, gpm = np.zeros((n, 3)), np.zeros((n, 3))
    capi.check(capi.hip.shq_grav_short_download(ctx.h, capi.ptr(acc), None, None, None))
    capi.check(capi.hip.shq_pm_download(ctx.h, capi.ptr(gpm), None))
    P["FullTreeGravAccel"], P["GravPM"] = acc, gpm

    times = capi.DriftKickTimes()
    times.Ti_Current, times.PM_kick = Ti, TB + (1 << 40)
    if pm_step:
        times.PM_start, times.PM_length = Ti - (1 << 40), 1 << 40
    else:
        times.PM_start, times.PM_length = TB + (1 << 40), 1 << 38
    times.mintimebin, times.maxtimebin, times.mingravtimebin = 31, 40, 31
    for b in range(capi.TIMEBINS + 1):
        times.Ti_kick[b] = Ti - (tl.dti_from_timebin(b) // 2 if tl.is_timebin_active(b, Ti) else 0)
    want_times = capi.DriftKickTimes.from_buffer_copy(times)
    kick = _gravkick(t)

    # ---- the restatement
    bg = P["TimeBinGravity"]
    isPM = Ti == want_times.PM_start + want_times.PM_length
    assert isPM == pm_step
    asmth = 1.5 * cm.BOX / Nmesh
    dti_max = want_times.PM_length
    if isPM:
        v, mim, cnt = tl.long_range_moments(Parts(P))
        dloga = tl.long_range_dloga(v, mim, cnt, atime, hubble, COSMO, tp, 2, asmth)
        dti_max = min(tl.round_down_power_of_two(t.dti_from_dloga(dloga, Ti)), t.find_next_ti_sync(Ti) - want_times.PM_kick)
        want_times.PM_length, want_times.PM_start = dti_max, want_times.PM_kick
    largest = next(b for b in range(tl.TIMEBINS, -1, -1) if tl.is_timebin_active(b, Ti) and tl.dti_from_timebin(b) <= want_times.PM_length)
    act_all = np.arange(n, dtype=np.int32) if isPM else np.array([i for i in range(n) if tl.is_timebin_active(int(bg[i]), Ti)], dtype=np.int32)
    top = act_all            # every active particle is gravity-active here (no gas)
    counts = tl.hier_gravity_bins(Parts(P), top, P["FullTreeGravAccel"], Ti, t, dti_max, atime, hubble, tp, soft, largest)
    largest = next((b for b in range(largest, 0, -1) if counts[b] > 0), largest)
    push = largest
    if isPM:
        for b in range(largest, 0, -1):
            if counts[b] // 3 > counts[b - 1]:
                break
            push = b - 1
            counts[b - 1] += counts[b]
    assert push > 0
    if push != largest:
        bg[top] = np.minimum(bg[top], push)
        largest = push
    want_times.maxtimebin = largest
    vel = P["Vel"].copy()

    def hkick(lst, A, ti):
        dti = tl.dti_from_timebin(ti)
        g = kick(want_times.Ti_kick[ti], want_times.Ti_kick[ti] + dti // 2)
        if ti < largest:
            g -= kick(want_times.Ti_kick[ti + 1], want_times.Ti_kick[ti + 1] + tl.dti_from_timebin(ti + 1) // 2)
        vel[lst] = vel[lst] + A[lst] * g

    hkick(top, P["FullTreeGravAccel"], largest)
    c2 = sq.Context(0)
    nlevels = 0
    try:
        forces(c2)
        capi.check(capi.hip.shq_grav_refresh_oldacc(c2.h, cm.G))
        bad = 0
        for ti in range(largest - 1, 0, -1):
            sub = np.array([i for i in act_all if bg[i] <= ti and tl.is_timebin_active(int(bg[i]), Ti)], dtype=np.int32)
            if len(sub) == 0:
                want_times.mingravtimebin = ti + 1
                break
            nlevels += 1
            sq.tree_build_device(c2, cm.BOX, active=sub)
            capi.check(capi.hip.shq_grav_short_run(c2.h, C.byref(gp), capi.ptr(sub), len(sub), 0, sq.WALK_EXACT))
            A = np.zeros((n, 3))
            capi.check(capi.hip.shq_grav_short_download(c2.h, capi.ptr(A), None, None, None))
            bad += tl.hier_refine(Parts(P), sub, A, Ti, t, dti_max, atime, hubble, tp, soft, ti)
            hkick(sub, A, ti)
    finally:
        c2.close()
    want_times.mintimebin = want_times.mingravtimebin

    # ---- the host mirror on the resident set
    cb = capi.GRAVKICK_CB(lambda a, b, u: kick(a, b))
    capi.host.shqh_timebinmgr_set_gravkick(tbm_host, cb, None)
    have_list = 0 if isPM else 1
    ainfo = sq.build_active_particles(ctx, Ti, isPM)
    nact = len(act_all)
    assert ainfo.NumActiveParticle == nact
    badc = C.c_int64(-1)
    cosmo = _cosmo(hubble)
    rc = capi.host.shqh_hierarchical_gravity_and_timesteps(ctx.h, have_list, nact, nact, cm.BOX, 1.5, Nmesh, cm.G, 0, C.byref(times), tbm_host, atime,
                                                           0x3f, 2, C.byref(cosmo), sq.WALK_EXACT, C.byref(badc))
    assert rc == 0, capi.host.shqh_last_error()
    got_bg, _ = _bins(ctx, n)
    sq.dynamics_download(ctx, pman)
    assert np.array_equal(got_bg, bg)
    assert np.array_equal(P["Vel"], vel)
    for f in ("mintimebin", "maxtimebin", "mingravtimebin", "PM_length", "PM_start"):
        assert getattr(times, f) == getattr(want_times, f), f
    assert badc.value == bad
    assert nlevels >= 3 and len(np.unique(bg)) >= 4
    print("hierarchical step (PM %s): %d levels, bins %s" % (pm_step, nlevels, np.unique(bg, return_counts=True)))
