"""CPU restatement (numpy / Python integers) of the reference's integer time line for the parity tests.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing in shenqi_amd/).  Each function cites what it follows in
/root/reference/libgadget.  Floating point is IEEE double in the reference's operation order (numpy scalars and arrays do
not contract to fma), integers are Python ints, so the bins it produces are what the reference's loops produce.

Pinned by the reference's own fixture tests/test_timebinmgr.cpp (sync points {0.1, 0.2, 0.8, 1.0}: every BOOST_TEST on
loga_from_ti, ti_from_loga, dti_from_dloga, get_dloga_for_bin and round_down_power_of_two is re-run in
tests/test_timeline_cpu.py).  The particle loops (find_timesteps ...) have no fixture in the reference: parity unpinned for
those, restated line by line."""
import math

import numpy as np

TIMEBINS = 46
TIMEBASE = 1 << TIMEBINS
GAMMA = 5.0 / 3
TI_ACCEL, TI_COURANT, TI_ACCRETE, TI_NEIGH, TI_HSML = range(5)  # enum TimeStepType, timestep.cpp:87-94


def dti_from_timebin(b):
    """timebinmgr.h:42-45"""
    return (1 << b) if b > 0 else 0


def is_timebin_active(i, current):
    """timestep.cpp:132-139"""
    if i <= 0 or current <= 0:
        return True
    return current % dti_from_timebin(i) == 0


def round_down_power_of_two(dti):
    """timebinmgr.cpp:189-203"""
    ti_min = TIMEBASE
    sign = 1
    if dti < 0:
        dti, sign = -dti, -1
    while ti_min > dti:
        ti_min >>= 1
    return ti_min * sign


def get_timestep_bin(dti):
    """timestep.cpp:1236-1251"""
    if dti <= 1:
        return 0
    b = -1
    while dti:
        b += 1
        dti >>= 1
    return b


def _to_int(x):
    """(inttime_t) of a double: C truncation; out of range is undefined in C (x86 gives INT64_MIN)"""
    if not (-9.2e18 < x < 9.2e18):
        return -(1 << 63)
    return int(x)


class TimeBinMgr:
    """timebinmgr.h:48-260, the conversions"""

    def __init__(self, sync_a):
        self.loga = [math.log(a) for a in sync_a]

    @property
    def N(self):
        return len(self.loga)

    def Dloga_interval_ti(self, ti):
        lastsnap = ti >> TIMEBINS
        if lastsnap >= self.N - 1:
            return 0.0
        return (self.loga[lastsnap + 1] - self.loga[lastsnap]) / TIMEBASE

    def loga_from_ti(self, ti):
        lastsnap = min(ti >> TIMEBINS, self.N - 1)
        return self.loga[lastsnap] + (ti & (TIMEBASE - 1)) * self.Dloga_interval_ti(ti)

    def ti_from_loga(self, loga):
        i = 1
        while i < self.N - 1 and not self.loga[i] > loga:
            i += 1
        logDTime = (self.loga[i] - self.loga[i - 1]) / TIMEBASE
        ti = (i - 1) << TIMEBINS
        return _to_int(float(ti) + (loga - self.loga[i - 1]) / logDTime)

    def ti_from_loga_snap(self, loga, lastsnap):
        logDTime = (self.loga[lastsnap + 1] - self.loga[lastsnap]) / TIMEBASE
        ti = lastsnap << TIMEBINS
        return _to_int(float(ti) + (loga - self.loga[lastsnap]) / logDTime)

    def dti_from_dloga(self, dloga, Ti_Current):
        lastsnap = min(Ti_Current >> TIMEBINS, self.N - 1)
        dti = Ti_Current & (TIMEBASE - 1)
        loga = self.loga[lastsnap] + dti * self.Dloga_interval_ti(Ti_Current)
        if lastsnap >= self.N - 1:
            lastsnap = self.N - 2
        if lastsnap < self.N - 2 and self.loga[lastsnap + 1] <= dloga + loga:
            lastsnap += 1
        return self.ti_from_loga_snap(dloga + loga, lastsnap) - Ti_Current

    def dloga_from_dti(self, dti, Ti_Current):
        sign = 1
        if dti < 0:
            dti, sign = -dti, -1
        dti = min(dti, TIMEBASE)
        return self.Dloga_interval_ti(Ti_Current) * dti * sign

    def get_dloga_for_bin(self, timebin, Ti_Current):
        return dti_from_timebin(timebin) * self.Dloga_interval_ti(Ti_Current)

    def find_next_ti_sync(self, ti):
        return ((ti >> TIMEBINS) + 1) << TIMEBINS


def convert_timestep_to_ti(dloga, dti_max, Ti_Current, tbm, MinSizeTimestep):
    """timestep.cpp:157-174"""
    if dti_max == 0:
        return 0
    if dloga < MinSizeTimestep:
        dloga = MinSizeTimestep
    dti = tbm.dti_from_dloga(dloga, Ti_Current)
    if dti > dti_max or dti < 0:
        dti = dti_max
    return dti


def get_timebin_from_dti(dti, binold, Ti_Current):
    """timestep.cpp:176-192"""
    dti = round_down_power_of_two(dti)
    b = get_timestep_bin(dti)
    if b > binold:
        while (not is_timebin_active(b, Ti_Current)) and b > binold and b > 1:
            b -= 1
    return b


def gravity_dloga(GravAccel, GravPM, atime, hubble, tp, soft):
    """grav_acceleration2 + get_timestep_gravity_dloga (timestep.cpp:1012-1040) for arrays [n][3]"""
    a2inv = 1 / (atime * atime)
    ax = a2inv * GravAccel[:, 0]
    ay = a2inv * GravAccel[:, 1]
    az = a2inv * GravAccel[:, 2]
    ay = ay + a2inv * GravPM[:, 1]
    ax = ax + a2inv * GravPM[:, 0]
    az = az + a2inv * GravPM[:, 2]
    ac2 = ax * ax + ay * ay + az * az
    ac2 = np.where(ac2 == 0, 1.0e-60, ac2)
    ac = np.sqrt(ac2)
    dt = np.sqrt(2 * tp["ErrTolIntAccuracy"] * atime * (soft / 2.8) / ac)
    return dt * hubble


def hydro_dloga(i, P, MaxSignalVel, bh_minTimeBin, Ti_Current, tbm, atime, hubble, tp):
    """get_timestep_hydro_dloga, timestep.cpp:1042-1081; bh_minTimeBin: dict particle index -> BHP.minTimeBin"""
    dt = 1.0
    titype = TI_ACCEL
    if P["Type"][i] == 0:
        fac3 = math.pow(atime, 3 * (1 - GAMMA) / 2.0)
        hs = float(P["Hsml"][i])
        dt = 2 * tp["CourantFac"] * atime * hs / (fac3 * float(MaxSignalVel[i]))
        titype = TI_COURANT
        dt_hsml = tp["CourantFac"] * atime * atime * abs(hs / (float(P["DtHsml"][i]) + 1e-20))
        if dt_hsml < dt:
            dt, titype = dt_hsml, TI_HSML
    elif P["Type"][i] == 5:
        mb = bh_minTimeBin.get(i, 0)
        if mb > 0 and mb + 1 < TIMEBINS:
            dt = tbm.get_dloga_for_bin(mb + 1, Ti_Current) / hubble
            titype = TI_NEIGH
    return dt * hubble, titype


def dynfric_dloga(i, P, DF_SurroundingVel, atime, hubble, tp):
    """get_timestep_dynfric_dloga, timestep.cpp:1085-1110"""
    bhvel = 0.0
    bhvel2 = 0.0
    for j in range(3):
        d = float(P["Vel"][i][j]) - float(DF_SurroundingVel[j])
        bhvel += d * d
        bhvel2 += float(P["Vel"][i][j]) * float(P["Vel"][i][j])
    if bhvel2 > bhvel:
        bhvel = bhvel2
    bhvel = math.sqrt(bhvel)
    hs = float(P["Hsml"][i])
    dt = 2 * tp["ErrTolIntAccuracy"] * atime * atime * hs / (bhvel + 1e-20)
    dt_hsml = tp["CourantFac"] * atime * atime * abs(hs / (float(P["DtHsml"][i]) + 1e-20))
    if dt_hsml < dt:
        dt = dt_hsml
    return dt * hubble


def _live(P, i):
    return not (P["IsGarbage"][i] or P["Swallowed"][i])


def find_global_timestep(P, MaxSignalVel, bh_minTimeBin, Ti_Current, tbm, dti_max, atime, hubble, tp, soft):
    """timestep.cpp:195-221"""
    dti_min = TIMEBASE
    dg = gravity_dloga(P["FullTreeGravAccel"], P["GravPM"], atime, hubble, tp, soft)
    for i in range(len(P)):
        if not _live(P, i):
            continue
        dloga = float(dg[i])
        dh, _ = hydro_dloga(i, P, MaxSignalVel, bh_minTimeBin, Ti_Current, tbm, atime, hubble, tp)
        if dh < dloga:
            dloga = dh
        dti = convert_timestep_to_ti(dloga, dti_max, Ti_Current, tbm, tp["MinSizeTimestep"])
        dti_min = min(dti_min, dti)
    return dti_min


def find_timesteps_loop(P, act, MaxSignalVel, bh_minTimeBin, Ti_Current, tbm, dti_max, atime, hubble, tp, soft, dti_min_global=None):
    """the particle loop of find_timesteps, timestep.cpp:733-792.  P is modified (TimeBinHydro, TimeBinGravity).
    Returns dict(badstepsizecount, mTimeBin, maxTimeBin, counts[5] by TimeStepType)."""
    bad = 0
    mTimeBin, maxTimeBin = TIMEBINS, 0
    counts = [0] * 5
    dg = gravity_dloga(P["FullTreeGravAccel"], P["GravPM"], atime, hubble, tp, soft)
    for i in (range(len(P)) if act is None else act):
        i = int(i)
        if not _live(P, i):
            continue
        titype = TI_ACCEL
        if tp["ForceEqualTimesteps"]:
            dti = dti_min_global
        else:
            dti = convert_timestep_to_ti(float(dg[i]), dti_max, Ti_Current, tbm, tp["MinSizeTimestep"])
            if P["Type"][i] == 0 or P["Type"][i] == 5:
                dh, th = hydro_dloga(i, P, MaxSignalVel, bh_minTimeBin, Ti_Current, tbm, atime, hubble, tp)
                dti_hydro = convert_timestep_to_ti(dh, dti_max, Ti_Current, tbm, tp["MinSizeTimestep"])
                if dti_hydro < dti:
                    dti, titype = dti_hydro, th
            counts[titype] += 1
        b = get_timebin_from_dti(dti, int(P["TimeBinHydro"][i]), Ti_Current)
        if b < 1:
            bad += 1
        if is_timebin_active(int(P["TimeBinHydro"][i]), Ti_Current) and is_timebin_active(b, Ti_Current):
            P["TimeBinHydro"][i] = b
            P["TimeBinGravity"][i] = b
        mTimeBin = min(mTimeBin, b)
        maxTimeBin = max(maxTimeBin, b)
    return dict(badstepsizecount=bad, mTimeBin=mTimeBin, maxTimeBin=maxTimeBin, counts=counts)


def find_hydro_timesteps_loop(P, act, MaxSignalVel, bh_minTimeBin, bh_dynfric, bh_dfsurr, Ti_Current, tbm, dti_max, atime, hubble, tp):
    """the particle loop of find_hydro_timesteps, timestep.cpp:596-658.  bh_dynfric (dict index -> TimeBinDynFric) is modified."""
    bad = 0
    mTimeBin = TIMEBINS
    counts = [0] * 5
    dynratio = nbh = maxdyndiff = 0
    for i in (range(len(P)) if act is None else act):
        i = int(i)
        if not _live(P, i):
            continue
        if P["Type"][i] != 0 and P["Type"][i] != 5:
            continue
        dh, titype = hydro_dloga(i, P, MaxSignalVel, bh_minTimeBin, Ti_Current, tbm, atime, hubble, tp)
        dti_hydro = convert_timestep_to_ti(dh, dti_max, Ti_Current, tbm, tp["MinSizeTimestep"])
        bin_hydro = get_timebin_from_dti(dti_hydro, int(P["TimeBinHydro"][i]), Ti_Current)
        bg = int(P["TimeBinGravity"][i])
        if bin_hydro > bg:
            bin_hydro, titype = bg, TI_ACCEL
        if bin_hydro < 1:
            bad += 1
        counts[titype] += 1
        if is_timebin_active(int(P["TimeBinHydro"][i]), Ti_Current) and is_timebin_active(bin_hydro, Ti_Current):
            P["TimeBinHydro"][i] = bin_hydro
        mTimeBin = min(mTimeBin, bin_hydro)
        if P["Type"][i] == 5 and i in bh_dynfric:
            dd = dynfric_dloga(i, P, bh_dfsurr[i], atime, hubble, tp)
            dti_d = convert_timestep_to_ti(dd, dti_max, Ti_Current, tbm, tp["MinSizeTimestep"])
            bd = get_timebin_from_dti(dti_d, bh_dynfric[i], Ti_Current)
            bh = int(P["TimeBinHydro"][i])
            if bd > bg:
                bd = bg
            if bd < bh:
                bd = bh
            bh_dynfric[i] = bd
            dynratio += bd - bh
            maxdyndiff = max(maxdyndiff, bd - bh)
            nbh += 1
    return dict(badstepsizecount=bad, mTimeBin=mTimeBin, counts=counts, dynratio=dynratio, nbh=nbh, maxdyndiff=maxdyndiff)


def hydro_mintimebin_fixups(mTimeBin, times_mintimebin, times_mingravtimebin, Ti_Current):
    """timestep.cpp:677-696 -> (mTimeBin for set_bh_first_timestep, times->mintimebin)"""
    if not is_timebin_active(mTimeBin, Ti_Current):
        mTimeBin = times_mintimebin
        if is_timebin_active(mTimeBin + 1, Ti_Current):
            mTimeBin += 1
    mint = mTimeBin
    if mint > times_mingravtimebin and times_mingravtimebin > 0:
        mint = times_mingravtimebin
    return mTimeBin, mint


def hier_gravity_bins(P, act, GravAccel, Ti_Current, tbm, dti_max, atime, hubble, tp, soft, largest_active):
    """timestep.cpp:356-380 -> timebincounts"""
    counts = [0] * (TIMEBINS + 1)
    dg = gravity_dloga(GravAccel, P["GravPM"], atime, hubble, tp, soft)
    for i in (range(len(P)) if act is None else act):
        i = int(i)
        if not _live(P, i):
            continue
        dti = convert_timestep_to_ti(float(dg[i]), dti_max, Ti_Current, tbm, tp["MinSizeTimestep"])
        dti = round_down_power_of_two(dti)
        b = min(get_timestep_bin(dti), largest_active)
        counts[b] += 1
        P["TimeBinGravity"][i] = b
    return counts


def hier_refine(P, act, GravAccel, Ti_Current, tbm, dti_max, atime, hubble, tp, soft, ti):
    """timestep.cpp:449-464 -> badstepsizecount"""
    bad = 0
    dg = gravity_dloga(GravAccel, P["GravPM"], atime, hubble, tp, soft)
    for i in act:
        i = int(i)
        if not _live(P, i):
            continue
        dti = convert_timestep_to_ti(float(dg[i]), dti_max, Ti_Current, tbm, tp["MinSizeTimestep"])
        if dti < dti_from_timebin(ti):
            P["TimeBinGravity"][i] = ti - 1
            if ti == 1:
                bad += 1
    return bad


def long_range_moments(P):
    """the particle loop of get_long_range_timestep_dloga, timestep.cpp:1153-1166 (sum in blocks of 256 in index order, the
    order the device sum is defined with; the reference's OpenMP reduction has none)"""
    v = np.zeros(6)
    mim = np.full(6, 1.0e30)
    count = np.zeros(6, dtype=np.int64)
    n = len(P)
    live = ~(P["IsGarbage"].astype(bool) | P["Swallowed"].astype(bool))
    v2 = P["Vel"][:, 0] * P["Vel"][:, 0] + P["Vel"][:, 1] * P["Vel"][:, 1] + P["Vel"][:, 2] * P["Vel"][:, 2]
    for t in range(6):
        sel = live & (P["Type"] == t)
        count[t] = sel.sum()
        m = P["Mass"][sel & (P["Mass"] > 0)]
        if len(m):
            mim[t] = float(m.min())
        tot = 0.0
        for b0 in range(0, n, 256):
            s = 0.0
            for x in v2[b0:b0 + 256][sel[b0:b0 + 256]]:
                s += float(x)
            tot += s
        v[t] = tot
    return v, mim, count


def long_range_dloga(v_sum, min_mass, count_sum, atime, hubble, cosmo, tp, FastParticleType, asmth):
    """timestep.cpp:1172-1219"""
    v_sum, min_mass, count_sum = v_sum.copy(), min_mass.copy(), count_sum.copy()
    dloga = tp["MaxSizeTimestep"]
    v_sum[0] += v_sum[4]
    count_sum[0] += count_sum[4]
    v_sum[4] = v_sum[0]
    count_sum[4] = count_sum[0]
    v_sum[0] += v_sum[5]
    count_sum[0] += count_sum[5]
    v_sum[5] = v_sum[0]
    count_sum[5] = count_sum[0]
    min_mass[5] = min_mass[0]
    for t in range(6):
        if count_sum[t] == 0:
            continue
        omega = cosmo["OmegaBaryon"] if t in (0, 4, 5) else (cosmo["OmegaNu1"] if t == 2 else cosmo["OmegaCDM"])
        dmean = math.pow(float(min_mass[t]) / (omega * cosmo["RhoCrit"]), 1.0 / 3)
        dloga1 = tp["MaxRMSDisplacementFac"] * hubble * atime * atime * min(asmth, dmean) / math.sqrt(float(v_sum[t]) / int(count_sum[t]))
        if t != FastParticleType and dloga1 < dloga:
            dloga = dloga1
    return max(dloga, tp["MinSizeTimestep"])
