"""The integer time line on the CPU: the numpy / Python restatement (oracle/timeline.py) against the values the reference's own
test holds (libgadget/tests/test_timebinmgr.cpp:23-146, sync points {0.1, 0.2, 0.8, 1.0}), and the C++ host mirror
(integration/reference_side/timestep.cpp) against the restatement."""
import ctypes as C
import math
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import timeline as tl  # noqa: E402
from shenqi_amd import capi  # noqa: E402

OUTS = [0.1, 0.2, 0.8, 1.0]
LOGOUTS = [math.log(a) for a in OUTS]
TB = tl.TIMEBASE


def close(a, b, tol=1e-6):
    """boost's tt::tolerance(1e-6): relative difference"""
    return abs(a - b) <= tol * max(abs(a), abs(b))


def test_oracle_conversions_reference_fixture():
    """test_timebinmgr.cpp:23-48 (test_conversions)"""
    t = tl.TimeBinMgr(OUTS)
    assert t.loga_from_ti(0) == LOGOUTS[0]
    assert t.loga_from_ti(TB) == LOGOUTS[1]
    assert close(t.loga_from_ti(TB - 1), LOGOUTS[0] + (LOGOUTS[1] - LOGOUTS[0]) * (TB - 1) / TB)
    assert close(t.loga_from_ti(TB + 1), LOGOUTS[1] + (LOGOUTS[2] - LOGOUTS[1]) / TB)
    assert t.loga_from_ti(2 * TB) == LOGOUTS[2]
    assert t.ti_from_loga(LOGOUTS[0]) == 0
    assert t.ti_from_loga(LOGOUTS[1]) == TB
    assert t.ti_from_loga(LOGOUTS[2]) == 2 * TB
    midpt = (LOGOUTS[2] + LOGOUTS[1]) / 2
    assert t.ti_from_loga(midpt) == TB + TB // 2
    assert close(t.loga_from_ti(TB + TB // 2), midpt)
    assert t.ti_from_loga(0) == 3 * TB
    assert close(t.loga_from_ti(t.ti_from_loga(math.log(0.1))), math.log(0.1))
    assert t.find_next_ti_sync(0) == TB and t.find_next_ti_sync(TB) == 2 * TB
    assert t.find_next_ti_sync(TB - 1) == TB and t.find_next_ti_sync(TB + 1) == 2 * TB


def test_oracle_dloga_reference_fixture():
    """test_timebinmgr.cpp:102-146 (test_dloga)"""
    t = tl.TimeBinMgr(OUTS)
    Ti = t.ti_from_loga(math.log(0.55))
    loga0 = t.loga_from_ti(Ti)
    dloga_in = (LOGOUTS[2] - LOGOUTS[1]) / 4
    assert t.dti_from_dloga(dloga_in, Ti) == t.ti_from_loga(loga0 + dloga_in) - t.ti_from_loga(loga0)
    assert TB // 4 - 2 <= t.dti_from_dloga(dloga_in, Ti) <= TB // 4 + 2
    dloga_cross = (LOGOUTS[2] - LOGOUTS[1]) / 2 + (LOGOUTS[3] - LOGOUTS[2]) / 4
    assert t.dti_from_dloga(dloga_cross, Ti) == t.ti_from_loga(loga0 + dloga_cross) - t.ti_from_loga(loga0)
    assert t.get_dloga_for_bin(0, Ti) < 1e-6
    assert close(t.get_dloga_for_bin(tl.TIMEBINS, Ti), LOGOUTS[2] - LOGOUTS[1])
    assert close(t.get_dloga_for_bin(tl.TIMEBINS - 2, Ti), (LOGOUTS[2] - LOGOUTS[1]) / 4)
    assert tl.round_down_power_of_two(TB) == TB
    assert tl.round_down_power_of_two(TB + 1) == TB
    assert tl.round_down_power_of_two(TB - 1) == TB // 2


@pytest.fixture(scope="module")
def tbm():
    la = (C.c_double * 4)(*LOGOUTS)
    h = capi.host.shqh_timebinmgr_create(la, 4)
    yield h
    capi.host.shqh_timebinmgr_destroy(h)


def test_host_mirror_equals_oracle(tbm):
    """the C++ TimeBinMgr / bin helpers of the host mirror, the reference fixture again and random arguments against the
    restatement (integers equal, doubles bit-equal: same operations)"""
    H = capi.host
    t = tl.TimeBinMgr(OUTS)
    assert H.shqh_tbm_ti_from_loga(tbm, LOGOUTS[1]) == TB and H.shqh_tbm_ti_from_loga(tbm, 0.0) == 3 * TB
    assert H.shqh_round_down_power_of_two(TB - 1) == TB // 2
    rng = np.random.default_rng(3)
    for _ in range(2000):
        Ti = int(rng.integers(0, 3 * TB + 5))
        dloga = float(10 ** rng.uniform(-9, 0.3))
        assert H.shqh_tbm_dti_from_dloga(tbm, dloga, Ti) == t.dti_from_dloga(dloga, Ti)
        assert H.shqh_tbm_loga_from_ti(tbm, Ti) == t.loga_from_ti(Ti)
        assert H.shqh_tbm_ti_from_loga(tbm, t.loga_from_ti(Ti) + dloga) == t.ti_from_loga(t.loga_from_ti(Ti) + dloga)
        b = int(rng.integers(0, tl.TIMEBINS + 1))
        assert H.shqh_tbm_get_dloga_for_bin(tbm, b, Ti) == t.get_dloga_for_bin(b, Ti)
        dti = int(rng.integers(-5, 2 * TB))
        assert H.shqh_tbm_dloga_from_dti(tbm, dti, Ti) == t.dloga_from_dti(dti, Ti)
        assert H.shqh_round_down_power_of_two(dti) == tl.round_down_power_of_two(dti)
        assert H.shqh_get_timestep_bin(dti) == tl.get_timestep_bin(dti)
        assert bool(H.shqh_is_timebin_active(b, Ti)) == tl.is_timebin_active(b, Ti)
        assert H.shqh_tbm_find_next_ti_sync(tbm, Ti) == t.find_next_ti_sync(Ti)


def test_timeline_struct_reproduces_dti_from_dloga(tbm):
    """shq_timeline (what the device loops get) carries enough of the table: evaluating the device formula on it equals
    dti_from_dloga, inside a segment, across a sync point and past the last one"""
    t = tl.TimeBinMgr(OUTS)
    rng = np.random.default_rng(5)
    line = capi.Timeline()
    for _ in range(3000):
        Ti = int(rng.integers(0, 3 * TB + 5))
        capi.host.shqh_tbm_timeline_at(tbm, Ti, C.byref(line))
        assert line.Ti_Current == Ti and line.loga_now == t.loga_from_ti(Ti) and line.Dloga_interval == t.Dloga_interval_ti(Ti)
        dloga = float(10 ** rng.uniform(-9, 0.3))
        target = dloga + line.loga_now
        s = 1 if (line.nseg == 2 and line.seg_loga[1] <= target) else 0
        logDTime = (line.seg_loga[s + 1] - line.seg_loga[s]) / TB
        ti = tl._to_int(float(line.seg_snap[s] << tl.TIMEBINS) + (target - line.seg_loga[s]) / logDTime)
        assert ti - Ti == t.dti_from_dloga(dloga, Ti)
