"""The set-ups of the reference's exchange tests (libgadget/tests/test_exchange.cpp:20-57, 84-205), every task in one process."""
import numpy as np

from shenqi_amd import capi

NUMPART1 = 8
SLOT_DTYPES = [capi.SPH_DTYPE, None, None, None, capi.STAR_DTYPE, capi.BH_DTYPE]   # slots_set_enabled(0 / 4 / 5), test_exchange.cpp:34-36


def setup_task(thistask, ntask, ntype, maxpart=1024, rng=None):
    """setup_particles, test_exchange.cpp:20-57: NType[ptype] particles per type in type order, PI = index inside the type
    (slots_setup_topology), ID = (i+1) + NumPart * ThisTask, ReverseLink = particle index (slots_setup_id).  The records get
    recognisable payloads (Pos carries the ID, slot fields carry ID and type) so that the exchange can be checked byte for byte."""
    numpart = int(sum(ntype))
    P = np.zeros(maxpart, dtype=capi.PARTICLE_DTYPE)
    slots, slot_size = [], []
    off = 0
    for t in range(6):
        n = int(ntype[t])
        P["Type"][off:off + n] = t
        if SLOT_DTYPES[t] is not None:
            P["PI"][off:off + n] = np.arange(n)
        off += n
    P["ID"][:numpart] = np.arange(1, numpart + 1) + numpart * thistask
    P["Pos"][:numpart] = P["ID"][:numpart, None] * np.array([1.0, 0.5, 0.25])
    P["Mass"][:numpart] = 1 + thistask
    for t in range(6):
        if SLOT_DTYPES[t] is None:
            slots.append(None)
            slot_size.append(0)
            continue
        S = np.zeros(maxpart, dtype=SLOT_DTYPES[t])
        n = int(ntype[t])
        idx = np.flatnonzero(P["Type"][:numpart] == t)
        S["ReverseLink"][P["PI"][idx]] = idx
        payload = "Density" if t in (0, 5) else "Metallicity"
        S[payload][P["PI"][idx]] = P["ID"][idx] + 0.125 * t
        slots.append(S)
        slot_size.append(n)
    return P, numpart, slots, slot_size


def layout_id_mod(P, numpart, ntask):
    """TestExchangePlan::layoutfunc, test_exchange.cpp:84-92"""
    t = np.full(len(P), -1, dtype=np.int32)
    t[:numpart] = P["ID"][:numpart] % ntask
    return t


def layout_uneven(P, numpart, ntask):
    """TestUnevenExchangePlan::layoutfunc, test_exchange.cpp:153-162"""
    t = layout_id_mod(P, numpart, ntask)
    t[:numpart][P["Type"][:numpart] == 0] = 0
    return t


def check_after(tasks, ntask, tot, uneven=False):
    """teardown_particles / the end of test_exchange_uneven, test_exchange.cpp:58-80, 178-203: every live particle sits on the
    task its ID names (gas on task 0 in the uneven case), none was lost or duplicated, PI and slot payloads still belong together"""
    seen = []
    for r, (P, numpart, slots, slot_size) in enumerate(tasks):
        live = np.flatnonzero((P["Flags"][:numpart] & 1) == 0)
        for i in live:
            if uneven and P["Type"][i] == 0:
                assert r == 0
            else:
                assert P["ID"][i] % ntask == r
            t = int(P["Type"][i])
            if slots[t] is not None:
                pi = int(P["PI"][i])
                assert 0 <= pi < slot_size[t]
                payload = "Density" if t in (0, 5) else "Metallicity"
                assert slots[t][payload][pi] == P["ID"][i] + 0.125 * t
        seen.extend(int(x) for x in P["ID"][live])
    assert len(seen) == tot and len(set(seen)) == tot
