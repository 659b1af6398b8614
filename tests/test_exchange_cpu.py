"""The restatement of the particle exchange (oracle/exchange.py) against the four cases of the reference's
libgadget/tests/test_exchange.cpp, with 1, 2, 3 and 5 tasks emulated in one process."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import exchange as oex  # noqa: E402
import exchange_fixtures as fx  # noqa: E402

CASES = {"all": [8] * 6, "zero_slots": [8, 0, 8, 0, 8, 0]}


def run(ntask, ntype, layout, garbage=False):
    tasks = [list(fx.setup_task(r, ntask, ntype)) for r in range(ntask)]
    tot = ntask * sum(ntype)
    if garbage:
        for T in tasks:                      # slots_mark_garbage(0, ...), test_exchange.cpp:136
            T[0]["Flags"][0] |= 1
            T[2][0]["ReverseLink"][T[0]["PI"][0]] = len(T[0]) + 100
        tot -= ntask
    objs = [oex.Task(T[0], T[1], T[2], T[3]) for T in tasks]
    targets = [layout(T[0], T[1], ntask) for T in tasks]
    oex.domain_exchange(objs, targets)
    return [(o.parts, o.numpart, o.slots, o.slot_size) for o in objs], tot


@pytest.mark.parametrize("ntask", [1, 2, 3, 5])
@pytest.mark.parametrize("case", ["all", "zero_slots"])
def test_exchange(ntask, case):
    tasks, tot = run(ntask, CASES[case], fx.layout_id_mod)
    fx.check_after(tasks, ntask, tot)


@pytest.mark.parametrize("ntask", [2, 4])
def test_exchange_with_garbage(ntask):
    tasks, tot = run(ntask, CASES["all"], fx.layout_id_mod, garbage=True)
    fx.check_after(tasks, ntask, tot)


@pytest.mark.parametrize("ntask", [2, 4])
def test_exchange_uneven(ntask):
    tasks, tot = run(ntask, CASES["all"], fx.layout_uneven)
    fx.check_after(tasks, ntask, tot, uneven=True)
    assert tasks[0][3][0] == fx.NUMPART1 * ntask       # the gas slots of task 0 grew to hold everybody's gas (test_exchange.cpp:179)


def test_slots_gc_reference_fixture():
    """tests/test_slotsmanager.cpp:65-85 (test_slots_gc): 6 x 128 particles, the first of every type marked garbage, all types
    compacted: 6 x 127 particles and 127 slots per enabled type remain, PI and payloads still belong together"""
    P, numpart, slots, slot_size = fx.setup_task(0, 1, [128] * 6)
    T = oex.Task(P, numpart, slots, slot_size)
    for i in range(6):
        k = 128 * i
        P["Flags"][k] |= 1
        t = int(P["Type"][k])
        if slots[t] is not None:
            slots[t]["ReverseLink"][P["PI"][k]] = len(P) + 100
    oex.slots_gc(T, [1] * 6)
    assert T.numpart == 127 * 6 and [T.slot_size[t] for t in (0, 4, 5)] == [127] * 3
    fx.check_after([(T.parts, T.numpart, T.slots, T.slot_size)], 1, 127 * 6)


def test_exchange_in_batches_with_gc():
    """a cap of 5 list entries per round: several rounds with a garbage collection between pack and receive in each
    (exchange.hpp:398-406), same end state as the fixture demands"""
    ntask = 3
    tasks = [oex.Task(*fx.setup_task(r, ntask, [8] * 6, maxpart=96)) for r in range(ntask)]
    lay = [lambda P, n, nt=ntask: fx.layout_id_mod(P, n, nt)] * ntask
    iters = oex.domain_exchange_batched(tasks, lay, 5)
    assert iters >= 4
    fx.check_after([(T.parts, T.numpart, T.slots, T.slot_size) for T in tasks], ntask, ntask * 48)


def _ref_keys(P, n, box):
    import ctypes as C
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libpeano_ref.so"))
    lib.ref_PEANO.restype = C.c_uint64
    lib.ref_PEANO.argtypes = [C.c_void_p, C.c_double]
    return np.array([lib.ref_PEANO(np.ascontiguousarray(P["Pos"][i]).ctypes.data, box) for i in range(n)], dtype=np.uint64)


def gc_sorted_setup(rng):
    P, numpart, slots, slot_size = fx.setup_task(0, 1, [128] * 6)
    P["Pos"][:numpart] = rng.random((numpart, 3)) * 25000.0
    for i in range(6):
        k = 128 * i
        P["Flags"][k] |= 1
        t = int(P["Type"][k])
        if slots[t] is not None:
            slots[t]["ReverseLink"][P["PI"][k]] = len(P) + 100
    return P, numpart, slots, slot_size


def check_gc_sorted(P, numpart, slots, slot_size, box=25000.0):
    """tests/test_slotsmanager.cpp:87-115 (test_slots_gc_sorted)"""
    assert numpart == 127 * 6 and [slot_size[t] for t in (0, 4, 5)] == [127] * 3
    keys = _ref_keys(P, numpart, box)
    ty = P["Type"][:numpart].astype(int)
    assert (np.diff(ty) >= 0).all()
    same = np.diff(ty) == 0
    assert (keys[1:][same] >= keys[:-1][same]).all()
    fx.check_after([(P, numpart, slots, slot_size)], 1, 127 * 6)


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libpeano_ref.so")), reason="needs the reference's peano key (make -C oracle ref)")
def test_slots_gc_sorted_reference_fixture():
    P, numpart, slots, slot_size = gc_sorted_setup(np.random.default_rng(5))
    T = oex.Task(P, numpart, slots, slot_size)
    oex.slots_gc_sorted(T, _ref_keys(P, numpart, 25000.0))
    check_gc_sorted(T.parts, T.numpart, T.slots, T.slot_size)


def _fork_setup():
    P, numpart, slots, slot_size = fx.setup_task(0, 1, [128] * 6)
    return oex.Task(P, numpart, slots, slot_size)


def test_slots_fork_reference_fixture():
    """tests/test_slotsmanager.cpp:268-286 (test_slots_fork): the first particle of every type splits off a child of mass 0 and is
    then converted to its own type: 6 x 129 particles, 129 slots of every enabled type"""
    T = _fork_setup()
    ids = T.parts["ID"].copy()
    for i in range(6):
        child = oex.slots_split_particle(T, 128 * i, 0.0)
        assert child == 128 * 6 + i and T.parts["PI"][child] == -1 and T.parts["Mass"][child] == 0
        assert T.parts["ID"][child] == ids[128 * i] + (1 << 56) and (T.parts["Flags"][128 * i] >> 4) == 1
        oex.slots_convert(T, 128 * i, int(T.parts["Type"][128 * i]))
    assert T.numpart == 129 * 6
    assert [T.slot_size[t] for t in (0, 4, 5)] == [129] * 3


def test_slots_convert_reference_fixture():
    """tests/test_slotsmanager.cpp:288-304 (test_slots_convert): conversion to the own type makes a new slot, no new particle; the
    old slot is garbage for the next collection"""
    T = _fork_setup()
    for i in range(6):
        oex.slots_convert(T, 128 * i, int(T.parts["Type"][128 * i]))
    assert T.numpart == 128 * 6
    assert [T.slot_size[t] for t in (0, 4, 5)] == [129] * 3
    for t in (0, 4, 5):
        k = int(np.flatnonzero(T.parts["Type"][:T.numpart] == t)[0])
        assert T.parts["PI"][k] == 128 and T.slots[t]["ReverseLink"][0] == T.maxpart + 100
        assert (T.slots[t][128:129].view(np.uint8) == 101).all()
    oex.slots_gc(T, [1] * 6)
    assert T.numpart == 128 * 6 and [T.slot_size[t] for t in (0, 4, 5)] == [128] * 3


def test_slots_split_and_convert_limits():
    T = _fork_setup()
    T.parts["Flags"][5] = 0xF0 | 0                       # Generation 15: the 4-bit field wraps to 0, as in the reference
    c = oex.slots_split_particle(T, 5, 0.25)
    assert (T.parts["Flags"][5] >> 4) == 0 and T.parts["ID"][c] == (int(T.parts["ID"][5]) & 0x00ffffffffffffff)
    assert T.parts["Mass"][5] == np.float32(0.75) and T.parts["Mass"][c] == np.float32(0.25)
    T.numpart = T.maxpart
    with pytest.raises(MemoryError):
        oex.slots_split_particle(T, 5, 0.1)
    with pytest.raises(MemoryError):
        oex.slots_convert(T, 5, 4, maxsize=[128] * 6)
    oex.slots_convert(T, 5, 2)                           # a type without slots: only the type changes, the gas slot is garbage
    assert T.parts["Type"][5] == 2 and T.slots[0]["ReverseLink"][5] == T.maxpart + 100
