"""GPU parity tests of the particle-exchange kernels (csrc/exchange.hip: shq_exchange_plan / _pack / _unpack) against the
restatement of libgadget/exchange.hpp in oracle/exchange.py: all tasks of the reference's tests/test_exchange.cpp cases live in one
process, each with its own device arrays, the alltoallv between pack and unpack is a set of device-to-device slice copies, and at
the end every task's particle and slot arrays must equal the oracle's field for field, garbage-marked sources included."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

import shenqi_amd as sq
from shenqi_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import exchange as oex  # noqa: E402
import exchange_fixtures as fx  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def layout_struct():
    f = capi.PARTICLE_DTYPE.fields
    L = capi.ExchangeLayout()
    L.part_elsize, L.off_flags, L.off_type, L.off_pi = capi.PARTICLE_DTYPE.itemsize, f["Flags"][1], f["Type"][1], f["PI"][1]
    for t in range(6):
        L.slot_elsize[t] = 0 if fx.SLOT_DTYPES[t] is None else fx.SLOT_DTYPES[t].itemsize
    L.off_reverselink = 0
    return L


def dev(a):
    return torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).to(DEV)


def same_records(a, b):
    """every field equal (numpy leaves the padding bytes of a copied structured array undefined, so no raw byte compare)"""
    return len(a) == len(b) and all(np.array_equal(a[f], b[f]) for f in a.dtype.names)


def entries(arr):
    e = (capi.ExchangeEntry * len(arr))()
    for k, row in enumerate(arr):
        e[k].base = int(row[0])
        for t in range(6):
            e[k].slots[t] = int(row[1 + t])
    return e


def gpu_exchange(ctx, tasks, targets, ntask, maxlast=0):
    L = layout_struct()
    esz = L.part_elsize
    d_parts = [dev(T[0]) for T in tasks]
    d_slots = [[None if s is None else dev(s) for s in T[2]] for T in tasks]
    d_tgt = [torch.from_numpy(t.copy()).to(DEV) for t in targets]
    togo = np.zeros((ntask, ntask, 7), dtype=np.int64)
    partbuf, slotbuf, last = [], [], []
    for r in range(ntask):
        tg = (capi.ExchangeEntry * ntask)()
        nex, la = C.c_int64(), C.c_int64()
        capi.check(capi.hip.shq_exchange_plan(ctx.h, C.byref(L), d_parts[r].data_ptr(), tasks[r][1], d_tgt[r].data_ptr(), r, ntask, maxlast, C.byref(nex),
                                              C.byref(la), tg))
        for t in range(ntask):
            togo[r, t, 0] = tg[t].base
            togo[r, t, 1:] = list(tg[t].slots)
        off = oex.offsets(togo[r])
        pb = torch.zeros(max(int(togo[r][:, 0].sum()), 1) * esz, dtype=torch.uint8, device=DEV)
        sb = [None if fx.SLOT_DTYPES[t] is None else torch.zeros(max(int(togo[r][:, 1 + t].sum()), 1) * fx.SLOT_DTYPES[t].itemsize, dtype=torch.uint8, device=DEV)
              for t in range(6)]
        sp = (C.c_void_p * 6)(*[None if s is None else s.data_ptr() for s in d_slots[r]])
        bp = (C.c_void_p * 6)(*[None if s is None else s.data_ptr() for s in sb])
        capi.check(capi.hip.shq_exchange_pack(ctx.h, C.byref(L), d_parts[r].data_ptr(), sp, len(tasks[r][0]), entries(off), ntask, pb.data_ptr(), bp))
        partbuf.append(pb)
        slotbuf.append(sb)
        last.append((nex.value, la.value))
    toget = np.stack([np.stack([togo[src][r] for src in range(ntask)]) for r in range(ntask)])
    out = []
    for r in range(ntask):
        P, numpart, slots, slot_size = tasks[r]
        goff = oex.offsets(toget[r])
        for src in range(ntask):                        # the alltoallv: slices of the senders' buffers behind NumPart / the slot sizes
            soff = oex.offsets(togo[src])
            nb = int(toget[r][src, 0])
            a = (numpart + int(goff[src, 0])) * esz
            d_parts[r][a:a + nb * esz] = partbuf[src][int(soff[r, 0]) * esz:(int(soff[r, 0]) + nb) * esz]
            for t in range(6):
                if fx.SLOT_DTYPES[t] is None:
                    continue
                ssz = fx.SLOT_DTYPES[t].itemsize
                ns = int(toget[r][src, 1 + t])
                a = (slot_size[t] + int(goff[src, 1 + t])) * ssz
                d_slots[r][t][a:a + ns * ssz] = slotbuf[src][t][int(soff[r, 1 + t]) * ssz:(int(soff[r, 1 + t]) + ns) * ssz]
        torch.cuda.synchronize()
        so = (C.c_int64 * 6)(*slot_size)
        capi.check(capi.hip.shq_exchange_unpack(ctx.h, C.byref(L), d_parts[r].data_ptr(), numpart, so, entries(toget[r]), entries(goff), ntask))
        newP = d_parts[r].cpu().numpy().view(capi.PARTICLE_DTYPE)
        newS = [None if s is None else s.cpu().numpy().view(fx.SLOT_DTYPES[t]) for t, s in enumerate(d_slots[r])]
        out.append((newP, numpart + int(toget[r][:, 0].sum()), newS, [slot_size[t] + (int(toget[r][:, 1 + t].sum()) if fx.SLOT_DTYPES[t] is not None else 0) for t in range(6)]))
    return out, togo, last, partbuf, slotbuf


def both(ctx, ntask, ntype, layout, garbage=False, swallowed=False, maxpart=1024):
    tasks = [list(fx.setup_task(r, ntask, ntype, maxpart=maxpart)) for r in range(ntask)]
    tot = ntask * sum(ntype)
    if garbage:
        for T in tasks:
            T[0]["Flags"][0] |= 1
            T[2][0]["ReverseLink"][T[0]["PI"][0]] = len(T[0]) + 100
        tot -= ntask
    targets = [layout(T[0], T[1], ntask) for T in tasks]
    if swallowed:                                        # a swallowed particle stays where it is (exchange.hpp:170)
        for r, T in enumerate(tasks):
            k = int(np.flatnonzero(targets[r][:T[1]] != r)[0])
            T[0]["Flags"][k] |= 2
    gout, togo, last, _, _ = gpu_exchange(ctx, tasks, targets, ntask)
    objs = [oex.Task(T[0].copy(), T[1], [None if s is None else s.copy() for s in T[2]], T[3]) for T in tasks]
    olists, otogo, _ = oex.domain_exchange(objs, targets)
    for r in range(ntask):
        assert np.array_equal(togo[r], otogo[r]) and last[r] == (len(olists[r]), len(olists[r]))
        gP, gn, gS, gsz = gout[r]
        o = objs[r]
        assert gn == o.numpart and gsz == o.slot_size
        assert same_records(gP[:gn], o.parts[:gn])
        for t in range(6):
            if o.slots[t] is not None:
                assert same_records(gS[t][:gsz[t]], o.slots[t][:gsz[t]]), (r, t)
    return gout, tot


@pytest.mark.parametrize("ntask", [1, 2, 3, 5])
@pytest.mark.parametrize("ntype", [[8] * 6, [8, 0, 8, 0, 8, 0], [40, 13, 0, 7, 25, 3]])
def test_exchange_equals_reference_loop(ctx, ntask, ntype):
    gout, tot = both(ctx, ntask, ntype, fx.layout_id_mod)
    fx.check_after(gout, ntask, tot)


def test_exchange_at_scale(ctx):
    """44 000 particles per task, four tasks: the sorts and the word-wise copies beyond toy sizes (several workgroups per task and
    type, slot records of 72 / 176 / 248 bytes), still field for field against the serial loop"""
    gout, tot = both(ctx, 4, [20000, 6500, 0, 3500, 12500, 1500], fx.layout_id_mod, garbage=True, maxpart=100000)
    fx.check_after(gout, 4, tot)
    assert all(g[1] > 40000 for g in gout)


def test_exchange_with_garbage_and_swallowed(ctx):
    gout, tot = both(ctx, 4, [8] * 6, fx.layout_id_mod, garbage=True)
    fx.check_after(gout, 4, tot)
    both(ctx, 3, [8] * 6, fx.layout_id_mod, swallowed=True)


def test_exchange_uneven(ctx):
    gout, tot = both(ctx, 4, [8] * 6, fx.layout_uneven)
    fx.check_after(gout, 4, tot, uneven=True)
    assert gout[0][3][0] == fx.NUMPART1 * 4


def test_exchange_batches_and_bad_target(ctx):
    """maxlast: only the first entries of the list go in this round (find_iter_space); a target outside [0, NTask) is refused"""
    ntask = 3
    tasks = [list(fx.setup_task(r, ntask, [8] * 6)) for r in range(ntask)]
    targets = [fx.layout_id_mod(T[0], T[1], ntask) for T in tasks]
    L = layout_struct()
    d_parts = dev(tasks[0][0])
    d_tgt = torch.from_numpy(targets[0].copy()).to(DEV)
    tg = (capi.ExchangeEntry * ntask)()
    nex, la = C.c_int64(), C.c_int64()
    capi.check(capi.hip.shq_exchange_plan(ctx.h, C.byref(L), d_parts.data_ptr(), tasks[0][1], d_tgt.data_ptr(), 0, ntask, 5, C.byref(nex), C.byref(la), tg))
    lst = oex.build_exchange_list(oex.Task(*tasks[0]), targets[0], 0)
    assert nex.value == len(lst) and la.value == 5
    want = oex.counts(oex.Task(*tasks[0]), lst[:5], targets[0], ntask)
    assert [[tg[t].base] + list(tg[t].slots) for t in range(ntask)] == want.tolist()
    targets[0][3] = 7
    d_tgt = torch.from_numpy(targets[0].copy()).to(DEV)
    assert capi.hip.shq_exchange_plan(ctx.h, C.byref(L), d_parts.data_ptr(), tasks[0][1], d_tgt.data_ptr(), 0, ntask, 0, C.byref(nex), C.byref(la), tg) != 0


def test_slots_gc_reference_fixture(ctx):
    """tests/test_slotsmanager.cpp:65-85 on the device, against the oracle field for field"""
    P, numpart, slots, slot_size = fx.setup_task(0, 1, [128] * 6)
    for i in range(6):
        k = 128 * i
        P["Flags"][k] |= 1
        t = int(P["Type"][k])
        if slots[t] is not None:
            slots[t]["ReverseLink"][P["PI"][k]] = len(P) + 100
    L = layout_struct()
    d_parts = dev(P)
    d_slots = [None if s is None else dev(s) for s in slots]
    sp = (C.c_void_p * 6)(*[None if s is None else s.data_ptr() for s in d_slots])
    n = C.c_int64(numpart)
    sz = (C.c_int64 * 6)(*slot_size)
    compact = (C.c_int * 6)(*[1] * 6)
    capi.check(capi.hip.shq_slots_gc(ctx.h, C.byref(L), d_parts.data_ptr(), C.byref(n), len(P), sp, sz, compact))
    T = oex.Task(P, numpart, slots, slot_size)
    oex.slots_gc(T, [1] * 6)
    assert n.value == T.numpart == 127 * 6 and list(sz) == T.slot_size
    gP = d_parts.cpu().numpy().view(capi.PARTICLE_DTYPE)
    assert same_records(gP[:n.value], T.parts[:T.numpart])
    for t in range(6):
        if slots[t] is not None:
            gS = d_slots[t].cpu().numpy().view(fx.SLOT_DTYPES[t])
            assert same_records(gS[:sz[t]], T.slots[t][:T.slot_size[t]]), t
    fx.check_after([(gP, n.value, [None if s is None else d_slots[t].cpu().numpy().view(fx.SLOT_DTYPES[t]) for t, s in enumerate(slots)], list(sz))], 1, 127 * 6)


def test_exchange_in_batches_with_gc_equals_oracle(ctx):
    """ExchangePlan::domain_exchange with a cap of 5 list entries per round: plan / pack / gc / receive / unpack per round on the
    device, the same rounds by the oracle; all arrays equal after every round"""
    ntask, maxlast = 3, 5
    L = layout_struct()
    esz = L.part_elsize
    host = [list(fx.setup_task(r, ntask, [8] * 6, maxpart=96)) for r in range(ntask)]
    otasks = [oex.Task(h[0].copy(), h[1], [None if s is None else s.copy() for s in h[2]], list(h[3])) for h in host]
    d_parts = [dev(h[0]) for h in host]
    d_slots = [[None if s is None else dev(s) for s in h[2]] for h in host]
    numpart = [h[1] for h in host]
    slot_size = [list(h[3]) for h in host]
    rounds = 0
    while True:
        rounds += 1
        assert rounds < 50
        cur = [d_parts[r].cpu().numpy().view(capi.PARTICLE_DTYPE) for r in range(ntask)]
        targets = [fx.layout_id_mod(cur[r], numpart[r], ntask) for r in range(ntask)]
        d_tgt = [torch.from_numpy(t.copy()).to(DEV) for t in targets]
        togo = np.zeros((ntask, ntask, 7), dtype=np.int64)
        nex, last, partbuf, slotbuf = [], [], [], []
        for r in range(ntask):
            tg = (capi.ExchangeEntry * ntask)()
            a, b = C.c_int64(), C.c_int64()
            capi.check(capi.hip.shq_exchange_plan(ctx.h, C.byref(L), d_parts[r].data_ptr(), numpart[r], d_tgt[r].data_ptr(), r, ntask, maxlast, C.byref(a), C.byref(b), tg))
            nex.append(a.value)
            last.append(b.value)
            for t in range(ntask):
                togo[r, t, 0] = tg[t].base
                togo[r, t, 1:] = list(tg[t].slots)
            off = oex.offsets(togo[r])
            pb = torch.zeros(max(int(togo[r][:, 0].sum()), 1) * esz, dtype=torch.uint8, device=DEV)
            sb = [None if fx.SLOT_DTYPES[t] is None else torch.zeros(max(int(togo[r][:, 1 + t].sum()), 1) * fx.SLOT_DTYPES[t].itemsize, dtype=torch.uint8, device=DEV)
                  for t in range(6)]
            sp = (C.c_void_p * 6)(*[None if s is None else s.data_ptr() for s in d_slots[r]])
            bp = (C.c_void_p * 6)(*[None if s is None else s.data_ptr() for s in sb])
            capi.check(capi.hip.shq_exchange_pack(ctx.h, C.byref(L), d_parts[r].data_ptr(), sp, 96, entries(off), ntask, pb.data_ptr(), bp))
            partbuf.append(pb)
            slotbuf.append(sb)
        if not any(nex):
            break
        toget = np.stack([np.stack([togo[src][r] for src in range(ntask)]) for r in range(ntask)])
        shall_gc = any(last[r] < nex[r] or numpart[r] + int(toget[r][:, 0].sum()) > 96 for r in range(ntask))
        if shall_gc:
            compact = [0] * 6
            for r in range(ntask):
                tmp = oex.Task(cur[r], numpart[r], host[r][2], slot_size[r])     # only sizes and capacities are read
                c = oex.shall_we_compact_slots(tmp, toget[r].sum(axis=0), togo[r].sum(axis=0))
                compact = [x | y for x, y in zip(compact, c)]
            for r in range(ntask):
                sp = (C.c_void_p * 6)(*[None if s is None else s.data_ptr() for s in d_slots[r]])
                n = C.c_int64(numpart[r])
                sz = (C.c_int64 * 6)(*slot_size[r])
                capi.check(capi.hip.shq_slots_gc(ctx.h, C.byref(L), d_parts[r].data_ptr(), C.byref(n), 96, sp, sz, (C.c_int * 6)(*compact)))
                numpart[r], slot_size[r] = n.value, list(sz)
        for r in range(ntask):
            goff = oex.offsets(toget[r])
            for src in range(ntask):
                soff = oex.offsets(togo[src])
                nb = int(toget[r][src, 0])
                a = (numpart[r] + int(goff[src, 0])) * esz
                d_parts[r][a:a + nb * esz] = partbuf[src][int(soff[r, 0]) * esz:(int(soff[r, 0]) + nb) * esz]
                for t in range(6):
                    if fx.SLOT_DTYPES[t] is None:
                        continue
                    ssz = fx.SLOT_DTYPES[t].itemsize
                    ns = int(toget[r][src, 1 + t])
                    a = (slot_size[r][t] + int(goff[src, 1 + t])) * ssz
                    d_slots[r][t][a:a + ns * ssz] = slotbuf[src][t][int(soff[r, 1 + t]) * ssz:(int(soff[r, 1 + t]) + ns) * ssz]
            torch.cuda.synchronize()
            so = (C.c_int64 * 6)(*slot_size[r])
            capi.check(capi.hip.shq_exchange_unpack(ctx.h, C.byref(L), d_parts[r].data_ptr(), numpart[r], so, entries(toget[r]), entries(goff), ntask))
            numpart[r] += int(toget[r][:, 0].sum())
            for t in range(6):
                if fx.SLOT_DTYPES[t] is not None:
                    slot_size[r][t] += int(toget[r][:, 1 + t].sum())
        if not any(last[r] < nex[r] for r in range(ntask)):
            break
    lay = [lambda P, n, nt=ntask: fx.layout_id_mod(P, n, nt)] * ntask
    oiters = oex.domain_exchange_batched(otasks, lay, maxlast)
    assert rounds == oiters >= 4
    out = []
    for r in range(ntask):
        o = otasks[r]
        gP = d_parts[r].cpu().numpy().view(capi.PARTICLE_DTYPE)
        assert numpart[r] == o.numpart and slot_size[r] == o.slot_size
        assert same_records(gP[:numpart[r]], o.parts[:o.numpart])
        gS = [None if s is None else d_slots[r][t].cpu().numpy().view(fx.SLOT_DTYPES[t]) for t, s in enumerate(host[r][2])]
        for t in range(6):
            if gS[t] is not None:
                assert same_records(gS[t][:slot_size[r][t]], o.slots[t][:o.slot_size[t]]), (r, t)
        out.append((gP, numpart[r], gS, slot_size[r]))
    fx.check_after(out, ntask, ntask * 48)


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libpeano_ref.so")), reason="needs the reference's peano key (make -C oracle ref)")
def test_slots_gc_sorted_reference_fixture(ctx):
    """tests/test_slotsmanager.cpp:87-115 on the device with the reference's own keys, against the oracle field for field"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_exchange_cpu as tc
    P, numpart, slots, slot_size = tc.gc_sorted_setup(np.random.default_rng(5))
    P["Pos"][3] = P["Pos"][2]                     # two particles in one key cell: the order of equals
    keys = tc._ref_keys(P, numpart, 25000.0)
    L = layout_struct()
    d_parts = dev(P)
    d_slots = [None if s is None else dev(s) for s in slots]
    d_keys = torch.from_numpy(keys.view(np.int64).copy()).to(DEV)
    sp = (C.c_void_p * 6)(*[None if s is None else s.data_ptr() for s in d_slots])
    n = C.c_int64(numpart)
    sz = (C.c_int64 * 6)(*slot_size)
    capi.check(capi.hip.shq_slots_gc_sorted(ctx.h, C.byref(L), d_parts.data_ptr(), C.byref(n), len(P), sp, sz, d_keys.data_ptr()))
    T = oex.Task(P, numpart, slots, slot_size)
    oex.slots_gc_sorted(T, keys)
    assert n.value == T.numpart and list(sz) == T.slot_size
    gP = d_parts.cpu().numpy().view(capi.PARTICLE_DTYPE)
    assert same_records(gP[:n.value], T.parts[:T.numpart])
    gS = [None if s is None else d_slots[t].cpu().numpy().view(fx.SLOT_DTYPES[t]) for t, s in enumerate(slots)]
    for t in range(6):
        if gS[t] is not None:
            assert same_records(gS[t][:sz[t]], T.slots[t][:T.slot_size[t]]), t
    tc.check_gc_sorted(gP, n.value, gS, list(sz))


# ---- slots_split_particle / slots_convert for lists ----------------------------------------------------------------------------
def spawn_layout():
    f = capi.PARTICLE_DTYPE.fields
    S = capi.SpawnLayout()
    S.off_id, S.off_mass, S.generation_shift = f["ID"][1], f["Mass"][1], 4
    return S


class DevTask:
    def __init__(self, T):
        self.P = dev(T.parts)
        self.S = [None if s is None else dev(s) for s in T.slots]
        self.sp = (C.c_void_p * 6)(*[None if s is None else s.data_ptr() for s in self.S])
        self.n = C.c_int64(T.numpart)
        self.sz = (C.c_int64 * 6)(*T.slot_size)
        self.maxpart = T.maxpart
        self.maxsize = (C.c_int64 * 6)(*[0 if s is None else len(s) for s in T.slots])

    def split(self, ctx, parents, masses, want_children=True):
        p = torch.from_numpy(np.asarray(parents, dtype=np.int32)).to(DEV)
        m = torch.from_numpy(np.asarray(masses, dtype=np.float64)).to(DEV)
        ch = torch.full((max(len(parents), 1),), -7, dtype=torch.int32, device=DEV)
        capi.check(capi.hip.shq_slots_split_particles(ctx.h, C.byref(layout_struct()), C.byref(spawn_layout()), self.P.data_ptr(), C.byref(self.n), self.maxpart,
                                                      p.data_ptr(), m.data_ptr(), len(parents), ch.data_ptr() if want_children else None))
        return ch.cpu().numpy()[:len(parents)]

    def convert(self, ctx, index, ptype, maxsize=None):
        ix = torch.from_numpy(np.asarray(index, dtype=np.int32)).to(DEV)
        ms = self.maxsize if maxsize is None else (C.c_int64 * 6)(*maxsize)
        capi.check(capi.hip.shq_slots_convert(ctx.h, C.byref(layout_struct()), self.P.data_ptr(), self.n.value, self.maxpart, self.sp, self.sz, ms, ix.data_ptr(),
                                              len(index), ptype))

    def equals(self, T):
        assert self.n.value == T.numpart and list(self.sz) == T.slot_size
        gP = self.P.cpu().numpy().view(capi.PARTICLE_DTYPE)
        assert same_records(gP[:T.numpart], T.parts[:T.numpart])
        for t in range(6):
            if T.slots[t] is not None:
                gS = self.S[t].cpu().numpy().view(fx.SLOT_DTYPES[t])
                # poisoned slots hold NaN-patterned floats: compare them as bytes, field by field
                for f in gS.dtype.names:
                    a = np.ascontiguousarray(gS[f][:T.slot_size[t]])
                    b = np.ascontiguousarray(T.slots[t][f][:T.slot_size[t]])
                    assert a.tobytes() == b.tobytes(), (t, f)


def fork_task():
    P, numpart, slots, slot_size = fx.setup_task(0, 1, [128] * 6)
    return oex.Task(P, numpart, slots, slot_size)


def test_slots_fork_reference_fixture(ctx):
    """tests/test_slotsmanager.cpp:268-286 through the list calls: six parents split, then converted to their own types"""
    T = fork_task()
    D = DevTask(T)
    parents = [128 * i for i in range(6)]
    ch = D.split(ctx, parents, [0.0] * 6)
    assert ch.tolist() == [768 + i for i in range(6)]
    for i in range(6):
        D.convert(ctx, [128 * i], i)
        oex.slots_split_particle(T, 128 * i, 0.0)
    for i in range(6):
        oex.slots_convert(T, 128 * i, i)
    assert D.n.value == 129 * 6 and [D.sz[t] for t in (0, 4, 5)] == [129] * 3
    D.equals(T)


def test_slots_convert_reference_fixture(ctx):
    """tests/test_slotsmanager.cpp:288-304, then the collection that sweeps the abandoned slots"""
    T = fork_task()
    D = DevTask(T)
    for i in range(6):
        D.convert(ctx, [128 * i], i)
        oex.slots_convert(T, 128 * i, i)
    assert D.n.value == 128 * 6 and [D.sz[t] for t in (0, 4, 5)] == [129] * 3
    D.equals(T)
    compact = (C.c_int * 6)(*[1] * 6)
    capi.check(capi.hip.shq_slots_gc(ctx.h, C.byref(layout_struct()), D.P.data_ptr(), C.byref(D.n), D.maxpart, D.sp, D.sz, compact))
    oex.slots_gc(T, [1] * 6)
    assert [D.sz[t] for t in (0, 4, 5)] == [128] * 3
    D.equals(T)


def test_star_formation_pattern_equals_serial_loop(ctx):
    """the shape of sfr_eff.cpp:344-372: some gas particles turn into stars whole, others split a star off first; then black-hole
    seeds from gas (blackhole.cpp:1040) and a conversion to a type without slots.  List entry k = the k-th call of the serial loop."""
    rng = np.random.default_rng(8)
    P, numpart, slots, slot_size = fx.setup_task(0, 1, [300, 20, 0, 5, 40, 6], maxpart=1024, rng=rng)
    T = oex.Task(P, numpart, slots, slot_size)
    T.parts["Mass"][:numpart] = rng.uniform(0.5, 2.0, numpart).astype(np.float32)
    T.parts["Flags"][:numpart] |= (rng.integers(0, 16, numpart).astype(np.uint8) << 4)      # generations 0..15, wrap included
    D = DevTask(T)
    gas = rng.permutation(300)
    whole, splitters, seeds, drop = gas[:60], gas[60:130], gas[130:137], gas[137:140]
    cm = rng.uniform(0.05, 0.4, len(splitters))
    ch = D.split(ctx, splitters, cm)
    och = [oex.slots_split_particle(T, int(p), float(m)) for p, m in zip(splitters, cm)]
    assert ch.tolist() == och
    newstars = np.concatenate([whole, ch])                          # NewStars: converted parents and spawned children alike
    D.convert(ctx, newstars, 4)
    for s in newstars:
        oex.slots_convert(T, int(s), 4)
    D.convert(ctx, seeds, 5)
    for s in seeds:
        oex.slots_convert(T, int(s), 5)
    D.convert(ctx, drop, 1)
    for s in drop:
        oex.slots_convert(T, int(s), 1)
    D.equals(T)
    assert D.n.value == numpart + len(splitters) and D.sz[4] == 40 + len(newstars) and D.sz[5] == 6 + len(seeds) and D.sz[0] == 300
    # the children carry PI = -1 until converted: the spawned stars were converted, so no live particle has PI -1 in a slotted type
    gP = D.P.cpu().numpy().view(capi.PARTICLE_DTYPE)[:D.n.value]
    assert (gP["PI"][np.isin(gP["Type"], (0, 4, 5))] >= 0).all()
    # an empty list is fine, children may be left unreported
    D.split(ctx, [], [])
    D.convert(ctx, [], 4)
    D.split(ctx, [int(gas[200])], [0.01], want_children=False)
    oex.slots_split_particle(T, int(gas[200]), 0.01)
    D.equals(T)


def test_slots_split_convert_errors(ctx):
    T = fork_task()
    D = DevTask(T)
    with pytest.raises(sq.ShqError):                     # no space left: 768 + 300 > 1024, nothing touched
        D.split(ctx, list(range(300)), [0.1] * 300)
    with pytest.raises(sq.ShqError):                     # more stars than reserved slots
        D.convert(ctx, [0, 1, 2], 4, maxsize=[1024, 0, 0, 0, 130, 1024])
    D.equals(T)
    with pytest.raises(sq.ShqError):
        D.convert(ctx, [5], 6)
    with pytest.raises(sq.ShqError):
        D.convert(ctx, [5000], 4)
    with pytest.raises(sq.ShqError):
        D.split(ctx, [-1], [0.1])


# ---- make_particle_star / blackhole_make_one for lists ----------------------------------------------------------------------
def star_layout():
    st, sp = capi.STAR_DTYPE.fields, capi.SPH_DTYPE.fields
    L = capi.StarSpawnLayout()
    L.star_formationtime, L.star_lastenrichmentmyr, L.star_totalmassreturned = st["FormationTime"][1], st["LastEnrichmentMyr"][1], st["TotalMassReturned"][1]
    L.star_birthdensity, L.star_vdisp, L.star_metallicity, L.star_metals = st["BirthDensity"][1], st["VDisp"][1], st["Metallicity"][1], st["Metals"][1]
    L.sph_density, L.sph_vdisp, L.sph_metallicity, L.sph_metals = sp["Density"][1], sp["VDisp"][1], sp["Metallicity"][1], sp["Metals"][1]
    L.nmetals = 9
    return L


def bh_seed_layout():
    b, p = capi.BH_DTYPE.fields, capi.PARTICLE_DTYPE.fields
    L = capi.BhSeedLayout()
    for key, name in (("bh_mass", "Mass"), ("bh_mseed", "Mseed"), ("bh_mdot", "Mdot"), ("bh_formationtime", "FormationTime"), ("bh_swallowid", "SwallowID"),
                      ("bh_density", "Density"), ("bh_timebindynfric", "TimeBinDynFric"), ("bh_minpotpos", "MinPotPos"), ("bh_dfaccel", "DFAccel"),
                      ("bh_df_surroundingvel", "DF_SurroundingVel"), ("bh_dragaccel", "DragAccel"), ("bh_df_surroundingrmsvel", "DF_SurroundingRmsVel"),
                      ("bh_df_surroundingdensity", "DF_SurroundingDensity"), ("bh_jumptominpot", "JumpToMinPot"), ("bh_countprogs", "CountProgs"),
                      ("bh_mtrack", "Mtrack"), ("bh_kineticfdbkenergy", "KineticFdbkEnergy"), ("bh_vdisp", "VDisp")):
        setattr(L, key, b[name][1])
    L.part_pos, L.part_mass, L.part_timebin_hydro = p["Pos"][1], p["Mass"][1], p["TimeBinHydro"][1]
    return L


def gas_task(rng, ngas=200):
    P, numpart, slots, slot_size = fx.setup_task(0, 1, [ngas, 30, 0, 0, 25, 4], maxpart=1024, rng=rng)
    T = oex.Task(P, numpart, slots, slot_size)
    S = T.slots[0]
    S["Density"][:ngas] = rng.uniform(0.1, 50, ngas)
    S["VDisp"][:ngas] = rng.uniform(1, 300, ngas)
    S["Metallicity"][:ngas] = rng.uniform(0, 0.05, ngas)
    S["Metals"][:ngas] = rng.uniform(0, 1e-3, (ngas, 9)).astype(np.float32)
    T.parts["Mass"][:numpart] = rng.uniform(0.5, 2.0, numpart).astype(np.float32)
    T.parts["TimeBinHydro"][:numpart] = rng.integers(20, 40, numpart)
    T.parts["Pos"][:numpart] = rng.random((numpart, 3)) * 1000
    return T


def test_make_particle_stars_equals_serial_loop(ctx):
    """sfr_eff.cpp:344-372: NewStars / NewParents with converted and spawned stars mixed, placement firststarslot + i; the star slots
    carry the PARENT's gas fields, also for a converted parent whose own PI has been overwritten by then"""
    rng = np.random.default_rng(31)
    T = gas_task(rng)
    D = DevTask(T)
    gas = rng.permutation(200)
    converted, splitters = gas[:40], gas[40:90]
    cm_ = rng.uniform(0.05, 0.3, len(splitters))
    ch = D.split(ctx, splitters, cm_)
    for p, m in zip(splitters, cm_):
        oex.slots_split_particle(T, int(p), float(m))
    # the reference's lists are in active-particle order: converted and spawned entries interleave
    order = rng.permutation(len(converted) + len(splitters))
    children = np.concatenate([converted, ch])[order].astype(np.int32)
    parents = np.concatenate([converted, splitters])[order].astype(np.int32)
    dch, dpa = torch.from_numpy(children).to(DEV), torch.from_numpy(parents).to(DEV)
    first = T.slot_size[4]
    capi.check(capi.hip.shq_make_particle_stars(ctx.h, C.byref(layout_struct()), C.byref(star_layout()), D.P.data_ptr(), D.n.value, D.maxpart, D.sp, D.sz, D.maxsize,
                                                dch.data_ptr(), dpa.data_ptr(), len(children), 0.3125))
    for i, (c, p) in enumerate(zip(children, parents)):
        oex.make_particle_star(T, int(c), int(p), first + i, 0.3125)
    T.slot_size[4] += len(children)
    D.equals(T)
    gS = D.S[4].cpu().numpy().view(capi.STAR_DTYPE)
    assert (gS["FormationTime"][first:first + len(children)] == np.float32(0.3125)).all() and (gS["BirthDensity"][first:first + len(children)] > 0).all()
    # a parent that is not gas (any more): refused before anything changes
    with pytest.raises(sq.ShqError):
        capi.check(capi.hip.shq_make_particle_stars(ctx.h, C.byref(layout_struct()), C.byref(star_layout()), D.P.data_ptr(), D.n.value, D.maxpart, D.sp, D.sz, D.maxsize,
                                                    dch.data_ptr(), dpa.data_ptr(), 1, 0.5))
    few = (C.c_int64 * 6)(1024, 0, 0, 0, D.sz[4] + 1, 1024)
    two = torch.from_numpy(gas[100:102].astype(np.int32)).to(DEV)
    with pytest.raises(sq.ShqError):                     # sfr_reserve_slots was not called
        capi.check(capi.hip.shq_make_particle_stars(ctx.h, C.byref(layout_struct()), C.byref(star_layout()), D.P.data_ptr(), D.n.value, D.maxpart, D.sp, D.sz, few,
                                                    two.data_ptr(), two.data_ptr(), 2, 0.5))
    D.equals(T)


@pytest.mark.parametrize("dynmass", [0.0, 7.5])
def test_blackhole_make_seeds_equals_serial_loop(ctx, dynmass):
    """blackhole.cpp:1029-1088 for a list of seeds, with and without SeedBHDynMass"""
    rng = np.random.default_rng(41)
    T = gas_task(rng)
    D = DevTask(T)
    seeds = rng.permutation(200)[:9].astype(np.int32)
    seedmass = rng.uniform(1e-5, 1e-3, len(seeds))
    ds, dm = torch.from_numpy(seeds).to(DEV), torch.from_numpy(seedmass).to(DEV)
    capi.check(capi.hip.shq_blackhole_make_seeds(ctx.h, C.byref(layout_struct()), C.byref(bh_seed_layout()), D.P.data_ptr(), D.n.value, D.maxpart, D.sp, D.sz, D.maxsize,
                                                 ds.data_ptr(), dm.data_ptr(), len(seeds), 0.125, dynmass))
    for s, m in zip(seeds, seedmass):
        oex.blackhole_make_one(T, int(s), 0.125, float(m), dynmass)
    D.equals(T)
    gP = D.P.cpu().numpy().view(capi.PARTICLE_DTYPE)
    gB = D.S[5].cpu().numpy().view(capi.BH_DTYPE)
    assert (gP["Type"][seeds] == 5).all() and np.array_equal(gB["MinPotPos"][gP["PI"][seeds]], gP["Pos"][seeds])
    assert (gB["Mtrack"][gP["PI"][seeds]] == -1).all() if dynmass == 0 else (gP["Mass"][seeds] == np.float32(dynmass)).all()
    with pytest.raises(sq.ShqError):                     # "Only Gas turns into blackholes": the first seed is a black hole now
        capi.check(capi.hip.shq_blackhole_make_seeds(ctx.h, C.byref(layout_struct()), C.byref(bh_seed_layout()), D.P.data_ptr(), D.n.value, D.maxpart, D.sp, D.sz,
                                                     D.maxsize, ds.data_ptr(), dm.data_ptr(), 1, 0.125, dynmass))
    D.equals(T)
