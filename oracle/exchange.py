"""CPU restatement (numpy) of the reference's particle exchange between tasks, all tasks in one process.

TEST INFRASTRUCTURE ONLY.  Follows /root/reference/libgadget/exchange.hpp:
  build_exchange_list      :158-176   particles whose layoutfunc names another task, in index order, garbage / swallowed excluded
  build_export_buffer      :178-240   toGo / toGet per (task, type), offsets in task order
  exchange_once            :334-535   pack in list order (slot first, then the base record), slots_mark_garbage on the source,
                                       received particles appended after NumPart in source-task order, slots appended per type,
                                       PI of every received particle renumbered in arrival order
  slots_mark_garbage       slotsmanager.cpp:590-599
  slots_split_particle, slots_convert   slotsmanager.cpp:27-126
  slots_gc                 slotsmanager.cpp:132-370, with shall_we_compact_slots (exchange.hpp:300-317): the collection the
                                       exchange runs between pack and receive when a round is capped or memory is short
Pinned by the four cases of the reference's tests/test_exchange.cpp and by test_slots_gc, test_slots_gc_sorted, test_slots_fork and
test_slots_convert of tests/test_slotsmanager.cpp
(tests/test_exchange_cpu.py)."""
import numpy as np


class Task:
    """one task's particle and slot arrays (structured numpy arrays with at least Flags, Type, PI / ReverseLink)"""

    def __init__(self, parts, numpart, slots, slot_size):
        self.parts, self.numpart = parts, int(numpart)
        self.slots = slots                      # list of 6 arrays or None (type not enabled)
        self.slot_size = [int(x) for x in slot_size]

    @property
    def maxpart(self):
        return len(self.parts)


def build_exchange_list(task, target, thistask):
    P = task.parts[:task.numpart]
    dead = (P["Flags"] & 3) != 0
    t = np.asarray(target[:task.numpart])
    return np.flatnonzero(~dead & (t != thistask) & (t >= 0)).astype(np.int64)


def counts(task, lst, target, ntask):
    """toGo[target] = (base, slots[6])"""
    togo = np.zeros((ntask, 7), dtype=np.int64)
    for i in lst:
        togo[target[i], 0] += 1
        togo[target[i], 1 + int(task.parts["Type"][i])] += 1
    return togo


def offsets(c):
    """exclusive prefix over tasks, exchange.hpp:206-224"""
    off = np.zeros_like(c)
    off[1:] = np.cumsum(c[:-1], axis=0)
    return off


def domain_exchange(tasks, targets):
    """one iteration of ExchangePlan::domain_exchange for all tasks at once (enough room everywhere, so `last == nexchange`).
    tasks: list of Task, targets: list of per-particle target arrays.  Modifies the tasks in place."""
    ntask = len(tasks)
    lists = [build_exchange_list(tasks[r], targets[r], r) for r in range(ntask)]
    togo = [counts(tasks[r], lists[r], targets[r], ntask) for r in range(ntask)]
    toget = [np.stack([togo[src][r] for src in range(ntask)]) for r in range(ntask)]          # MPI_Alltoall of the entries
    partbuf, slotbuf = [], []
    for r in range(ntask):
        T = tasks[r]
        off = offsets(togo[r])
        pb = np.zeros(int(togo[r][:, 0].sum()), dtype=T.parts.dtype)
        sb = [None if T.slots[t] is None else np.zeros(int(togo[r][:, 1 + t].sum()), dtype=T.slots[t].dtype) for t in range(6)]
        ptr = np.zeros((ntask, 7), dtype=np.int64)
        for i in lists[r]:
            tgt = int(targets[r][i])
            ty = int(T.parts["Type"][i])
            bufpi = ptr[tgt, 1 + ty]
            ptr[tgt, 1 + ty] += 1
            if T.slots[ty] is not None:
                sb[ty][bufpi + off[tgt, 1 + ty]] = T.slots[ty][T.parts["PI"][i]]
            pb[off[tgt, 0] + ptr[tgt, 0]] = T.parts[i]
            ptr[tgt, 0] += 1
            # slots_mark_garbage
            T.parts["Flags"][i] |= 1
            if T.slots[ty] is not None:
                T.slots[ty]["ReverseLink"][T.parts["PI"][i]] = T.maxpart + 100
        partbuf.append(pb)
        slotbuf.append(sb)
    for r in range(ntask):
        T = tasks[r]
        goff = offsets(toget[r])
        newnum = T.numpart + int(toget[r][:, 0].sum())
        assert newnum <= T.maxpart, "NumPart > MaxPart"
        for src in range(ntask):
            soff = offsets(togo[src])
            nb = int(toget[r][src, 0])
            T.parts[T.numpart + goff[src, 0]:T.numpart + goff[src, 0] + nb] = partbuf[src][soff[r, 0]:soff[r, 0] + nb]
            for t in range(6):
                if T.slots[t] is None:
                    continue
                ns = int(toget[r][src, 1 + t])
                assert T.slot_size[t] + goff[src, 1 + t] + ns <= len(T.slots[t]), "slot array too small"
                T.slots[t][T.slot_size[t] + goff[src, 1 + t]:T.slot_size[t] + goff[src, 1 + t] + ns] = slotbuf[src][t][soff[r, 1 + t]:soff[r, 1 + t] + ns]
        # PI of the arrivals, exchange.hpp:483-511
        for src in range(ntask):
            newpi = [T.slot_size[t] + int(goff[src, 1 + t]) for t in range(6)]
            for i in range(T.numpart + int(goff[src, 0]), T.numpart + int(goff[src, 0]) + int(toget[r][src, 0])):
                ty = int(T.parts["Type"][i])
                T.parts["PI"][i] = newpi[ty]
                newpi[ty] += 1
        T.numpart = newnum
        for t in range(6):
            if T.slots[t] is not None:
                T.slot_size[t] += int(toget[r][:, 1 + t].sum())
    return lists, togo, toget


def slots_gc(task, compact):
    """slots_gc, slotsmanager.cpp:132-370: slots_gc_base (garbage particles squeezed out, order kept), slots_gc_mark, then per
    compacted type slots_gc_sweep + slots_gc_collect.  Modifies the task in place."""
    P = task.parts
    n = task.numpart
    keep = np.flatnonzero((P["Flags"][:n] & 1) == 0)
    P[:len(keep)] = P[keep]
    task.numpart = n = len(keep)
    if not any(compact):
        return
    invalid = task.maxpart + 100
    for i in range(n):                                   # slots_gc_mark (no garbage is left after the base pass)
        t = int(P["Type"][i])
        if task.slots[t] is None:
            continue
        pi = int(P["PI"][i])
        assert 0 <= pi < task.slot_size[t]
        task.slots[t]["ReverseLink"][pi] = invalid if (P["Flags"][i] & 1) else i
    for t in range(6):
        if not compact[t] or task.slots[t] is None:
            continue
        S = task.slots[t]
        used = task.slot_size[t]
        live = np.flatnonzero(S["ReverseLink"][:used] <= task.maxpart)
        S[:len(live)] = S[live]
        task.slot_size[t] = len(live)
        P["PI"][S["ReverseLink"][:len(live)]] = np.arange(len(live))


def shall_we_compact_slots(task, toget_sum, togo_sum):
    """exchange.hpp:300-317 for one task (the caller ORs over tasks)"""
    c = [0] * 6
    for t in range(6):
        if task.slots[t] is None:
            continue
        if task.slot_size[t] + toget_sum[1 + t] > 0.95 * len(task.slots[t]):
            c[t] = 1
        if togo_sum[1 + t] > 0.1 * task.slot_size[t]:
            c[t] = 1
    return c


def domain_exchange_batched(tasks, layouts, maxlast, maxiter=10000):
    """ExchangePlan::domain_exchange (exchange.hpp:88-153) with a cap of `maxlast` list entries per task and iteration
    (find_iter_space), the garbage collection between pack and receive included (exchange_once :398-406).
    layouts: per task a function parts, numpart -> target array.  Returns the number of iterations."""
    ntask = len(tasks)
    it = 0
    while True:
        assert it < maxiter
        targets = [layouts[r](tasks[r].parts, tasks[r].numpart) for r in range(ntask)]
        lists = [build_exchange_list(tasks[r], targets[r], r) for r in range(ntask)]
        if not any(len(l) for l in lists):
            break
        lasts = [min(len(l), maxlast) for l in lists]
        sub = [lists[r][:lasts[r]] for r in range(ntask)]
        togo = [counts(tasks[r], sub[r], targets[r], ntask) for r in range(ntask)]
        toget = [np.stack([togo[src][r] for src in range(ntask)]) for r in range(ntask)]
        partbuf, slotbuf = [], []
        for r in range(ntask):
            T = tasks[r]
            off = offsets(togo[r])
            pb = np.zeros(int(togo[r][:, 0].sum()), dtype=T.parts.dtype)
            sb = [None if T.slots[t] is None else np.zeros(int(togo[r][:, 1 + t].sum()), dtype=T.slots[t].dtype) for t in range(6)]
            ptr = np.zeros((ntask, 7), dtype=np.int64)
            for i in sub[r]:
                tgt = int(targets[r][i])
                ty = int(T.parts["Type"][i])
                bufpi = ptr[tgt, 1 + ty]
                ptr[tgt, 1 + ty] += 1
                if T.slots[ty] is not None:
                    sb[ty][bufpi + off[tgt, 1 + ty]] = T.slots[ty][T.parts["PI"][i]]
                pb[off[tgt, 0] + ptr[tgt, 0]] = T.parts[i]
                ptr[tgt, 0] += 1
                T.parts["Flags"][i] |= 1
                if T.slots[ty] is not None:
                    T.slots[ty]["ReverseLink"][T.parts["PI"][i]] = T.maxpart + 100
            partbuf.append(pb)
            slotbuf.append(sb)
        shall_gc = any(lasts[r] < len(lists[r]) or tasks[r].numpart + int(toget[r][:, 0].sum()) > tasks[r].maxpart for r in range(ntask))
        if shall_gc:
            compact = [0] * 6
            for r in range(ntask):
                c = shall_we_compact_slots(tasks[r], toget[r].sum(axis=0), togo[r].sum(axis=0))
                compact = [a | b for a, b in zip(compact, c)]
            for r in range(ntask):
                slots_gc(tasks[r], compact)
        for r in range(ntask):
            T = tasks[r]
            goff = offsets(toget[r])
            newnum = T.numpart + int(toget[r][:, 0].sum())
            assert newnum <= T.maxpart
            for src in range(ntask):
                soff = offsets(togo[src])
                nb = int(toget[r][src, 0])
                T.parts[T.numpart + goff[src, 0]:T.numpart + goff[src, 0] + nb] = partbuf[src][soff[r, 0]:soff[r, 0] + nb]
                for t in range(6):
                    if T.slots[t] is None:
                        continue
                    ns = int(toget[r][src, 1 + t])
                    a = T.slot_size[t] + goff[src, 1 + t]
                    assert a + ns <= len(T.slots[t])
                    T.slots[t][a:a + ns] = slotbuf[src][t][soff[r, 1 + t]:soff[r, 1 + t] + ns]
            for src in range(ntask):
                newpi = [T.slot_size[t] + int(goff[src, 1 + t]) for t in range(6)]
                for i in range(T.numpart + int(goff[src, 0]), T.numpart + int(goff[src, 0]) + int(toget[r][src, 0])):
                    ty = int(T.parts["Type"][i])
                    T.parts["PI"][i] = newpi[ty]
                    newpi[ty] += 1
            T.numpart = newnum
            for t in range(6):
                if T.slots[t] is not None:
                    T.slot_size[t] += int(toget[r][:, 1 + t].sum())
        it += 1
        if not any(lasts[r] < len(lists[r]) for r in range(ntask)):
            break
    return it


def slots_gc_sorted(task, keys):
    """slots_gc_sorted, slotsmanager.cpp:417-510: keys[i] = PEANO(Pos[i]).  Equal (TypeKey, Key) pairs keep their order (a
    stable sort; the reference's is not, which only matters for particles in the same 2^-21 cell)."""
    P = task.parts
    n = task.numpart
    garbage = (P["Flags"][:n] & 1) != 0
    typekey = np.where(garbage, 255, P["Type"][:n].astype(np.int64))
    order = np.lexsort((np.asarray(keys[:n], dtype=np.uint64), typekey))
    P[:n] = P[:n][order]
    invalid = task.maxpart + 100
    for i in range(n):                                   # slots_gc_mark, garbage included
        t = int(P["Type"][i])
        if task.slots[t] is None:
            continue
        pi = int(P["PI"][i])
        assert 0 <= pi < task.slot_size[t]
        task.slots[t]["ReverseLink"][pi] = invalid if (P["Flags"][i] & 1) else i
    task.numpart = n - int(garbage.sum())
    for t in range(6):
        if task.slots[t] is None:
            continue
        S = task.slots[t]
        used = task.slot_size[t]
        so = np.argsort(S["ReverseLink"][:used], kind="stable")
        S[:used] = S[:used][so]
        live = int((S["ReverseLink"][:used] <= task.maxpart).sum())
        task.slot_size[t] = live
        P["PI"][S["ReverseLink"][:live]] = np.arange(live)


def slots_split_particle(task, parent, childmass):
    """slots_split_particle, slotsmanager.cpp:102-126: a new particle split off `parent` at index NumPart (serial: the atomic
    counter's order is the call order).  Generation is the 4-bit field in the upper half of the flag byte (it wraps at 16: the
    reference's "generation wrapped" test compares a 4-bit value with 256 and never fires).  Returns the child's index."""
    P = task.parts
    child = task.numpart
    if child >= task.maxpart:
        raise MemoryError("Tried to spawn: NumPart=%d MaxPart = %d. Sorry, no space left." % (child, task.maxpart))
    task.numpart += 1
    f = int(P["Flags"][parent])
    g = ((f >> 4) + 1) & 15
    P["Flags"][parent] = (f & 15) | (g << 4)
    P[child] = P[parent]
    P["ID"][child] = (int(P["ID"][parent]) & 0x00ffffffffffffff) + (g << 56)
    P["Mass"][child] = np.float32(childmass)
    P["Mass"][parent] = np.float32(np.float64(P["Mass"][parent]) - np.float64(childmass))
    P["PI"][child] = -1
    return child


def slots_convert(task, parent, ptype, placement=-1, maxsize=None):
    """slots_convert + slots_connect_new_slot, slotsmanager.cpp:27-84: the old slot becomes garbage (ReverseLink = MaxPart + 100),
    a new slot of ptype (at `placement`, or at the end of the array) is filled with the poison byte 101 and linked, Type = ptype"""
    P = task.parts
    oldtype, oldpi = int(P["Type"][parent]), int(P["PI"][parent])
    if oldpi >= 0 and task.slots[oldtype] is not None:
        task.slots[oldtype]["ReverseLink"][oldpi] = task.maxpart + 100
    if task.slots[ptype] is not None:
        newpi = placement
        if placement < 0:
            newpi = task.slot_size[ptype]
            task.slot_size[ptype] += 1
        cap = len(task.slots[ptype]) if maxsize is None else maxsize[ptype]
        if newpi >= cap:
            raise MemoryError("Tried to use non-allocated slot %d (> %d)" % (newpi, cap))
        task.slots[ptype][newpi:newpi + 1].view(np.uint8)[:] = 101
        P["PI"][parent] = newpi
    P["Type"][parent] = ptype
    return parent


NMETALS = 9


def make_particle_star(task, child, parent, placement, Time):
    """make_particle_star, sfr_eff.cpp:604-630 (no fixture in the reference's tests: a field-by-field restatement)"""
    P = task.parts
    if P["Type"][parent] != 0:
        raise ValueError("Only gas forms stars, what's wrong?")
    old = task.slots[0][int(P["PI"][parent])].copy()
    slots_convert(task, child, 4, placement)
    S = task.slots[4]
    pi = int(P["PI"][child])
    S["FormationTime"][pi] = np.float32(Time)
    S["LastEnrichmentMyr"][pi] = 0
    S["TotalMassReturned"][pi] = 0
    S["BirthDensity"][pi] = np.float32(old["Density"])
    S["VDisp"][pi] = np.float32(old["VDisp"])
    S["Metallicity"][pi] = old["Metallicity"]
    S["Metals"][pi] = old["Metals"]


def blackhole_make_one(task, index, atime, seedmass, SeedBHDynMass):
    """blackhole_make_one, blackhole.cpp:1029-1088; seedmass is the caller's BHP.Mass (SeedBlackHoleMass or the power-law draw)"""
    P = task.parts
    if P["Type"][index] != 0:
        raise ValueError("Only Gas turns into blackholes, what's wrong?")
    slots_convert(task, index, 5, -1)
    B = task.slots[5]
    pi = int(P["PI"][index])
    B["Mass"][pi] = B["Mseed"][pi] = seedmass
    B["Mdot"][pi] = 0
    B["FormationTime"][pi] = atime
    B["SwallowID"][pi] = np.uint64(0xffffffffffffffff)
    B["Density"][pi] = 0
    B["TimeBinDynFric"][pi] = P["TimeBinHydro"][index]
    B["MinPotPos"][pi] = P["Pos"][index]
    B["DFAccel"][pi] = 0
    B["DF_SurroundingVel"][pi] = 0
    B["DragAccel"][pi] = 0
    B["DF_SurroundingRmsVel"][pi] = 0
    B["DF_SurroundingDensity"][pi] = 0
    B["JumpToMinPot"][pi] = 0
    B["CountProgs"][pi] = 1
    if SeedBHDynMass > 0:
        B["Mtrack"][pi] = np.float64(P["Mass"][index])
        P["Mass"][index] = np.float32(SeedBHDynMass)
    else:
        B["Mtrack"][pi] = -1
    B["KineticFdbkEnergy"][pi] = 0
    B["VDisp"][pi] = 0
