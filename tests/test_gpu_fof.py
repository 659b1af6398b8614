"""GPU parity tests of the friends-of-friends finder (csrc/fof.hip) through the C-ABI: the reference's own fixtures
(libgadget/tests/test_fof.cpp, at their full size) with every BOOST_TEST of theirs, and equality with the restatement
(oracle/fof.py) — labels, group numbers and lengths as integers, the group sums to the bit (same order of the same operations)."""
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
import fof as ofof  # noqa: E402
import fof_fixtures as fx  # noqa: E402

pytestmark = pytest.mark.gpu


def gpu_fof(ctx, pos, vel, mass, types, ids, box, linkl, minlength, flags=None, hsml=None, primary=2, secondary=1 + 16 + 32):
    n = len(pos)
    pman = cm.make_partmanager(pos, box=box)
    P = pman.Base
    P["Type"], P["Mass"], P["Vel"], P["ID"] = types, mass, vel, ids
    if flags is not None:
        P["Flags"] = flags
    if hsml is not None:
        P["Hsml"] = hsml
    pv = pman.view()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    sq.dynamics_upload(ctx, pman)
    fp = capi.FofParams(box, linkl, primary, secondary, minlength, 0)
    minid = np.zeros(n, dtype=np.uint64)
    grnr = np.zeros(n, dtype=np.int32)
    ng = C.c_int64()
    idarr = np.ascontiguousarray(ids, dtype=np.uint64)
    capi.check(capi.hip.shq_fof(ctx.h, C.byref(fp), capi.ptr(idarr), capi.ptr(minid), capi.ptr(grnr), C.byref(ng)))
    groups = np.zeros(ng.value, dtype=capi.FOF_GROUP_DTYPE)
    capi.check(capi.hip.shq_fof_groups_download(ctx.h, capi.ptr(groups), len(groups)))
    nm = C.c_int64()
    capi.check(capi.hip.shq_fof_members(ctx.h, None, 0, C.byref(nm)))
    members = np.zeros(nm.value, dtype=np.int32)
    capi.check(capi.hip.shq_fof_members(ctx.h, capi.ptr(members), len(members), C.byref(nm)))
    return minid, grnr, groups, members, P["Mass"].astype(np.float64)


def as_dicts(groups):
    return [dict(MinID=int(g["MinID"]), Length=int(g["Length"]), GrNr=int(g["GrNr"]), LenType=list(g["LenType"]), MassType=list(g["MassType"]),
                 Mass=float(g["Mass"]), CM=g["CM"], Vel=g["Vel"]) for g in groups]


def compare(groups, members, ogroups, grnr, ogrnr, minid, ominid):
    assert np.array_equal(minid, ominid)
    assert np.array_equal(grnr, ogrnr)
    assert len(groups) == len(ogroups)
    off = 0
    for g, o in zip(groups, ogroups):
        assert g["MinID"] == o["MinID"] and g["Length"] == o["Length"] and g["GrNr"] == o["GrNr"]
        assert list(g["LenType"]) == o["LenType"] and g["seed_index"] == o["seed_index"]
        assert tuple(g["FirstPos"]) == tuple(o["FirstPos"])
        assert list(g["MassType"]) == o["MassType"] and g["Mass"] == o["Mass"] and g["MaxDens"] == o["MaxDens"]
        for f in ("CM", "Vel", "Jmom", "Imom"):
            assert np.array_equal(g[f], o[f]), (f, g[f], o[f])
        assert g["first_member"] == off
        mem = members[off:off + g["Length"]]
        assert (np.diff(mem) > 0).all() and (ominid[mem] == o["MinID"]).all()
        off += g["Length"]
    assert off == len(members)


def test_fof_line_reference_fixture(ctx):
    f = fx.line()                                   # 512^2 particles, as the reference runs it
    n = len(f["pos"])
    minid, grnr, groups, members, _ = gpu_fof(ctx, f["pos"], f["vel"], f["mass"], f["types"], f["ids"], f["box"], f["linkl"], f["minlength"])
    fx.check_line(as_dicts(groups), grnr, f)
    assert (minid == 1).all() and len(members) == n and np.array_equal(members, np.arange(n))


def test_fof_halos_reference_fixture(ctx):
    f = fx.halos()
    n = len(f["pos"])
    minid, grnr, groups, members, massd = gpu_fof(ctx, f["pos"], f["vel"], f["mass"], f["types"], f["ids"], f["box"], f["linkl"], f["minlength"])
    fx.check_halos(as_dicts(groups), grnr, f)
    ominid, ogroups, ogrnr = ofof.fof(f["pos"], f["vel"], massd, f["types"], f["ids"], np.zeros(n, dtype=bool), np.zeros(n), f["box"], f["linkl"], f["minlength"])
    compare(groups, members, ogroups, grnr, ogrnr, minid, ominid)


@pytest.mark.parametrize("seed", [1, 2])
def test_fof_clustered_mixed_types_equals_oracle(ctx, seed):
    """dark matter in clumps of all sizes + a uniform part, gas / stars / black holes scattered around them with their own
    smoothing lengths, a few garbage and swallowed particles, shuffled IDs: labels, group numbers and sums against the oracle"""
    rng = np.random.default_rng(seed)
    box = 1000.0
    centres = rng.random((60, 3)) * box
    sizes = rng.integers(3, 400, size=60)
    dm = np.concatenate([c + rng.normal(size=(s, 3)) * rng.uniform(0.5, 3.0) for c, s in zip(centres, sizes)] + [rng.random((4000, 3)) * box])
    nsec = 3000
    sec = np.concatenate([centres[rng.integers(0, 60, size=nsec // 2)] + rng.normal(size=(nsec // 2, 3)) * 6.0, rng.random((nsec - nsec // 2, 3)) * box])
    pos = np.mod(np.concatenate([dm, sec]), box)
    n = len(pos)
    types = np.concatenate([np.ones(len(dm), dtype=np.uint8), rng.choice([0, 4, 5], size=nsec, p=[0.7, 0.25, 0.05]).astype(np.uint8)])
    perm = rng.permutation(n)
    pos, types = pos[perm], types[perm]
    ids = rng.permutation(n).astype(np.uint64) + 1000
    vel = rng.normal(size=(n, 3)) * 100
    mass = rng.choice([1.0, 0.5, 0.125], size=n).astype(np.float32).astype(np.float64)
    hsml = rng.uniform(0.2, 12.0, size=n)
    flags = np.zeros(n, dtype=np.uint8)
    flags[rng.random(n) < 0.01] |= 1
    flags[rng.random(n) < 0.01] |= 2
    dead = (flags & 3) != 0
    linkl = 2.0
    minid, grnr, groups, members, massd = gpu_fof(ctx, pos, vel, mass, types, ids, box, linkl, 8, flags=flags, hsml=hsml)
    ominid, ogroups, ogrnr = ofof.fof(pos, vel, massd, types, ids, dead, hsml, box, linkl, 8)
    compare(groups, members, ogroups, grnr, ogrnr, minid, ominid)
    nsec_attached = int((ominid[types != 1] != ids[types != 1]).sum())
    assert len(groups) > 20 and 100 < nsec_attached < nsec
    assert max(g["Length"] for g in groups) > 300


def test_fof_long_group_is_summed_by_a_workgroup(ctx):
    """a group of more than 4096 members: 256 slices of the member list summed in parallel and reduced in slice order (as the
    reference reduces the parts of a group spread over tasks): integers exact, sums to rounding"""
    rng = np.random.default_rng(9)
    box = 500.0
    blob = 250 + rng.normal(size=(9000, 3)) * 4.0
    pos = np.mod(np.concatenate([blob, rng.random((3000, 3)) * box]), box)
    n = len(pos)
    ids = rng.permutation(n).astype(np.uint64) + 1
    vel = rng.normal(size=(n, 3)) * 50
    mass = rng.choice([1.0, 0.5], size=n)
    types = np.ones(n, dtype=np.uint8)
    minid, grnr, groups, members, massd = gpu_fof(ctx, pos, vel, mass, types, ids, box, 1.0, 8)
    ominid, ogroups, ogrnr = ofof.fof(pos, vel, massd, types, ids, np.zeros(n, dtype=bool), np.zeros(n), box, 1.0, 8)
    assert np.array_equal(minid, ominid) and np.array_equal(grnr, ogrnr) and len(groups) == len(ogroups)
    big = 0
    for g, o in zip(groups, ogroups):
        assert g["MinID"] == o["MinID"] and g["Length"] == o["Length"] and g["GrNr"] == o["GrNr"] and list(g["LenType"]) == o["LenType"]
        if g["Length"] > 4096:
            big += 1
            scale = {"CM": box, "Vel": 50.0, "Jmom": 50.0 * 4.0 * g["Mass"], "Imom": 16.0 * g["Mass"]}
            assert abs(g["Mass"] - o["Mass"]) < 1e-12 * o["Mass"]
            for f in ("CM", "Vel", "Jmom", "Imom"):
                assert np.abs(g[f] - o[f]).max() < 1e-11 * scale[f], f
        else:
            for f in ("CM", "Vel", "Jmom", "Imom"):
                assert np.array_equal(g[f], o[f]), f
    assert big == 1


def test_fof_requires_run_and_handles_empty_and_no_groups(ctx):
    rng = np.random.default_rng(4)
    pos = rng.random((500, 3)) * cm.BOX
    n = len(pos)
    minid, grnr, groups, members, _ = gpu_fof(ctx, pos, np.zeros((n, 3)), np.ones(n), np.ones(n, dtype=np.uint8), np.arange(1, n + 1, dtype=np.uint64), cm.BOX,
                                              1e-4, 5)
    assert len(groups) == 0 and len(members) == 0 and (grnr == -1).all() and np.array_equal(minid, np.arange(1, n + 1, dtype=np.uint64))
    c2 = sq.Context(0)
    try:
        assert capi.hip.shq_fof_groups_download(c2.h, None, 0) != 0
    finally:
        c2.close()


def test_fof_leaves_no_resident_tree_behind(ctx):
    """shq_fof builds its tree over the primary link types only (fof.cpp:176-178) and, like fof_fof (:254), keeps none: a gravity walk
    of a mixed-type set right after it must fail loudly (SHQ_ERR_STATE) rather than walk a tree without the gas, star and BH particles,
    and work again once the caller has built its own tree."""
    rng = np.random.default_rng(11)
    n = 4000
    pos = rng.random((n, 3)) * cm.BOX
    types = rng.integers(0, 6, n).astype(np.uint8)
    types[types == 2] = 1
    types[types == 3] = 1
    gpu_fof(ctx, pos, np.zeros((n, 3)), np.ones(n), types, np.arange(1, n + 1, dtype=np.uint64), cm.BOX, 0.2, 5)
    sq.set_gravshort_treepar(ErrTolForceAcc=0.005, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=1, Rcut=6.0)
    sq.gravshort_set_softenings(cm.BOX / 16)
    gp = sq.make_grav_params(cm.BOX, 1.5, 48, cm.G, cm.RHO0)
    rc = capi.hip.shq_grav_short_run(ctx.h, C.byref(gp), None, 0, 1, 0)
    assert rc == 4 and b"tree" in capi.hip.shq_last_error()      # SHQ_ERR_STATE
    sq.tree_build_device(ctx, cm.BOX)
    capi.check(capi.hip.shq_grav_short_run(ctx.h, C.byref(gp), None, 0, 1, 0))
    nint = np.zeros(n, dtype=np.int64)
    capi.check(capi.hip.shq_grav_short_download(ctx.h, None, None, capi.ptr(nint), None))
    assert (nint > 0).all()                                       # every type is back in the tree


def test_fof_seed_candidates_and_seed_select(ctx):
    """the gas state resident (shq_sph_state_upload): MaxDens / seed_index per group as add_particle_to_group finds them (densest gas
    member that is not a decoupled wind particle, fof.cpp:619-628), then fof_seed's marking loop (fof.cpp:1290-1302) against the
    restatement, with thresholds that keep some groups and drop others"""
    rng = np.random.default_rng(21)
    box = 400.0
    centres = rng.random((40, 3)) * box
    sizes = rng.integers(10, 300, size=40)
    dm = np.concatenate([c + rng.normal(size=(s, 3)) * 1.5 for c, s in zip(centres, sizes)])
    nsec = 2500
    sec = centres[rng.integers(0, 40, size=nsec)] + rng.normal(size=(nsec, 3)) * 2.5
    pos = np.mod(np.concatenate([dm, sec]), box)
    n = len(pos)
    types = np.concatenate([np.ones(len(dm), dtype=np.uint8), rng.choice([0, 4, 5], size=nsec, p=[0.6, 0.39, 0.01]).astype(np.uint8)])
    perm = rng.permutation(n)
    pos, types = pos[perm], types[perm]
    ids = rng.permutation(n).astype(np.uint64) + 5
    vel = rng.normal(size=(n, 3)) * 50
    mass = rng.choice([1.0, 0.25], size=n).astype(np.float32).astype(np.float64)
    hsml = rng.uniform(0.5, 4.0, size=n)
    gas = np.flatnonzero(types == 0)
    S = np.zeros(len(gas), dtype=capi.SPH_DTYPE)
    S["Density"] = rng.uniform(1.0, 100.0, len(gas))
    S["DelayTime"] = np.where(rng.random(len(gas)) < 0.3, 1.0, 0.0)      # winds: never a seed when WindsDecoupleSph is on
    pman = cm.make_partmanager(pos, box=box)
    P = pman.Base
    P["Type"], P["Mass"], P["Vel"], P["ID"], P["Hsml"] = types, mass, vel, ids, hsml
    P["PI"][gas] = np.arange(len(gas))
    density = np.zeros(n)
    density[gas] = S["Density"]
    delayed = np.zeros(n, dtype=bool)
    delayed[gas] = S["DelayTime"] > 0
    pv, sv = pman.view(), capi.sph_view(S)
    for winds in (0, 1):
        capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
        sq.dynamics_upload(ctx, pman)
        capi.check(capi.hip.shq_sph_state_upload(ctx.h, C.byref(pv), C.byref(sv)))
        fp = capi.FofParams(box, 1.2, 2, 1 + 16 + 32, 8, winds)
        minid = np.zeros(n, dtype=np.uint64)
        grnr = np.zeros(n, dtype=np.int32)
        ng = C.c_int64()
        capi.check(capi.hip.shq_fof(ctx.h, C.byref(fp), capi.ptr(np.ascontiguousarray(ids)), capi.ptr(minid), capi.ptr(grnr), C.byref(ng)))
        groups = np.zeros(ng.value, dtype=capi.FOF_GROUP_DTYPE)
        capi.check(capi.hip.shq_fof_groups_download(ctx.h, capi.ptr(groups), len(groups)))
        ominid, ogroups, _ = ofof.fof(pos, vel, mass, types, ids, np.zeros(n, dtype=bool), hsml, box, 1.2, 8, density=density,
                                      decoupled=delayed if winds else None)
        assert np.array_equal(minid, ominid) and len(groups) == len(ogroups) > 20
        assert [int(g["seed_index"]) for g in groups] == [o["seed_index"] for o in ogroups]
        assert [float(g["MaxDens"]) for g in groups] == [o["MaxDens"] for o in ogroups]
        assert sum(o["seed_index"] >= 0 for o in ogroups) > 10
        import torch
        for minmass, minstar in ((0.0, 0.0), (150.0, 0.0), (60.0, 8.0), (1e9, 0.0)):
            want = ofof.seed_marks(ogroups, minmass, minstar)
            out = torch.full((max(len(groups), 1),), -9, dtype=torch.int32, device="cuda:0")
            ns = C.c_int64(-1)
            capi.check(capi.hip.shq_fof_seed_select(ctx.h, minmass, minstar, None, 0, C.byref(ns)))
            assert ns.value == len(want)
            capi.check(capi.hip.shq_fof_seed_select(ctx.h, minmass, minstar, out.data_ptr(), len(groups), C.byref(ns)))
            ctx.synchronize()
            assert out.cpu().numpy()[:ns.value].tolist() == want
            if len(want) > 1:
                with pytest.raises(sq.ShqError):
                    capi.check(capi.hip.shq_fof_seed_select(ctx.h, minmass, minstar, out.data_ptr(), 1, C.byref(ns)))
        assert 0 < len(ofof.seed_marks(ogroups, 60.0, 8.0)) < len(ofof.seed_marks(ogroups, 0.0, 0.0))


def test_fof_group_sums_of_slot_quantities(ctx):
    """shq_fof_group_sums: per-group sums of caller-supplied per-particle columns (Sfr, metal masses, BH_Mass ... of
    add_particle_to_group, fof.cpp:599-618) in member order: small groups exactly the serial sum, the long one to rounding"""
    rng = np.random.default_rng(33)
    box = 300.0
    centres = rng.random((30, 3)) * box
    sizes = rng.integers(5, 200, size=30)
    sizes[0] = 3000                                         # one group beyond 256 members per slice... and beyond 256 members
    pos = np.mod(np.concatenate([c + rng.normal(size=(s, 3)) * 1.2 for c, s in zip(centres, sizes)] + [rng.random((2000, 3)) * box]), box)
    n = len(pos)
    perm = rng.permutation(n)
    pos = pos[perm]
    types = np.ones(n, dtype=np.uint8)
    ids = rng.permutation(n).astype(np.uint64) + 1
    vel = np.zeros((n, 3))
    mass = np.ones(n)
    minid, grnr, groups, members, _ = gpu_fof(ctx, pos, vel, mass, types, ids, box, 1.5, 8)
    ncol = 5
    vals = np.ascontiguousarray(rng.normal(size=(n, ncol)) * np.array([1.0, 1e3, 1e-3, 5.0, 0.0]))
    sums = np.full((len(groups), ncol), np.nan)
    capi.check(capi.hip.shq_fof_group_sums(ctx.h, capi.ptr(vals), ncol, capi.ptr(sums)))
    assert len(groups) > 15 and groups["Length"].max() > 2500
    for g, G in enumerate(groups):
        mem = members[G["first_member"]:G["first_member"] + G["Length"]]
        if G["Length"] <= 256:
            want = np.zeros(ncol)
            for i in mem:                                    # the serial loop of fof_compile_catalogue over the sorted members
                want += vals[i]
            assert np.array_equal(sums[g], want), g
        else:
            want = vals[mem].sum(axis=0)
            assert np.abs(sums[g] - want).max() <= 1e-12 * np.abs(vals[mem]).sum(axis=0).max()
    assert (sums[:, 4] == 0).all()
    with pytest.raises(sq.ShqError):
        capi.check(capi.hip.shq_fof_group_sums(ctx.h, None, 3, capi.ptr(sums)))


def test_fof_properties_at_scale_on_the_cluster(ctx):
    """128^3 particles of the bench's S-cluster (two groups of ~10^6 members around a caustic, cliques many levels deep): what a
    friends-of-friends catalogue must satisfy whatever its size — friends share a label (a sample of particles against a k-d tree of
    all), small groups are connected and complete, lengths add up, the label is the smallest member ID"""
    from scipy.spatial import cKDTree
    n1 = 128
    n = n1**3
    L = 1.0
    pos = sq.synth_positions("cluster", n, L=L)
    rng = np.random.default_rng(2)
    ids = rng.permutation(n).astype(np.uint64) + 1
    types = np.ones(n, dtype=np.uint8)
    linkl = 0.2 * L / n1
    minid, grnr, groups, members, _ = gpu_fof(ctx, pos, np.zeros((n, 3)), np.ones(n), types, ids, L, linkl, 2)
    assert len(groups) >= 2 and groups["Length"].max() > n // 20
    # lengths, numbering and labels
    assert int(groups["Length"].sum()) == len(members) == int((grnr > 0).sum())
    assert np.array_equal(np.sort(groups["GrNr"]), np.arange(1, len(groups) + 1))
    order = np.argsort(groups["GrNr"])
    assert (np.diff(groups["Length"][order]) <= 0).all()
    low = np.full(len(groups), np.iinfo(np.uint64).max, dtype=np.uint64)
    gi = np.repeat(np.arange(len(groups)), groups["Length"])
    np.minimum.at(low, gi, ids[members])
    assert np.array_equal(low, groups["MinID"]) and np.array_equal(minid[members], groups["MinID"][gi])
    # friends share a label: a sample from the dense core, the outskirts and the field
    tree = cKDTree(np.mod(pos, L), boxsize=L)
    sample = np.concatenate([rng.choice(n, 1500, replace=False), members[rng.choice(len(members), 1500, replace=False)]])
    for i in sample:
        nb = tree.query_ball_point(np.mod(pos[i], L), linkl * (1 - 1e-12))
        assert (minid[nb] == minid[i]).all()
    # small groups: connected through links <= linkl, and nothing within linkl of a member is left outside
    small = np.flatnonzero(groups["Length"] <= 64)[:40]
    assert len(small) >= 5
    for g in small:
        mem = members[groups["first_member"][g]:groups["first_member"][g] + groups["Length"][g]]
        sub = cKDTree(np.mod(pos[mem], L), boxsize=L)
        pairs = sub.query_pairs(linkl * (1 + 1e-12), output_type="ndarray")
        reach = {0}
        frontier = [0]
        adj = {}
        for a, b in pairs:
            adj.setdefault(int(a), []).append(int(b))
            adj.setdefault(int(b), []).append(int(a))
        while frontier:
            x = frontier.pop()
            for y in adj.get(x, []):
                if y not in reach:
                    reach.add(y)
                    frontier.append(y)
        assert len(reach) == len(mem)
        for i in mem:
            nb = tree.query_ball_point(np.mod(pos[i], L), linkl * (1 - 1e-12))
            assert np.isin(nb, mem).all()
