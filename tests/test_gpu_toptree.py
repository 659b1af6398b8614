"""GPU parity tests of the export detection (csrc/toptree.hip): the device's export table
against the oracle's restatement of GravTopTreeWalk::toptree_visit / TopTreeWalk::toptree_visit +
export_particle, entry for entry, and the closure property of the distributed walk: primary walk over the
local tree + secondary walks at the exported NodeLists = the single-domain walk, interaction for interaction."""
import ctypes as C

import numpy as np
import pytest

import shenqi_amd as sq
from shenqi_amd import capi
import orc
import common as cm

pytestmark = pytest.mark.gpu


def _setup(n1=20, kind="cluster", seed=3):
    n = n1**3
    pos = sq.synth_positions(kind, n, L=cm.BOX)
    pos = pos[sq.hilbert_order(pos, cm.BOX)]
    pman = cm.make_partmanager(pos)
    rng = np.random.default_rng(seed)
    P = pman.Base
    P["FullTreeGravAccel"][:, 0] = 10 ** rng.uniform(1, 4, size=n)   # |a| exact: OldAcc = a / G on host and device alike
    P["Hsml"] = 0.03 * cm.BOX * (0.3 + rng.random(n))
    return pman, pos, rng


@pytest.mark.parametrize("use_bh", [0, 1])
def test_grav_export_table_equals_oracle(ctx, use_bh):
    pman, pos, rng = _setup()
    n = len(pos)
    dom = sq.force_tree_full(pman)
    tl = cm.make_domain(dom, ntask=4, me=2, depth=2)
    cm.reference_treepar(ErrTolForceAcc=0.005, MaxBHOpeningAngle=0.9, Rcut=6.0, TreeUseBH=use_bh)
    sq.gravshort_set_softenings(cm.BOX / 20)
    gp = sq.make_grav_params(cm.BOX, 1.5, 60, cm.G, cm.RHO0)
    pv = pman.view()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    capi.check(capi.hip.shq_grav_refresh_oldacc(ctx.h, cm.G))     # OldAcc = |FullTreeGravAccel + GravPM| / G
    sq.toptree_upload(ctx, dom, tl)
    counts, table = sq.grav_toptree_exports(ctx, gp, n)
    oldacc = pman.Base["FullTreeGravAccel"][:, 0] / cm.G
    ocounts, otable = orc.grav_toptree(dom.Nodes_base, dom.firstnode, dom.lastnode, tl, pos, oldacc, gp)
    assert np.array_equal(counts, np.cumsum(ocounts)) and counts[-1] == len(table) > 0
    assert np.array_equal(table, otable)
    # entries of a target are contiguous, tasks never the local one, node lists filled from the front
    assert not np.any(table["Task"] == 2)
    assert np.all(table["NodeList"][:, 0] >= dom.firstnode)
    filled = table["NodeList"] >= 0
    assert np.all(filled[:, :-1] >= filled[:, 1:])
    # an active list: same entries for those targets
    act = np.sort(rng.choice(n, size=n // 7, replace=False)).astype(np.int32)
    c2, t2 = sq.grav_toptree_exports(ctx, gp, len(act), act)
    oc2, ot2 = orc.grav_toptree(dom.Nodes_base, dom.firstnode, dom.lastnode, tl, pos, oldacc, gp, act)
    assert np.array_equal(c2, np.cumsum(oc2)) and np.array_equal(t2, ot2)


@pytest.mark.parametrize("symmetric", [0, 1])
def test_ngb_export_table_equals_oracle(ctx, symmetric):
    pman, pos, rng = _setup(kind="uniform")
    n = len(pos)
    dom = sq.force_tree_full(pman)
    nodes = dom.Nodes_base
    nodes["hmax"] = 0.02 * cm.BOX * rng.random(len(nodes))     # what force_tree_update_hmax would leave on the top nodes
    tl = cm.make_domain(dom, ntask=5, me=0, depth=3)
    pv = pman.view()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    sq.dynamics_upload(ctx, pman)                               # Hsml resident
    sq.toptree_upload(ctx, dom, tl)
    counts, table = sq.ngb_toptree_exports(ctx, symmetric, cm.BOX, n)
    ocounts, otable = orc.ngb_toptree(nodes, dom.firstnode, dom.lastnode, tl, pos, pman.Base["Hsml"], symmetric, cm.BOX)
    assert np.array_equal(counts, np.cumsum(ocounts)) and len(table) > 0
    assert np.array_equal(table, otable)


def test_resident_export_detection_and_device_query_packing(ctx):
    """shq_grav_toptree_exports_resident + shq_grav_export_pack: the table stays in HBM; only the total and the per-task send counts
    come back, and the GravTreeQuery records are written on the device in task order (entries of one task in table order) together
    with the `place` list of the reduce step — against the host-table path of the same walk, for all targets and for a device list."""
    import torch
    pman, pos, rng = _setup()
    n = len(pos)
    dom = sq.force_tree_full(pman)
    ntask = 6
    tl = cm.make_domain(dom, ntask=ntask, me=1, depth=3)
    cm.reference_treepar(ErrTolForceAcc=0.005, MaxBHOpeningAngle=0.9, Rcut=6.0, TreeUseBH=0)
    sq.gravshort_set_softenings(cm.BOX / 20)
    gp = sq.make_grav_params(cm.BOX, 1.5, 60, cm.G, cm.RHO0)
    pv = pman.view()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    capi.check(capi.hip.shq_grav_refresh_oldacc(ctx.h, cm.G))
    sq.toptree_upload(ctx, dom, tl)
    oldacc = pman.Base["FullTreeGravAccel"][:, 0] / cm.G
    act = np.sort(rng.choice(n, size=n // 5, replace=False)).astype(np.int32)
    for active in (None, act):
        nt = n if active is None else len(active)
        _, table = sq.grav_toptree_exports(ctx, gp, nt, active)
        nexp = C.c_int64(-1)
        tc = np.zeros(ntask, dtype=np.int64)
        d_act = None if active is None else torch.from_numpy(active).cuda()
        capi.check(capi.hip.shq_grav_toptree_exports_resident(ctx.h, C.byref(gp), None if d_act is None else d_act.data_ptr(), nt,
                                                              0 if d_act is None else 1, ntask, C.byref(nexp), capi.ptr(tc)))
        assert nexp.value == len(table) > 0
        assert np.array_equal(tc, np.bincount(table["Task"], minlength=ntask)) and tc[1] == 0
        d_q = torch.zeros(nexp.value * capi.GRAV_QUERY_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
        d_place = torch.zeros(nexp.value, dtype=torch.int32, device="cuda")
        capi.check(capi.hip.shq_grav_export_pack(ctx.h, d_q.data_ptr(), d_place.data_ptr()))
        ctx.synchronize()
        q = d_q.cpu().numpy().view(capi.GRAV_QUERY_DTYPE)
        place = d_place.cpu().numpy()
        order = np.argsort(table["Task"], kind="stable")
        want = table[order]
        assert np.array_equal(place, want["Index"])
        assert np.array_equal(q["NodeList"], want["NodeList"])
        assert np.array_equal(q["Pos"], pos[want["Index"]]) and np.array_equal(q["OldAcc"], oldacc[want["Index"]])


def test_primary_plus_secondary_equals_single_domain_walk(ctx):
    """The reference's distributed walk, assembled from this library's three pieces: primary walk on the tree
    with pseudo nodes, export table from the top-tree walk, secondary walks (on the owner's tree) of the
    exported queries.  Their sum must be the walk over the undivided tree: same interactions, same forces."""
    pman, pos, rng = _setup(n1=24)
    n = len(pos)
    me = 1
    full = sq.force_tree_full(pman)
    cm.make_domain(full, ntask=3, me=me, depth=2, pseudo=False)
    dom = sq.force_tree_full(pman)
    tl = cm.make_domain(dom, ntask=3, me=me, depth=2)
    cm.reference_treepar(ErrTolForceAcc=0.005, MaxBHOpeningAngle=0.9, Rcut=6.0, TreeUseBH=0)
    sq.gravshort_set_softenings(cm.BOX / 24)
    gp = sq.make_grav_params(cm.BOX, 1.5, 72, cm.G, cm.RHO0)
    pv = pman.view()
    oldacc = pman.Base["FullTreeGravAccel"][:, 0] / cm.G

    def walk(tree):
        tv = tree.view()
        capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
        capi.check(capi.hip.shq_tree_upload(ctx.h, C.byref(tv)))
        capi.check(capi.hip.shq_grav_refresh_oldacc(ctx.h, cm.G))
        capi.check(capi.hip.shq_grav_short_run(ctx.h, C.byref(gp), None, 0, 0, sq.WALK_EXACT))
        acc = np.zeros((n, 3)); nint = np.zeros(n, dtype=np.int64)
        capi.check(capi.hip.shq_grav_short_download(ctx.h, capi.ptr(acc), None, capi.ptr(nint), None))
        return acc, nint

    acc_full, nint_full = walk(full)
    acc_loc, nint_loc = walk(dom)
    sq.toptree_upload(ctx, dom, tl)
    counts, table = sq.grav_toptree_exports(ctx, gp, n)
    assert len(table) > n // 10 and nint_loc.sum() < nint_full.sum()
    q = np.zeros(len(table), dtype=capi.GRAV_QUERY_DTYPE)
    q["Pos"] = pos[table["Index"]]
    q["OldAcc"] = oldacc[table["Index"]]
    q["NodeList"] = table["NodeList"]
    tv = full.view()
    capi.check(capi.hip.shq_tree_upload(ctx.h, C.byref(tv)))     # the owners' side: every leaf is a real subtree here
    res = np.zeros(len(q), dtype=capi.GRAV_RESULT_DTYPE)
    nint2 = np.zeros(len(q), dtype=np.int64)
    capi.check(capi.hip.shq_grav_short_secondary(ctx.h, C.byref(gp), capi.ptr(q), len(q), capi.ptr(res), capi.ptr(nint2), 0))
    nint_sum = nint_loc.copy()
    np.add.at(nint_sum, table["Index"], nint2)
    assert np.array_equal(nint_sum, nint_full)
    acc_sum = acc_loc.copy()
    np.add.at(acc_sum, table["Index"], res["Acc"] * cm.G)
    assert np.abs(acc_sum - acc_full).max() < 1e-11 * np.abs(acc_full).max()


def test_distributed_walk_through_the_abi(ctx):
    """The same closure with the library's own reduction: deferred postprocess, shq_grav_reduce_export_results
    (entries of a target added in table order, ev_reduce_export_result), shq_grav_postprocess — against the one-shot
    run on the undivided tree, potential included."""
    pman, pos, rng = _setup(n1=20)
    n = len(pos)
    full = sq.force_tree_full(pman)
    cm.make_domain(full, ntask=4, me=0, depth=2, pseudo=False)
    dom = sq.force_tree_full(pman)
    tl = cm.make_domain(dom, ntask=4, me=0, depth=2)
    cm.reference_treepar(ErrTolForceAcc=0.005, MaxBHOpeningAngle=0.9, Rcut=6.0, TreeUseBH=0)
    sq.gravshort_set_softenings(cm.BOX / 20)
    gp = sq.make_grav_params(cm.BOX, 1.5, 60, cm.G, cm.RHO0)
    pv = pman.view()
    oldacc = pman.Base["FullTreeGravAccel"][:, 0] / cm.G

    def download():
        acc = np.zeros((n, 3)); pot = np.zeros(n)
        capi.check(capi.hip.shq_grav_short_download(ctx.h, capi.ptr(acc), capi.ptr(pot), None, None))
        return acc, pot

    tv = full.view()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    capi.check(capi.hip.shq_tree_upload(ctx.h, C.byref(tv)))
    capi.check(capi.hip.shq_grav_refresh_oldacc(ctx.h, cm.G))
    capi.check(capi.hip.shq_grav_short_run(ctx.h, C.byref(gp), None, 0, 1, sq.WALK_EXACT))
    acc_full, pot_full = download()
    # "remote" side first (same device, the undivided tree is still loaded): secondary walks of the exports
    sq.toptree_upload(ctx, dom, tl)
    counts, table = sq.grav_toptree_exports(ctx, gp, n)
    q = np.zeros(len(table), dtype=capi.GRAV_QUERY_DTYPE)
    q["Pos"], q["OldAcc"], q["NodeList"] = pos[table["Index"]], oldacc[table["Index"]], table["NodeList"]
    res = np.zeros(len(q), dtype=capi.GRAV_RESULT_DTYPE)
    capi.check(capi.hip.shq_grav_short_secondary(ctx.h, C.byref(gp), capi.ptr(q), len(q), capi.ptr(res), None, 1))
    # local side: primary walk with the postprocess deferred, reduce, postprocess
    tv = dom.view()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    capi.check(capi.hip.shq_tree_upload(ctx.h, C.byref(tv)))
    capi.check(capi.hip.shq_grav_refresh_oldacc(ctx.h, cm.G))
    assert capi.hip.shq_grav_postprocess(ctx.h, C.byref(gp), None, 0, 1) != 0       # nothing deferred yet
    capi.check(capi.hip.shq_grav_short_run(ctx.h, C.byref(gp), None, 0, 1, sq.WALK_EXACT | sq.WALK_DEFER_POSTPROCESS))
    place = np.ascontiguousarray(table["Index"])
    perm = rng.permutation(len(place))                 # any order of the table must give the grouped result
    capi.check(capi.hip.shq_grav_reduce_export_results(ctx.h, capi.ptr(np.ascontiguousarray(place[perm])),
                                                       capi.ptr(np.ascontiguousarray(res[perm])), len(place), 1))
    capi.check(capi.hip.shq_grav_postprocess(ctx.h, C.byref(gp), None, 0, 1))
    acc, pot = download()
    assert np.abs(acc - acc_full).max() < 1e-11 * np.abs(acc_full).max()
    assert np.abs(pot - pot_full).max() < 1e-11 * np.abs(pot_full).max()
    assert np.abs(acc_full).max() > 0 and len(table) > 0


def test_toptree_edge_cases(ctx):
    pman, pos, rng = _setup(n1=10)
    tree = sq.force_tree_full(pman)
    pv = pman.view()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    cm.reference_treepar(ErrTolForceAcc=0.005, MaxBHOpeningAngle=0.9, Rcut=6.0, TreeUseBH=0)
    gp = sq.make_grav_params(cm.BOX, 1.5, 30, cm.G, cm.RHO0)
    with pytest.raises(sq.ShqError):                 # no top tree yet
        sq.grav_toptree_exports(ctx, gp, len(pos))
    # the single-leaf domain of the host builder (tests/test_forcetree.cpp:294-314): the root is the only top leaf
    sq.toptree_upload(ctx, tree, np.zeros(1, dtype=capi.TOPLEAF_DTYPE))
    counts, table = sq.grav_toptree_exports(ctx, gp, len(pos))
    assert counts[-1] == 0 and len(table) == 0
    nodes = tree.Nodes_base
    nodes["flags"][0] &= ~np.uint32(3)                # no TopLevel flag at all: not a tree of a domain decomposition
    with pytest.raises(sq.ShqError):
        sq.toptree_upload(ctx, tree, np.zeros(1, dtype=capi.TOPLEAF_DTYPE))
    nodes["flags"][0] |= 2
    tl = cm.make_domain(tree, ntask=1, me=0, depth=2)  # everything local: no exports, empty table
    sq.toptree_upload(ctx, tree, tl)
    counts, table = sq.grav_toptree_exports(ctx, gp, len(pos))
    assert counts[-1] == 0 and len(table) == 0
    tl = cm.make_domain(sq.force_tree_full(pman), ntask=2, me=0, depth=1)
    bad = tl[:1]                                      # pseudo nodes refer to leaves the table does not have
    dom = sq.force_tree_full(pman)
    cm.make_domain(dom, ntask=2, me=0, depth=1)
    with pytest.raises(sq.ShqError):
        sq.toptree_upload(ctx, dom, bad)
    # a too small table is refused, the count still comes back
    sq.toptree_upload(ctx, dom, tl)
    nexp = C.c_int64()
    tab = np.zeros(1, dtype=capi.DATA_INDEX_DTYPE)
    rc = capi.hip.shq_grav_toptree_exports(ctx.h, C.byref(gp), None, 0, None, capi.ptr(tab), 1, C.byref(nexp))
    assert rc != 0 and nexp.value > 1
