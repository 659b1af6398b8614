"""GPU parity tests of the device tree build (shq_tree_build / shq_tree_download, csrc/tree_build.hip)
against the host builder, which test_oracle_cpu.py pins to the reference's insertion build."""
import ctypes as C

import numpy as np
import pytest

import shenqi_amd as sq
from shenqi_amd import capi
import orc
import common as cm

pytestmark = pytest.mark.gpu


def preorder(nodes, firstnode):
    """Depth-first pre-order of a reference-format tree (open every node: suns[0], else sibling)."""
    order = []
    no = firstnode
    nn = len(nodes)
    while firstnode <= no < firstnode + nn:
        order.append(no - firstnode)
        nd = nodes[no - firstnode]
        if ((nd["flags"] >> 3) & 3) == 1:
            no = nd["suns"][0]
        else:
            no = nd["sibling"]
    return np.array(order)


def canonical(nodes, firstnode):
    """Per node in pre-order: everything the reference defines, with links as pre-order ranks."""
    order = preorder(nodes, firstnode)
    rank = np.full(len(nodes), -1, dtype=np.int64)
    rank[order] = np.arange(len(order))
    nd = nodes[order]
    link = lambda x: np.where((x >= firstnode) & (x < firstnode + len(nodes)), rank[np.clip(x - firstnode, 0, len(nodes) - 1)], -1)
    ctype = (nd["flags"] >> 3) & 3
    suns = nd["suns"].astype(np.int64).copy()
    internal = ctype == 1
    suns[internal] = np.where(suns[internal] >= 0, link(suns[internal]), -1)
    return dict(ctype=ctype, noccupied=nd["noccupied"], len=nd["len"], center=nd["center"], cofm=nd["cofm"], mass=nd["mass"],
                hmax=nd["hmax"], sibling=link(nd["sibling"].astype(np.int64)), father=link(nd["father"].astype(np.int64)),
                suns=suns, toplevel=(nd["flags"] & 6))


def _positions(kind, n):
    if kind == "close":
        return cm.close_positions(round(n ** (1 / 3)))
    if kind == "random":
        return cm.random_positions(orc.boost_mt19937_uniform(0, 3 * n), n)
    return sq.synth_positions(kind, n, L=cm.BOX)


@pytest.mark.parametrize("kind", ["grid", "uniform", "cluster", "close", "random"])
def test_device_tree_equals_host_tree(ctx, kind):
    """Same cells, same leaves in the same particle order, same moments to the bit."""
    n = 16**3 if kind in ("close", "random") else 24**3
    pos = _positions(kind, n)
    n = len(pos)
    pman = cm.make_partmanager(pos)
    rng = np.random.default_rng(3)
    pman.Base["Mass"] = (1.0 + rng.random(n)).astype(np.float32)  # unequal masses exercise the centre of mass
    host = sq.force_tree_full(pman)
    pv = pman.view()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    st = sq.tree_build_device(ctx, cm.BOX)
    assert st.nparticles == n and st.numnodes == len(preorder(host.Nodes_base, host.firstnode))
    dnodes, father = sq.tree_download(ctx, host.firstnode, n)
    a, b = canonical(host.Nodes_base, host.firstnode), canonical(dnodes, host.firstnode)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    # Father: the leaf that lists the particle
    leaf = (dnodes["flags"] >> 3 & 3) == 0
    for j in np.nonzero(leaf)[0][:200]:
        cnt = dnodes["noccupied"][j]
        assert np.all(father[dnodes["suns"][j][:cnt]] == host.firstnode + j)


def test_device_tree_mask_and_active(ctx):
    """Type mask, garbage / swallowed particles and an active list select the same particles as
    force_tree_rebuild_mask does (forcetree.cpp:692-705)."""
    n = 20**3
    pos = sq.synth_positions("cluster", n, L=cm.BOX)
    pman = cm.make_partmanager(pos)
    rng = np.random.default_rng(11)
    P = pman.Base
    P["Type"] = rng.choice([0, 1, 4, 5], size=n).astype(np.uint8)
    flags = np.zeros(n, dtype=np.uint8)
    flags[rng.random(n) < 0.05] |= 1   # IsGarbage
    flags[rng.random(n) < 0.05] |= 2   # Swallowed
    P["Flags"] = flags
    P["Hsml"] = 0.02 * cm.BOX * (0.5 + rng.random(n))   # hmax of gas / BH particles (forcetree.cpp:985-1005)
    active = np.sort(rng.choice(n, size=n // 2, replace=False)).astype(np.int32)
    pv = pman.view()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    sq.dynamics_upload(ctx, pman)                        # makes Hsml resident
    for mask, act in ((sq.GASMASK, None), (sq.GASMASK + sq.BHMASK, active), (sq.ALLMASK, active)):
        host = sq.force_tree_rebuild_mask(pman, mask, act)
        st = sq.tree_build_device(ctx, cm.BOX, mask, act)
        dnodes, _ = sq.tree_download(ctx, host.firstnode, 0)
        a, b = canonical(host.Nodes_base, host.firstnode), canonical(dnodes, host.firstnode)
        assert st.numnodes == len(a["len"])
        assert a["hmax"].max() > 0
        for k in a:
            assert np.array_equal(a[k], b[k]), (mask, k)


def test_walk_on_device_tree_is_bit_identical(ctx):
    """The walk pool written by the device build equals the one shq_tree_upload packs from the host
    tree: forces, potentials and interaction counts agree to the bit."""
    n = 32**3
    L = 1.0
    pos = sq.synth_positions("cluster", n, L=L)
    pos = pos[sq.hilbert_order(pos, L)]
    pman = cm.make_partmanager(pos, box=L)
    host = sq.force_tree_full(pman)
    cm.reference_treepar(ErrTolForceAcc=0.005, MaxBHOpeningAngle=0.9, Rcut=6.0, TreeUseBH=1)
    sq.gravshort_set_softenings(L / 32)
    gp = sq.make_grav_params(L, 1.5, 96, cm.G, cm.RHO0)
    pv, tv = pman.view(), host.view()
    out = []
    for device_tree in (False, True):
        capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
        if device_tree:
            sq.tree_build_device(ctx, L)
        else:
            capi.check(capi.hip.shq_tree_upload(ctx.h, C.byref(tv)))
        capi.check(capi.hip.shq_grav_short_run(ctx.h, C.byref(gp), None, 0, 1, sq.WALK_EXACT))
        acc = np.zeros((n, 3)); pot = np.zeros(n); nint = np.zeros(n, dtype=np.int64)
        capi.check(capi.hip.shq_grav_short_download(ctx.h, capi.ptr(acc), capi.ptr(pot), capi.ptr(nint), None))
        out.append((acc, pot, nint))
    for x, y in zip(out[0], out[1]):
        assert np.array_equal(x, y)


def test_device_tree_refuses_too_deep(ctx):
    """More than 8 coincident particles cannot be separated in 21 levels: the build must fail loudly
    (the reference ends the run for the same input, forcetree.cpp:393-401)."""
    n = 64
    pos = np.full((n, 3), 0.3 * cm.BOX)
    pos[:40] += np.random.default_rng(0).random((40, 3)) * cm.BOX * 0.5
    pman = cm.make_partmanager(pos)
    pv = pman.view()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    with pytest.raises(sq.ShqError):
        sq.tree_build_device(ctx, cm.BOX)


@pytest.mark.parametrize("n", [0, 1, 5, 8, 9, 65])
def test_device_tree_tiny_inputs(ctx, n):
    """Empty, single-leaf (<= 8 particles: the root is the leaf) and just-split inputs; ragged target counts."""
    rng = np.random.default_rng(n)
    pos = rng.random((n, 3)) * cm.BOX
    pman = cm.make_partmanager(pos)
    pv = pman.view()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    st = sq.tree_build_device(ctx, cm.BOX)
    assert st.nparticles == n and st.numnodes >= 1
    dnodes, father = sq.tree_download(ctx, n, n)
    if n > 0:
        host = sq.force_tree_full(pman)
        a, b = canonical(host.Nodes_base, host.firstnode), canonical(dnodes, n)
        for k in a:
            assert np.array_equal(a[k], b[k]), k
        assert np.all(father >= n)
    else:
        assert dnodes["noccupied"][0] == 0 and ((dnodes["flags"][0] >> 3) & 3) == 0
    # a walk over the tiny tree runs and counts n - 1 ... n interactions per target (direct sum inside Rcut or monopoles)
    cm.reference_treepar(ErrTolForceAcc=0.005, MaxBHOpeningAngle=0.9, Rcut=6.0, TreeUseBH=1)
    sq.gravshort_set_softenings(cm.BOX / 4)
    gp = sq.make_grav_params(cm.BOX, 1.5, 16, cm.G, cm.RHO0)
    capi.check(capi.hip.shq_grav_short_run(ctx.h, C.byref(gp), None, 0, 1, sq.WALK_EXACT | sq.WALK_TREE_ORDER))
    acc = np.zeros((n, 3)); nint = np.zeros(n, dtype=np.int64)
    capi.check(capi.hip.shq_grav_short_download(ctx.h, capi.ptr(acc), None, capi.ptr(nint), None))
    if n > 0:
        oacc, _, onint = orc.grav_walk(dnodes, n, pos, pman.Base["Mass"], np.zeros(n), gp)
        assert np.array_equal(nint, onint)
        assert np.abs(acc - oacc * cm.G).max() <= 1e-11 * max(np.abs(oacc * cm.G).max(), 1e-300)


@pytest.mark.parametrize("kind,ncbrt", [("flat", 128), ("close", 64), ("random2", 64)])
def test_device_tree_reference_invariants(ctx, kind, ncbrt):
    """The reference's own tree tests (tests/test_forcetree.cpp: check_tree :111-168, check_moments :21-109, check_hmax :237-253
    and the root hmax gate :369 on the 128^3 lattice) on the DEVICE-built tree, downloaded in the reference's NODE format."""
    import forcetree_checks as ft
    from test_forcetree_cpu import positions, hsml_table
    pos = positions(kind, ncbrt)
    n = len(pos)
    pman = cm.make_partmanager(pos, ptype=0)
    pman.Base["Hsml"] = hsml_table(n)
    pv = pman.view()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    sq.dynamics_upload(ctx, pman)                        # Hsml: the moments pass computes hmax from it
    st = sq.tree_build_device(ctx, cm.BOX, mask=sq.GASMASK)
    assert st.nparticles == n and st.numnodes < 0.7 * n
    nodes, father = sq.tree_download(ctx, n, n)
    nreal = ft.check_tree(nodes, n, father, pos)
    assert abs(nodes["mass"][0] - n) < 0.5
    ft.check_moments(nodes, n, father, pman.Base["Mass"], cm.BOX, nreal)
    ft.check_hmax(nodes, n, father, pos, pman.Base["Hsml"])
    if kind == "flat":
        assert nodes["hmax"][0] >= 0.0584                # test_forcetree.cpp:369
