"""CPU tests pinning the oracle: reference golden values, reference-built pieces (oracle/_ref),
and the reference's own accuracy gates, plus host-logic checks (tree builder, params)."""
import ctypes as C
import os

import numpy as np
import pytest

import shenqi_amd as sq
from shenqi_amd import capi
import orc
import common as cm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")

# tests/test_densitykernel.cpp:13-36 (exact golden values)
KERNEL_GOLDEN = {
    1: dict(desnumngb=33.510321638291124, wk=0.079577471545947673, dwk=-0.238732414637843),
    4: dict(desnumngb=65.449846949787357, wk=0.075283862851696096, dwk=-0.29142140458721072),
    2: dict(desnumngb=113.09733552923254, wk=0.066304197971682174, dwk=-0.3147351169541876),
}


@pytest.mark.parametrize("ktype", [1, 2, 4])
def test_density_kernel_golden(ktype):
    k = orc.density_kernel(ktype, 2.0, 0.5, 1.0)
    g = KERNEL_GOLDEN[ktype]
    assert abs(k["desnumngb"] - g["desnumngb"]) < 1e-9 * g["desnumngb"]
    assert k["volume"] == 4.0 / 3.0 * np.pi * 2.0**3
    assert k["wk"] == g["wk"]
    assert k["dwk"] == g["dwk"]


@pytest.mark.skipif(not os.path.exists(os.path.join(REFDIR, "libdensitykernel_ref.so")), reason="oracle/_ref not built")
def test_density_kernel_vs_reference_build():
    ref = C.CDLL(os.path.join(REFDIR, "libdensitykernel_ref.so"))
    ref.ref_density_kernel.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p]
    rng = np.random.default_rng(7)
    for ktype in (1, 2, 4):
        for _ in range(200):
            H, u, eta = rng.uniform(0.01, 5), rng.uniform(0, 1.05), rng.uniform(0.5, 2)
            out = np.zeros(5)
            assert ref.ref_density_kernel(ktype, H, u, eta, capi.ptr(out)) == 0
            k = orc.density_kernel(ktype, H, u, eta)
            got = np.array([k["desnumngb"], k["volume"], k["wk"], k["dwk"], k["dW"]])
            assert np.array_equal(got, out), (ktype, H, u, got, out)


@pytest.mark.skipif(not os.path.exists(os.path.join(REFDIR, "libshortrange_ref.so")), reason="oracle/_ref not built")
def test_shortrange_table_fixture_matches_reference_data():
    ref = C.CDLL(os.path.join(REFDIR, "libshortrange_ref.so"))
    arr = (C.c_double * (512 * 5)).in_dll(ref, "shortrange_force_kernels")
    tab = np.frombuffer(arr, dtype=np.float64).reshape(512, 5)
    assert np.array_equal(tab, capi.load_kernel_table())


def test_shortrange_table_properties():
    tab = capi.load_kernel_table()
    assert tab.shape == (512, 5)
    assert tab[1, 0] == 0.029354207436397859 or abs(tab[1, 0] - 0.029354207436397859) < 1e-15
    assert abs(tab[-1, 0] - 15.0) < 1e-12
    # GravShortTable (gravity.h:32-61): exact window = columns 2 (force) and 1 (potential)
    cm.reference_treepar()
    gp = sq.make_grav_params(8.0, 1.5, 48, cm.G, cm.RHO0)
    assert np.array_equal(np.array(gp.shortrange_table), tab[:, 2].astype(np.float32))
    assert np.array_equal(np.array(gp.shortrange_table_potential), tab[:, 1].astype(np.float32))
    assert gp.dx == tab[1, 0]
    # Asmth != 1.5 with the exact window is an error (gravshort-tree2.cpp:42-46)
    with pytest.raises(sq.ShqError):
        sq.make_grav_params(8.0, 1.25, 48, cm.G, cm.RHO0)


def test_grav_params_mirror_reference_ctor():
    """GravTreeParams ctor, gravshort2.hpp:45-54."""
    cm.reference_treepar(ErrTolForceAcc=0.005, MaxBHOpeningAngle=0.9, Rcut=6.0, TreeUseBH=0)
    sq.gravshort_set_softenings(8.0 / 16)
    gp = sq.make_grav_params(8.0, 1.5, 48, cm.G, cm.RHO0)
    assert gp.cellsize == 8.0 / 48
    assert gp.Rcut == 6.0 * 1.5 * (8.0 / 48)
    assert gp.ForceSoftening == 2.8 * (1.0 / 30.0) * (8.0 / 16)
    assert gp.BHOpeningAngle2 == 0.9 * 0.9      # TreeUseBH == 0 -> MaxBHOpeningAngle
    cm.reference_treepar(TreeUseBH=1)
    gp = sq.make_grav_params(8.0, 1.5, 48, cm.G, cm.RHO0)
    assert gp.BHOpeningAngle2 == 0.175 * 0.175


def _apply_accn_numpy(dx, mass, gp):
    """independent float64 restatement for the L0 check"""
    r2 = float(np.dot(dx, dx))
    r = np.sqrt(r2)
    h = gp.ForceSoftening
    fac = mass / (r2 * r)
    if r2 < h * h:
        u = r / h
        if u < 0.5:
            fac = mass / h**3 * (10.666666666667 + u * u * (32.0 * u - 38.4))
        else:
            fac = mass / h**3 * (21.333333333333 - 48.0 * u + 38.4 * u * u - 10.666666666667 * u**3 - 0.066666666667 / u**3)
    i = r / gp.cellsize / gp.dx
    ti = int(np.floor(i))
    if ti >= 511:
        return np.zeros(3)
    t = np.array(gp.shortrange_table, dtype=np.float64)
    fac *= (ti + 1 - i) * t[ti] + (i - ti) * t[ti + 1]
    return dx * fac


def test_apply_accn_l0():
    cm.reference_treepar()
    sq.gravshort_set_softenings(8.0 / 16)
    gp = sq.make_grav_params(8.0, 1.5, 48, cm.G, cm.RHO0)
    rng = np.random.default_rng(3)
    for _ in range(500):
        dx = rng.normal(size=3) * 10 ** rng.uniform(-3, 0.5)
        applied, acc, pot = orc.apply_accn(dx, float(np.dot(dx, dx)), 1.5, gp)
        want = _apply_accn_numpy(dx, 1.5, gp)
        assert np.allclose(acc, want, rtol=1e-13, atol=0)
        assert applied == (1 if np.linalg.norm(dx) / gp.cellsize / gp.dx < 511 else 0)
    # beyond the table (15 cells) nothing is added: gravity.h:52-54
    applied, acc, pot = orc.apply_accn(np.array([15.1 * gp.cellsize, 0, 0]), (15.1 * gp.cellsize) ** 2, 1.0, gp)
    assert applied == 0 and np.all(acc == 0) and pot == 0


def _canon(nodes, firstnode, no):
    """canonical recursive description of a tree: (len, center, sorted particle list | children)"""
    nd = nodes[no - firstnode]
    ct = (nd["flags"] >> 3) & 3
    if ct == 0:
        return ("L", float(nd["len"]), tuple(nd["center"]), tuple(sorted(nd["suns"][: nd["noccupied"]])))
    kids = [c for c in nd["suns"] if c >= 0]
    return ("N", float(nd["len"]), tuple(nd["center"]), tuple(_canon(nodes, firstnode, c) for c in kids))


def _walk_order(nodes, firstnode):
    """follow the threaded walk opening every node; returns visited node list"""
    out = []
    no = firstnode
    while no >= 0:
        nd = nodes[no - firstnode]
        out.append(no)
        ct = (nd["flags"] >> 3) & 3
        no = nd["suns"][0] if ct == 1 else nd["sibling"]
    return out


@pytest.mark.parametrize("kind", ["grid", "random", "close"])
def test_host_tree_equals_reference_insertion_tree(kind):
    """The product's top-down builder must yield the tree the reference's insertion algorithm
    yields (restated in oracle/grav.cpp from forcetree.cpp)."""
    n = 16**3
    if kind == "grid":
        pos = cm.grid_positions(16)
    elif kind == "close":
        pos = cm.close_positions(16)
    else:
        pos = cm.random_positions(orc.boost_mt19937_uniform(0, 3 * n), n)
    pman = cm.make_partmanager(pos)
    tree = sq.force_tree_full(pman)
    hn = tree.Nodes_base
    on, ofirst, ofather = orc.tree_build(pos, pman.Base["Mass"], cm.BOX)
    assert tree.firstnode == ofirst == n
    assert _canon(hn, n, n) == _canon(on, n, n)
    # threaded traversal visits the same sequence of cells
    ho = [(_c["len"], tuple(_c["center"])) for _c in (hn[i - n] for i in _walk_order(hn, n))]
    oo = [(_c["len"], tuple(_c["center"])) for _c in (on[i - n] for i in _walk_order(on, n))]
    assert ho == oo
    # moments: mass conservation and matching centre of mass (tests/test_forcetree.cpp:117-169)
    assert hn[0]["mass"] == n
    hm = {(_c["len"], tuple(_c["center"])): (_c["mass"], tuple(_c["cofm"])) for _c in hn[: tree.numnodes] if _c["father"] >= -1 and _c["mass"] > 0}
    for _c in on:
        if _c["mass"] > 0 and (_c["len"], tuple(_c["center"])) in hm:
            m, cofm = hm[(_c["len"], tuple(_c["center"]))]
            assert m == _c["mass"]
            assert np.allclose(cofm, _c["cofm"], rtol=1e-14, atol=1e-14)


def _oracle_treepm(pos, ErrTol=0.002, Nmesh=48, Rcut=7.0, MaxBH=0.0, nodes=None):
    """do_force_test, tests/test_gravity.cpp:197-247, with the oracle."""
    n = len(pos)
    mass = np.ones(n, dtype=np.float32)
    gpm, pot, _, _ = orc.pm_force(pos, mass, Nmesh, cm.BOX, 1.5, cm.G)
    if nodes is None:
        nodes, first, _ = orc.tree_build(pos, mass, cm.BOX)
    first = n
    cm.reference_treepar(ErrTolForceAcc=ErrTol, MaxBHOpeningAngle=MaxBH, Rcut=Rcut)
    sq.gravshort_set_softenings(cm.BOX / np.cbrt(n))
    gp = sq.make_grav_params(cm.BOX, 1.5, Nmesh, cm.G, cm.RHO0)          # TreeUseBH = 2: BH pass
    oldacc = np.linalg.norm(gpm, axis=1) / cm.G                        # FullTreeGravAccel = 0 initially
    acc, tpot, nint1 = orc.grav_walk(nodes, first, pos, mass, oldacc, gp)
    orc.grav_postprocess(mass, gp, acc, tpot, True)
    cm.reference_treepar(ErrTolForceAcc=ErrTol, MaxBHOpeningAngle=MaxBH, Rcut=Rcut, TreeUseBH=0)   # gravshort-tree2.cpp:170-171
    gp = sq.make_grav_params(cm.BOX, 1.5, Nmesh, cm.G, cm.RHO0)
    oldacc = np.linalg.norm(acc + gpm, axis=1) / cm.G
    acc2, tpot2, nint2 = orc.grav_walk(nodes, first, pos, mass, oldacc, gp)
    orc.grav_postprocess(mass, gp, acc2, tpot2, True)
    return gpm, acc2, (nint1, nint2)


def test_reference_gate_force_flat():
    """tests/test_gravity.cpp:249-292: homogeneous grid => total force ~ 0."""
    pos = cm.grid_positions(16)
    gpm, acc, _ = _oracle_treepm(pos)
    tot = np.abs(gpm + acc)
    assert tot.max() < 0.015
    assert tot.mean() < 0.005


@pytest.mark.parametrize("kind", ["close", "random0", "random1"])
def test_reference_gate_force_vs_direct(kind):
    """tests/test_gravity.cpp:294-355 with check_against_force_direct (:145-169)."""
    n = 16**3
    if kind == "close":
        pos = cm.close_positions(16)
    else:
        k = int(kind[-1])
        u = orc.boost_mt19937_uniform(0, 3 * n, skip=3 * n * k)   # the engine is shared by both iterations (:349-353)
        pos = cm.random_positions(u, n)
    gpm, acc, nint = _oracle_treepm(pos)
    sq.gravshort_set_softenings(cm.BOX / np.cbrt(n))
    direct = orc.force_direct(pos, np.ones(n, dtype=np.float32), cm.BOX, cm.G, sq.FORCE_SOFTENING(), 1)
    meanerr, maxerr = cm.check_accns(direct, gpm + acc)
    assert maxerr < 3 * 0.002, (meanerr, maxerr)
    assert meanerr < 0.8 * 0.002, (meanerr, maxerr)


def test_oracle_walk_relative_criterion_vs_open_tree():
    """runtests.cpp:304-315: tree (relative criterion, MaxBHOpeningAngle 0.9) vs fully open tree:
    mean error <= 1.2 ErrTolForceAcc."""
    n = 16**3
    pos = cm.random_positions(orc.boost_mt19937_uniform(0, 3 * n), n)
    gpm, acc_open, _ = _oracle_treepm(pos, MaxBH=0.0)
    gpm, acc_rel, (n1, n2) = _oracle_treepm(pos, MaxBH=0.9)
    err = cm.force_err(acc_rel, acc_open)
    assert err.mean() <= 1.2 * 0.002
    assert n2.max() < n  # not fully open


def test_oracle_fft_matches_numpy():
    rng = np.random.default_rng(11)
    for N in (12, 48):
        a = rng.normal(size=(N, N, N))
        f = orc.fft_r2c(a)
        ref = np.fft.rfftn(a)
        assert np.abs(f - ref).max() < 1e-12 * np.abs(ref).max()
        back = orc.fft_c2r(f)
        assert np.abs(back - a * N**3).max() < 1e-12 * N**3


def test_oracle_pm_stencil_equals_kspace_difference():
    """one c2r + 4-point real-space difference == three extra c2r with i*K(w) (gravpm.cpp:448-488)."""
    n = 16**3
    pos = cm.random_positions(orc.boost_mt19937_uniform(0, 3 * n), n)
    m = np.ones(n, dtype=np.float32)
    g0, p0, rho, phi = orc.pm_force(pos, m, 48, cm.BOX, 1.5, cm.G, use_stencil=0, want_mesh=True)
    g1, p1, _, _ = orc.pm_force(pos, m, 48, cm.BOX, 1.5, cm.G, use_stencil=1)
    assert abs(rho.sum() - n) < 1e-9        # mass conservation, petapm.cpp:1225-1255
    assert np.abs(g0 - g1).max() < 1e-12 * np.abs(g0).max()
    assert np.abs(p0 - p1).max() < 1e-12 * np.abs(p0).max()   # deposit order (omp atomic) is not fixed
    # fixed-point deposit: quantised at 2^-e, force agrees to ~1e-11
    g2, _, rho2, _ = orc.pm_force(pos, m, 48, cm.BOX, 1.5, cm.G, fixed_point_log2scale=48, use_stencil=1, want_mesh=True)
    assert np.abs(rho2 - rho).max() < 8 * 2.0**-48 * 64
    assert np.abs(g2 - g1).max() < 1e-10 * np.abs(g1).max()


@pytest.mark.parametrize("kind", ["uniform", "cluster"])
def test_space_filling_orders(kind):
    """Both particle orders are permutations; the Peano-Hilbert one (what the reference sorts by,
    domain.cpp:268) is a continuous curve: consecutive particles of a dense sample are neighbours."""
    n, L = 32768, 1.0
    pos = sq.synth_positions(kind, n, L=L)
    for fn in (sq.morton_order, sq.hilbert_order):
        o = fn(pos, L)
        assert np.array_equal(np.sort(o), np.arange(n))
    if kind == "uniform":
        step_h = np.linalg.norm(np.diff(pos[sq.hilbert_order(pos, L)], axis=0), axis=1)
        step_m = np.linalg.norm(np.diff(pos[sq.morton_order(pos, L)], axis=0), axis=1)
        assert step_h.max() < 0.25 * L < step_m.max()  # no long jumps along the Hilbert curve
        assert step_h.mean() < step_m.mean()


def test_toptree_exports_close_the_distributed_walk():
    """Pins oracle/toptree.cpp: primary walk over a tree with pseudo nodes + secondary walks at the exported
    NodeLists visit exactly the interactions of the walk over the undivided tree (gravshort2.hpp:243-322, 362-438),
    and the neighbour variant exports every remote top leaf that holds a particle within the search radius."""
    import shenqi_amd as sq
    n1 = 14
    n = n1**3
    pos = sq.synth_positions("cluster", n, L=cm.BOX)
    pman = cm.make_partmanager(pos)
    full = sq.force_tree_full(pman)
    cm.make_domain(full, ntask=3, me=1, depth=2, pseudo=False)
    dom = sq.force_tree_full(pman)
    tl = cm.make_domain(dom, ntask=3, me=1, depth=2)
    cm.reference_treepar(ErrTolForceAcc=0.005, MaxBHOpeningAngle=0.9, Rcut=6.0, TreeUseBH=0)
    sq.gravshort_set_softenings(cm.BOX / n1)
    gp = sq.make_grav_params(cm.BOX, 1.5, 3 * n1, cm.G, cm.RHO0)
    rng = np.random.default_rng(8)
    oldacc = 10 ** rng.uniform(0, 3, size=n)
    mass = pman.Base["Mass"]
    a_full, _, n_full = orc.grav_walk(full.Nodes_base, full.firstnode, pos, mass, oldacc, gp)
    a_loc, _, n_loc = orc.grav_walk(dom.Nodes_base, dom.firstnode, pos, mass, oldacc, gp)
    counts, table = orc.grav_toptree(dom.Nodes_base, dom.firstnode, dom.lastnode, tl, pos, oldacc, gp)
    assert counts.sum() == len(table) > 0
    # entries of one target are contiguous and in target order
    assert np.array_equal(table["Index"], np.repeat(np.arange(n), counts))
    a2, _, n2 = orc.grav_walk_secondary(full.Nodes_base, full.firstnode, pos, mass, pos[table["Index"]], table["NodeList"],
                                        oldacc[table["Index"]], gp)
    nsum = n_loc.copy()
    np.add.at(nsum, table["Index"], n2)
    assert np.array_equal(nsum, n_full)
    asum = a_loc.copy()
    np.add.at(asum, table["Index"], a2)
    assert np.abs(asum - a_full).max() < 1e-12 * np.abs(a_full).max()
    # neighbour search: no remote neighbour may be missed
    hsml = 0.05 * cm.BOX * (0.5 + rng.random(n))
    counts, table = orc.ngb_toptree(dom.Nodes_base, dom.firstnode, dom.lastnode, tl, pos, hsml, 0, cm.BOX)
    nodes, fn = full.Nodes_base, full.firstnode
    leaf_of = np.full(n, -1)
    for k, no in enumerate(tl["treenode"]):            # particles below each top leaf, by its cell
        nd = nodes[no - fn]
        inside = np.all(np.abs(pos - nd["center"]) <= 0.5 * nd["len"], axis=1)
        leaf_of[inside & (leaf_of < 0)] = k
    start = np.concatenate([[0], np.cumsum(counts)])
    for i in rng.choice(n, size=60, replace=False):
        d = pos - pos[i]
        d -= cm.BOX * np.rint(d / cm.BOX)
        ngb = np.nonzero(np.sum(d * d, axis=1) < hsml[i] ** 2)[0]
        need = {int(tl["treenode"][leaf_of[j]]) for j in ngb if leaf_of[j] >= 0 and tl["Task"][leaf_of[j]] != 1}
        got = {int(x) for x in table["NodeList"][start[i]:start[i + 1]].ravel() if x >= 0}
        assert need <= got


def _stars_in_gas(n1=14, nstar=300, seed=4):
    """Gas on a jittered lattice with a density gradient in its SPH densities, plus star particles at random places."""
    import shenqi_amd as sq
    rng = np.random.default_rng(seed)
    ng = n1**3
    gas = cm.grid_positions(n1) + rng.normal(size=(ng, 3)) * 0.15 * cm.BOX / n1
    gas = np.mod(gas, cm.BOX)
    gas[gas == 0] = 1e-9
    stars = rng.random((nstar, 3)) * cm.BOX
    pos = np.concatenate([gas, stars])
    pman = sq.PartManager(ng + nstar, cm.BOX)
    P = pman.Base
    P["Pos"] = pos
    P["Mass"] = 1.0 + 0.5 * rng.random(ng + nstar)
    P["Type"][:ng] = 0
    P["Type"][ng:] = 4
    P["PI"][:ng] = np.arange(ng)
    P["PI"][ng:] = np.arange(nstar)
    P["Hsml"] = cm.BOX / n1 * rng.uniform(0.8, 3.0, size=ng + nstar)
    SphP = np.zeros(ng, dtype=sq.SPH_DTYPE)
    SphP["Density"] = 1.0 + 3.0 * pos[:ng, 0] / cm.BOX
    SphP["Entropy"] = 1.0
    return pman, SphP, ng, nstar


def test_stellar_density_oracle_against_brute_force():
    """oracle/sph.cpp orc_stellar_density has no reference fixture: check what it must deliver — every star ends with
    DesNumNgb +- MaxNgbDeviation kernel-weighted neighbours at its final Hsml, and StarVolumeSPH is the brute-force sum
    of m_j / rho_j (times w_k) over the gas inside that radius."""
    import shenqi_amd as sq
    pman, SphP, ng, nstar = _stars_in_gas()
    P = pman.Base
    tree = sq.force_tree_rebuild_mask(pman, sq.GASMASK)
    kernel = 1
    des = 4.0 / 3 * np.pi * 2.0**3          # cubic spline, eta = 1
    for weighting in (0, 1):
        st = orc.SphState(P, SphP)
        queue = np.arange(ng, ng + nstar, dtype=np.int32)
        rc, vol, niter, nint = orc.stellar_density(tree.Nodes_base, tree.firstnode, st, queue, cm.BOX, des, 2.0, weighting, kernel)
        assert rc == 0 and 1 < niter < 60 and nint > 0
        out = np.zeros(5)
        for i in queue[:80]:
            h = st.hsml[i]
            d = P["Pos"][:ng] - P["Pos"][i]
            d -= cm.BOX * np.rint(d / cm.BOX)
            r = np.sqrt(np.sum(d * d, axis=1))
            inside = np.nonzero(r < h)[0]
            ngb, v = 0.0, 0.0
            for j in inside:
                orc.lib.orc_density_kernel(kernel, h, r[j] / h, 1.0, orc.ptr(out))
                wk, kvol = out[2], out[1]
                ngb += wk * kvol
                v += P["Mass"][j] / SphP["Density"][j] * (wk if weighting else 1.0)
            assert abs(ngb - des) <= 2.0 + 1e-9, (i, ngb)
            assert abs(v - vol[i]) <= 1e-9 * max(v, 1e-300), (i, v, vol[i])


def _bhs_in_dm(n1=14, nbh=60, seed=6):
    """Dark matter with random velocities and accelerations plus a few black holes; one DM particle is garbage."""
    import shenqi_amd as sq
    rng = np.random.default_rng(seed)
    nd = n1**3
    pos = np.concatenate([rng.random((nd, 3)) * cm.BOX, rng.random((nbh, 3)) * cm.BOX])
    pman = sq.PartManager(nd + nbh, cm.BOX)
    P = pman.Base
    P["Pos"] = pos
    P["Mass"] = 1.0
    P["Type"][:nd] = 1
    P["Type"][nd:] = 5
    P["PI"][nd:] = np.arange(nbh)[::-1]               # slots in another order than the particles
    P["Hsml"] = cm.BOX / n1 * rng.uniform(1.5, 3.5, size=nd + nbh)
    P["Vel"] = rng.normal(size=(nd + nbh, 3)) * 100.0
    P["FullTreeGravAccel"] = rng.normal(size=(nd + nbh, 3)) * 1e3
    P["GravPM"] = rng.normal(size=(nd + nbh, 3)) * 1e2
    P["TimeBinGravity"] = rng.integers(20, 24, size=nd + nbh).astype(np.uint8)
    P["Flags"][5] = 1
    kf = sq.KickFactors()
    kf.FgravkickB = 3e-3
    for b in range(20, 24):
        kf.gravkicks[b] = 1e-3 * (b - 18)
    return pman, kf, nd, nbh


def test_bh_veldisp_oracle_against_brute_force():
    import shenqi_amd as sq
    pman, kf, nd, nbh = _bhs_in_dm()
    P = pman.Base
    tree = sq.force_tree_rebuild_mask(pman, sq.DMMASK)
    st = orc.SphState(P, np.zeros(1, dtype=sq.SPH_DTYPE))
    queue = np.arange(nd, nd + nbh, dtype=np.int32)
    out, vd = orc.bh_veldisp(tree.Nodes_base, tree.firstnode, st, queue, cm.BOX, kf)
    gk = np.array([kf.gravkicks[b] for b in P["TimeBinGravity"][:nd]])
    vp = P["Vel"][:nd] + gk[:, None] * P["FullTreeGravAccel"][:nd] + P["GravPM"][:nd] * kf.FgravkickB
    for q, i in enumerate(queue):
        d = P["Pos"][:nd] - P["Pos"][i]
        d -= cm.BOX * np.rint(d / cm.BOX)
        r2 = np.sum(d * d, axis=1)
        sel = (r2 > 0) & (r2 < P["Hsml"][i] ** 2) & (P["Flags"][:nd] == 0)
        rel = vp[sel] - P["Vel"][i]
        assert out[q, 0] == sel.sum() > 0
        assert np.allclose(out[q, 1:4], rel.sum(axis=0), rtol=1e-11, atol=1e-9)
        assert np.isclose(out[q, 4], (rel * rel).sum(), rtol=1e-12)
        var = (rel * rel).sum() / sel.sum() - np.sum((rel.sum(axis=0) / sel.sum()) ** 2)
        assert np.isclose(vd[q], np.sqrt(var / 3), rtol=1e-10)


def _gas_in_dm(n1=14, ngas=200, seed=12):
    """Dark matter with velocities / accelerations and a few gas particles whose DM neighbourhoods are wanted."""
    pman, kf, nd, _ = _bhs_in_dm(n1=n1, nbh=ngas, seed=seed)
    P = pman.Base
    P["Type"][nd:] = 0
    P["PI"][nd:] = np.arange(ngas)[::-1]
    P["Hsml"][nd:] = cm.BOX / n1 * np.random.default_rng(seed).uniform(0.8, 4.0, size=ngas)
    return pman, kf, nd, ngas


def test_wind_veldisp_oracle_against_brute_force():
    """orc_wind_veldisp: every gas particle ends with 39..41 dark-matter neighbours inside its DMRadius, and VDisp is the
    dispersion of their (Hubble-flow corrected) predicted velocities."""
    import shenqi_amd as sq
    pman, kf, nd, ngas = _gas_in_dm()
    P = pman.Base
    tree = sq.force_tree_rebuild_mask(pman, sq.DMMASK)
    st = orc.SphState(P, np.zeros(ngas, dtype=sq.SPH_DTYPE))
    queue = np.arange(nd, nd + ngas, dtype=np.int32)
    Time, hubble = 0.25, 3.0
    rc, vd, dm, niter = orc.wind_veldisp(tree.Nodes_base, tree.firstnode, st, queue, cm.BOX, kf, Time, hubble)
    assert rc == 0 and 1 < niter < 80 and np.all(np.isfinite(vd))
    gk = np.array([kf.gravkicks[b] for b in P["TimeBinGravity"][:nd]])
    vp = P["Vel"][:nd] + gk[:, None] * P["FullTreeGravAccel"][:nd] + P["GravPM"][:nd] * kf.FgravkickB
    ok = P["Flags"][:nd] == 0
    for q, i in enumerate(queue[:60]):
        d = P["Pos"][i] - P["Pos"][:nd]
        d -= cm.BOX * np.rint(d / cm.BOX)
        r = np.sqrt(np.sum(d * d, axis=1))
        # VDisp belongs to the trial radius that enclosed 39..41 neighbours, i.e. to the 39, 40 or 41 nearest dark-matter
        # particles (the stored DMRadius may have been extrapolated past it)
        r[~ok] = np.inf
        r[r <= 0] = np.inf
        nearest = np.argsort(r)
        cands = []
        for nn in (39, 40, 41):
            sel = nearest[:nn]
            rel = vp[sel] - P["Vel"][i] + hubble * Time * Time * d[sel]
            var = (rel * rel).sum() / nn - np.sum((rel.sum(axis=0) / nn) ** 2)
            cands.append(np.sqrt(var / 3))
        assert any(np.isclose(vd[q], c, rtol=1e-9) for c in cands), (q, vd[q], cands)
        assert dm[q] > 0


def _bhs_in_stars_and_dm(n1=14, nbh=50, seed=15):
    pman, kf, nd, nbh = _bhs_in_dm(n1=n1, nbh=nbh, seed=seed)
    P = pman.Base
    rng = np.random.default_rng(seed + 1)
    P["Type"][:nd] = rng.choice([0, 1, 4], size=nd, p=[0.2, 0.5, 0.3]).astype(np.uint8)
    P["Mass"][:nd] = 1.0 + rng.random(nd)
    P["Potential"] = rng.normal(size=len(P)) * 1e4
    return pman, kf, nd, nbh


def test_bh_dynfric_oracle_against_brute_force():
    import shenqi_amd as sq
    pman, kf, nd, nbh = _bhs_in_stars_and_dm()
    P = pman.Base
    n = len(P)
    queue = np.arange(nd, nd + nbh, dtype=np.int32)
    gk = np.array([kf.gravkicks[b] for b in P["TimeBinGravity"]])
    vp = P["Vel"] + gk[:, None] * P["FullTreeGravAccel"] + P["GravPM"] * kf.FgravkickB
    kout = np.zeros(5)
    for method, mask in ((0, sq.ALLMASK), (1, sq.STARMASK + sq.BHMASK), (2, sq.STARMASK + sq.BHMASK + sq.DMMASK)):
        tree = sq.force_tree_rebuild_mask(pman, mask)
        st = orc.SphState(P, np.zeros(1, dtype=sq.SPH_DTYPE))
        out = orc.bh_dynfric(tree.Nodes_base, tree.firstnode, st, P["Potential"], queue, cm.BOX, kf, method, 1, mask)
        intree = ((1 << P["Type"].astype(np.int64)) & mask) != 0
        intree &= (P["Flags"] & 1) == 0
        for q, i in enumerate(queue[:25]):
            d = P["Pos"][i] - P["Pos"]
            d -= cm.BOX * np.rint(d / cm.BOX)
            r = np.sqrt(np.sum(d * d, axis=1))
            sel = np.nonzero((r < P["Hsml"][i]) & intree)[0]
            j = sel[np.argmin(P["Potential"][sel])]
            assert out[q, 0] == P["Potential"][j] and np.array_equal(out[q, 1:4], P["Pos"][j]) and np.array_equal(out[q, 4:7], P["Vel"][j])
            if method > 0:
                cnt = sel[(P["Type"][sel] == 4) | ((P["Type"][sel] == 1) & (method > 1))]
                dens = 0.0
                svel = np.zeros(3)
                rms = 0.0
                for jj in cnt:
                    orc.lib.orc_density_kernel(1, P["Hsml"][i], r[jj] / P["Hsml"][i], 1.0, orc.ptr(kout))
                    mw = P["Mass"][jj] * kout[2]
                    dens += mw
                    svel += mw * vp[jj]
                    rms += mw * np.sum(vp[jj] ** 2)
                assert np.isclose(out[q, 7], dens, rtol=1e-11) and np.allclose(out[q, 8:11], svel, rtol=1e-9, atol=1e-9 * abs(dens) * 1e3)
                assert np.isclose(out[q, 11], rms, rtol=1e-11)



def _runtests_sequence(walk, pmforce, n):
    """The force checks of runtests.cpp:289-352 (run_gravity_test) with callables for the tree walk and the PM:
    open tree -> default tree (two calls) -> Rcut 9.5 (two calls) -> Nmesh / 2.  Returns the (mean, max) errors of
    check_accns (:126-170: | |F| / |F_open| - 1 | on the TOTAL force GravPM + FullTreeGravAccel)."""
    errtol = 0.002
    gpm = pmforce(48)
    z = np.zeros((n, 3))
    f_open = walk(48, dict(ErrTolForceAcc=0.0, MaxBHOpeningAngle=0.0, Rcut=6.0, TreeUseBH=0), z, gpm)          # :297-300
    pair = gpm + f_open
    par = dict(ErrTolForceAcc=errtol, MaxBHOpeningAngle=0.9, Rcut=6.0, TreeUseBH=0)
    f = walk(48, par, walk(48, par, f_open, gpm), gpm)                                                       # :304-306
    e_def = cm.force_err(gpm + f, pair)
    par95 = dict(par, Rcut=9.5)
    f95 = walk(48, par95, walk(48, par95, f, gpm), gpm)                                                     # :320-323
    e_rcut = cm.force_err(gpm + f95, pair)
    gpm2 = pmforce(24)                                                                                       # :337-343
    f2 = walk(24, par, walk(24, par, f95, gpm2), gpm2)
    e_nmesh = cm.force_err(gpm2 + f2, pair)
    return errtol, e_def, e_rcut, e_nmesh


def check_runtests_gates(errtol, e_def, e_rcut, e_nmesh):
    assert e_def.mean() <= 1.2 * errtol                       # runtests.cpp:314
    assert e_rcut.mean() <= errtol                            # :330 (a larger Rcut must stay within the tolerance)
    assert not (e_nmesh.max() < e_def.max() or e_nmesh.mean() < e_def.mean())     # :351 (half the mesh must not be more accurate)


def test_oracle_runtests_force_gates():
    """runtests.cpp:304-352 on the oracle: default tree vs open tree, larger Rcut, coarser mesh."""
    n = 16**3
    pos = cm.random_positions(orc.boost_mt19937_uniform(0, 3 * n), n)
    mass = np.ones(n, dtype=np.float32)
    nodes, first, _ = orc.tree_build(pos, mass, cm.BOX)

    def pmforce(nmesh):
        return orc.pm_force(pos, mass, nmesh, cm.BOX, 1.5, cm.G)[0]

    def walk(nmesh, par, treeacc, gpm):
        cm.reference_treepar(**par)
        sq.gravshort_set_softenings(cm.BOX / np.cbrt(n))
        gp = sq.make_grav_params(cm.BOX, 1.5, nmesh, cm.G, cm.RHO0)
        acc, pot, _ = orc.grav_walk(nodes, first, pos, mass, np.linalg.norm(treeacc + gpm, axis=1) / cm.G, gp)
        orc.grav_postprocess(mass, gp, acc, pot, True)
        return acc

    check_runtests_gates(*_runtests_sequence(walk, pmforce, n))


def test_oracle_pm_with_scipy_fft_hook():
    """orc_set_fft (what bench.py's cpu_baseline times its PM leg with): scipy's pocketfft behind the oracle's deposit, transfer
    functions and readout gives the oracle's own forces and potentials to rounding, with the hook installed and after it is removed"""
    import common as cm
    n = 12**3
    pos = cm.random_positions(orc.boost_mt19937_uniform(5, 3 * n), n)
    mass = np.ones(n, dtype=np.float32)
    g0, p0, _, _ = orc.pm_force(pos, mass, 24, cm.BOX, 1.5, cm.G)
    orc.use_scipy_fft(2)
    try:
        g1, p1, _, _ = orc.pm_force(pos, mass, 24, cm.BOX, 1.5, cm.G)
    finally:
        orc.use_scipy_fft(0)
    g2, p2, _, _ = orc.pm_force(pos, mass, 24, cm.BOX, 1.5, cm.G)
    assert np.abs(g1 - g0).max() < 1e-12 * np.abs(g0).max() and np.abs(p1 - p0).max() < 1e-12 * np.abs(p0).max()
    # (not bit for bit: the oracle's deposit adds f64 values with omp atomic, as the reference does, in whatever order its threads arrive)
    assert np.abs(g2 - g0).max() < 1e-12 * np.abs(g0).max() and np.abs(p2 - p0).max() < 1e-12 * np.abs(p0).max()
