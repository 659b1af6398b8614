"""Shared set-up of the reference's force tests (tests/test_gravity.cpp:25-247)."""
import numpy as np

import shenqi_amd as sq

G = 43.0071                      # tests/test_gravity.cpp:23
BOX = 8.0                        # :30
OMEGA0, HUBBLE = 0.3, 0.1        # CP.Omega0 (:211) and CP.Hubble in internal units (init_cosmology)
RHO0 = OMEGA0 * 3 * HUBBLE * HUBBLE / (8 * np.pi * G)   # :223


def grid_positions(ncbrt, box=BOX):
    i = np.arange(ncbrt**3)
    pos = np.empty((ncbrt**3, 3))
    pos[:, 0] = (box / ncbrt) * (i // ncbrt // ncbrt)
    pos[:, 1] = (box / ncbrt) * ((i // ncbrt) % ncbrt)
    pos[:, 2] = (box / ncbrt) * (i % ncbrt)
    return pos


def close_positions(ncbrt, close=5000.0):
    """tests/test_gravity.cpp:294-314 test_force_close"""
    i = np.arange(ncbrt**3)
    pos = np.empty((ncbrt**3, 3))
    pos[:, 0] = 4.0 + (i // ncbrt // ncbrt) / close
    pos[:, 1] = 4.0 + ((i // ncbrt) % ncbrt) / close
    pos[:, 2] = 4.0 + (i % ncbrt) / close
    return pos


def random_positions(u, numpart, box=BOX):
    """tests/test_gravity.cpp:316-341 do_random_test; u = 3*numpart uniform variates in draw order."""
    u = u.reshape(numpart, 3)
    pos = np.empty((numpart, 3))
    a, b = numpart // 4, 3 * numpart // 4
    pos[:a] = box * u[:a]
    pos[a:b] = box / 2 + box / 8 * np.exp((u[a:b] - 0.5) ** 2)
    pos[b:] = box * 0.1 + box / 32 * np.exp((u[b:] - 0.5) ** 2)
    return pos


def make_partmanager(pos, box=BOX, mass=1.0, ptype=1):
    """setup_particles, tests/test_gravity.cpp:174-195"""
    n = len(pos)
    pm = sq.PartManager(n, box)
    P = pm.Base
    P["Pos"] = pos
    P["Type"] = ptype
    P["Mass"] = mass
    P["ID"] = np.arange(1, n + 1)
    return pm


def reference_treepar(ErrTolForceAcc=0.002, MaxBHOpeningAngle=0.0, Rcut=7.0, TreeUseBH=2):
    """treeacc of tests/test_gravity.cpp:225-236 (MaxBHOpeningAngle is zero-initialised there)."""
    sq.set_gravshort_treepar(ErrTolForceAcc=ErrTolForceAcc, BHOpeningAngle=0.175, MaxBHOpeningAngle=MaxBHOpeningAngle,
                             TreeUseBH=TreeUseBH, Rcut=Rcut, FractionalGravitySoftening=1.0 / 30.0,
                             ShortRangeForceWindowType=sq.SHORTRANGE_FORCE_WINDOW_TYPE_EXACT)


def check_accns(pair, total):
    """check_accns + find_means, tests/test_gravity.cpp:78-119: errors in units of mean |a_direct|."""
    meanacc = np.abs(pair).mean()
    err = np.abs(pair - total) / meanacc
    return err.mean(), err.max()


def force_err(a, ref):
    """runtests.cpp:126-170: | |F|/|F_ref| - 1 | per particle."""
    na = np.linalg.norm(a, axis=1)
    nr = np.linalg.norm(ref, axis=1)
    ok = nr > 0
    return np.abs(na[ok] / nr[ok] - 1)


# ---- SPH set-up (tests/test_density.cpp:28-132) ---------------------------------------------
def density_params(BoxSize=BOX, kernel=1, eta=1.0, MaxNumNgbDeviation=0.5, BlackHoleNgbFactor=2.0, update_hsml=1,
                   DoEgyDensity=0, BlackHoleOn=0, MinGasHsml=None):
    """setup_density, tests/test_density.cpp:28-70: cubic kernel, eta 1, MinGasHsml = 0.006 * (FORCE_SOFTENING()/2.8)
    with FractionalGravitySoftening = 1 and mean separation 1."""
    from shenqi_amd import capi
    support = {1: 4, 2: 6, 4: 5}[kernel]
    des = 4.0 / 3 * np.pi * (support / 2.0 * eta) ** 3          # DensityKrnl::desnumngb, densitykernel.hpp:36-41
    dp = capi.DensityParams()
    dp.BoxSize = BoxSize
    dp.DesNumNgb = des
    dp.DesNumNgbBH = des * BlackHoleNgbFactor
    dp.MinGasHsml = 0.006 * 1.0 if MinGasHsml is None else MinGasHsml
    dp.MaxNumNgbDeviation = MaxNumNgbDeviation
    dp.update_hsml, dp.BlackHoleOn, dp.DoEgyDensity, dp.WindsDecouple = update_hsml, BlackHoleOn, DoEgyDensity, 0
    dp.DensityKernelType = kernel
    return dp                                                    # kick factors all zero: DriftKickTimes kick = {0}


def hydro_params(BoxSize=BOX, atime=0.1, hubble=HUBBLE, kernel=1, DensityIndependentSphOn=1, DensityContrastLimit=100.0,
                 ArtBulkViscConst=0.75):
    """HydroPriv ctor, hydratree2.hpp:83-119, with set_hydropar{1, 100, 0.75}."""
    from shenqi_amd import capi
    g = 5.0 / 3.0
    hp = capi.HydroParams()
    hp.BoxSize, hp.atime = BoxSize, atime
    hp.fac_mu = atime ** (3 * (g - 1) / 2) / atime
    hp.fac_vsic_fix = hubble * atime ** (3 * (g - 1))
    hp.hubble_a2 = hubble * atime * atime
    hp.ArtBulkViscConst, hp.DensityContrastLimit = ArtBulkViscConst, DensityContrastLimit
    hp.DensityIndependentSphOn, hp.DensityKernelType = DensityIndependentSphOn, kernel
    return hp


def make_gas(pos, hsml, box=BOX, lastisbh=False):
    """setup_density_particles, tests/test_density.cpp:98-132: Mass 1, Vel 1.5, Entropy 1, Density 1."""
    n = len(pos)
    pm = sq.PartManager(n, box)
    P = pm.Base
    P["Pos"] = pos
    P["Mass"] = 1.0
    P["Hsml"] = hsml
    P["Vel"] = 1.5
    P["ID"] = np.arange(1, n + 1)
    ngas = n - (1 if lastisbh else 0)
    P["Type"][:ngas] = 0
    P["PI"][:ngas] = np.arange(ngas)
    if lastisbh:
        P["Type"][ngas:] = 5
        P["PI"][ngas:] = 0
    SphP = np.zeros(ngas, dtype=sq.SPH_DTYPE)
    SphP["Entropy"] = 1
    SphP["Density"] = 1
    BhP = np.zeros(2, dtype=[("Density", "<f8"), ("DivVel", "<f8")])
    return pm, SphP, BhP


def density_close_positions(ncbrt=32, box=BOX, close=500.0):
    """tests/test_density.cpp:240-262 test_density_close (last particle is a black hole)."""
    numpart = ncbrt**3
    pos = np.empty((numpart, 3))
    hsml = np.empty(numpart)
    i = np.arange(numpart // 4)
    hsml[: numpart // 4] = 4 * box / np.cbrt(numpart / 8)
    pos[: numpart // 4, 0] = (box / ncbrt) * (i / (ncbrt / 2.0) / (ncbrt / 2.0))
    pos[: numpart // 4, 1] = (box / ncbrt) * ((i * 2 // ncbrt) % (ncbrt // 2))
    pos[: numpart // 4, 2] = (box / ncbrt) * (i % (ncbrt // 2))
    i = np.arange(numpart // 4, numpart)
    hsml[numpart // 4:] = 2 * ncbrt / close
    pos[numpart // 4:, 0] = 4.1 + (i // ncbrt // ncbrt) / close
    pos[numpart // 4:, 1] = 4.1 + ((i // ncbrt) % ncbrt) / close
    pos[numpart // 4:, 2] = 4.1 + (i % ncbrt) / close
    return pos, hsml


def make_domain(tree, ntask=3, me=1, depth=2, pseudo=True):
    """Give a single-domain host tree the top tree a domain decomposition would (forcetree.cpp:1155-1206):
    the nodes down to `depth` below the root become TopLevel (internal ones InternalTopLevel too), the deepest
    of them (or shallower leaves) are the top leaves, dealt to `ntask` ranks in contiguous pre-order runs.
    With pseudo=True the leaves of other ranks than `me` become pseudo nodes (ChildType 2, suns[0] = lastnode + leaf index):
    the tree rank `me` would hold; with pseudo=False only the flags are set: the tree a remote rank's secondary walk
    sees.  Returns the TopLeaves table (capi.TOPLEAF_DTYPE)."""
    from shenqi_amd import capi
    nodes = tree.Nodes_base
    fn, ln = tree.firstnode, tree.lastnode
    leaves = []

    def visit(no, d):
        nd = nodes[no - fn]
        ctype = (nd["flags"] >> 3) & 3
        if ctype == 1 and d < depth:
            nodes["flags"][no - fn] |= 3          # InternalTopLevel | TopLevel
            for s in nd["suns"]:
                if s >= fn:
                    visit(int(s), d + 1)
        else:
            nodes["flags"][no - fn] |= 2          # TopLevel leaf
            leaves.append(no)

    visit(fn, 0)
    tl = np.zeros(len(leaves), dtype=capi.TOPLEAF_DTYPE)
    for k, no in enumerate(leaves):
        tl[k] = ((k * ntask) // len(leaves), k, no)
        if pseudo and tl["Task"][k] != me:
            f = int(nodes["flags"][no - fn])
            nodes["flags"][no - fn] = (f & ~(3 << 3)) | (2 << 3)
            nodes["suns"][no - fn, 0] = ln + k
    return tl


def make_topnodes(rng, ntask, maxdepth=3, psplit=0.6, shuffle=True):
    """A top tree as a domain decomposition would hand it over (struct topnode_data, domain.h:12-18, reduced to geometry):
    TopNodes refined at random down to `maxdepth`, the root always; the eight daughters of a node are consecutive table
    entries (Daughter + sub) in a shuffled octant assignment, as the Peano-Hilbert numbering of the reference shuffles them;
    top leaves numbered in table order and dealt to `ntask` tasks in contiguous runs (Tasks[].StartLeaf / EndLeaf).
    Returns (geo: capi.TOPNODE_GEO_DTYPE, topleaves: capi.TOPLEAF_DTYPE)."""
    from shenqi_amd import capi
    nodes = [dict(d=[-1] * 8, leaf=-1, depth=0)]
    queue = [0]
    while queue:
        t = queue.pop(0)
        nd = nodes[t]
        if nd["depth"] == 0 or (nd["depth"] < maxdepth and rng.random() < psplit):
            base = len(nodes)
            perm = rng.permutation(8) if shuffle else np.arange(8)
            for s in range(8):
                nodes.append(dict(d=[-1] * 8, leaf=-1, depth=nd["depth"] + 1))
            for octant in range(8):
                nd["d"][octant] = base + int(perm[octant])
            queue.extend(range(base, base + 8))
    nleaf = 0
    for nd in nodes:
        if nd["d"][0] < 0:
            nd["leaf"] = nleaf
            nleaf += 1
    geo = np.zeros(len(nodes), dtype=capi.TOPNODE_GEO_DTYPE)
    for t, nd in enumerate(nodes):
        geo["daughter"][t] = nd["d"]
        geo["leaf"][t] = nd["leaf"]
    tl = np.zeros(nleaf, dtype=capi.TOPLEAF_DTYPE)
    tl["Task"] = (np.arange(nleaf) * ntask) // nleaf
    tl["topnode"] = [t for t, nd in enumerate(nodes) if nd["d"][0] < 0]
    tl["treenode"] = -1
    return geo, tl


def topleaf_of(pos, geo, box):
    """the top leaf every position falls in: descent from the root cell (centre Box/2, len 1.001 Box) by `pos > centre`, child
    centres centre +- len/4 (get_subnode / init_internal_node, forcetree.cpp:277-328)"""
    out = np.empty(len(pos), dtype=np.int32)
    for i, p in enumerate(pos):
        t, c, ln = 0, [box / 2.0] * 3, box * 1.001
        while geo["daughter"][t][0] >= 0:
            s = int(p[0] > c[0]) + (int(p[1] > c[1]) << 1) + (int(p[2] > c[2]) << 2)
            lh = 0.25 * ln
            c = [c[j] + (lh if (s >> j) & 1 else -lh) for j in range(3)]
            ln = 0.5 * ln
            t = int(geo["daughter"][t][s])
        out[i] = geo["leaf"][t]
    return out
