"""CPU restatement (numpy / scipy) of the reference's friends-of-friends group finder for one task.

TEST INFRASTRUCTURE ONLY.  Follows /root/reference/libgadget/fof.cpp:
  fof_label_primary / fofp_merge / fof_primary_ngbiter   :368-581   the END STATE of the union-find: connected components of
        "r2 <= LinkL^2" among the primary types (neighbour test of treewalk_visit_ngbiter, treewalk.c:946-961), labelled with
        the smallest particle ID of the component
  fof_label_secondary and its ngbiter / postprocess       :1142-1270 nearest primary particle within a search radius that starts
        at max(0.4 LinkL, 0.5 Hsml) (float) and doubles while it is below 4 LinkL
  fof_fof, fof_compile_base, fof_assign_grnr               :159-256, 710-766, 1048-1096 (one task)
  add_particle_to_group, fof_finish_group_properties       :583-705
The reference sorts HaloLabel by MinID with an unstable sort, so which member of a group comes first (FirstPos) and the order
the members are added in are unspecified there; this restatement fixes them as "lowest particle index first" (a stable sort).
Pinned by the reference's own fixtures tests/test_fof.cpp (test_fof_line, test_fof_halos): tests/test_fof_cpu.py runs both
particle set-ups through this file and checks every BOOST_TEST of theirs."""
import numpy as np
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components
from scipy.spatial import cKDTree

LARGE = 1e29


def nearest(x, box):
    """NEAREST, partmanager.h:99"""
    return np.where(x > 0.5 * box, x - box, np.where(x < -0.5 * box, x + box, x))


def pair_r2(pi, pj, box):
    """r2 as treewalk_visit_ngbiter accumulates it (treewalk.c:953-958); rows of positions"""
    d = nearest(pi - pj, box)
    return d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]


def label_primary(pos, ids, types, dead, box, linkl, primary_mask):
    """MinID per particle after fof_label_primary (non-primary and dead particles keep their own ID)"""
    n = len(pos)
    minid = ids.astype(np.uint64).copy()
    prim = np.flatnonzero((((1 << types.astype(np.int64)) & primary_mask) != 0) & ~dead)
    if len(prim) == 0:
        return minid
    tree = cKDTree(np.mod(pos[prim], box), boxsize=box)
    pairs = tree.query_pairs(linkl * (1 + 1e-9), output_type="ndarray")
    if len(pairs):
        r2 = pair_r2(pos[prim[pairs[:, 0]]], pos[prim[pairs[:, 1]]], box)
        pairs = pairs[r2 <= linkl * linkl]
    g = coo_matrix((np.ones(len(pairs), dtype=np.int8), (pairs[:, 0], pairs[:, 1])), shape=(len(prim), len(prim)))
    ncomp, comp = connected_components(g, directed=False)
    low = np.full(ncomp, np.iinfo(np.uint64).max, dtype=np.uint64)
    np.minimum.at(low, comp, ids[prim].astype(np.uint64))
    minid[prim] = low[comp]
    return minid


def label_secondary(pos, types, dead, hsml, minid, box, linkl, primary_mask, secondary_mask):
    """fof_label_secondary: MinID of the nearest primary particle for every secondary-type particle that finds one.
    Returns the updated MinID array and the number of attached particles."""
    n = len(pos)
    minid = minid.copy()
    prim = np.flatnonzero((((1 << types.astype(np.int64)) & primary_mask) != 0) & ~dead)
    sec = np.flatnonzero((((1 << types.astype(np.int64)) & secondary_mask) != 0) & ~dead)
    attached = 0
    if len(prim) == 0 or len(sec) == 0:
        return minid, 0
    tree = cKDTree(np.mod(pos[prim], box), boxsize=box)
    for p in sec:
        h = np.float32(0.4 * linkl)
        if types[p] in (0, 4, 5) and h < 0.5 * hsml[p]:
            h = np.float32(0.5 * hsml[p])
        while True:
            cand = tree.query_ball_point(np.mod(pos[p], box), float(h) * (1 + 1e-9))
            best, bestr = -1, LARGE
            if cand:
                cand = np.array(sorted(cand))
                r2 = pair_r2(np.repeat(pos[p][None, :], len(cand), 0), pos[prim[cand]], box)
                ok = r2 <= float(h) * float(h)
                if ok.any():
                    r = np.sqrt(r2[ok])
                    k = int(np.argmin(r))
                    best, bestr = int(prim[cand[ok][k]]), float(r[k])
            if best >= 0:
                minid[p] = minid[best]
                attached += 1
                break
            if h < 4 * linkl:
                h = np.float32(h * np.float32(2.0))
            else:
                break
    return minid, attached


def crossproduct(a, b):
    """densitykernel.h:63-75"""
    return np.array([a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1]])


def catalogue(pos, vel, mass, types, minid, box, minlength, density=None, decoupled=None):
    """fof_compile_base + fof_assign_grnr + fof_compile_catalogue + fof_finish_group_properties for one task.
    Returns (groups: list of dicts ordered by MinID, GrNr per particle)."""
    n = len(pos)
    order = np.argsort(minid, kind="stable")
    sm = minid[order]
    starts = np.flatnonzero(np.concatenate([[True], sm[1:] != sm[:-1]]))
    lens = np.diff(np.concatenate([starts, [n]]))
    keep = lens >= minlength
    starts, lens = starts[keep], lens[keep]
    ng = len(starts)
    # GrNr: by decreasing length, then MinID (fof_radix_Group_TotalCountTaskDiffMinID, one task)
    rank = np.lexsort((sm[starts], -lens.astype(np.int64)))
    grnr = np.empty(ng, dtype=np.int64)
    grnr[rank] = np.arange(1, ng + 1)
    part_grnr = np.full(n, -1, dtype=np.int64)
    groups = []
    for g in range(ng):
        members = order[starts[g]:starts[g] + lens[g]]
        part_grnr[members] = grnr[g]
        first = pos[members[0]].astype(np.float32).astype(np.float64)       # BaseGroup.FirstPos is float[3]
        G = dict(MinID=int(sm[starts[g]]), Length=0, GrNr=int(grnr[g]), LenType=[0] * 6, MassType=[0.0] * 6, Mass=0.0, CM=np.zeros(3), Vel=np.zeros(3),
                 Imom=np.zeros((3, 3)), Jmom=np.zeros(3), MaxDens=0.0, seed_index=-1, FirstPos=first.astype(np.float32))
        for i in members:
            m = float(mass[i])
            t = int(types[i])
            G["Length"] += 1
            G["Mass"] += m
            G["LenType"][t] += 1
            G["MassType"][t] += m
            if t == 0 and density is not None and not (decoupled is not None and decoupled[i]):
                if density[i] > G["MaxDens"]:
                    G["MaxDens"] = float(density[i])
                    G["seed_index"] = int(i)
            rel = nearest(pos[i] - first, box)
            xyz = rel + first
            jm = crossproduct(rel, vel[i])
            for d1 in range(3):
                G["CM"][d1] += m * xyz[d1]
                G["Vel"][d1] += m * vel[i][d1]
                G["Jmom"][d1] += m * jm[d1]
                for d2 in range(3):
                    G["Imom"][d1][d2] += m * rel[d1] * rel[d2]
        # fof_finish_group_properties
        vcm = np.zeros(3)
        cm = np.zeros(3)
        rel = np.zeros(3)
        for d1 in range(3):
            G["Vel"][d1] /= G["Mass"]
            vcm[d1] = G["Vel"][d1]
            cm[d1] = G["CM"][d1] / G["Mass"]
            rel[d1] = float(nearest(np.float64(cm[d1] - first[d1]), box))
            c = cm[d1]
            while c >= box:
                c -= box
            while c < 0:
                c += box
            G["CM"][d1] = c
        jcm = crossproduct(rel, vcm)
        for d1 in range(3):
            G["Jmom"][d1] -= jcm[d1] * G["Mass"]
        for d1 in range(3):
            for d2 in range(3):
                G["Imom"][d1][d2] -= G["Mass"] * (rel[d1] * rel[d2])
        groups.append(G)
    return groups, part_grnr


def fof(pos, vel, mass, types, ids, dead, hsml, box, linkl, minlength, primary_mask=2, secondary_mask=1 + 16 + 32, density=None, decoupled=None):
    """fof_fof for one task: (MinID per particle, groups, GrNr per particle)"""
    minid = label_primary(pos, ids, types, dead, box, linkl, primary_mask)
    minid, _ = label_secondary(pos, types, dead, hsml, minid, box, linkl, primary_mask, secondary_mask)
    groups, part_grnr = catalogue(pos, vel, mass, types, minid, box, minlength, density, decoupled)
    return minid, groups, part_grnr


def seed_marks(groups, min_fof_mass, min_mstar):
    """the marking loop of fof_seed, fof.cpp:1290-1302 (one task): seed_index of the marked groups in catalogue order"""
    return [G["seed_index"] for G in groups
            if G["Mass"] >= min_fof_mass and G["MassType"][4] >= min_mstar and G["LenType"][5] == 0 and G["seed_index"] >= 0]
