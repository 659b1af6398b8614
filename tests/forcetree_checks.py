"""The invariants of the reference's tests/test_forcetree.cpp (check_tree :111-168, check_moments :21-109, check_hmax :237-253,
root hmax gate :369), restated in numpy for a reference-format NODE array (the host mirror's tree or the device-built one
after shq_tree_download).  The reference runs check_tree on the tree before the moments pass removes its empty leaves; the
trees here are final, so the child checks apply to the children that are there."""
import numpy as np

NMAXCHILD = 8


def ranlux48_uniform(seed, n, discard=100):
    """boost::random::ranlux48(seed) through uniform_real_distribution<double>(0, 1), as set_random_numbers does
    (libgadget/utils/system.cpp:64-84): 48-bit draws / 2^48 after `discard` draws."""
    # subtract_with_carry_engine<uint64, 48, 5, 12> seeded by the LCG 40014 x mod 2147483563, two 32-bit words per state word
    M = 1 << 48
    e = seed if seed != 0 else 19780503
    e %= 2147483563

    def lcg():
        nonlocal e
        e = (40014 * e) % 2147483563
        return e

    x = []
    for _ in range(12):
        lo = lcg()
        hi = lcg()
        x.append((lo + (hi << 32)) % M)
    carry = 1 if x[11] == 0 else 0
    k = 0

    def base():
        nonlocal carry, k
        ps = (k + 12 - 5) % 12
        xi = x[ps] - x[k] - carry
        if xi < 0:
            xi += M
            carry = 1
        else:
            carry = 0
        x[k] = xi
        k = (k + 1) % 12
        return xi

    out = []
    used = 0
    for _ in range(discard + n):
        if used >= 11:                 # discard_block_engine<..., 389, 11>
            for _ in range(389 - 11):
                base()
            used = 0
        out.append(base() / float(M))
        used += 1
    return np.array(out[discard:])


def get_father(nodes, firstnode, father, idx):
    """force_get_father: particles through Father, nodes through their father link"""
    idx = np.asarray(idx)
    out = np.full(idx.shape, -1, dtype=np.int64)
    isp = (idx >= 0) & (idx < firstnode)
    out[isp] = father[idx[isp]]
    isn = idx >= firstnode
    out[isn] = nodes["father"][idx[isn] - firstnode]
    return out


def check_tree(nodes, firstnode, father, pos):
    """check_tree (test_forcetree.cpp:111-168)"""
    nn = len(nodes)
    numpart = len(pos)
    ctype = (nodes["flags"] >> 3) & 3
    leaf = ctype == 0
    internal = ctype == 1
    seen = np.zeros(numpart, dtype=np.int64)
    assert np.all(nodes["noccupied"][leaf] <= NMAXCHILD) and np.all(nodes["noccupied"][leaf] >= 1)   # empty leaves are removed by the moments pass
    for c in range(NMAXCHILD):
        sel = leaf & (nodes["noccupied"] > c)
        child = nodes["suns"][sel, c]
        assert np.all(child >= 0) and np.all(child < firstnode)
        np.add.at(seen, child, 1)
        assert np.all(father[child] == firstnode + np.nonzero(sel)[0])
    assert np.all(seen == 1), "every particle sits in exactly one leaf"
    for c in range(8):
        sel = internal & (nodes["suns"][:, c] >= 0)
        child = nodes["suns"][sel, c].astype(np.int64)
        assert np.all(child >= firstnode) and np.all(child < firstnode + nn)
        ch = nodes[child - firstnode]
        par = nodes[sel]
        assert np.all(np.abs(ch["len"] / par["len"] - 0.5) < 1e-4)
        # the child's centre lies in one octant of the parent: +- len/4 per axis, and the octant index orders the children
        d = ch["center"] - par["center"]
        assert np.all(np.abs(np.abs(d) - 0.25 * par["len"][:, None]) < 1e-9 * par["len"][:, None])
    # the surviving children of a node come in octant order (bit k set <=> centre above the parent's along k)
    cen = nodes["center"]
    octant = np.full((nn, 8), -1, dtype=np.int64)
    for c in range(8):
        sel = internal & (nodes["suns"][:, c] >= 0)
        child = nodes["suns"][sel, c].astype(np.int64) - firstnode
        d = cen[child] - cen[sel]
        octant[sel, c] = (d[:, 0] > 0) + 2 * (d[:, 1] > 0) + 4 * (d[:, 2] > 0)
    for c in range(7):
        both = (octant[:, c] >= 0) & (octant[:, c + 1] >= 0)
        assert np.all(octant[both, c] < octant[both, c + 1])
    return int(leaf.sum() + internal.sum())


def check_moments(nodes, firstnode, father, mass, BoxSize, nrealnode):
    """check_moments (test_forcetree.cpp:21-109): masses add up along the father chains, siblings are siblings or children
    of an ancestor, centres of mass lie in the box"""
    nn = len(nodes)
    resid = nodes["mass"].astype(np.float64).copy()
    f = father.astype(np.int64).copy()
    m = mass.astype(np.float64)
    assert np.all(f >= firstnode) and np.all(f < firstnode + nn)
    alive = np.ones(len(f), dtype=bool)
    while alive.any():
        np.subtract.at(resid, f[alive] - firstnode, m[alive])
        f[alive] = nodes["father"][f[alive] - firstnode]
        alive &= f >= 0
        assert np.all(f[alive] >= firstnode) and np.all(f[alive] < firstnode + nn)
    assert np.all(np.abs(resid) < 0.5), "a node's mass is the mass of the particles below it"
    sib = nodes["sibling"].astype(np.int64)
    assert np.all(sib >= -1) and np.all(sib < firstnode + nn)
    fath = nodes["father"].astype(np.int64)
    has = sib >= 0
    sfather = np.full(nn, -1, dtype=np.int64)
    sfather[has] = fath[sib[has] - firstnode]
    ok = ~has | (sfather == fath)
    anc = fath.copy()
    for _ in range(64):
        todo = ~ok & (anc >= 0)
        if not todo.any():
            break
        anc[todo] = fath[anc[todo] - firstnode]
        ok |= todo & (anc == sfather)
    assert np.all(ok), "a sibling is a true sibling or the child of an ancestor"
    assert (~has).sum() < max(1, nn // 100) or nn < 200
    assert np.all(nodes["cofm"] <= BoxSize) and np.all(nodes["cofm"] >= 0)
    assert nn <= nrealnode


def check_hmax(nodes, firstnode, father, pos, hsml):
    """check_hmax (test_forcetree.cpp:237-253): every particle lies inside all its ancestors and within their hmax"""
    j = father.astype(np.int64).copy()
    idx = np.arange(len(pos))
    alive = j >= 0
    assert np.all(nodes["hmax"] >= 0)
    while alive.any():
        nd = nodes[j[alive] - firstnode]
        p = pos[idx[alive]]
        d = np.abs(p - nd["center"])
        assert np.all(d <= 0.5 * nd["len"][:, None]), "particle outside an ancestor"
        dist = (d + hsml[idx[alive], None] - 0.5 * nd["len"][:, None]).max(axis=1)
        assert np.all(dist <= nd["hmax"] + 1e-5)
        j[alive] = nd["father"]
        alive = j >= 0
