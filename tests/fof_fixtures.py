"""The particle set-ups of the reference's FOF tests (libgadget/tests/test_fof.cpp:69-99 and :152-300, one task) and their
expectations, as data for the oracle (CPU) and the GPU parity tests."""
import numpy as np

NHALO, NSINGLE, NPAIR, NBGGRID = 64, 64, 64, 16


def line(numpart=512 * 512, box=20000.0):
    """setup_particles, test_fof.cpp:69-99: a wrapped diagonal line.  fof_init(BoxSize / cbrt(NumPart)), linking length 0.2"""
    ids = np.arange(1, numpart + 1, dtype=np.uint64)
    pos = np.empty((numpart, 3))
    vel = np.empty((numpart, 3))
    for j in range(3):
        p = box * (j + 1) * ids.astype(np.float64) / numpart          # BoxSize * (j+1) * ID / (NumPart * NTask), left to right
        for _ in range(4):
            p = np.where(p > box, p - box, p)
        pos[:, j] = p
        vel[:, j] = j + 1
    linkl = 0.2 * (box / np.cbrt(numpart))
    return dict(pos=pos, vel=vel, mass=np.ones(numpart), types=np.ones(numpart, dtype=np.uint8), ids=ids, box=box, linkl=linkl, minlength=5)


def halo_size(h):
    return 5 + h


def halo_first(h):
    return 5 * h + (h * (h - 1)) // 2


def halos(box=8000.0):
    """test_fof_halos, test_fof.cpp:205-300: 64 halos of 5..68 members on a 4-kpc sub-grid, 64 singles, 64 pairs, a 16^3 background
    grid; fof_init(100) => linking length 20."""
    nhalopart = halo_first(NHALO)
    nglobal = nhalopart + NSINGLE + 2 * NPAIR + NBGGRID ** 3
    gpos = np.zeros((nglobal, 3))
    expcm = np.zeros((NHALO, 3))
    for h in range(NHALO):
        center = np.array([(box / 4.) * (h % 4), (box / 4.) * ((h // 4) % 4), (box / 4.) * (h // 16)])
        for k in range(halo_size(h)):
            off = np.array([4. * (k % 4) - 6., 4. * ((k // 4) % 4) - 6., 4. * (k // 16) - 6.])
            expcm[h] += off / halo_size(h)
            pp = center + off
            pp = np.where(pp < 0, pp + box, pp)
            pp = np.where(pp >= box, pp - box, pp)
            gpos[halo_first(h) + k] = pp
        expcm[h] += center
        expcm[h] = np.where(expcm[h] < 0, expcm[h] + box, expcm[h])
        expcm[h] = np.where(expcm[h] >= box, expcm[h] - box, expcm[h])
    for i in range(NSINGLE):
        gpos[nhalopart + i] = [(box / 4.) * (i % 4) + box / 8., (box / 4.) * ((i // 4) % 4) + box / 8., (box / 4.) * (i // 16) + box / 8.]
    for i in range(NPAIR):
        for j in range(2):
            gpos[nhalopart + NSINGLE + 2 * i + j] = [(box / 4.) * (i % 4) + box / 8., (box / 4.) * ((i // 4) % 4) + box / 8., (box / 4.) * (i // 16) + 5. * j]
    for i in range(NBGGRID ** 3):
        gpos[nhalopart + NSINGLE + 2 * NPAIR + i] = [(box / NBGGRID) * (i % NBGGRID) + box / (2. * NBGGRID),
                                                    (box / NBGGRID) * ((i // NBGGRID) % NBGGRID) + box / (2. * NBGGRID),
                                                    (box / NBGGRID) * (i // (NBGGRID * NBGGRID)) + box / (2. * NBGGRID)]
    ids = np.arange(1, nglobal + 1, dtype=np.uint64)
    halo_of = np.full(nglobal, -1)
    for h in range(NHALO):
        halo_of[halo_first(h):halo_first(h + 1)] = h
    vel = np.zeros((nglobal, 3))
    for j in range(3):
        vel[:, j] = np.where(halo_of >= 0, (j + 1) * (halo_of + 1), 0.0)
    return dict(pos=gpos, vel=vel, mass=np.full(nglobal, 1.5), types=np.ones(nglobal, dtype=np.uint8), ids=ids, box=box, linkl=0.2 * 100, minlength=5,
                expcm=expcm, halo_of=halo_of)


def periodic_dist(a, b, box):
    d = abs(a - b)
    return box - d if d > box / 2 else d


def check_line(groups, part_grnr, fx):
    """test_fof.cpp:125-149"""
    n = len(fx["pos"])
    assert len(groups) == 1
    g = groups[0]
    assert g["Length"] == n and g["GrNr"] == 1 and g["MinID"] == 1
    assert g["LenType"][1] == n and g["LenType"][0] == 0
    assert abs(g["Mass"] - n) < 1e-3 * n and abs(g["MassType"][1] - g["Mass"]) < 1e-6 * g["Mass"]
    for j in range(3):
        assert abs(g["Vel"][j] - (j + 1)) < 1e-5
    assert (part_grnr == 1).all()


def check_halos(groups, part_grnr, fx):
    """test_fof.cpp:310-361"""
    assert len(groups) == NHALO
    grnr_of_halo = [-1] * NHALO
    found = [0] * NHALO
    for g in groups:
        hh = int(fx["halo_of"][g["MinID"] - 1])
        assert hh >= 0 and g["MinID"] == halo_first(hh) + 1
        found[hh] += 1
        grnr_of_halo[hh] = g["GrNr"]
        assert g["Length"] == halo_size(hh) and g["LenType"][1] == halo_size(hh) and g["LenType"][0] == 0
        assert abs(g["Mass"] - 1.5 * halo_size(hh)) < 1e-6 * g["Mass"] and abs(g["MassType"][1] - g["Mass"]) < 1e-6 * g["Mass"]
        for j in range(3):
            assert abs(g["Vel"][j] - (j + 1) * (hh + 1)) < 1e-5 * (hh + 1)
            assert periodic_dist(g["CM"][j], fx["expcm"][hh][j], fx["box"]) < 0.01
    assert all(f == 1 for f in found)
    assert all(grnr_of_halo[h] == NHALO - h for h in range(NHALO))
    want = np.where(fx["halo_of"] >= 0, np.array(grnr_of_halo + [-1])[fx["halo_of"]], -1)
    assert np.array_equal(part_grnr, want)
