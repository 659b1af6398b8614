"""world_size-1 / -2 / -3 gloo tests of the sharded friends-of-friends finder (shenqi_amd/dist.py:DistFOF) on CPU: every rank owns
an x-slab, imports the neighbours' particles within the halo, labels local + ghost particles (the oracle stands in for shq_fof) and
lowers labels through the shared particles until nothing changes.  Must equal the undivided oracle: MinID per particle, GrNr,
lengths as integers, the group sums to rounding.  The reference does this with ghost queries inside fof_label_primary's loop and
fof_reduce_groups (fof.cpp:404-470, 903-1040)."""
import os
import pickle
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

BOX, NMESH, LINKL, MINLEN = 200.0, 48, 1.0, 6


def global_set():
    """clumps and filaments that cross the slab boundaries (and the periodic edge), gas / stars / a few BHs around them"""
    from shenqi_amd import capi
    rng = np.random.default_rng(11)
    centres = rng.random((40, 3)) * BOX
    centres[:6, 0] = [0.2, BOX / 2, BOX / 3, 2 * BOX / 3, BOX - 0.3, BOX / 2 + 0.4]      # on the cuts of 2 and 3 slabs and on the edge
    dm = [c + rng.normal(size=(int(s), 3)) * 0.8 for c, s in zip(centres, rng.integers(4, 120, size=40))]
    t = np.linspace(0, 1, 400)[:, None]
    # a filament through every slab, shorter than half the box (beyond that the reference's centre of mass, taken about whichever
    # member is FirstPos, is not defined)
    dm.append(np.array([40.0, 50, 50]) + t * np.array([96.0, 3, 2]) + rng.normal(size=(400, 3)) * 0.05)
    dm.append(rng.random((1500, 3)) * BOX)
    dm = np.concatenate(dm)
    sec = np.concatenate([centres[rng.integers(0, 40, size=600)] + rng.normal(size=(600, 3)) * 3.0, rng.random((300, 3)) * BOX])
    pos = np.mod(np.concatenate([dm, sec]), BOX)
    n = len(pos)
    P = np.zeros(n, dtype=capi.PARTICLE_DTYPE)
    P["Pos"] = pos
    P["Type"] = np.concatenate([np.ones(len(dm), dtype=np.uint8), rng.choice([0, 4, 5], size=len(sec), p=[0.7, 0.25, 0.05]).astype(np.uint8)])
    P["ID"] = rng.permutation(n).astype(np.uint64) + 10
    P["Vel"] = rng.normal(size=(n, 3)) * 30
    P["Mass"] = rng.choice([1.0, 0.5, 0.25], size=n).astype(np.float32)
    P["Hsml"] = rng.uniform(0.2, 5.0, size=n)
    fl = np.zeros(n, dtype=np.uint8)
    fl[rng.random(n) < 0.005] |= 1
    P["Flags"] = fl
    return P


class OracleFofOps:
    def labels(self, Pall, ids, linkl, primary_mask, secondary_mask):
        import fof as ofof
        dead = (Pall["Flags"] & 3) != 0
        pos = np.ascontiguousarray(Pall["Pos"])
        lab = ofof.label_primary(pos, ids, Pall["Type"], dead, BOX, linkl, primary_mask)
        if secondary_mask:
            lab, _ = ofof.label_secondary(pos, Pall["Type"], dead, Pall["Hsml"], lab, BOX, linkl, primary_mask, secondary_mask)
        return lab


def monolithic():
    import fof as ofof
    P = global_set()
    dead = (P["Flags"] & 3) != 0
    return P, ofof.fof(np.ascontiguousarray(P["Pos"]), np.ascontiguousarray(P["Vel"]), P["Mass"].astype(np.float64), P["Type"], P["ID"].astype(np.uint64),
                       dead, P["Hsml"], BOX, LINKL, MINLEN)


def _worker(rank, world, initfile, outdir):
    os.environ["OMP_NUM_THREADS"] = "2"
    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    try:
        from shenqi_amd import dist as sd
        comm = sd.Comm()
        decomp = sd.SlabDecomp(comm, NMESH, BOX)
        Pg = global_set()
        mine = (decomp.owner_of(torch.from_numpy(np.ascontiguousarray(Pg["Pos"][:, 0]))) == rank).numpy()
        P = Pg[mine].copy()
        drv = sd.DistFOF(comm, decomp, OracleFofOps())
        minid, groups, grnr = drv.fof(P, LINKL, MINLEN)
        with open(os.path.join(outdir, "r%d.pkl" % rank), "wb") as f:
            pickle.dump(dict(ids=P["ID"], minid=minid, groups=groups, grnr=grnr, rounds=drv.rounds, nghost=drv.nghost), f)
    finally:
        dist.destroy_process_group()


def check(results, P, ref):
    ominid, ogroups, ogrnr = ref
    key = {int(i): k for k, i in enumerate(P["ID"])}
    seen = 0
    got_groups = []
    for res in results:
        idx = np.array([key[int(i)] for i in res["ids"]])
        seen += len(idx)
        assert np.array_equal(res["minid"], ominid[idx])
        assert np.array_equal(res["grnr"], ogrnr[idx])
        got_groups.extend(res["groups"])
    assert seen == len(P)
    got_groups.sort(key=lambda G: G["MinID"])
    assert len(got_groups) == len(ogroups) > 10
    for g, o in zip(got_groups, ogroups):
        assert g["MinID"] == o["MinID"] and g["Length"] == o["Length"] and g["GrNr"] == o["GrNr"] and g["LenType"] == o["LenType"]
        assert abs(g["Mass"] - o["Mass"]) < 1e-12 * o["Mass"] and np.allclose(g["MassType"], o["MassType"], rtol=1e-12, atol=1e-12)
        d = np.abs(g["CM"] - o["CM"])
        assert np.minimum(d, BOX - d).max() < 1e-9
        assert np.abs(g["Vel"] - o["Vel"]).max() < 1e-10 * 30
        # Imom / Jmom are taken about the centre of mass, so they do not depend on which member served as FirstPos
        assert np.abs(g["Imom"] - o["Imom"]).max() < 1e-8 * (1 + np.abs(o["Imom"]).max())
        assert np.abs(g["Jmom"] - o["Jmom"]).max() < 1e-8 * (1 + np.abs(o["Jmom"]).max())


def test_dist_fof_one_rank_without_process_group():
    from shenqi_amd import dist as sd
    P, ref = monolithic()
    comm = sd.Comm()
    drv = sd.DistFOF(comm, sd.SlabDecomp(comm, NMESH, BOX), OracleFofOps())
    minid, groups, grnr = drv.fof(P, LINKL, MINLEN)
    check([dict(ids=P["ID"], minid=minid, groups=groups, grnr=grnr)], P, ref)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_fof_equals_monolithic_gloo(world):
    P, ref = monolithic()
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(world, os.path.join(tmp, "init"), tmp), nprocs=world, join=True)
        results = []
        for r in range(world):
            with open(os.path.join(tmp, "r%d.pkl" % r), "rb") as f:
                results.append(pickle.load(f))
        check(results, P, ref)
        assert all(res["nghost"] > 0 for res in results)
        assert max(res["rounds"] for res in results) >= 2          # the filament needs more than one exchange of labels
