"""world_size-1 / -2 / -3 gloo tests of the sharded wind step (shenqi_amd/dist.py:DistWinds) on CPU: every rank owns an x-slab,
imports the neighbours' gas within the largest Hsml of a new star, runs the two walks for its own new stars (the restatement stands
in for shq_winds_candidates / shq_winds_apply) and sends the kick candidates that fell on ghosts to their owners.  The outcome must be
the undivided restatement's: the same particles kicked by the same stars (nearest star, then smaller ID — decomposition independent),
the same velocities, entropies and delay times, TotalWeight to rounding (the sums run in another order)."""
import os
import pickle
import sys
import tempfile
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

NMESH = 48


def global_set():
    import test_gpu_winds as tw
    pman, S, ST, rnd, new = tw.setup(13, ngrid=12, nstar=300, nnew=120)
    return pman.Base.copy(), S, ST, rnd, new, tw.params()[1]


def to_kicks(lst):
    from shenqi_amd import capi
    k = np.zeros(len(lst), dtype=capi.WIND_KICK_DTYPE)
    for j, (p, d, sid, v, th) in enumerate(lst):
        k[j] = (p, 0, d, sid, v, th)
    return k


class OracleWindOps:
    def candidates(self, Pall, Sall, StarP, newstars, prm, rnd):
        import winds as ow
        tw, kicks = ow.candidates(Pall, Sall, StarP, np.ascontiguousarray(Pall["ID"]), newstars, prm, rnd)
        return tw, to_kicks(kicks)

    def apply(self, P, SphP, kicks, prm, rnd):
        import winds as ow
        lst = [(int(k["part_index"]), float(k["StarDistance"]), int(k["StarID"]), float(k["StarKickVelocity"]), float(k["StarTherm"])) for k in kicks]
        return ow.apply(P, SphP, np.ascontiguousarray(P["ID"]), lst, prm, rnd)


def local_part(rank, decomp, Pg, Sg, STg, newg):
    """this rank's share of the global set: its particles, their slots renumbered locally, its new stars"""
    owner = decomp.owner_of(torch.from_numpy(np.ascontiguousarray(Pg["Pos"][:, 0]))).numpy()
    mine = np.flatnonzero(owner == rank)
    P = Pg[mine].copy()
    gas, star = P["Type"] == 0, P["Type"] == 4
    S = Sg[P["PI"][gas]].copy()
    ST = STg[P["PI"][star]].copy()
    P["PI"][gas] = np.arange(gas.sum())
    P["PI"][star] = np.arange(star.sum())
    back = {int(g): k for k, g in enumerate(mine)}
    new = np.array([back[int(g)] for g in newg if int(g) in back], dtype=np.int32)
    return P, S, ST, new


def run_rank(comm, rank, ops):
    from shenqi_amd import dist as sd
    Pg, Sg, STg, rnd, newg, prm = global_set()
    decomp = sd.SlabDecomp(comm, NMESH, prm.BoxSize)
    P, S, ST, new = local_part(rank, decomp, Pg, Sg, STg, newg)
    drv = sd.DistWinds(comm, decomp, ops)
    tw, applied = drv.run(P, S, ST, new, prm, rnd)
    star = P["Type"] == 4
    gas = P["Type"] == 0
    return dict(ids=P["ID"], vel=P["Vel"], gas_ids=P["ID"][gas], entropy=S["Entropy"][P["PI"][gas]], delay=S["DelayTime"][P["PI"][gas]],
                star_ids=P["ID"][new] if len(new) else np.zeros(0, dtype=np.uint64), tw=tw[P["PI"][new]] if len(new) else np.zeros(0), applied=applied,
                nghost=drv.nghost, nkicks=drv.nkicks)


def monolithic():
    import winds as ow
    Pg, Sg, STg, rnd, newg, prm = global_set()
    P, S = Pg.copy(), Sg.copy()
    tw, kicks, applied = ow.winds_and_feedback(P, S, STg, np.ascontiguousarray(P["ID"]), newg, prm, rnd)
    return Pg, P, S, tw, kicks, applied, newg


def check(results, exact=True):
    Pg, P, S, tw, kicks, applied, newg = monolithic()
    by_id = {int(i): k for k, i in enumerate(P["ID"])}
    assert sum(r["applied"] for r in results) == applied > 10
    assert sum(r["nkicks"] for r in results) == len(kicks)
    seen = 0
    for r in results:
        idx = np.array([by_id[int(i)] for i in r["ids"]])
        seen += len(idx)
        if exact:
            assert np.array_equal(r["vel"], P["Vel"][idx])
        else:
            assert np.abs(r["vel"] - P["Vel"][idx]).max() < 1e-11
            assert np.array_equal(np.any(r["vel"] != Pg["Vel"][idx], axis=1), np.any(P["Vel"][idx] != Pg["Vel"][idx], axis=1))
        gidx = np.array([by_id[int(i)] for i in r["gas_ids"]], dtype=np.int64)
        assert np.abs(r["entropy"] / S["Entropy"][P["PI"][gidx]] - 1).max() < (1e-15 if exact else 1e-13)
        assert np.array_equal(r["delay"], S["DelayTime"][P["PI"][gidx]])
        sidx = np.array([by_id[int(i)] for i in r["star_ids"]], dtype=np.int64)
        if len(sidx):
            want = tw[P["PI"][sidx]]
            assert np.abs(r["tw"] - want).max() <= 1e-12 * max(want.max(), 1.0)
    assert seen == len(P)


def _worker(rank, world, initfile, outdir):
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    try:
        from shenqi_amd import dist as sd
        res = run_rank(sd.Comm(), rank, OracleWindOps())
        with open(os.path.join(outdir, "r%d.pkl" % rank), "wb") as f:
            pickle.dump(res, f)
    finally:
        dist.destroy_process_group()


def test_dist_winds_one_rank_without_process_group():
    from shenqi_amd import dist as sd
    res = run_rank(sd.Comm(), 0, OracleWindOps())
    assert res["nghost"] == 0
    check([res])


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_winds_equal_monolithic_gloo(world):
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(world, os.path.join(tmp, "init"), tmp), nprocs=world, join=True)
        results = []
        for r in range(world):
            with open(os.path.join(tmp, "r%d.pkl" % r), "rb") as f:
                results.append(pickle.load(f))
        check(results)
        assert all(r["nghost"] > 0 for r in results)
