"""world_size-2 / -4 gloo tests of the multi-GPU orchestration (shenqi_amd/dist.py) on CPU:
slab PM with ghost planes and all-to-all transposes vs the monolithic oracle PM, domain exchange,
and ghost-particle import for the tree vs the monolithic oracle walk."""
import os
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

NPART = 16**3
NMESH = 48
BOX = 8.0
G = 43.0071


def _global_particles():
    import orc
    import common as cm
    pos = cm.random_positions(orc.boost_mt19937_uniform(0, 3 * NPART), NPART)
    posm = np.concatenate([pos, np.ones((NPART, 1))], axis=1)
    return posm


def _worker(rank, world, initfile, outdir, balanced):
    os.environ["OMP_NUM_THREADS"] = "2"
    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    try:
        import shenqi_amd as sq
        from shenqi_amd import dist as sd
        import orc
        import common as cm
        from cpu_ops import CpuOps

        comm = sd.Comm()
        posm_g = _global_particles()
        mine = torch.from_numpy(posm_g[rank::world].copy())          # any initial distribution
        bounds, ycuts = None, None
        if balanced == "split":       # boundaries cut below the plane: a plane's particles shared by y, its mesh plane not
            bounds, ycuts = sd.balanced_bounds(comm, NMESH, BOX, mine[:, 0], y=mine[:, 1])
            assert max(ycuts) > 0
        elif balanced:
            bounds = sd.balanced_bounds(comm, NMESH, BOX, mine[:, 0])
        decomp = sd.SlabDecomp(comm, NMESH, BOX, bounds, ycuts)
        local = sd.exchange_to_owner(comm, decomp, mine)
        nloc = local.shape[0]
        total = comm.allreduce_sum(float(nloc))
        assert int(total) == NPART
        x = local[:, 0].numpy()
        pl = np.floor(x / (BOX / NMESH)) % NMESH
        if balanced == "split":
            yy = local[:, 1].numpy()
            nxt = decomp.bounds[rank + 1] % NMESH
            assert np.all(((pl >= decomp.plane0) & (pl < decomp.plane0 + decomp.nxl)) | ((pl == nxt) & (yy < decomp.ycuts[rank + 1])))
            assert not np.any((pl == decomp.plane0) & (yy < decomp.ycuts[rank]))
            assert np.any(pl == nxt) or rank == world - 1
            # (how much closer to equal shares the cut by rows of cells comes: test_balanced_bounds_weighted_gloo — this
            # set's cluster sits inside one cell)
        else:
            assert np.all((pl >= decomp.plane0) & (pl < decomp.plane0 + decomp.nxl))
        if balanced and balanced != "split":
            assert nloc < 0.6 * NPART, (nloc, bounds)      # plane granularity limits the balance on this tiny mesh
        # ---- PM ----
        ops = CpuOps(NMESH, BOX, 1.5, G)
        ops.set_deposit_scale(comm.allreduce_sum(float(local[:, 3].sum())))
        ops.set_particles(local, nloc)
        pm = sd.SlabPM(comm, NMESH, BOX, 1.5, G, ops, bounds, ycuts)
        pm.force()
        gpm, ppot = ops.results(nloc)
        # ---- tree with imported ghosts ----
        cm.reference_treepar(ErrTolForceAcc=0.005, MaxBHOpeningAngle=0.9, Rcut=6.0, TreeUseBH=0)
        sq.gravshort_set_softenings(BOX / np.cbrt(NPART))
        gp = sq.make_grav_params(BOX, 1.5, NMESH, G, cm.RHO0)
        halo = 1.3 * gp.Rcut
        ghosts = sd.ghost_exchange(comm, decomp, local, halo)
        allp = torch.cat([local, ghosts], dim=0).numpy()
        nodes, first, _ = orc.tree_build(allp[:, :3].copy(), allp[:, 3].astype(np.float32), BOX)
        np.save(os.path.join(outdir, "r%d.npy" % rank), np.concatenate([local.numpy(), gpm, ppot[:, None]], axis=1))
        np.save(os.path.join(outdir, "t%d.npy" % rank), allp)
        np.save(os.path.join(outdir, "n%d.npy" % rank), np.array([nloc, len(ghosts)]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,balanced", [(2, False), (4, False), (3, True), (4, True), (2, "split"), (3, "split")])
def test_slab_pm_and_ghost_tree_gloo(world, balanced):
    import orc
    import common as cm
    import shenqi_amd as sq
    with tempfile.TemporaryDirectory() as tmp:
        initfile = os.path.join(tmp, "init")
        mp.spawn(_worker, args=(world, initfile, tmp, balanced), nprocs=world, join=True)
        posm_g = _global_particles()
        mass = posm_g[:, 3].astype(np.float32)
        e = 61 - int(np.frexp(float(NPART))[1])
        og, opot, _, _ = orc.pm_force(posm_g[:, :3].copy(), mass, NMESH, BOX, 1.5, G, fixed_point_log2scale=e, use_stencil=1)
        key = {tuple(p): i for i, p in enumerate(map(tuple, posm_g[:, :3]))}
        seen = 0
        # monolithic tree forces (relative criterion with OldAcc from a BH pass)
        nodes, first, _ = orc.tree_build(posm_g[:, :3].copy(), mass, BOX)
        cm.reference_treepar(ErrTolForceAcc=0.005, MaxBHOpeningAngle=0.9, Rcut=6.0, TreeUseBH=1)
        sq.gravshort_set_softenings(BOX / np.cbrt(NPART))
        gp_bh = sq.make_grav_params(BOX, 1.5, NMESH, G, cm.RHO0)
        acc_bh, _, _ = orc.grav_walk(nodes, first, posm_g[:, :3].copy(), mass, np.zeros(NPART), gp_bh)
        oldacc = np.linalg.norm(acc_bh * G + og, axis=1) / G
        cm.reference_treepar(ErrTolForceAcc=0.005, MaxBHOpeningAngle=0.9, Rcut=6.0, TreeUseBH=0)
        gp = sq.make_grav_params(BOX, 1.5, NMESH, G, cm.RHO0)
        acc_mono, _, _ = orc.grav_walk(nodes, first, posm_g[:, :3].copy(), mass, oldacc, gp)
        num = den = 0.0
        for r in range(world):
            a = np.load(os.path.join(tmp, "r%d.npy" % r))
            idx = np.array([key[tuple(p)] for p in a[:, :3]])
            seen += len(idx)
            # PM: sharded == monolithic to FFT-decomposition rounding
            assert np.abs(a[:, 4:7] - og[idx]).max() < 1e-11 * np.abs(og).max()
            assert np.abs(a[:, 7] - opot[idx]).max() < 1e-11 * np.abs(opot).max()
            # tree: local targets on the (local + ghost) tree
            allp = np.load(os.path.join(tmp, "t%d.npy" % r))
            nloc, ngh = np.load(os.path.join(tmp, "n%d.npy" % r))
            lnodes, lfirst, _ = orc.tree_build(allp[:, :3].copy(), allp[:, 3].astype(np.float32), BOX)
            gidx = np.array([key[tuple(p)] for p in allp[:, :3]])
            assert len(set(gidx.tolist())) == len(gidx)           # no particle imported twice
            acc_loc, _, _ = orc.grav_walk(lnodes, lfirst, allp[:, :3].copy(), allp[:, 3].astype(np.float32), oldacc[gidx], gp,
                                          targets=np.arange(nloc, dtype=np.int32))
            ref = acc_mono[gidx[:nloc]]
            num += np.sum((acc_loc - ref) ** 2)
            den += np.sum(ref**2)
            assert 0 < ngh < NPART
        assert seen == NPART
        rms = np.sqrt(num / den)
        print("world %d balanced %s: sharded tree vs monolithic rms |dF|/|F| = %.3e" % (world, balanced, rms))
        assert rms < 1e-3


def _weights_worker(rank, world, initfile):
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    try:
        from shenqi_amd import dist as sd
        comm = sd.Comm()
        N, L = 64, 1.0
        rng = np.random.default_rng(5 + rank)
        x = torch.from_numpy(rng.random(20000) * L)
        # unit weights = the count balance; uniform particles split the planes evenly
        assert sd.balanced_bounds(comm, N, L, x) == sd.balanced_bounds(comm, N, L, x, weights=torch.ones(20000))
        b = sd.balanced_bounds(comm, N, L, x)
        assert abs(b[1] - N // 2) <= 2
        # particles of the left half weigh three times as much: the cut moves to the plane splitting the weight in halves
        w = torch.where(x < 0.5 * L, 3.0, 1.0)
        bw = sd.balanced_bounds(comm, N, L, x, weights=w)
        assert abs(bw[1] - N // 3) <= 2, bw      # 3 t = (3 / 2 + 1 / 2) / 2 -> t = 1 / 3 of the box
        # a plane cost that dwarfs the particle work brings back equal widths
        assert sd.balanced_bounds(comm, N, L, x, weights=w, plane_cost=1e9) == [0, N // 2, N]
        # a sheet thinner than a plane (what a cluster's core is to a 1024 mesh): three quarters of everything in plane 40.
        # Cut by planes, a rank gets the sheet or does not; cut by rows of cells inside the sheet's plane the shares meet.
        xs = torch.cat([x[:5000], torch.from_numpy((40.5 + 0.1 * rng.standard_normal(15000)) * (L / N))])
        ys = torch.from_numpy(rng.random(20000) * L)
        bp = sd.balanced_bounds(comm, N, L, xs)
        bs, yc = sd.balanced_bounds(comm, N, L, xs, y=ys)
        assert bs[1] == 40 and yc[0] == 0 and yc[2] == 0 and 0.2 * L < yc[1] < 0.8 * L, (bs, yc)

        def share(decomp):
            own = decomp.owner_of(xs, ys)
            return comm.allreduce_sum(float((own == 0).sum())) / 40000.0
        plain, cut = share(sd.SlabDecomp(comm, N, L, bp)), share(sd.SlabDecomp(comm, N, L, bs, yc))
        assert abs(plain - 0.5) > 0.2 and abs(cut - 0.5) < 0.01, (plain, cut)
        # weights move the cut inside the plane too
        wy = torch.where(ys < 0.25 * L, 5.0, 1.0)
        _, ycw = sd.balanced_bounds(comm, N, L, xs, weights=wy, y=ys)
        assert ycw[1] < yc[1] - 0.1 * L, (ycw, yc)
        with pytest.raises(ValueError):
            sd.SlabDecomp(comm, N, L, bs, yc).owner_of(xs)
    finally:
        dist.destroy_process_group()


def test_balanced_bounds_weighted_gloo():
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_weights_worker, args=(2, os.path.join(tmp, "init")), nprocs=2, join=True)


def _empty_slab_worker(rank, world, initfile, outdir):
    os.environ["OMP_NUM_THREADS"] = "2"
    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    try:
        from shenqi_amd import dist as sd
        from cpu_ops import CpuOps
        comm = sd.Comm()
        posm_g = _global_particles()
        posm_g = posm_g[(posm_g[:, 0] < BOX / 3) | (posm_g[:, 0] >= 2 * BOX / 3)]     # nothing in the middle third of the box
        mine = torch.from_numpy(posm_g[rank::world].copy())
        bounds = [0, NMESH // 3, 2 * NMESH // 3, NMESH]
        decomp = sd.SlabDecomp(comm, NMESH, BOX, bounds)
        local = sd.exchange_to_owner(comm, decomp, mine)
        nloc = local.shape[0]
        assert (nloc == 0) == (rank == 1)
        ops = CpuOps(NMESH, BOX, 1.5, G)
        ops.set_deposit_scale(comm.allreduce_sum(float(local[:, 3].sum())))
        # the empty rank still imports ghosts for nobody's benefit: they must neither deposit nor be read out
        ghosts = sd.ghost_exchange(comm, decomp, local, 0.2 * BOX)
        ops.set_particles(torch.cat([local, ghosts], dim=0), nloc)
        pm = sd.SlabPM(comm, NMESH, BOX, 1.5, G, ops, bounds)
        pm.force()
        gpm, ppot = ops.results(nloc)
        np.save(os.path.join(outdir, "e%d.npy" % rank), np.concatenate([local.numpy().reshape(-1, 4), gpm.reshape(-1, 3), ppot.reshape(-1, 1)], axis=1))
    finally:
        dist.destroy_process_group()


def test_slab_pm_with_an_empty_slab_gloo():
    """a rank whose slab holds no particle (nlocal = 0) takes part in the transposes and the ghost-plane exchange, deposits
    nothing - not even the ghosts it imported - and the others' forces equal the monolithic PM"""
    import orc
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_empty_slab_worker, args=(3, os.path.join(tmp, "init"), tmp), nprocs=3, join=True)
        posm_g = _global_particles()
        posm_g = posm_g[(posm_g[:, 0] < BOX / 3) | (posm_g[:, 0] >= 2 * BOX / 3)]
        n = len(posm_g)
        e = 61 - int(np.frexp(float(n))[1])
        og, opot, _, _ = orc.pm_force(posm_g[:, :3].copy(), posm_g[:, 3].astype(np.float32), NMESH, BOX, 1.5, G, fixed_point_log2scale=e,
                                      use_stencil=1)
        key = {tuple(p): i for i, p in enumerate(map(tuple, posm_g[:, :3]))}
        seen = 0
        for r in range(3):
            a = np.load(os.path.join(tmp, "e%d.npy" % r))
            if r == 1:
                assert len(a) == 0
                continue
            idx = np.array([key[tuple(p)] for p in a[:, :3]])
            seen += len(idx)
            assert np.abs(a[:, 4:7] - og[idx]).max() < 1e-11 * np.abs(og).max()
            assert np.abs(a[:, 7] - opot[idx]).max() < 1e-11 * np.abs(opot).max()
        assert seen == n
