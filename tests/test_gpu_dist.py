"""GPU tests of the multi-GPU path on the one-GPU box: the slab PM / ghost-tree driver with one rank
(device kernels through the slab entry points) and with two gloo ranks sharing the GPU (exchanges
staged through the host), both against the monolithic single-GPU path and the oracle."""
import ctypes as C
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

pytestmark = pytest.mark.gpu

NPART, NMESH, BOX, G = 16**3, 48, 8.0, 43.0071


def _global_particles():
    import orc
    import common as cm
    pos = cm.random_positions(orc.boost_mt19937_uniform(0, 3 * NPART), NPART)
    return np.concatenate([pos, np.ones((NPART, 1))], axis=1)


def _run_rank(rank, world, outdir, split=False):
    import shenqi_amd as sq
    from shenqi_amd import capi, dist as sd
    import common as cm
    dev = torch.device("cuda", 0)
    comm = sd.Comm()
    posm_g = _global_particles()
    mine = torch.from_numpy(posm_g[rank::world].copy()).to(dev)
    ctx = sq.Context(0)
    cm.reference_treepar(ErrTolForceAcc=0.005, MaxBHOpeningAngle=0.9, Rcut=6.0, TreeUseBH=1)
    sq.gravshort_set_softenings(BOX / np.cbrt(NPART))
    gp_bh = sq.make_grav_params(BOX, 1.5, NMESH, G, cm.RHO0)
    cm.reference_treepar(ErrTolForceAcc=0.005, MaxBHOpeningAngle=0.9, Rcut=6.0, TreeUseBH=0)
    gp = sq.make_grav_params(BOX, 1.5, NMESH, G, cm.RHO0)
    bounds = sd.balanced_bounds(comm, NMESH, BOX, mine[:, 0]) if world > 1 else None
    # split: the first plane of rank 1 (the plane through the cluster, which sits at y = 5.0 ... 5.3) shared by y — its particles
    # below y = 5.06 are rank 0's
    ycuts = [0.0, 5.06, 0.0] if split else None
    drv = sd.DistTreePM(comm, ctx, NMESH, BOX, 1.5, G, dev, halo_factor=1.3, bounds=bounds, ycuts=ycuts)
    local = sd.exchange_to_owner(comm, drv.decomp, mine)
    if split:
        pl = torch.floor(local[:, 0] / (BOX / NMESH)).to(torch.int64) % NMESH
        shared = int(((pl == bounds[1]) & (local[:, 1] < ycuts[1])).sum()) if rank == 0 else int((pl == bounds[1]).sum())
        assert shared > 100, (rank, shared, bounds)       # both ranks hold a good part of that plane
    drv.setup(local, gp.Rcut)
    drv.step(gp_bh)
    drv.step(gp)
    acc, pot, gpm, ppot = drv.download()
    np.save(os.path.join(outdir, "g%d.npy" % rank), np.concatenate([drv.local.cpu().numpy(), acc, gpm, ppot[:, None]], axis=1))
    ctx.close()


def _worker(rank, world, initfile, outdir, backend="gloo", split=False):
    os.environ["OMP_NUM_THREADS"] = "2"
    if backend == "nccl":
        os.environ["SHQ_COMM_FORCE"] = "1"
        os.environ["SHQ_COMM_MAX_MSG"] = "300000"   # the 48 x 48 x 25 spectrum is 0.9 MB: sent as 4 row-chunked rounds
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", init_method="file://" + initfile, rank=rank, world_size=world,
                                device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group(backend, init_method="file://" + initfile, rank=rank, world_size=world)
    try:
        _run_rank(rank, world, outdir, split)
    finally:
        dist.destroy_process_group()


def _reference():
    """monolithic oracle: PM (stencil, fixed point) + BH pass + relative-criterion pass"""
    import orc
    import common as cm
    import shenqi_amd as sq
    posm_g = _global_particles()
    pos, mass = posm_g[:, :3].copy(), posm_g[:, 3].astype(np.float32)
    e = 61 - int(np.frexp(float(NPART))[1])
    og, opot, _, _ = orc.pm_force(pos, mass, NMESH, BOX, 1.5, G, fixed_point_log2scale=e, use_stencil=1)
    nodes, first, _ = orc.tree_build(pos, mass, BOX)
    cm.reference_treepar(ErrTolForceAcc=0.005, MaxBHOpeningAngle=0.9, Rcut=6.0, TreeUseBH=1)
    sq.gravshort_set_softenings(BOX / np.cbrt(NPART))
    gp_bh = sq.make_grav_params(BOX, 1.5, NMESH, G, cm.RHO0)
    a1, _, _ = orc.grav_walk(nodes, first, pos, mass, np.zeros(NPART), gp_bh)
    cm.reference_treepar(ErrTolForceAcc=0.005, MaxBHOpeningAngle=0.9, Rcut=6.0, TreeUseBH=0)
    gp = sq.make_grav_params(BOX, 1.5, NMESH, G, cm.RHO0)
    a2, _, _ = orc.grav_walk(nodes, first, pos, mass, np.linalg.norm(a1 * G + og, axis=1) / G, gp)
    return pos, og, opot, a2 * G


def _check(tmp, world):
    pos, og, opot, oacc = _reference()
    key = {tuple(p): i for i, p in enumerate(map(tuple, pos))}
    seen = 0
    num = den = 0.0
    for r in range(world):
        a = np.load(os.path.join(tmp, "g%d.npy" % r))
        idx = np.array([key[tuple(p)] for p in a[:, :3]])
        seen += len(idx)
        assert np.abs(a[:, 7:10] - og[idx]).max() < 1e-10 * np.abs(og).max()          # PM
        assert np.abs(a[:, 10] - opot[idx]).max() < 1e-10 * np.abs(opot).max()
        num += np.sum((a[:, 4:7] - oacc[idx]) ** 2)
        den += np.sum(oacc[idx] ** 2)
    assert seen == NPART
    rms = np.sqrt(num / den)
    print("world %d on one GPU: tree rms vs monolithic oracle %.3e" % (world, rms))
    assert rms < 1e-3


def test_dist_driver_single_rank():
    with tempfile.TemporaryDirectory() as tmp:
        _run_rank(0, 1, tmp)
        _check(tmp, 1)


def test_dist_driver_two_gloo_ranks_one_gpu():
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(2, os.path.join(tmp, "init"), tmp), nprocs=2, join=True)
        _check(tmp, 2)


def test_dist_driver_two_ranks_sharing_a_plane():
    """slabs cut below the plane (SlabDecomp ycuts): rank 0 deposits into two planes of rank 1 and reads four of its potential
    planes (shq_pm_slab2_deposit_ghosts, [nxl + 6] buffers).  Same oracle bounds as the plane-aligned run, and the PM force of
    every particle has the plane-aligned run's bits: the mesh is a fixed-point sum, whoever deposits."""
    with tempfile.TemporaryDirectory() as tmp, tempfile.TemporaryDirectory() as tmp2:
        mp.spawn(_worker, args=(2, os.path.join(tmp, "init"), tmp, "gloo", True), nprocs=2, join=True)
        _check(tmp, 2)
        mp.spawn(_worker, args=(2, os.path.join(tmp2, "init"), tmp2), nprocs=2, join=True)
        rows = [np.concatenate([np.load(os.path.join(t, "g%d.npy" % r)) for r in range(2)]) for t in (tmp, tmp2)]
        assert len(np.load(os.path.join(tmp, "g0.npy"))) != len(np.load(os.path.join(tmp2, "g0.npy")))
        a, b = [r[np.lexsort(r[:, :3].T)] for r in rows]
        assert np.array_equal(a[:, :3], b[:, :3]) and np.array_equal(a[:, 7:11], b[:, 7:11])


def test_dist_driver_one_rccl_rank_collectives_forced():
    """The RCCL calls of the N-GPU run (device-side all_to_all_single of particle rows and of the complex mesh
    transposes, the scalar all-reduce) on the one GPU there is: a one-rank nccl group with SHQ_COMM_FORCE=1."""
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(1, os.path.join(tmp, "init"), tmp, "nccl"), nprocs=1, join=True)
        _check(tmp, 1)


def test_hilbert_order_on_device_equals_host(ctx):
    """shq_hilbert_order (what DistTreePM.setup orders a rank's particles with) = the host mirror's order, entry for entry;
    equal keys (coincident particles) stay in index order; n = 0 and n = 1 are fine"""
    import shenqi_amd as sq
    from shenqi_amd import capi
    rng = np.random.default_rng(5)
    for n in (0, 1, 77, 50000):
        pos = rng.random((n, 4)) * BOX
        if n > 10:
            pos[5, :3] = pos[9, :3] = pos[2, :3]            # coincident: same key
            pos[7, 0] = 0.0
            pos[8, :3] = BOX * (1 - 1e-16)
        t = torch.from_numpy(pos).to("cuda:0")
        out = torch.full((n,), -1, dtype=torch.int64, device="cuda:0")
        torch.cuda.synchronize()
        capi.check(capi.hip.shq_hilbert_order(ctx.h, C.c_void_p(t.data_ptr()), n, BOX, C.c_void_p(out.data_ptr())))
        ctx.synchronize()
        dev = out.cpu().numpy()
        assert sorted(dev.tolist()) == list(range(n))
        host = sq.hilbert_order(np.ascontiguousarray(pos[:, :3]), BOX).astype(np.int64)
        if n > 10:
            k = [int(np.flatnonzero(dev == i)[0]) for i in (2, 5, 9)]
            assert k[0] < k[1] < k[2] and k[2] - k[0] == 2
            same = ~np.isin(host, (2, 5, 9))                 # the host sort need not be stable
            assert np.array_equal(dev[same], host[same])
        else:
            assert np.array_equal(dev, host)
    with pytest.raises(sq.ShqError):
        capi.check(capi.hip.shq_hilbert_order(ctx.h, None, 5, BOX, None))


def test_device_particle_set_semantics(ctx):
    """shq_particles_set_device: nlocal = 0 is a rank that owns nothing (no deposit, no targets); a new set drops the tree
    (a walk is refused until the next build) unless the caller vouches the positions are the tree's"""
    import shenqi_amd as sq
    from shenqi_amd import capi
    import common as cm
    posm = torch.from_numpy(_global_particles()[:2000].copy()).to("cuda:0")
    pm = capi.PMParams(NMESH, 0, BOX, 1.5, G)
    cm.reference_treepar(ErrTolForceAcc=0.005, MaxBHOpeningAngle=0.9, Rcut=6.0, TreeUseBH=1)
    sq.gravshort_set_softenings(BOX / 16)
    gp = sq.make_grav_params(BOX, 1.5, NMESH, G, cm.RHO0)
    torch.cuda.synchronize()
    # all 2000 rows are ghosts: nothing may be deposited (the old "0 means all" reading put every ghost on the mesh)
    capi.check(capi.hip.shq_particles_set_device(ctx.h, C.c_void_p(posm.data_ptr()), 2000, 0, 0))
    capi.check(capi.hip.shq_pm_set_deposit_log2scale(ctx.h, 40))
    mesh = torch.ones((NMESH, NMESH, NMESH + 2), dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    capi.check(capi.hip.shq_pm_slab_deposit(ctx.h, C.byref(pm), 0, NMESH, C.c_void_p(mesh.data_ptr())))
    ctx.synchronize()
    assert int(mesh.abs().sum().item()) == 0
    sq.tree_build_device(ctx, BOX)
    capi.check(capi.hip.shq_grav_short_run(ctx.h, C.byref(gp), None, 0, 1, sq.WALK_EXACT))
    st = sq.WalkStats()
    capi.check(capi.hip.shq_grav_short_download(ctx.h, None, None, None, C.byref(st)))
    assert st.ntargets == 0 and st.ninteractions == 0
    # own particles: the first 1500
    capi.check(capi.hip.shq_particles_set_device(ctx.h, C.c_void_p(posm.data_ptr()), 2000, 1500, 0))
    assert capi.hip.shq_grav_short_run(ctx.h, C.byref(gp), None, 0, 1, sq.WALK_EXACT) != 0      # same count, but the tree was dropped
    sq.tree_build_device(ctx, BOX)
    capi.check(capi.hip.shq_grav_short_run(ctx.h, C.byref(gp), None, 0, 1, sq.WALK_EXACT))
    capi.check(capi.hip.shq_grav_short_download(ctx.h, None, None, None, C.byref(st)))
    assert st.ntargets == 1500
    n1 = st.ninteractions
    capi.check(capi.hip.shq_particles_set_device(ctx.h, C.c_void_p(posm.data_ptr()), 2000, 1500, 1))    # frozen positions: the tree stays
    capi.check(capi.hip.shq_grav_short_run(ctx.h, C.byref(gp), None, 0, 1, sq.WALK_EXACT))
    capi.check(capi.hip.shq_grav_short_download(ctx.h, None, None, None, C.byref(st)))
    assert st.ntargets == 1500 and st.ninteractions == n1
    capi.check(capi.hip.shq_pm_set_deposit_log2scale(ctx.h, -1))     # the session context goes back to its own scale rule
    cm.reference_treepar()


def _sph_worker(rank, world, initfile, outdir):
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    try:
        import shenqi_amd as sq
        from shenqi_amd import dist as sd
        from test_dist_sph_cpu import global_gas
        comm = sd.Comm()
        decomp = sd.SlabDecomp(comm, NMESH, BOX)
        Pg, Sg = global_gas()
        mine = (decomp.owner_of(torch.from_numpy(np.ascontiguousarray(Pg["Pos"][:, 0]))) == rank).numpy()
        P = Pg[mine].copy()
        SphP = Sg[Pg["PI"][mine]].copy()
        P["PI"] = np.arange(len(P))
        sq.set_densitypar(DensityResolutionEta=1.0, MaxNumNgbDeviation=0.5, DensityKernelType=1, BlackHoleNgbFactor=2.0, MinGasHsml=0.006)
        sq.set_hydropar(DensityIndependentSphOn=1, DensityContrastLimit=100.0, ArtBulkViscConst=0.75)
        with sq.Context(0) as ctx:
            drv = sd.DistSPH(comm, decomp, sd.GpuSphOps(ctx, BOX))
            drv.density(P, SphP)
            drv.hydro(P, SphP, atime=0.1, hubble=0.1)
        np.save(os.path.join(outdir, "p%d.npy" % rank), P)
        np.save(os.path.join(outdir, "s%d.npy" % rank), SphP)
    finally:
        dist.destroy_process_group()


def test_dist_sph_two_gloo_ranks_one_gpu():
    """sharded density (Hsml loop) + hydro with the DEVICE operators on two ranks sharing the GPU (records exchanged over gloo)
    against the oracle on the undivided gas"""
    from test_dist_sph_cpu import monolithic
    Pm, Sm, _ = monolithic(1.5)
    key = {int(i): k for k, i in enumerate(Pm["ID"])}
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_sph_worker, args=(2, os.path.join(tmp, "init"), tmp), nprocs=2, join=True)
        seen = 0
        for r in range(2):
            P = np.load(os.path.join(tmp, "p%d.npy" % r))
            S = np.load(os.path.join(tmp, "s%d.npy" % r))
            idx = np.array([key[int(i)] for i in P["ID"]])
            seen += len(idx)
            assert np.abs(P["Hsml"] / Pm["Hsml"][idx] - 1).max() < 1e-9
            for name in ("Density", "EgyWtDensity", "DivVel", "CurlVel"):
                assert np.abs(S[name] - Sm[name][idx]).max() < 1e-8 * np.abs(Sm[name]).max(), name
            assert np.abs(S["HydroAccel"] - Sm["HydroAccel"][idx]).max() < 1e-7 * np.abs(Sm["HydroAccel"]).max()
            assert np.abs(S["MaxSignalVel"] / Sm["MaxSignalVel"][idx] - 1).max() < 1e-8
        assert seen == 16**3


def _fof_worker(rank, world, initfile, outdir):
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    try:
        import pickle
        import shenqi_amd as sq
        from shenqi_amd import dist as sd
        import test_dist_fof_cpu as tf
        comm = sd.Comm()
        decomp = sd.SlabDecomp(comm, tf.NMESH, tf.BOX)
        Pg = tf.global_set()
        mine = (decomp.owner_of(torch.from_numpy(np.ascontiguousarray(Pg["Pos"][:, 0]))) == rank).numpy()
        P = Pg[mine].copy()
        with sq.Context(0) as ctx:
            drv = sd.DistFOF(comm, decomp, sd.GpuFofOps(ctx, tf.BOX))
            minid, groups, grnr = drv.fof(P, tf.LINKL, tf.MINLEN)
        with open(os.path.join(outdir, "r%d.pkl" % rank), "wb") as f:
            pickle.dump(dict(ids=P["ID"], minid=minid, groups=groups, grnr=grnr, rounds=drv.rounds, nghost=drv.nghost), f)
    finally:
        dist.destroy_process_group()


def test_dist_fof_two_gloo_ranks_one_gpu():
    """sharded friends-of-friends with the DEVICE labelling (shq_fof) on two ranks sharing the GPU, labels exchanged over gloo,
    against the oracle on the undivided set"""
    import pickle
    import test_dist_fof_cpu as tf
    P, ref = tf.monolithic()
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_fof_worker, args=(2, os.path.join(tmp, "init"), tmp), nprocs=2, join=True)
        results = []
        for r in range(2):
            with open(os.path.join(tmp, "r%d.pkl" % r), "rb") as f:
                results.append(pickle.load(f))
        tf.check(results, P, ref)
        assert max(res["rounds"] for res in results) >= 2


def _exchange_worker(rank, world, initfile, outdir, maxlast):
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    try:
        import pickle
        import shenqi_amd as sq
        from shenqi_amd import capi, dist as sd
        import exchange_fixtures as fx
        P, numpart, slots, slot_size = fx.setup_task(rank, world, [40, 13, 0, 7, 25, 3], maxpart=256)
        f = capi.PARTICLE_DTYPE.fields
        L = capi.ExchangeLayout()
        L.part_elsize, L.off_flags, L.off_type, L.off_pi = capi.PARTICLE_DTYPE.itemsize, f["Flags"][1], f["Type"][1], f["PI"][1]
        for t in range(6):
            L.slot_elsize[t] = 0 if fx.SLOT_DTYPES[t] is None else fx.SLOT_DTYPES[t].itemsize
        L.off_reverselink = 0
        esz, idoff = int(L.part_elsize), f["ID"][1]
        dev = "cuda:0"
        d_parts = torch.from_numpy(P.view(np.uint8).reshape(-1).copy()).to(dev)
        d_slots = [None if s is None else torch.from_numpy(s.view(np.uint8).reshape(-1).copy()).to(dev) for s in slots]

        def layoutfn(parts, n):                              # TestExchangePlan::layoutfunc: ID % NTask, on the device records
            ids = parts.view(-1, esz)[:, idoff:idoff + 8].contiguous().view(torch.int64).reshape(-1)
            return (ids % world).to(torch.int32)
        comm = sd.Comm()
        with sq.Context(0) as ctx:
            ex = sd.DistExchange(comm, ctx, L)
            n, sz = ex.exchange(d_parts, numpart, d_slots, slot_size, layoutfn, maxlast=maxlast)
        with open(os.path.join(outdir, "x%d.pkl" % rank), "wb") as fh:
            pickle.dump(dict(P=d_parts.cpu().numpy().view(capi.PARTICLE_DTYPE), n=n, sz=sz, rounds=ex.rounds,
                             S=[None if s is None else s.cpu().numpy().view(fx.SLOT_DTYPES[t]) for t, s in enumerate(d_slots)]), fh)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("maxlast", [0, 9])
def test_dist_exchange_two_gloo_ranks_one_gpu(maxlast):
    """DistExchange: the device exchange loops (plan / pack / gc / unpack) with the collectives over a real process group — two ranks
    sharing the GPU — end in the same particle and slot arrays as the restatement of ExchangePlan::domain_exchange; with a cap of 9
    list entries per round the same number of rounds, a collection in each"""
    import pickle
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import exchange as oex
    import exchange_fixtures as fx
    world = 2
    otasks = []
    for r in range(world):
        P, numpart, slots, slot_size = fx.setup_task(r, world, [40, 13, 0, 7, 25, 3], maxpart=256)
        otasks.append(oex.Task(P, numpart, slots, slot_size))
    lay = [lambda P, n: fx.layout_id_mod(P, n, world)] * world
    oit = oex.domain_exchange_batched(otasks, lay, maxlast if maxlast else 10**9)
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_exchange_worker, args=(world, os.path.join(tmp, "init"), tmp, maxlast), nprocs=world, join=True)
        out = []
        for r in range(world):
            with open(os.path.join(tmp, "x%d.pkl" % r), "rb") as fh:
                d = pickle.load(fh)
            o = otasks[r]
            assert d["n"] == o.numpart and d["sz"] == o.slot_size and d["rounds"] == oit
            assert all(np.array_equal(d["P"][k][:d["n"]], o.parts[k][:o.numpart]) for k in o.parts.dtype.names)
            for t in range(6):
                if o.slots[t] is not None:
                    assert all(np.array_equal(d["S"][t][k][:d["sz"][t]], o.slots[t][k][:o.slot_size[t]]) for k in o.slots[t].dtype.names), (r, t)
            out.append((d["P"], d["n"], d["S"], d["sz"]))
        fx.check_after(out, world, world * 88)
        assert oit == (1 if maxlast == 0 else oit) and (maxlast == 0 or oit >= 4)


def _winds_worker(rank, world, initfile, outdir):
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    try:
        import pickle
        import shenqi_amd as sq
        from shenqi_amd import dist as sd
        import test_dist_winds_cpu as tdw
        with sq.Context(0) as ctx:
            res = tdw.run_rank(sd.Comm(), rank, sd.GpuWindOps(ctx, tdw.global_set()[5].BoxSize))
        with open(os.path.join(outdir, "r%d.pkl" % rank), "wb") as f:
            pickle.dump(res, f)
    finally:
        dist.destroy_process_group()


def test_dist_winds_two_gloo_ranks_one_gpu():
    """DistWinds with the DEVICE walks and kicks (shq_winds_candidates / shq_winds_apply) on two ranks sharing the GPU: the kick
    candidates that fall on imported ghosts travel to their owners; same kicks as the undivided restatement"""
    import pickle
    import test_dist_winds_cpu as tdw
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_winds_worker, args=(2, os.path.join(tmp, "init"), tmp), nprocs=2, join=True)
        results = []
        for r in range(2):
            with open(os.path.join(tmp, "r%d.pkl" % r), "rb") as f:
                results.append(pickle.load(f))
        tdw.check(results, exact=False)
        assert all(r["nghost"] > 0 for r in results)


@pytest.mark.parametrize("ngrid,total,shared", [(32, 40, False), (26, 32, True)])
def test_bench_driver_command_two_gloo_ranks(ngrid, total, shared):
    """The driver's own multi-GPU command — `python bench.py --gpus 2`, which starts its ranks through torch.distributed.run
    (bench.launch_ranks) — rehearsed on the one-GPU box with SHQ_BENCH_BACKEND=gloo (both ranks share device 0, exchanges staged
    through the host): one JSON line on stdout, whole-job value, weak scaling, the sharded walk's roofline, and the sampled
    force check of the global particle set against direct summation.  ngrid 26 -> 32^3 particles on a 96 mesh, which has the
    library's own transforms: there the rebalanced slabs share the plane the cut falls in (config.slab_ycuts)."""
    import json
    import subprocess
    env = dict(os.environ, SHQ_BENCH_BACKEND="gloo", OMP_NUM_THREADS="2")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--ngrid", str(ngrid), "--steps", "1", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["unit"] == "particle-steps/s" and out["value"] > 0
    assert out["config"]["particles_total"] == total**3       # ngrid^3 per GPU, rounded to a multiple of 2 x ranks per dimension
    yc = out["config"]["slab_ycuts"]
    assert (yc is not None and len(yc) == 3 and yc[0] == 0 and yc[2] == 0) if shared else yc is None, yc
    assert out["roofline"]["bound"] == "valu-f64" and 0 < out["roofline"]["frac"] < 1
    fe = out["force_error"]
    assert "mean" in fe, fe
    # the reference's own limits for PM + tree against the +-1 image sum (tests/test_gravity.cpp:294-355): mean < 0.8 x, max < 3 x ErrTol
    # hold at its 16^3 size; at this size and ErrTolForceAcc 0.005 the check is a guard against gross errors (wrong ghosts, a missing slab)
    assert fe["mean"] < 0.02 and fe["max"] < 0.1, fe


def _sph_device_worker(rank, world, initfile, outdir):
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    try:
        import shenqi_amd as sq
        from shenqi_amd import dist as sd
        import common as cm
        from test_dist_sph_cpu import global_gas
        comm = sd.Comm()
        decomp = sd.SlabDecomp(comm, NMESH, BOX)
        Pg, Sg = global_gas()
        mine = (decomp.owner_of(torch.from_numpy(np.ascontiguousarray(Pg["Pos"][:, 0]))) == rank).numpy()
        P = Pg[mine].copy()
        SphP = Sg[Pg["PI"][mine]].copy()
        P["PI"] = np.arange(len(P))
        rows = torch.from_numpy(sd.gas_rows_from_records(P, SphP)).cuda()
        with sq.Context(0) as ctx:
            drv = sd.DistSPHDevice(comm, decomp, ctx, BOX)
            rounds = drv.density(rows, cm.density_params(update_hsml=1, DoEgyDensity=1))
            drv.hydro(rows, cm.hydro_params())
            ctx.synchronize()
        sd.gas_rows_to_records(rows.cpu().numpy(), P, SphP)
        np.save(os.path.join(outdir, "p%d.npy" % rank), P)
        np.save(os.path.join(outdir, "s%d.npy" % rank), SphP)
        np.save(os.path.join(outdir, "r%d.npy" % rank), np.array([rounds, drv.nghost]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2])
def test_dist_sph_device_resident(world):
    """sharded density (Hsml loop) + hydro with the gas RESIDENT on the device (DistSPHDevice: ghost rows as device tensors, tree built on
    the device, shq_density_resident / shq_hydro_resident, results gathered into the rows) on one rank and on two gloo ranks sharing the
    GPU, against the oracle on the undivided gas — the same bar as the host-staged DistSPH above"""
    from test_dist_sph_cpu import monolithic
    Pm, Sm, _ = monolithic(1.5)
    key = {int(i): k for k, i in enumerate(Pm["ID"])}
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_sph_device_worker, args=(world, os.path.join(tmp, "init"), tmp), nprocs=world, join=True)
        seen = 0
        for r in range(world):
            P = np.load(os.path.join(tmp, "p%d.npy" % r))
            S = np.load(os.path.join(tmp, "s%d.npy" % r))
            rounds, nghost = np.load(os.path.join(tmp, "r%d.npy" % r))
            assert rounds >= 1 and (nghost > 0) == (world > 1)
            idx = np.array([key[int(i)] for i in P["ID"]])
            seen += len(idx)
            assert np.abs(P["Hsml"] / Pm["Hsml"][idx] - 1).max() < 1e-9
            for name in ("Density", "EgyWtDensity", "DivVel", "CurlVel"):
                assert np.abs(S[name] - Sm[name][idx]).max() < 1e-8 * np.abs(Sm[name]).max(), name
            assert np.abs(S["HydroAccel"] - Sm["HydroAccel"][idx]).max() < 1e-7 * np.abs(Sm["HydroAccel"]).max()
            assert np.abs(S["MaxSignalVel"] / Sm["MaxSignalVel"][idx] - 1).max() < 1e-8
        assert seen == 16**3


def test_bench_c5_mode_two_gloo_ranks():
    """`bench.py --gpus 2 --workload c5`: the C5-shaped step (gas + dark matter: sharded TreePM + device-resident sharded density and
    hydro) through the driver's launch path, rehearsed over gloo on one GPU: one JSON line with the three operators' rooflines"""
    import json
    import subprocess
    env = dict(os.environ, SHQ_BENCH_BACKEND="gloo", OMP_NUM_THREADS="2")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--ngrid", "16", "--steps", "1", "--warmup", "0",
                        "--workload", "c5"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["config"]["particles_total"] == 2 * 20**3
    k = out["kernels"]
    assert k["gas_ghosts_density"] > 0 and k["gas_ghosts_hydro"] > 0 and k["grav_ghosts"] > 0
    assert 1 <= k["density_iterations"] <= 3          # steady state: Hsml converged by the set-up call
    for r in ("roofline", "roofline_sph_density", "roofline_sph_hydro"):
        assert out[r]["achieved"] > 0, r


def test_bench_c5_mode_one_rank_rccl():
    """`bench.py --workload c5` as a ONE-rank RCCL group (SHQ_COMM_FORCE=1: every exchange of the sharded TreePM and of the
    device-resident sharded SPH is a real RCCL collective on the one GPU - self-copies, so no scaling figure, but the nccl code
    paths of Comm, the device-tensor ghost rows and the stream hand-overs all run): one JSON line, the three operators'
    rooflines, a force check of the global set against direct summation."""
    import json
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, SHQ_COMM_FORCE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("SHQ_BENCH_BACKEND", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--ngrid", "32", "--steps", "1", "--warmup", "0",
                        "--workload", "c5"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["config"]["particles_total"] == 2 * 32**3
    k = out["kernels"]
    assert 1 <= k["density_iterations"] <= 3
    for r in ("roofline", "roofline_sph_density", "roofline_sph_hydro"):
        assert out[r]["achieved"] > 0, r
