"""debug: DistSPHDevice on 2 gloo ranks sharing the GPU, prints operator statistics per rank"""
import os, sys, tempfile
import numpy as np, torch, torch.distributed as dist, torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker(rank, world, initfile):
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    import shenqi_amd as sq
    from shenqi_amd import dist as sd
    import common as cm
    from test_dist_sph_cpu import global_gas, NMESH, BOX
    comm = sd.Comm()
    decomp = sd.SlabDecomp(comm, NMESH, BOX)
    Pg, Sg = global_gas()
    mine = (decomp.owner_of(torch.from_numpy(np.ascontiguousarray(Pg["Pos"][:, 0]))) == rank).numpy()
    P = Pg[mine].copy(); SphP = Sg[Pg["PI"][mine]].copy(); P["PI"] = np.arange(len(P))
    rows = torch.from_numpy(sd.gas_rows_from_records(P, SphP)).cuda()
    with sq.Context(0) as ctx:
        drv = sd.DistSPHDevice(comm, decomp, ctx, BOX)
        r = drv.density(rows, cm.density_params(update_hsml=1, DoEgyDensity=1))
        print(rank, "density rounds", r, drv.stats["density"], "hsml max", float(rows[:, 7].max()), "rho", float(rows[:, 20].mean()), flush=True)
        drv.hydro(rows, cm.hydro_params())
        ctx.synchronize()
        print(rank, "hydro", drv.stats["hydro"], "max |hacc|", float(rows[:, 14:17].abs().max()), "maxsig", float(rows[:, 25].max()), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(worker, args=(world, os.path.join(tmp, "init")), nprocs=world, join=True)
