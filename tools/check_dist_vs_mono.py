"""One GPU: the sharded driver (shenqi_amd/dist.py, one rank, no process group) against the monolithic entry points on the
same particles: interactions per target, walk time, PM and tree forces.  Usage: check_dist_vs_mono.py [n1]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import shenqi_amd as sq  # noqa: E402
from shenqi_amd import capi, dist as sd  # noqa: E402

n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 128
G = 43.0071
RHO0 = 0.3 * 3 * 0.1 * 0.1 / (8 * np.pi * G)
n, L, nmesh = n1**3, 1.0, 3 * n1
pos = sq.synth_positions("cluster", n, seed=20240601, L=L)
pos = pos[sq.hilbert_order(pos, L)]
sq.set_gravshort_treepar(ErrTolForceAcc=0.005, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=1, Rcut=6.0)
sq.gravshort_set_softenings(L / n1)
gp_bh = sq.make_grav_params(L, 1.5, nmesh, G, RHO0)
sq.set_gravshort_treepar(ErrTolForceAcc=0.005, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=0, Rcut=6.0)
gp_rel = sq.make_grav_params(L, 1.5, nmesh, G, RHO0)


def stats(ctx):
    st = sq.WalkStats()
    capi.check(capi.hip.shq_grav_short_download(ctx.h, None, None, None, C.byref(st)))
    return st


# monolithic
pman = sq.PartManager(n, L)
pman.Base["Pos"] = pos
pman.Base["Type"] = 1
pman.Base["Mass"] = 1.0
ctx = sq.Context(0)
pv = pman.view()
capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
sq.tree_build_device(ctx, L)
pmp = sq.PMParams(nmesh, 0, L, 1.5, G)
import time  # noqa: E402
for gp in (gp_bh, gp_rel, gp_rel, gp_rel):
    ctx.synchronize()
    t0 = time.perf_counter()
    capi.check(capi.hip.shq_pm_run(ctx.h, C.byref(pmp)))
    capi.check(capi.hip.shq_grav_short_run(ctx.h, C.byref(gp), None, 0, 1, 0))
    capi.check(capi.hip.shq_grav_refresh_oldacc(ctx.h, G))
    ctx.synchronize()
    dt = time.perf_counter() - t0
    st = stats(ctx)
    print("mono: interactions/target %.1f walk %.2f ms, step %.2f ms" % (st.ninteractions / n, st.kernel_ms, 1e3 * dt), flush=True)
acc0 = np.zeros((n, 3))
capi.check(capi.hip.shq_grav_short_download(ctx.h, capi.ptr(acc0), None, None, None))
gpm0 = np.zeros((n, 3))
ppot0 = np.zeros(n)
capi.check(capi.hip.shq_pm_download(ctx.h, capi.ptr(gpm0), capi.ptr(ppot0)))
ctx.close()

# sharded driver, one rank; SHQ_COMM_FORCE=1: in a one-rank RCCL group, every exchange a real collective
dev = torch.device("cuda", 0)
if os.environ.get("SHQ_COMM_FORCE", "0") == "1":
    import torch.distributed as tdist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    tdist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
work_stream = torch.cuda.Stream(device=dev)     # shared by torch and the library: no host synchronisation inside a step
torch.cuda.set_stream(work_stream)
ctx = sq.Context(0, stream=work_stream.cuda_stream)
comm = sd.Comm()
posm = torch.from_numpy(np.concatenate([pos, np.ones((n, 1))], axis=1)).to(dev)
drv = sd.DistTreePM(comm, ctx, nmesh, L, 1.5, G, dev, halo_factor=1.5, bounds=None)
drv.setup(sd.exchange_to_owner(comm, drv.decomp, posm), gp_rel.Rcut)
for gp in (gp_bh, gp_rel, gp_rel, gp_rel):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    drv.step(gp)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = stats(ctx)
    print("dist: interactions/target %.1f walk %.2f ms, step %.2f ms" % (st.ninteractions / n, st.kernel_ms, 1e3 * dt), flush=True)
acc1, pot1, gpm1, ppot1 = drv.download()
p1 = drv.local.cpu().numpy()[:, :3]
print("same particle order:", bool(np.array_equal(p1, pos)))
print("PM  max |d| / max |f|: %.3e" % (np.abs(gpm1 - gpm0).max() / np.abs(gpm0).max()))
print("tree max |d| / max |f|: %.3e" % (np.abs(acc1 - acc0).max() / np.abs(acc0).max()))
