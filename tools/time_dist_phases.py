"""Where the sharded driver's step goes on one GPU (one-rank RCCL group, SHQ_COMM_FORCE=1): wall time per phase, every phase
bracketed by device synchronisation (so nothing overlaps here).  Usage: time_dist_phases.py [n1]"""
import collections
import ctypes as C
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as tdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SHQ_COMM_FORCE"] = "1"
os.environ["SHQ_DIST_OVERLAP"] = "0"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29534")
import shenqi_amd as sq  # noqa: E402
from shenqi_amd import capi, dist as sd  # noqa: E402

n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 256
G = 43.0071
RHO0 = 0.3 * 3 * 0.1 * 0.1 / (8 * np.pi * G)
n, L, nmesh = n1**3, 1.0, 3 * n1
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
tdist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
pos = sq.synth_positions("cluster", n, seed=20240601, L=L)
sq.set_gravshort_treepar(ErrTolForceAcc=0.005, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=1, Rcut=6.0)
sq.gravshort_set_softenings(L / n1)
gp_bh = sq.make_grav_params(L, 1.5, nmesh, G, RHO0)
sq.set_gravshort_treepar(ErrTolForceAcc=0.005, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=0, Rcut=6.0)
gp_rel = sq.make_grav_params(L, 1.5, nmesh, G, RHO0)
ctx = sq.Context(0, stream=torch.cuda.current_stream().cuda_stream)
comm = sd.Comm()
posm = torch.from_numpy(np.concatenate([pos, np.ones((n, 1))], axis=1)).to(dev)
drv = sd.DistTreePM(comm, ctx, nmesh, L, 1.5, G, dev)
drv.setup(sd.exchange_to_owner(comm, drv.decomp, posm), gp_rel.Rcut)
drv.step(gp_bh)
drv.step(gp_rel)

acc = collections.OrderedDict()


def timed(obj, name, label=None):
    f = getattr(obj, name)

    def g(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = f(*a, **k)
        if hasattr(r, "wait") and not isinstance(r, torch.Tensor):
            r = sd._Done(r.wait())
        torch.cuda.synchronize()
        ctx.synchronize()
        acc[label or name] = acc.get(label or name, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, name, g)


timed(drv, "_load_particles")
for m in ("mesh_buffer", "deposit2", "fft_yz", "xgreen", "readout2"):
    timed(drv.ops, m)
timed(comm, "all_to_all_rows_start", "transposes (collective only)")
timed(drv.pm, "force", "pm.force total")
NS = 3
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(NS):
    drv.step(gp_rel)
torch.cuda.synchronize()
ctx.synchronize()
tot = (time.perf_counter() - t0) / NS
for k, v in acc.items():
    print("%-32s %7.2f ms" % (k, 1e3 * v / NS))
print("%-32s %7.2f ms (walk + oldacc = the rest)" % ("step", 1e3 * tot))
tdist.destroy_process_group()
