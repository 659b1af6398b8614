"""Time the sharded driver (slab PM + ghost tree) with ONE rank at full per-GPU size, phase by phase:
what every rank of a multi-GPU run computes, without the exchanges."""
import ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import shenqi_amd as sq
from shenqi_amd import capi, dist as sd
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = n1**3; L = 1.0; G = 43.0071; RHO0 = 0.3 * 3 * 0.1 * 0.1 / (8 * np.pi * G); nmesh = 3 * n1
dev = torch.device("cuda", 0)
pos = sq.synth_positions("cluster", n, L=L)
posm = torch.from_numpy(np.concatenate([pos, np.ones((n, 1))], axis=1)).to(dev)
comm = sd.Comm()
ctx = sq.Context(0)
sq.set_gravshort_treepar(ErrTolForceAcc=0.005, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=1, Rcut=6.0)
sq.gravshort_set_softenings(L / n1)
gp_bh = sq.make_grav_params(L, 1.5, nmesh, G, RHO0)
sq.set_gravshort_treepar(ErrTolForceAcc=0.005, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=0, Rcut=6.0)
gp = sq.make_grav_params(L, 1.5, nmesh, G, RHO0)
drv = sd.DistTreePM(comm, ctx, nmesh, L, 1.5, G, dev, halo_factor=1.5)
local = sd.exchange_to_owner(comm, drv.decomp, posm)
drv.setup(local, gp.Rcut)
drv.step(gp_bh)
torch.cuda.synchronize(); ctx.synchronize()
for it in range(3):
    t = {}
    def T(name, fn):
        torch.cuda.synchronize(); ctx.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ctx.synchronize()
        t[name] = round(1e3 * (time.perf_counter() - t0), 2)
    T("ghosts+set_particles", drv._load_particles)
    T("pm", drv.pm.force)
    T("walk", lambda: capi.check(capi.hip.shq_grav_short_run(ctx.h, C.byref(gp), None, 0, 1, 0)))
    T("oldacc", lambda: capi.check(capi.hip.shq_grav_refresh_oldacc(ctx.h, G)))
    print(t, "sum", round(sum(t.values()), 2), flush=True)
if hasattr(drv.pm, "last_phase_ms"):
    print(drv.pm.last_phase_ms)
