"""One resident 256^3 (or n1^3) S-cluster workload, Barnes-Hut seeding with the exact walk, then `reps` relative-criterion walks in
the given mode: the command the walk kernels are profiled with (rocprofv3 --kernel-trace / --pmc)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", str(len(os.sched_getaffinity(0))))
import shenqi_amd as sq  # noqa: E402
from shenqi_amd import capi  # noqa: E402

n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
G = 43.0071
RHO0 = 0.3 * 3 * 0.1 * 0.1 / (8 * np.pi * G)
n = n1**3
L = 1.0
nmesh = 3 * n1
pos = sq.synth_positions("cluster", n, L=L)
pos = pos[sq.hilbert_order(pos, L)]
pman = sq.PartManager(n, L)
pman.Base["Pos"] = pos
pman.Base["Type"] = 1
pman.Base["Mass"] = 1.0
sq.set_gravshort_treepar(ErrTolForceAcc=0.005, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=1, Rcut=6.0)
sq.gravshort_set_softenings(L / n1)
gp_bh = sq.make_grav_params(L, 1.5, nmesh, G, RHO0)
sq.set_gravshort_treepar(ErrTolForceAcc=0.005, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=0, Rcut=6.0)
gp_rel = sq.make_grav_params(L, 1.5, nmesh, G, RHO0)
pmp = sq.PMParams(nmesh, 0, L, 1.5, G)
pv = pman.view()
c = sq.Context(0)
capi.check(capi.hip.shq_particles_upload(c.h, C.byref(pv)))
sq.tree_build_device(c, L)
capi.check(capi.hip.shq_pm_run(c.h, C.byref(pmp)))
capi.check(capi.hip.shq_grav_short_run(c.h, C.byref(gp_bh), None, 0, 1, 0))
capi.check(capi.hip.shq_grav_refresh_oldacc(c.h, G))
capi.check(capi.hip.shq_grav_short_run(c.h, C.byref(gp_rel), None, 0, 1, 0))
capi.check(capi.hip.shq_grav_refresh_oldacc(c.h, G))
s = sq.WalkStats()
for r in range(reps):
    capi.check(capi.hip.shq_grav_short_run(c.h, C.byref(gp_rel), None, 0, 1, mode))
    capi.check(capi.hip.shq_grav_short_download(c.h, None, None, None, C.byref(s)))
    print("mode %d: walk %.2f ms, %.1f interactions/target, %.1f node tests/target" % (mode, s.kernel_ms, s.ninteractions / n, s.nnodes_visited / n),
          flush=True)
c.close()
if len(sys.argv) > 4:
    # sparse active lists: every k-th particle (what a deep time bin of the hierarchical integrator looks like)
    c = sq.Context(0)
    capi.check(capi.hip.shq_particles_upload(c.h, C.byref(pv)))
    sq.tree_build_device(c, L)
    capi.check(capi.hip.shq_pm_run(c.h, C.byref(pmp)))
    capi.check(capi.hip.shq_grav_short_run(c.h, C.byref(gp_bh), None, 0, 1, 0))
    capi.check(capi.hip.shq_grav_refresh_oldacc(c.h, G))
    for stride in (8, 64, 512):
        act = np.arange(0, n, stride, dtype=np.int32)
        for m in (0, 1):
            for r in range(2):
                capi.check(capi.hip.shq_grav_short_run(c.h, C.byref(gp_rel), capi.ptr(act), len(act), 0, m))
                capi.check(capi.hip.shq_grav_short_download(c.h, None, None, None, C.byref(s)))
            print("stride %d (%d targets) mode %d: walk %.2f ms, %.1f interactions/target" % (stride, len(act), m, s.kernel_ms, s.ninteractions / len(act)),
                  flush=True)
    c.close()
