"""256^3 S-cluster, resident: the exact walk in particle-index order against the same walk with targets in tree order
(WALK_TREE_ORDER), with and without the potential, before and after a drift — what the resident loop of bench.py pays for."""
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
pman.Base["Vel"] = np.random.default_rng(7).normal(size=(n, 3))
sq.set_gravshort_treepar(ErrTolForceAcc=0.005, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=1, Rcut=6.0)
sq.gravshort_set_softenings(L / n1)
gp_bh = sq.make_grav_params(L, 1.5, nmesh, G, RHO0)
sq.set_gravshort_treepar(ErrTolForceAcc=0.005, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=0, Rcut=6.0)
gp = sq.make_grav_params(L, 1.5, nmesh, G, RHO0)
pmp = sq.PMParams(nmesh, 0, L, 1.5, G)
pv = pman.view()
c = sq.Context(0)
capi.check(capi.hip.shq_particles_upload(c.h, C.byref(pv)))
sq.dynamics_upload(c, pman)
sq.tree_build_device(c, L)
capi.check(capi.hip.shq_pm_run(c.h, C.byref(pmp)))
capi.check(capi.hip.shq_grav_short_run(c.h, C.byref(gp_bh), None, 0, 1, 0))
capi.check(capi.hip.shq_grav_refresh_oldacc(c.h, G))
capi.check(capi.hip.shq_grav_short_run(c.h, C.byref(gp), None, 0, 1, 0))
capi.check(capi.hip.shq_grav_refresh_oldacc(c.h, G))
s = sq.WalkStats()


def t(label, pot, mode):
    ms = []
    for _ in range(3):
        capi.check(capi.hip.shq_grav_short_run(c.h, C.byref(gp), None, 0, pot, mode))
        capi.check(capi.hip.shq_grav_short_download(c.h, None, None, None, C.byref(s)))
        ms.append(s.kernel_ms)
    print("%-46s %.2f ms, %.1f interactions/target" % (label, min(ms), s.ninteractions / n), flush=True)


t("index order, potential", 1, 0)
t("index order, no potential", 0, 0)
t("tree order, no potential", 0, sq.WALK_TREE_ORDER)
for k in range(3):
    sq.drift(c, 1e-4 * L / n1, L)
    sq.tree_build_device(c, L)
    capi.check(capi.hip.shq_pm_run(c.h, C.byref(pmp)))
    t("after drift %d: tree order, no potential" % (k + 1), 0, sq.WALK_TREE_ORDER)
    t("after drift %d: index order, no potential" % (k + 1), 0, 0)
    capi.check(capi.hip.shq_grav_refresh_oldacc(c.h, G))
c.close()
