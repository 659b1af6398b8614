"""A/B the walk-kernel variants on one resident workload (one process, interleaved rounds)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", str(len(os.sched_getaffinity(0))))
os.environ.setdefault("SHQ_WALK_STATS", "1")  # these tools print the wave-level counters
import shenqi_amd as sq  # noqa: E402
from shenqi_amd import capi  # noqa: E402

n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kind = sys.argv[2] if len(sys.argv) > 2 else "cluster"
variants = [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else "0,1,2,3".split(","))]
G = 43.0071
RHO0 = 0.3 * 3 * 0.1 * 0.1 / (8 * np.pi * G)
n = n1**3
L = 1.0
nmesh = 3 * n1
pos = sq.synth_positions(kind, n, L=L)
pos = pos[(sq.hilbert_order if os.environ.get("TUNE_ORDER", "morton") == "hilbert" else sq.morton_order)(pos, L)]
pman = sq.PartManager(n, L)
pman.Base["Pos"] = pos
pman.Base["Type"] = 1
pman.Base["Mass"] = 1.0
tree = sq.force_tree_full(pman)
sq.set_gravshort_treepar(ErrTolForceAcc=0.005, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=1, Rcut=6.0)
sq.gravshort_set_softenings(L / n1 * float(os.environ.get("TUNE_SOFT", "1")))
gp_bh = sq.make_grav_params(L, 1.5, nmesh, G, RHO0)
sq.set_gravshort_treepar(ErrTolForceAcc=0.005, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=0, Rcut=6.0)
gp_rel = sq.make_grav_params(L, 1.5, nmesh, G, RHO0)
pmp = sq.PMParams(nmesh, 0, L, 1.5, G)
pv, tv = pman.view(), tree.view()

ctxs = {}
ks = [int(k) for k in os.environ.get("TUNE_XCD_K", "16").split(",")]
for st in ks:
    for v in variants:
        os.environ["SHQ_WALK_VARIANT"] = str(v)
        os.environ["SHQ_XCD_K"] = str(st)
        c = sq.Context(0)
        capi.check(capi.hip.shq_particles_upload(c.h, C.byref(pv)))
        capi.check(capi.hip.shq_tree_upload(c.h, C.byref(tv)))
        capi.check(capi.hip.shq_pm_run(c.h, C.byref(pmp)))
        capi.check(capi.hip.shq_grav_short_run(c.h, C.byref(gp_bh), None, 0, 1, 0))
        capi.check(capi.hip.shq_grav_refresh_oldacc(c.h, G))
        c.synchronize()
        ctxs[(v, st)] = c
res = {k: [] for k in ctxs}
for rnd in range(3):
    for k, c in ctxs.items():
        capi.check(capi.hip.shq_grav_short_run(c.h, C.byref(gp_rel), None, 0, 1, 0))
        s = sq.WalkStats()
        capi.check(capi.hip.shq_grav_short_download(c.h, None, None, None, C.byref(s)))
        res[k].append(s.kernel_ms)
        last = s
for k in sorted(res):
    print("variant %d xcdK %d: walk ms min %.2f med %.2f" % (k[0], k[1], min(res[k]), sorted(res[k])[1]), flush=True)
print("order %s: visits/wave %.1f node rounds/wave %.1f lane eff %.3f" % (
    os.environ.get("TUNE_ORDER", "morton"), last.nnodes_visited / (n / 64.0), last.nwave_node_interactions / (n / 64.0),
    last.ninteractions / max(1.0, 64.0 * last.nwave_interactions)), flush=True)
