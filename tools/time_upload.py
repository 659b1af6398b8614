"""Wall time of the host -> device hand-over of the one-shot calls: shq_particles_upload and shq_tree_upload at 256^3."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import shenqi_amd as sq  # noqa: E402
from shenqi_amd import capi  # noqa: E402

n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = n1**3
L = 1.0
pos = sq.synth_positions("cluster", n, L=L)
pos = pos[sq.hilbert_order(pos, L)]
pman = sq.PartManager(n, L)
pman.Base["Pos"] = pos
pman.Base["Type"] = 1
pman.Base["Mass"] = 1.0
t0 = time.perf_counter()
tree = sq.force_tree_full(pman)
print("host tree build %.2f s, %d nodes" % (time.perf_counter() - t0, tree.numnodes), flush=True)
ctx = sq.Context(0)
pv, tv = pman.view(), tree.view()
for it in range(3):
    t0 = time.perf_counter()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    t1 = time.perf_counter()
    capi.check(capi.hip.shq_tree_upload(ctx.h, C.byref(tv)))
    t2 = time.perf_counter()
    print("pass %d: particles_upload %.1f ms (%.1f GB/s of the 160-byte records), tree_upload %.1f ms" %
          (it, 1e3 * (t1 - t0), 160.0 * n / (t1 - t0) / 1e9, 1e3 * (t2 - t1)), flush=True)

# the whole one-shot call (hand-over both ways + kernels)
sq.set_gravshort_treepar(ErrTolForceAcc=0.005, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=1, Rcut=6.0)
sq.gravshort_set_softenings(L / n1)
gp = sq.make_grav_params(L, 1.5, 3 * n1, 43.0071, 1.0)
acc = np.zeros((n, 3))
for it in range(2):
    t0 = time.perf_counter()
    st = sq.WalkStats()
    capi.check(capi.hip.shq_grav_short_tree(ctx.h, C.byref(tv), C.byref(pv), None, 0, C.byref(gp), capi.ptr(acc), 1, 0, C.byref(st)))
    t1 = time.perf_counter()
    print("one-shot shq_grav_short_tree (BH walk): %.1f ms wall, kernel %.1f ms" % (1e3 * (t1 - t0), st.kernel_ms), flush=True)
