"""shq_fof on the bench's dm-only S-cluster box (n1^3 particles, linking length 0.2 mean separations): wall time of the whole call and
the size of the catalogue."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", str(len(os.sched_getaffinity(0))))
import shenqi_amd as sq  # noqa: E402
from shenqi_amd import capi  # noqa: E402

n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kind = sys.argv[2] if len(sys.argv) > 2 else "cluster"
n = n1 ** 3
L = 1.0
pos = sq.synth_positions(kind, n, L=L)
pman = sq.PartManager(n, L)
pman.Base["Pos"] = pos
pman.Base["Type"] = 1
pman.Base["Mass"] = 1.0
pv = pman.view()
c = sq.Context(0)
capi.check(capi.hip.shq_particles_upload(c.h, C.byref(pv)))
fp = capi.FofParams(L, 0.2 * L / n1, 2, 1 + 16 + 32, 32, 0)
ids = np.arange(1, n + 1, dtype=np.uint64)
ng = C.c_int64()
for rep in range(3):
    t0 = time.perf_counter()
    capi.check(capi.hip.shq_fof(c.h, C.byref(fp), capi.ptr(ids), None, None, C.byref(ng)))
    dt = time.perf_counter() - t0
    groups = np.zeros(ng.value, dtype=capi.FOF_GROUP_DTYPE)
    capi.check(capi.hip.shq_fof_groups_download(c.h, capi.ptr(groups), len(groups)))
    print("%s %d^3: shq_fof %.1f ms, %d groups of >= 32, largest %d, %d particles in groups" %
          (kind, n1, dt * 1e3, ng.value, groups["Length"].max() if len(groups) else 0, groups["Length"].sum()), flush=True)
c.close()
