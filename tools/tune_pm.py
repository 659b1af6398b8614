"""Time the PM pipeline phases for a few settings of an environment knob (one process, several contexts)."""
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
knob = sys.argv[2] if len(sys.argv) > 2 else "SHQ_FFT_XCD_K"
vals = sys.argv[3].split(",") if len(sys.argv) > 3 else ["0", "4", "8", "32"]
n = n1**3
pos = sq.synth_positions("cluster", n, L=1.0)
pos = pos[sq.hilbert_order(pos, 1.0)]
pman = sq.PartManager(n, 1.0)
pman.Base["Pos"] = pos
pman.Base["Type"] = 1
pman.Base["Mass"] = 1.0
pv = pman.view()
pmp = sq.PMParams(3 * n1, 0, 1.0, 1.5, 43.0071)
for v in vals:
    os.environ[knob] = v
    c = sq.Context(0)
    capi.check(capi.hip.shq_particles_upload(c.h, C.byref(pv)))
    best = None
    for rep in range(4):
        capi.check(capi.hip.shq_pm_run(c.h, C.byref(pmp)))
        ph = (C.c_double * 6)()
        capi.check(capi.hip.shq_pm_phase_ms(c.h, C.byref(ph)))
        ph = list(ph)
        if best is None or ph[5] < best[5]:
            best = ph
    print("%s=%s: deposit %.2f fft %.2f readout %.2f total %.2f ms" % (knob, v, best[0], best[1] + best[2] + best[3], best[4], best[5]), flush=True)
    c.close()
