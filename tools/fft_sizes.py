"""Round-trip check + timing of the bespoke PM pipeline for the large mesh sizes (device only, random density)."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import shenqi_amd as sq
from shenqi_amd import capi
for nmesh, n1 in ((960, 160), (1200, 200), (1536, 256)):
    n = n1**3
    pos = sq.synth_positions("uniform", n, L=1.0)
    pman = sq.PartManager(n, 1.0); P = pman.Base
    P["Pos"] = pos; P["Type"] = 1; P["Mass"] = 1.0
    with sq.Context(0) as ctx:
        pv = pman.view()
        capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
        pmp = sq.PMParams(nmesh, 0, 1.0, 1.5, 43.0071)
        for _ in range(2):
            capi.check(capi.hip.shq_pm_run(ctx.h, C.byref(pmp)))
        ph = (C.c_double * 6)(); capi.check(capi.hip.shq_pm_phase_ms(ctx.h, C.byref(ph)))
        g = np.zeros((n, 3)); capi.check(capi.hip.shq_pm_download(ctx.h, capi.ptr(g), None))
        zp = 2 * (((nmesh // 2 + 1) + 3) // 4 * 4)
        gb = 10 * 8.0 * nmesh * nmesh * zp / 1e9
        print("Nmesh %d: fft %.2f ms (%.2f TB/s), total %.2f ms, |g| rms %.3e finite %s" % (nmesh, ph[1], gb / ph[1], ph[5], np.sqrt((g**2).mean()), np.isfinite(g).all()), flush=True)
