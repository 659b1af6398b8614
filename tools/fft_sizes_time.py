#!/usr/bin/env python3
"""tools/fft_sizes_time.py [Nmesh ...] — the five-pass FFT pipeline's time per mesh size (a PM run on a handful of particles; HIP events
of the library's phases), with the bytes it moves per second.  The multi-GPU bench's per-rank meshes are 960 / 1200 / 1024 planes of
those sizes; 768 is the one-GPU headline."""
import ctypes as C
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import shenqi_amd as sq
from shenqi_amd import capi

sizes = [int(a) for a in sys.argv[1:]] or [384, 768, 960, 1024, 1200, 1536]
n = 4096
pos = np.random.default_rng(1).random((n, 3))
pman = sq.PartManager(n, 1.0)
P = pman.Base
P["Pos"], P["Type"], P["Mass"] = pos, 1, 1.0
for N in sizes:
    ctx = sq.Context()
    pv = pman.view()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    pmp = sq.PMParams(N, 0, 1.0, 1.5, 43.0071)
    best = None
    for it in range(4):
        capi.check(capi.hip.shq_pm_run(ctx.h, C.byref(pmp)))
        ctx.synchronize()
        ph = (C.c_double * 6)()
        capi.check(capi.hip.shq_pm_phase_ms(ctx.h, C.byref(ph)))
        t = ph[1] + ph[2] + ph[3]
        best = t if best is None else min(best, t)
    zp = 2 * (((N // 2 + 1) + 3) // 4 * 4)
    gb = 5 * 2 * 8.0 * N * N * zp / 1e9
    print("Nmesh %5d: FFT pipeline %7.2f ms  (%.1f GB moved, %.2f TB/s)" % (N, best, gb, gb / best), flush=True)
    del ctx
