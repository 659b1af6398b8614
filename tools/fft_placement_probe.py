#!/usr/bin/env python3
"""tools/fft_placement_probe.py [Nmesh] [nbuf] — is the FFT pipeline's speed a property of the buffer the mesh landed in?  nbuf contexts
alive at once (each with its own mesh), the pipeline timed on each in turn, three rounds."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import shenqi_amd as sq
from shenqi_amd import capi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 768
nbuf = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = 4096
pman = sq.PartManager(n, 1.0)
P = pman.Base
P["Pos"], P["Type"], P["Mass"] = np.random.default_rng(1).random((n, 3)), 1, 1.0
ctxs = []
for b in range(nbuf):
    ctx = sq.Context()
    pv = pman.view()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    ctxs.append(ctx)
pmp = sq.PMParams(N, 0, 1.0, 1.5, 43.0071)
for rnd in range(3):
    row = []
    for ctx in ctxs:
        best = 1e9
        for it in range(3):
            capi.check(capi.hip.shq_pm_run(ctx.h, C.byref(pmp)))
            ctx.synchronize()
            ph = (C.c_double * 6)()
            capi.check(capi.hip.shq_pm_phase_ms(ctx.h, C.byref(ph)))
            best = min(best, ph[1] + ph[2] + ph[3])
        row.append("%.2f" % best)
    print("round", rnd, " ".join(row), flush=True)
