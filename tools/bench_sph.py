"""Time SPH density (with the Hsml loop) and hydro on the device for BASELINE configs[2]-like input:
n^3 gas particles (S-cluster or uniform), quintic kernel, pressure-entropy SPH.  Reports the HIP-event
time of the walk kernels (the one-shot C-ABI calls also move the AoS arrays over PCIe; that is not timed)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OMP_NUM_THREADS", str(len(os.sched_getaffinity(0))))
import shenqi_amd as sq  # noqa: E402

n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 128
kind = sys.argv[2] if len(sys.argv) > 2 else "cluster"
kernel = int(sys.argv[3]) if len(sys.argv) > 3 else 2
n = n1**3
L = 1.0
pos = sq.synth_positions(kind, n, L=L)
pos = pos[sq.hilbert_order(pos, L)]
pman = sq.PartManager(n, L)
P = pman.Base
P["Pos"] = pos
P["Mass"] = 1.0
P["Type"] = 0
P["PI"] = np.arange(n)
P["Vel"] = np.random.default_rng(1).normal(size=(n, 3)) * 0.01
P["Hsml"] = 1.5 * L / n1
SphP = np.zeros(n, dtype=sq.SPH_DTYPE)
SphP["Entropy"] = 1
SphP["Density"] = 1
BhP = np.zeros(2, dtype=sq.BH_SLOT_DTYPE)
sq.set_densitypar(DensityResolutionEta=1.0, MaxNumNgbDeviation=0.5, DensityKernelType=kernel, BlackHoleNgbFactor=2.0,
                  MinGasHsml=1e-6)
t0 = time.time()
tree = sq.force_tree_rebuild_mask(pman, sq.GASMASK)
sq.set_init_hsml(tree, L / n1, pman)
tree = sq.force_tree_rebuild_mask(pman, sq.GASMASK)
print("tree+init hsml %.1f s, DesNumNgb %.1f" % (time.time() - t0, sq.GetNumNgb()), flush=True)
with sq.Context(0) as ctx:
    for rep in range(2):
        h0 = P["Hsml"].copy()
        t0 = time.time()
        evp, st = sq.density(ctx, None, 1, 1, 0, None, tree, pman, SphP, BhP)
        t1 = time.time() - t0
        print("density: %d iterations, %.1f candidates/target(all iters), kernels %.2f ms (call %.2f s) -> %.3g particles/s"
              % (st.niterations, st.ninteractions / st.ntargets, st.kernel_ms, t1, n / (st.kernel_ms * 1e-3)), flush=True)
    sq.force_tree_update_hmax(tree, pman)
    sq.set_hydropar(1, 100.0, 0.75)
    for rep in range(2):
        t0 = time.time()
        hs = sq.hydro_force(ctx, None, 0.1, 0.1, evp, None, tree, pman, SphP)
        t1 = time.time() - t0
        print("hydro: %.1f candidates/target, kernels %.2f ms (call %.2f s) -> %.3g particles/s"
              % (hs.ninteractions / hs.ntargets, hs.kernel_ms, t1, n / (hs.kernel_ms * 1e-3)), flush=True)
print("mean Hsml %.5f  mean rho %.4g  max|a| %.3g" % (P["Hsml"].mean(), SphP["Density"].mean(), np.abs(SphP["HydroAccel"]).max()))
