"""Wall time of the sub-grid tree walks on a gas box: shq_bh_accretion / shq_bh_feedback, shq_winds_and_feedback, shq_metal_return
(with shq_stellar_density before it).  n1^3 gas particles on a jittered grid, nstar stars, nbh black holes.  The calls are one-shot
(views in, arrays out): the wall times include the host marshalling and the uploads; the kernel times come from a rocprofv3
--kernel-trace --stats run of this script (profiles/r02_subgrid_kernel_stats.csv)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OMP_NUM_THREADS", str(len(os.sched_getaffinity(0))))
import shenqi_amd as sq  # noqa: E402
from shenqi_amd import capi  # noqa: E402

n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 96
nstar = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
nbh = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
CURRENT = len(sys.argv) > 4 and sys.argv[4] == "current"      # shq_set_inputs_current after the first call: later calls skip the uploads
BOX = 8.0
rng = np.random.default_rng(1)
ngas = n1 ** 3
sp = BOX / n1
g = np.arange(ngas)
gas = np.stack([g // n1 // n1, (g // n1) % n1, g % n1], axis=1) * sp
gas = np.mod(gas + rng.normal(size=(ngas, 3)) * 0.25 * sp, BOX)
pos = np.concatenate([gas, rng.random((nstar, 3)) * BOX, rng.random((nbh, 3)) * BOX])
n = len(pos)
pman = sq.PartManager(n, BOX)
P = pman.Base
P["Pos"] = pos
P["Type"][:ngas] = 0
P["Type"][ngas:ngas + nstar] = 4
P["Type"][ngas + nstar:] = 5
P["Mass"] = 1.0
P["ID"] = rng.permutation(n).astype(np.uint64) * 3 + 5
P["Vel"] = rng.normal(size=(n, 3)) * 30
P["FullTreeGravAccel"] = rng.normal(size=(n, 3)) * 50
P["Hsml"] = sp * 1.6
P["Hsml"][ngas + nstar:] = sp * 2.2
P["TimeBinGravity"] = 20
P["TimeBinHydro"] = 18
P["PI"][:ngas] = np.arange(ngas)
P["PI"][ngas:ngas + nstar] = np.arange(nstar)
P["PI"][ngas + nstar:] = np.arange(nbh)
S = np.zeros(ngas, dtype=capi.SPH_DTYPE)
S["Entropy"] = 100.0
S["Density"] = ngas / BOX ** 3
S["Metallicity"] = 0.01
ST = np.zeros(nstar, dtype=capi.STAR_DTYPE)
ST["VDisp"] = 30.0
B = np.zeros(nbh, dtype=capi.BH_DTYPE)
B["Mass"] = 4.0
B["Density"] = ngas / BOX ** 3
B["Mtrack"] = 1.0
B["CountProgs"] = 1
kf = sq.KickFactors()
for b in range(47):
    kf.gravkicks[b] = 1e-4
    kf.hydrokicks[b] = 1e-4
    kf.dloga_for_bin[b] = 1e-7 * 2.0 ** (b - 16) if b > 0 else 0.0
rnd = rng.random(8191)
ids = np.ascontiguousarray(P["ID"])
c = sq.Context(0)
t0 = time.perf_counter()
tree_gb = sq.force_tree_rebuild_mask(pman, sq.GASMASK + sq.BHMASK)
sq.force_tree_update_hmax(tree_gb, pman)
tree_g = sq.force_tree_rebuild_mask(pman, sq.GASMASK)
print("%d^3 gas, %d stars, %d black holes; host trees %.0f ms" % (n1, nstar, nbh, (time.perf_counter() - t0) * 1e3), flush=True)
pv, sv = pman.view(), capi.sph_view(S)


def timed(name, fn, reps=2):
    for r in range(reps):
        t0 = time.perf_counter()
        out = fn()
        print("  %-22s %8.1f ms  %s" % (name, (time.perf_counter() - t0) * 1e3, out if out is not None else ""), flush=True)


# ---- black holes
sys.path.insert(0, os.path.join(ROOT, "tests"))
from blackhole_fixtures import params as bh_params, make_work  # noqa: E402
cp, _ = bh_params(BoxSize=BOX)
w, cw = make_work(ngas, nbh)
bv = capi.bh_slot_view(B)
queue = np.ascontiguousarray(np.arange(ngas + nstar, n, dtype=np.int32))
tv = tree_gb.view()


def acc():
    capi.check(capi.hip.shq_bh_accretion(c.h, C.byref(tv), C.byref(pv), C.byref(sv), C.byref(bv), capi.ptr(ids), capi.ptr(queue), len(queue), C.byref(kf), C.byref(cp),
                                         1 << 18, capi.ptr(rnd), len(rnd), C.byref(cw)))
    return "%d gas marked" % int((w["SPH_SwallowID"] != 0).sum())


def fb():
    ns, nb = C.c_int64(), C.c_int64()
    capi.check(capi.hip.shq_bh_feedback(c.h, C.byref(tv), C.byref(pv), C.byref(sv), C.byref(bv), capi.ptr(ids), capi.ptr(queue), len(queue), C.byref(kf), C.byref(cp),
                                        n + 100, capi.ptr(rnd), len(rnd), None, C.byref(cw), C.byref(ns), C.byref(nb)))
    return "%d gas swallowed" % ns.value


timed("shq_bh_accretion", acc, 1)
if CURRENT:
    capi.check(capi.hip.shq_set_inputs_current(c.h, 15))
timed("shq_bh_accretion", acc, 1)          # the second call: no first-use allocations; with `current`, no uploads either
timed("shq_bh_feedback", fb, 1)
# the feedback changed the particle set (garbage): rebuild the gas tree
tree_g = sq.force_tree_rebuild_mask(pman, sq.GASMASK)
tvg = tree_g.view()
# ---- winds
wp = capi.WindParams(BOX, 0.2, 20.0, 0.6, 2.0, 350.0, 353.0, 3.7, 100.0, 0.0, 4 + 2, 0)
stv = capi.StarView(ST.ctypes.data, ST.dtype.itemsize, len(ST), ST.dtype.fields["VDisp"][1])
new = np.ascontiguousarray(np.arange(ngas, ngas + nstar, dtype=np.int32))


def winds():
    nk, na = C.c_int64(), C.c_int64()
    capi.check(capi.hip.shq_winds_and_feedback(c.h, C.byref(tvg), C.byref(pv), C.byref(sv), C.byref(stv), capi.ptr(ids), capi.ptr(new), len(new), C.byref(wp),
                                               capi.ptr(rnd), len(rnd), None, None, 0, C.byref(nk), C.byref(na)))
    return "%d candidates, %d kicked" % (nk.value, na.value)


timed("shq_winds_and_feedback", winds, 1)
# ---- metal return
f = capi.SPH_DTYPE.fields
gv = capi.GasMetalView(S.ctypes.data, S.dtype.itemsize, len(S), f["Density"][1], f["Metallicity"][1], f["Metals"][1], 9, 0)
starvol = np.full(nstar, 1.0)                           # sum of wk * volume over the kernel
massgen = np.full(nstar, 0.01)
metalgen = massgen * 0.02
species = np.ascontiguousarray(np.repeat(metalgen[:, None] / 9, 9, axis=1))
mret = np.zeros(nstar)


def metals():
    npairs = C.c_int64()
    capi.check(capi.hip.shq_metal_return(c.h, C.byref(tvg), C.byref(pv), C.byref(gv), capi.ptr(new), len(new), capi.ptr(starvol), capi.ptr(massgen), capi.ptr(metalgen),
                                         capi.ptr(species), 4.0, 1, 1, capi.ptr(mret), C.byref(npairs)))
    return "%d pairs, %.3f returned" % (npairs.value, mret.sum())


timed("shq_metal_return", metals, 2)
c.close()
