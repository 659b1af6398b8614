"""Time the phases of a fully resident step (drift, device tree build, PM, walk, kicks) at 256^3."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import shenqi_amd as sq
from shenqi_amd import capi
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 256
MODE = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0
n = n1**3; L = 1.0; G = 43.0071; RHO0 = 0.3 * 3 * 0.1 * 0.1 / (8 * np.pi * G); nmesh = 3 * n1
pos = sq.synth_positions("cluster", n, L=L); pos = pos[sq.hilbert_order(pos, L)]
pman = sq.PartManager(n, L); P = pman.Base
P["Pos"] = pos; P["Type"] = 1; P["Mass"] = 1.0; P["Vel"] = np.random.default_rng(7).normal(size=(n, 3))
ctx = sq.Context(0)
sq.set_gravshort_treepar(ErrTolForceAcc=0.005, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=1, Rcut=6.0)
sq.gravshort_set_softenings(L / n1)
gp_bh = sq.make_grav_params(L, 1.5, nmesh, G, RHO0)
sq.set_gravshort_treepar(ErrTolForceAcc=0.005, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=0, Rcut=6.0)
gp = sq.make_grav_params(L, 1.5, nmesh, G, RHO0)
pmp = sq.PMParams(nmesh, 0, L, 1.5, G)
pv = pman.view()
capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
sq.dynamics_upload(ctx, pman)
sq.tree_build_device(ctx, L)
capi.check(capi.hip.shq_pm_run(ctx.h, C.byref(pmp)))
capi.check(capi.hip.shq_grav_short_run(ctx.h, C.byref(gp_bh), None, 0, 1, 0))
capi.check(capi.hip.shq_grav_refresh_oldacc(ctx.h, G))
gk = np.full(capi.TIMEBINS + 1, 1e-9)
names = ["drift", "tree_build", "pm", "walk", "oldacc", "kick_short", "kick_pm"]
acc = {k: [] for k in names}
for it in range(4):
    def t(name, fn):
        ctx.synchronize(); t0 = time.perf_counter(); fn(); ctx.synchronize(); acc[name].append(1e3 * (time.perf_counter() - t0))
    if not os.environ.get("NODRIFT"):
        t("drift", lambda: sq.drift(ctx, float(os.environ.get("DDRIFT", "1e-4")) * L / n1, L))
    t("tree_build", lambda: sq.tree_build_device(ctx, L))
    t("pm", lambda: capi.check(capi.hip.shq_pm_run(ctx.h, C.byref(pmp))))
    t("walk", lambda: capi.check(capi.hip.shq_grav_short_run(ctx.h, C.byref(gp), None, 0, 1, MODE)))
    st = sq.WalkStats()
    capi.check(capi.hip.shq_grav_short_download(ctx.h, None, None, None, C.byref(st)))
    print("iter", it, "walk kernel %.2f ms, interactions/target %.1f, visits/wave %.1f" % (st.kernel_ms, st.ninteractions / n, st.nnodes_visited / (n / 64)), flush=True)
    t("oldacc", lambda: capi.check(capi.hip.shq_grav_refresh_oldacc(ctx.h, G)))
    if not os.environ.get("NOKICK"):
        t("kick_short", lambda: sq.kick_short(ctx, gk))
        t("kick_pm", lambda: sq.kick_pm(ctx, 1e-9))
print({k: [round(x, 2) for x in v] for k, v in acc.items()})
print({k: round(min(v), 2) for k, v in acc.items() if v}, "sum", round(sum(min(v) for v in acc.values() if v), 2))
