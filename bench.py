#!/usr/bin/env python3
"""bench.py — particle-steps/s of one TreePM force evaluation (PM + short-range tree walk).

Contract: python bench.py --gpus N --steps K --warmup W  prints ONE JSON line on rank 0.
A "step" is one pass of the hot path over the resident particle set: shq_pm_run (CIC deposit,
r2c, potential transfer, c2r, readout) + shq_grav_short_run (relative-criterion walk for every
particle) + OldAcc refresh.  Inputs are resident in HBM when the timed region starts; tree
build and host packing are outside it.

Workload at N=1: BASELINE.json configs[1] — dm-only 256^3, Nmesh 768, S-cluster positions,
Asmth 1.5, TreeRcut 6, softening 2.8 L/(30 n), exact window, ErrTolForceAcc 0.005, after a
theta=0.175 Barnes-Hut seeding walk.  For N>1 the box grows with the GPU count at fixed
particles per GPU (n = 320, 400, 512 per dimension for N = 2, 4, 8; Nmesh = 3 n) and is sharded
over x-slabs, one per rank (shenqi_amd/dist.py): RCCL all-to-all for the slab-FFT transposes,
ghost mesh planes and ghost particles for the tree => "scaling": "weak".
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# host-side OpenMP (tree build, CPU baseline) uses the cores this process may run on
os.environ.setdefault("OMP_NUM_THREADS", str(len(os.sched_getaffinity(0))))
# idle OpenMP workers sleep instead of spinning: after the CPU baseline they otherwise keep every core busy and slow
# the host thread that launches the GPU work (measured: resident step 95 ms instead of 75 ms)
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

G = 43.0071
RHO0 = 0.3 * 3 * 0.1 * 0.1 / (8 * np.pi * G)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VECTOR_PEAK_TF = 78.6


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--ngrid", dest="n", type=int, default=256, help="particles per dimension (default 256 = configs[1])")
    ap.add_argument("--errtol", type=float, default=0.005)
    ap.add_argument("--kind", default="cluster", choices=["cluster", "uniform", "grid"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sph", action="store_true", help="skip the SPH operator figures (kernels.sph_*)")
    ap.add_argument("--sph-ngrid", type=int, default=128,
                    help="gas particles per dimension of the SPH / C3 figures (128 = BASELINE configs[2]; 256 = the particle set of configs[4])")
    ap.add_argument("--walk-mode", type=int, default=0)
    ap.add_argument("--separate-calls", action="store_true",
                    help="time shq_pm_run + shq_grav_refresh_oldacc + shq_grav_short_run instead of the one-call shq_treepm_step")
    ap.add_argument("--workload", default="dm", choices=["dm", "c5"],
                    help="N > 1 only: dm = dm-only TreePM (the metric's configuration, default); c5 = BASELINE configs[4]-shaped step, "
                         "n^3 gas + n^3 dark matter: sharded TreePM over all + device-resident sharded SPH density and hydro of the gas")
    return ap.parse_args()


def cpu_baseline(pos, mass, tree, gp_rel, oldacc, L, acc_gpu=None):
    """Oracle (CPU port of the reference algorithm) on a bounded 1/64 sample of the workload:
    the tree walk of every 64th 64-target group on the SAME tree, plus a full PM step of a
    64^3-particle / 192^3-mesh sub-problem (1/64 of the particles and of the mesh)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc  # noqa: E402  (oracle = checker / reported baseline only)

    n = len(pos)
    groups = np.arange(0, n // 64, 64)
    targets = (groups[:, None] * 64 + np.arange(64)[None, :]).ravel().astype(np.int32)
    if len(targets) == 0:
        targets = np.arange(n, dtype=np.int32)
    t0 = time.perf_counter()
    oacc, _, _ = orc.grav_walk(tree.Nodes_base, tree.firstnode, pos, mass, oldacc, gp_rel, targets=targets)
    t_tree = time.perf_counter() - t0
    # the metric's second half: rms |dF| / |F| of the device's short-range forces against the CPU walk, same targets
    ferr = None
    if acc_gpu is not None:
        ref = oacc * G
        rel = np.linalg.norm(acc_gpu[targets] - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-300)
        ferr = {"rms_rel_tree_force_error_vs_cpu_walk": float(np.sqrt(np.mean(rel**2))), "max_rel": float(rel.max()),
                "targets": int(len(targets)), "tolerance_north_star": 1e-3}
    nsub = min(n, 64**3)
    sub = pos[:: max(1, n // nsub)][:nsub]
    # the PM leg: the oracle's deposit / transfer functions / readout around scipy's multi-threaded pocketfft (the oracle's own mixed-radix
    # FFT is a checker, not a baseline; the reference runs FFTW / heffte, absent here)
    fft_threads = orc.lib.orc_num_threads()
    fft_name = "scipy.fft (pocketfft, %d threads)" % fft_threads
    try:
        orc.use_scipy_fft(fft_threads)
    except Exception:
        fft_name = "the oracle's own mixed-radix FFT (scipy.fft unavailable)"
    t0 = time.perf_counter()
    orc.pm_force(sub, np.ones(len(sub), dtype=np.float32), 192, L, 1.5, G)
    t_pm = time.perf_counter() - t0
    orc.use_scipy_fft(0)
    frac_tree = len(targets) / n
    frac_pm = len(sub) / n
    est_full = t_tree / frac_tree + t_pm / frac_pm
    model, phys = cpu_info()
    return {
        "value": n / est_full, "unit": "particle-steps/s", "cores": orc.lib.orc_num_threads(), "kind": "port",
        "cpu_model": model, "physical_cores": phys,
        # the two legs separately; the tree leg is the like-for-like part
        "tree_value": n / (t_tree / frac_tree), "pm_value": n / (t_pm / frac_pm),
        "sample": "oracle tree walk of %d of %d targets (every 64th 64-target group, same tree) in %.2f s + oracle PM step "
                  "(reference structure, 5 FFTs through %s) on a %d-particle/192^3-mesh sub-problem in %.2f s; each scaled by its "
                  "fraction of the full job" % (len(targets), n, t_tree, fft_name, len(sub), t_pm),
        "tree_s_sample": t_tree, "pm_s_sample": t_pm, "force_error": ferr,
    }


def cpu_info():
    """CPU model string and number of physical cores of the host (/proc/cpuinfo), as SURVEY.md section 8(d) asks"""
    model, cores = None, set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model is None:
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    return model, (len(cores) or None)


def load_profile(name):
    """the newest committed rocprof summary profiles/rNN_<name> (tools/pmc_summary.py output), or None"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + name)))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def production_walk_key(prof):
    """the profile row of the production walk (potential on, no prefetch, leaf batch 2, no counters, relative criterion, primary
    walk; whatever launch form the round used: the row with the most dispatches)"""
    keys = [k for k in prof if k.startswith("grav_walk_exact_kernel<true, false, 2, 0, false, false")]
    return max(keys, key=lambda k: prof[k].get("dispatches", 0)) if keys else None


def hbm_bytes(e):
    return e.get("fetch_bytes_corrected", 0.0) + e.get("write_bytes", 0.0)


def distributed_walk_figures(ctx, sq, capi, tree, pos, oldacc, gp_rel, n):
    """The pieces shenqi's own multi-rank walk would call, timed on the bench's tree turned into the local tree
    of rank 0 of an 8-rank domain (512 top leaves, 7/8 of them pseudo nodes): export detection for all local targets and the
    secondary walk of that many imported queries.  Extra figures (kernels.dist_*), host-inclusive wall times."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import common as cm  # test helper that fabricates the domain flags (no oracle involved)
    nodes = tree.Nodes_base
    keep_flags, keep_suns = nodes["flags"].copy(), nodes["suns"].copy()
    tl = cm.make_domain(tree, ntask=8, me=0, depth=3)
    fn = tree.firstnode
    leaf_nodes = tl["treenode"][tl["Task"] == 0]
    # local targets: the particles under rank 0's top leaves
    local = np.zeros(n, dtype=bool)
    for no in leaf_nodes:
        nd = nodes[no - fn]
        local |= np.all(np.abs(pos - nd["center"]) <= 0.5 * nd["len"], axis=1)
    act = np.nonzero(local)[0].astype(np.int32)
    tv = tree.view()
    capi.check(capi.hip.shq_tree_upload(ctx.h, C.byref(tv)))
    sq.toptree_upload(ctx, tree, tl)
    ctx.synchronize()
    t0 = time.perf_counter()
    counts, table = sq.grav_toptree_exports(ctx, gp_rel, len(act), act)
    t_top_host = time.perf_counter() - t0
    # the resident form: target list and table stay in HBM, send counts per task come back, queries are packed on the device
    import torch
    d_act = torch.from_numpy(act).cuda()
    nexp, tc = C.c_int64(0), np.zeros(8, dtype=np.int64)
    d_q = torch.empty(max(1, len(table)) * capi.GRAV_QUERY_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    d_place = torch.empty(max(1, len(table)), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ctx.synchronize()
    t0 = time.perf_counter()
    capi.check(capi.hip.shq_grav_toptree_exports_resident(ctx.h, C.byref(gp_rel), d_act.data_ptr(), len(act), 1, 8, C.byref(nexp), capi.ptr(tc)))
    capi.check(capi.hip.shq_grav_export_pack(ctx.h, d_q.data_ptr(), d_place.data_ptr()))
    ctx.synchronize()
    t_top = time.perf_counter() - t0
    assert nexp.value == len(table)
    # the owners' side: restore the undivided tree (flags only) and walk the exported queries
    nodes["flags"][:] = keep_flags | (nodes["flags"] & 3)
    nodes["suns"][:] = keep_suns
    capi.check(capi.hip.shq_tree_upload(ctx.h, C.byref(tv)))
    q = np.zeros(len(table), dtype=capi.GRAV_QUERY_DTYPE)
    q["Pos"], q["OldAcc"], q["NodeList"] = pos[table["Index"]], oldacc[table["Index"]], table["NodeList"]
    res = np.zeros(len(q), dtype=capi.GRAV_RESULT_DTYPE)
    nint = np.zeros(len(q), dtype=np.int64)
    ctx.synchronize()
    t0 = time.perf_counter()
    capi.check(capi.hip.shq_grav_short_secondary(ctx.h, C.byref(gp_rel), capi.ptr(q), len(q), capi.ptr(res), capi.ptr(nint), 1))
    t_sec = time.perf_counter() - t0
    nodes["flags"][:] = keep_flags
    return {"dist_workload": "rank 0 of a fabricated 8-rank domain (512 top leaves) over the bench's tree", "dist_local_targets": int(len(act)),
            "dist_exports": int(len(table)), "dist_export_detection_ms": 1e3 * t_top,
            "dist_export_detection_note": "resident: count + scan + fill + per-task send counts + query packing on the device (wall time); "
                                          "with the list up and the counts and the table down over PCIe: dist_export_detection_host_table_ms",
            "dist_export_detection_host_table_ms": 1e3 * t_top_host, "dist_secondary_queries": int(len(q)),
            "dist_secondary_walk_ms": 1e3 * t_sec, "dist_secondary_interactions_per_query": float(nint.mean()) if len(q) else 0.0}


def sph_run(ctx, kind, n1, kernel):
    """density (first call: the whole Hsml loop from the initial guess; second call: steady state) and hydro of n1^3 gas"""
    import shenqi_amd as sq
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
    tree = sq.force_tree_rebuild_mask(pman, sq.GASMASK)
    sq.set_init_hsml(tree, L / n1, pman)
    tree = sq.force_tree_rebuild_mask(pman, sq.GASMASK)
    evp, st0 = sq.density(ctx, None, 1, 1, 0, None, tree, pman, SphP, BhP)   # converges Hsml from the initial guess
    evp, st = sq.density(ctx, None, 1, 1, 0, None, tree, pman, SphP, BhP)    # steady state: one iteration
    sq.force_tree_update_hmax(tree, pman)
    sq.set_hydropar(1, 100.0, 0.75)
    hs = sq.hydro_force(ctx, None, 0.1, 0.1, evp, None, tree, pman, SphP)
    hs = sq.hydro_force(ctx, None, 0.1, 0.1, evp, None, tree, pman, SphP)
    return pos, st0, st, hs


def sph_figures(ctx, n1=128, kernel=2):
    """SPH operators of BASELINE configs[2] (density with the Hsml loop, hydro force) on n1^3 uniformly
    placed gas particles, quintic kernel, pressure-entropy SPH: HIP-event time of the walk kernels through
    the one-shot C-ABI calls (tools/bench_sph.py is the stand-alone version).  Reported under kernels.sph_*;
    not part of `value`.  kernels.sph_cluster_* are the same operators on S-cluster gas (a density caustic: a few targets
    next to it have 10^5 neighbours; they are walked by a wave or a workgroup each)."""
    import shenqi_amd as sq
    n = n1**3
    pos, st0, st, hs = sph_run(ctx, "uniform", n1, kernel)
    out = c3_step(ctx, pos, n1, float(st.kernel_ms) / max(1, int(st.niterations)), float(hs.kernel_ms))
    out.update({"sph_workload": "%d^3 gas, uniform, quintic kernel, %.0f neighbours" % (n1, sq.GetNumNgb()),
            "sph_density_first_call_iterations": int(st0.niterations), "sph_density_first_call_ms": float(st0.kernel_ms),
            "sph_density_iteration_ms": float(st.kernel_ms) / max(1, int(st.niterations)),
            "sph_density_particles_per_s": n / (1e-3 * float(st.kernel_ms) / max(1, int(st.niterations))),
            "sph_hydro_ms": float(hs.kernel_ms), "sph_hydro_particles_per_s": n / (1e-3 * float(hs.kernel_ms))})
    try:
        _, c0, c1, ch = sph_run(ctx, "cluster", n1, kernel)
        out.update({"sph_cluster_density_first_call_iterations": int(c0.niterations), "sph_cluster_density_first_call_ms": float(c0.kernel_ms),
                    "sph_cluster_density_iteration_ms": float(c1.kernel_ms) / max(1, int(c1.niterations)),
                    "sph_cluster_density_candidates_per_target": float(c1.ninteractions) / max(1, int(c1.ntargets)) / max(1, int(c1.niterations)),
                    "sph_cluster_hydro_ms": float(ch.kernel_ms),
                    "sph_cluster_hydro_candidates_per_target": float(ch.ninteractions) / max(1, int(ch.ntargets))})
    except Exception as e:  # an extra figure: never fatal
        out["sph_cluster_note"] = "skipped: %s" % e
    # ---- rooflines of the two SPH operators: gather-bound, so algorithmic STREAM bytes against the HBM peak:
    # density per target and iteration = candidates x 32 B (pos, mass, type / flags) + NumNgb neighbours x 114 B (Vel,
    # FullTreeGravAccel, GravPM, HydroAccel, bins, Entropy, DtEntropy); hydro per target = NumNgb pairs x 200 B (the lower end of
    # 8(d)'s 113-300 pairs).  `traffic` = HBM bytes of the walk + evaluation kernels from the committed counter passes.
    ngb = float(sq.GetNumNgb())
    dens_iter_s = 1e-3 * float(st.kernel_ms) / max(1, int(st.niterations))
    cand = float(st.ninteractions) / max(1, int(st.ntargets)) / max(1, int(st.niterations))
    dens_bytes = n * (cand * 32.0 + ngb * 114.0)
    hyd_bytes = n * ngb * 200.0
    pj = load_profile("sph128_pmc_hbm.json") or {}

    def traffic_of(prefixes):
        ks = [k for k in pj if any(k.startswith(p) for p in prefixes)]
        return sum(hbm_bytes(pj[k]) for k in ks) if ks else None
    # These two operators are gather / VALU bound: the stream of 8(d) is served from L2 and LDS, not from HBM.  `achieved` / `frac`
    # keep 8(d)'s stream figure over the HBM peak (the contract's definition); `hbm_achieved` / `hbm_frac` are the HBM bytes the
    # counters saw (committed FETCH_SIZE / WRITE_SIZE passes of the same kernels) over the same time: the real use of the memory.
    note = ("gather / VALU bound, not HBM bound: `achieved` is SURVEY 8(d)'s neighbour stream (served from L2 / LDS) over the HBM peak; "
            "`hbm_achieved` = counter traffic / kernel time is what actually crosses the HBM interface")
    dtr, htr = traffic_of(("sph_density_kernel",)), traffic_of(("sph_hydro_kernel",))
    hyd_s = 1e-3 * float(hs.kernel_ms)
    out["roofline_sph_density"] = {"bound": "gather-valu (stream from L2/LDS)", "kernel": "sph_density_kernel, walk + evaluation launches of one Hsml iteration",
                                   "achieved": dens_bytes / dens_iter_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": dens_bytes / dens_iter_s / 1e9 / HBM_PEAK_GBS, "traffic": dtr,
                                   "hbm_achieved": None if dtr is None else dtr / dens_iter_s / 1e9,
                                   "hbm_frac": None if dtr is None else dtr / dens_iter_s / 1e9 / HBM_PEAK_GBS,
                                   "algorithmic_bytes": dens_bytes, "candidates_per_target": cand, "neighbours": ngb, "note": note}
    out["roofline_sph_hydro"] = {"bound": "gather-valu (stream from L2/LDS)", "kernel": "sph_hydro_kernel, walk + evaluation launches",
                                 "achieved": hyd_bytes / hyd_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": hyd_bytes / hyd_s / 1e9 / HBM_PEAK_GBS, "traffic": htr,
                                 "hbm_achieved": None if htr is None else htr / hyd_s / 1e9,
                                 "hbm_frac": None if htr is None else htr / hyd_s / 1e9 / HBM_PEAK_GBS,
                                 "algorithmic_bytes": hyd_bytes,
                                 "candidates_per_target": float(hs.ninteractions) / max(1, int(hs.ntargets)), "neighbours": ngb, "note": note}
    return out


def c3_step(ctx, gas_pos, n1, density_ms, hydro_ms):
    """BASELINE configs[2] as one force step: 2 x n1^3 gas + dark matter (uniform, as the z = 99 initial conditions of
    examples/small nearly are), Nmesh 3 n1: PM + short-range walk over all particles, plus the density and hydro times
    of the gas measured above.  Extra figure (kernels.c3_*), not part of `value`."""
    import shenqi_amd as sq
    from shenqi_amd import capi
    n = len(gas_pos)
    L = 1.0
    dm = sq.synth_positions("uniform", n, seed=77, L=L)
    pos = np.concatenate([gas_pos, dm])
    order = sq.hilbert_order(pos, L)
    pos = pos[order]
    pman = sq.PartManager(2 * n, L)
    P = pman.Base
    P["Pos"] = pos
    P["Type"] = np.where(order < n, 0, 1).astype(np.uint8)
    P["Mass"] = np.where(order < n, 0.16, 0.84)       # Omega_b / Omega_m of a gas + dark matter pair
    nmesh = 3 * n1
    sq.set_gravshort_treepar(ErrTolForceAcc=0.005, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=1, Rcut=6.0)
    sq.gravshort_set_softenings(L / n1)
    gp_bh = sq.make_grav_params(L, 1.5, nmesh, G, RHO0)
    sq.set_gravshort_treepar(ErrTolForceAcc=0.005, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=0, Rcut=6.0)
    gp_rel = sq.make_grav_params(L, 1.5, nmesh, G, RHO0)
    pmp = sq.PMParams(nmesh, 0, L, 1.5, G)
    pv = pman.view()
    capi.check(capi.hip.shq_set_walk_stats(ctx.h, 1))   # keeps these walks out of the production walk's row of a profile
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    try:
        sq.tree_build_device(ctx, L)
    except sq.ShqError:
        tree = sq.force_tree_full(pman)
        tv = tree.view()
        capi.check(capi.hip.shq_tree_upload(ctx.h, C.byref(tv)))
    for k, gp in enumerate((gp_bh, gp_rel, gp_rel)):
        capi.check(capi.hip.shq_pm_run(ctx.h, C.byref(pmp)))
        capi.check(capi.hip.shq_grav_short_run(ctx.h, C.byref(gp), None, 0, 1, 0))
        capi.check(capi.hip.shq_grav_refresh_oldacc(ctx.h, G))
    st = sq.WalkStats()
    capi.check(capi.hip.shq_grav_short_download(ctx.h, None, None, None, C.byref(st)))
    ph = (C.c_double * 6)()
    capi.check(capi.hip.shq_pm_phase_ms(ctx.h, C.byref(ph)))
    total = float(st.kernel_ms) + float(ph[5]) + density_ms + hydro_ms
    return {"c3_workload": "2 x %d^3 gas + dark matter, uniform, Nmesh %d: PM + walk over all, density + hydro of the gas" % (n1, nmesh),
            "c3_tree_walk_ms": float(st.kernel_ms), "c3_tree_interactions_per_target": st.ninteractions / max(1, st.ntargets),
            "c3_pm_ms": float(ph[5]), "c3_density_iteration_ms": density_ms, "c3_hydro_ms": hydro_ms, "c3_force_step_ms": total,
            "c3_particle_steps_per_s": 2 * n / (1e-3 * total)}


def run_sharded_c5(args, rank, world, backend, dev, cdev, json_fd, sq, capi, sd, dist, torch):
    """BASELINE configs[4]-shaped step on N ranks (examples/hydro: 2 x n^3, gas + dark matter): sharded TreePM over all particles +
    sharded SPH density (Hsml loop) and hydro force of the gas with the gas resident on the device (dist.DistSPHDevice).  Positions as
    SURVEY 8(d) prescribes for gas configurations: dark matter uniform (the z = 99 initial conditions nearly are), gas = dark matter
    shifted by (L / 2n)(1, 1, 1), m_gas / m_dm = 0.0464 / 0.2350, Vel ~ N(0, 0.01 L), Entropy 1, Hsml 1.5 L / n, quintic kernel.
    Sub-grid physics (cooling, star formation) is out of scope; time bins all zero (every particle active)."""
    n1 = int(round(args.n * world ** (1.0 / 3.0) / (2 * world))) * 2 * world if args.n != 256 else {2: 320, 4: 400, 8: 512}.get(world, 256)
    n1 = max(n1, 2 * world)
    nmesh = 3 * n1
    L = 1.0
    ndm = n1**3
    mine = ndm // world + (1 if rank < ndm % world else 0)
    first = rank * (ndm // world) + min(rank, ndm % world)
    comm = sd.Comm()
    t0 = time.perf_counter()
    dm = sq.synth_positions_range("uniform", ndm, first, mine, seed=20240601, L=L)
    gas = np.mod(dm + 0.5 * L / n1, L)
    mg, md = 0.0464 / 0.2814, 0.2350 / 0.2814
    posm = np.concatenate([np.concatenate([gas, np.full((mine, 1), mg)], axis=1), np.concatenate([dm, np.full((mine, 1), md)], axis=1)], axis=0)
    rows = np.zeros((mine, capi.GAS_NCOL))
    rows[:, 0:3], rows[:, 3] = gas, mg
    rows[:, 4:7] = np.random.default_rng(100 + rank).normal(size=(mine, 3)) * 0.01 * L
    rows[:, 7], rows[:, 17], rows[:, 20] = 1.5 * L / n1, 1.0, 1.0
    posm = torch.from_numpy(posm).to(dev)
    rows = torch.from_numpy(rows).to(dev)
    bounds = sd.balanced_bounds(comm, nmesh, L, posm[:, 0])
    work_stream = torch.cuda.Stream(device=dev)
    work_stream.wait_stream(torch.cuda.current_stream(dev))
    torch.cuda.set_stream(work_stream)
    ctx = sq.Context(dev.index, stream=work_stream.cuda_stream)
    ctx_sph = sq.Context(dev.index, stream=work_stream.cuda_stream)   # the gas set and its tree live beside the gravity set
    sq.set_gravshort_treepar(ErrTolForceAcc=args.errtol, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=1, Rcut=6.0)
    sq.gravshort_set_softenings(L / n1)
    gp_bh = sq.make_grav_params(L, 1.5, nmesh, G, RHO0)
    sq.set_gravshort_treepar(ErrTolForceAcc=args.errtol, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=0, Rcut=6.0)
    gp_rel = sq.make_grav_params(L, 1.5, nmesh, G, RHO0)
    drv = sd.DistTreePM(comm, ctx, nmesh, L, 1.5, G, dev, halo_factor=1.5, bounds=bounds)
    local = sd.exchange_to_owner(comm, drv.decomp, posm)
    rows = sd.exchange_to_owner(comm, drv.decomp, rows).contiguous()
    del posm
    drv.setup(local, gp_rel.Rcut)
    drv.step(gp_bh)
    sph = sd.DistSPHDevice(comm, drv.decomp, ctx_sph, L)
    dp = sq.make_density_params(L, kernel=2, MinGasHsml=1e-6)
    hp = sq.make_hydro_params(L, kernel=2)
    rounds0 = sph.density(rows, dp)            # converges Hsml from the initial guess; the timed steps start from the converged values
    sph.hydro(rows, hp)
    t_setup = time.perf_counter() - t0

    def step():
        drv.step(gp_rel)
        sph.density(rows, dp)
        sph.hydro(rows, hp)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.synchronize()
    ctx_sph.synchronize()
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = time.perf_counter() - t0
    tt = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())
    # per-operator figures of one more, untimed step (walk as a single launch)
    t1 = time.perf_counter()
    drv.step(gp_rel, overlap=False)
    ctx.synchronize()
    t_grav = time.perf_counter() - t1
    st = sq.WalkStats()
    capi.check(capi.hip.shq_grav_short_download(ctx.h, None, None, None, C.byref(st)))
    t1 = time.perf_counter()
    sph.density(rows, dp)
    ctx_sph.synchronize()
    t_dens = time.perf_counter() - t1
    dstat = dict(sph.stats["density"])
    t1 = time.perf_counter()
    sph.hydro(rows, hp)
    ctx_sph.synchronize()
    t_hyd = time.perf_counter() - t1
    hstat = dict(sph.stats["hydro"])
    ngas = int(rows.shape[0])
    ngb = float(dp.DesNumNgb)
    dens_s = 1e-3 * dstat["kernel_ms"] / max(1, dstat["iterations"])
    cand = dstat["ninteractions"] / max(1, ngas) / max(1, dstat["iterations"])
    dens_bytes = ngas * (cand * 32.0 + ngb * 114.0)
    hyd_bytes = ngas * ngb * 200.0
    hyd_s = 1e-3 * hstat["kernel_ms"]
    walk_s = max(st.kernel_ms * 1e-3, 1e-12)
    nglobal = 2 * ndm
    note = "gather / VALU bound: SURVEY 8(d)'s neighbour stream (served from L2 / LDS) over the HBM peak; rank 0"
    out = {
        "metric": "particle-steps/sec (grav+PM+SPH) at 256^3; rms force error vs ref",
        "metric_note": "C5-shaped extra mode (--workload c5): n^3 gas + n^3 dark matter, TreePM over all + SPH density and hydro of the gas",
        "value": nglobal * args.steps / elapsed, "unit": "particle-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "2 x %d^3 gas + dark matter (uniform + shifted copy), Nmesh %d, sharded over %d x-slabs: TreePM over all, "
                               "density (quintic kernel, Hsml loop) + hydro of the gas resident on the device" % (n1, nmesh, world),
                   "particles_total": nglobal, "nmesh": nmesh, "parallelism": "x-slabs x%d, RCCL all-to-all + ghost particles + ghost gas rows" % world},
        "roofline": {"bound": "valu-f64", "kernel": "grav_walk_exact_kernel (rank 0)", "achieved": 45.0 * st.ninteractions / walk_s / 1e12,
                     "peak": FP64_VECTOR_PEAK_TF, "unit": "TFLOP/s", "frac": 45.0 * st.ninteractions / walk_s / 1e12 / FP64_VECTOR_PEAK_TF,
                     "traffic": None, "algorithmic_flops": 45.0 * st.ninteractions},
        "roofline_sph_density": {"bound": "gather-valu (stream from L2/LDS)", "kernel": "sph_density_kernel, one Hsml iteration (rank 0)",
                                 "achieved": dens_bytes / max(dens_s, 1e-12) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": dens_bytes / max(dens_s, 1e-12) / 1e9 / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes": dens_bytes,
                                 "note": note},
        "roofline_sph_hydro": {"bound": "gather-valu (stream from L2/LDS)", "kernel": "sph_hydro_kernel (rank 0)",
                               "achieved": hyd_bytes / max(hyd_s, 1e-12) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": hyd_bytes / max(hyd_s, 1e-12) / 1e9 / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes": hyd_bytes,
                               "note": note},
        "kernels": {"treepm_step_ms": 1e3 * t_grav, "tree_walk_ms": float(st.kernel_ms), "tree_interactions_per_target": st.ninteractions / max(1, st.ntargets),
                    "density_call_ms": 1e3 * t_dens, "density_kernels_ms": dstat["kernel_ms"], "density_iterations": dstat["iterations"],
                    "density_first_call_import_rounds": rounds0, "hydro_call_ms": 1e3 * t_hyd, "hydro_kernels_ms": hstat["kernel_ms"],
                    "gas_local": ngas, "gas_ghosts_density": dstat["nghost"], "gas_ghosts_hydro": hstat["nghost"],
                    "grav_local": int(drv.nloc), "grav_ghosts": int(drv.nghost)},
        "setup_s": {"generate_exchange_tree_first_density": t_setup},
    }
    ctx_sph.close()
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()
    sys.stdout.flush()
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)


def run_sharded(args, rank, local_rank, world):
    """N > 1: one rank per GPU over RCCL; x-slab sharded TreePM (shenqi_amd/dist.py)."""
    # torchrun exports OMP_NUM_THREADS=1 when the variable is unset; the host side (Hilbert order, tree hand-over) is threaded:
    # give every rank its share of the node's cores (must happen before anything loads libgomp)
    if os.environ.get("OMP_NUM_THREADS", "1") == "1":
        os.environ["OMP_NUM_THREADS"] = str(max(1, len(os.sched_getaffinity(0)) // max(1, world)))
    import torch
    import torch.distributed as dist
    if args.gpus not in (1, world):      # --gpus keeps its default under a bare torchrun; anything else must agree
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d" % (args.gpus, world))

    # SHQ_BENCH_BACKEND=gloo is a rehearsal knob for the one-GPU box: every rank shares device 0 and the
    # exchanges are staged through the host.  The driver's runs use nccl (= RCCL), one GPU per rank.
    backend = os.environ.get("SHQ_BENCH_BACKEND", "nccl")
    # RCCL prints a version banner on stdout when its communicator comes up: everything but the one JSON line goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    devidx = local_rank if backend == "nccl" else 0
    torch.cuda.set_device(devidx)
    dev = torch.device("cuda", devidx)
    cdev = dev if backend == "nccl" else torch.device("cpu")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    import shenqi_amd as sq
    from shenqi_amd import capi, dist as sd

    if args.workload == "c5":
        return run_sharded_c5(args, rank, world, backend, dev, cdev, json_fd, sq, capi, sd, dist, torch)
    # 256^3 particles per GPU (weak scaling); meshes 3 n1 = 960, 1200, 1536 all have a bespoke FFT
    n1 = {2: 320, 4: 400, 8: 512}.get(world)
    if n1 is None:
        n1 = int(round(args.n * world ** (1.0 / 3.0) / (2 * world))) * 2 * world
    if args.n != 256:      # reduced rehearsal sizes keep the same per-GPU share
        n1 = int(round(args.n * world ** (1.0 / 3.0) / (2 * world))) * 2 * world
    # Nmesh: the reference's rule 3 * 2^floor(log2(N_dm) / 3) (run.cpp:225-226) = 3 n1, as at N = 1 - except for the named
    # 8-GPU configuration, BASELINE configs[3] = benchmarks/dm-50-512, whose parameter file fixes Nmesh 1024
    # (paramfile.gadget:8): the same choice the 1-GPU branch makes for --ngrid 512
    nmesh = 1024 if n1 == 512 else 3 * n1
    L = 1.0
    nglobal = n1**3
    nmine = nglobal // world + (1 if rank < nglobal % world else 0)
    first = rank * (nglobal // world) + min(rank, nglobal % world)
    comm = sd.Comm()
    t0 = time.perf_counter()
    # ONE global particle set (one S-cluster in the box, whatever the rank count): this rank generates its index range of it
    pos = sq.synth_positions_range(args.kind, nglobal, first, nmine, seed=20240601, L=L)
    posm = torch.from_numpy(np.concatenate([pos, np.ones((nmine, 1))], axis=1)).to(dev)
    del pos
    bounds = sd.balanced_bounds(comm, nmesh, L, posm[:, 0])
    # one stream for torch and the library (a real one: the null stream cannot be handed over), so that a step needs no host
    # synchronisation between torch operators, RCCL rounds and library kernels
    work_stream = torch.cuda.Stream(device=dev)
    work_stream.wait_stream(torch.cuda.current_stream(dev))   # posm and the bounds were produced on the default stream
    torch.cuda.set_stream(work_stream)
    ctx = sq.Context(devidx, stream=work_stream.cuda_stream)
    sq.set_gravshort_treepar(ErrTolForceAcc=args.errtol, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=1, Rcut=6.0)
    sq.gravshort_set_softenings(L / n1)
    gp_bh = sq.make_grav_params(L, 1.5, nmesh, G, RHO0)
    sq.set_gravshort_treepar(ErrTolForceAcc=args.errtol, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=0, Rcut=6.0)
    gp_rel = sq.make_grav_params(L, 1.5, nmesh, G, RHO0)
    drv = sd.DistTreePM(comm, ctx, nmesh, L, 1.5, G, dev, halo_factor=1.5, bounds=bounds)
    local = sd.exchange_to_owner(comm, drv.decomp, posm)
    del posm
    drv.setup(local, gp_rel.Rcut)
    drv.step(gp_bh)
    bounds_count, ycuts = bounds, None
    if os.environ.get("SHQ_BENCH_REBALANCE", "1") == "1":
        # like the reference's domain decomposition (domain.cpp:620-700), balance the work counted in the previous force
        # evaluation rather than the particle count: one relative-criterion step, new boundaries, particles to their new owners
        drv.step(gp_rel)
        # SHQ_BENCH_SUBPLANE=0: boundaries on whole planes; otherwise the plane a rank's share ends in is shared by y
        # (the shared plane needs the slab entry points of the bespoke transforms: shq_pm_slab2_deposit_ghosts)
        if os.environ.get("SHQ_BENCH_SUBPLANE", "1") == "1" and world > 1 and int(capi.hip.shq_pm_slab_pitch(nmesh)) > 0:
            bounds, ycuts = sd.cost_balanced_bounds(comm, drv, subplane=True)
        else:
            bounds = sd.cost_balanced_bounds(comm, drv)
        if bounds != bounds_count or ycuts is not None:
            local = drv.local
            drv = sd.DistTreePM(comm, ctx, nmesh, L, 1.5, G, dev, halo_factor=1.5, bounds=bounds, ycuts=ycuts)
            local = sd.exchange_to_owner(comm, drv.decomp, local)
            drv.setup(local, gp_rel.Rcut)
            drv.step(gp_bh)
    t_setup = time.perf_counter() - t0
    for _ in range(args.warmup):
        drv.step(gp_rel)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        drv.step(gp_rel)
    ctx.synchronize()
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = time.perf_counter() - t0
    tt = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())
    # the timed steps cut the walk in two pieces around the mesh transposes; the per-rank walk figures below come from one
    # more, untimed step with the walk as a single launch
    drv.step(gp_rel, overlap=False)
    ctx.synchronize()
    st = sq.WalkStats()
    capi.check(capi.hip.shq_grav_short_download(ctx.h, None, None, None, C.byref(st)))
    loads = torch.tensor([float(drv.nloc), float(drv.nghost), st.kernel_ms], device=cdev, dtype=torch.float64)
    gathered = [torch.zeros_like(loads) for _ in range(world)]
    dist.all_gather(gathered, loads)
    walk_s = max(float(g[2]) for g in gathered) * 1e-3
    tree_bytes = 68.0 * drv.nloc + 76.0 * drv.tree.numnodes
    ach = tree_bytes / max(st.kernel_ms * 1e-3, 1e-12) / 1e9
    out = {
        "metric": "particle-steps/sec (grav+PM+SPH) at 256^3; rms force error vs ref",
        "metric_note": "BASELINE.json's metric name; this N-GPU line runs the dm-only TreePM step of config.workload (dm-only as "
                       "BASELINE configs[1] is; the sharded SPH operators of dist.py, DistSPH, are covered by tests, not timed here)",
        "value": nglobal * args.steps / elapsed, "unit": "particle-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "dm-only %d^3 TreePM sharded over %d x-slabs (S-%s, Nmesh %d, Asmth 1.5, Rcut 6, ErrTolForceAcc %g, "
                               "exact window), %.3g particles per GPU" % (n1, world, args.kind, nmesh, args.errtol, nglobal / world),
                   "particles_total": nglobal, "nmesh": nmesh, "parallelism": "x-slabs x%d, RCCL all-to-all + ghost exchange" % world,
                   "slab_bounds": bounds, "slab_ycuts": ycuts,
                   "slab_bounds_by_count": bounds_count, "walk": "exact (per-target reference opening decisions)"},
        "roofline": {"bound": "valu-f64", "kernel": "grav_walk_exact_kernel (rank 0)",
                     "achieved": 45.0 * st.ninteractions / max(st.kernel_ms * 1e-3, 1e-12) / 1e12, "peak": FP64_VECTOR_PEAK_TF,
                     "unit": "TFLOP/s", "frac": 45.0 * st.ninteractions / max(st.kernel_ms * 1e-3, 1e-12) / 1e12 / FP64_VECTOR_PEAK_TF,
                     "traffic": None, "algorithmic_flops": 45.0 * st.ninteractions, "algorithmic_bytes": tree_bytes,
                     "hbm_algorithmic_GBs": ach,
                     "note": "FP64-issue bound walk: 45 flop per interaction against the 78.6 TFLOP/s FP64 peak (see the N=1 line)"},
        "kernels": {"per_rank_local_particles": [int(g[0]) for g in gathered], "per_rank_ghost_particles": [int(g[1]) for g in gathered],
                    "per_rank_walk_ms": [float(g[2]) for g in gathered], "slowest_walk_ms": walk_s * 1e3},
        "setup_s": {"generate_exchange_tree_upload": t_setup},
    }
    # sampled force check (the metric's second half on the N > 1 line): total acceleration (tree + PM) of the sharded step at up
    # to 64 of rank 0's particles against direct summation over ALL particles and their 27 periodic images, every rank summing
    # its own particles (shq_direct_force_sample, the reference's own check: tests/test_gravity.cpp:41-119, errors in units of the
    # mean |a_direct|; its limits there: mean < 0.8 ErrTol... for PM + tree against this image sum)
    try:
        ns = 64
        acc_l, _, gpm_l, _ = drv.download()
        nl = torch.tensor([min(ns, drv.nloc) if rank == 0 else 0], dtype=torch.int64, device=cdev)
        dist.all_reduce(nl, op=dist.ReduceOp.MAX)
        ns = int(nl.item())
        samp = torch.zeros((ns, 3), dtype=torch.float64, device=cdev)
        if rank == 0:
            pick = np.linspace(0, drv.nloc - 1, ns).astype(np.int64)
            samp = drv.local[pick, :3].to(cdev).contiguous()
        dist.broadcast(samp, 0)
        sp = np.ascontiguousarray(samp.cpu().numpy())
        part = np.zeros((ns, 3))
        capi.check(capi.hip.shq_direct_force_sample(ctx.h, capi.ptr(sp), ns, drv.nloc, L, G, float(gp_rel.ForceSoftening), 1, capi.ptr(part)))
        tot = torch.from_numpy(part).to(cdev)
        dist.all_reduce(tot)
        if rank == 0:
            direct = tot.cpu().numpy()
            total = acc_l[pick] + gpm_l[pick]
            meanacc = np.abs(direct).mean()
            err = np.abs(direct - total) / meanacc
            out["force_error"] = {"check": "tree + PM of the sharded step vs direct summation over all %d particles and 27 images at %d targets "
                                           "of rank 0 (check_accns of tests/test_gravity.cpp:78-119: |a_direct - a| / mean |a_direct|)" % (nglobal, ns),
                                  "mean": float(err.mean()), "max": float(err.max()), "targets": ns}
    except Exception as e:  # an extra figure: never fatal
        out["force_error"] = {"note": "sampled force check skipped: %s" % e}
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()
    sys.stdout.flush()
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks the way the driver does (torch.distributed.run, one rank
    per GPU over RCCL) as a child process, before this process touches torch or the GPU, and pass its JSON line through."""
    import subprocess
    # --standalone: torchrun's own single-node rendezvous on a port it binds itself (no pick-then-release race), on the loopback
    # address (the container hostname may not resolve)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node",
           str(args.gpus), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and os.environ.get("SHQ_COMM_FORCE", "0") != "1":
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 or os.environ.get("SHQ_COMM_FORCE", "0") == "1":   # the latter: one-rank RCCL rehearsal (needs MASTER_ADDR/PORT)
        return run_sharded(args, rank, local_rank, world)
    dist = None
    sys.stdout.flush()   # stdout carries the one JSON line only; anything a library prints goes to stderr
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import shenqi_amd as sq
    from shenqi_amd import capi

    n1 = args.n
    n = n1**3
    L = 1.0
    nmesh = 3 * n1 if n1 != 512 else 1024
    t_setup = time.perf_counter()
    pos = sq.synth_positions(args.kind, n, seed=20240601 + rank, L=L)
    pos = pos[sq.hilbert_order(pos, L)]  # Peano-Hilbert order, as the reference keeps its particles (domain.cpp:268)
    pman = sq.PartManager(n, L)
    P = pman.Base
    P["Pos"] = pos
    P["Type"] = 1
    P["Mass"] = 1.0
    t_setup = time.perf_counter() - t_setup

    ctx = sq.Context(local_rank if world > 1 else 0)
    sq.set_gravshort_treepar(ErrTolForceAcc=args.errtol, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=1, Rcut=6.0)
    sq.gravshort_set_softenings(L / n1)
    gp_bh = sq.make_grav_params(L, 1.5, nmesh, G, RHO0)
    sq.set_gravshort_treepar(ErrTolForceAcc=args.errtol, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=0, Rcut=6.0)
    gp_rel = sq.make_grav_params(L, 1.5, nmesh, G, RHO0)
    pmp = sq.PMParams(nmesh, 0, L, 1.5, G)

    pv = pman.view()
    t0 = time.perf_counter()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    t_upload = time.perf_counter() - t0
    # the tree is built on the device (shq_tree_build: bit-identical to the host / reference tree, see
    # tests/test_gpu_treebuild.py); outside the timed region like the host build it replaces
    try:
        sq.tree_build_device(ctx, L)         # first call allocates the scratch
        t0 = time.perf_counter()
        tb = sq.tree_build_device(ctx, L)
        t_tree_build = time.perf_counter() - t0
        numnodes = int(tb.numnodes)
    except sq.ShqError as e:                 # deeper than 21 levels: host build (8 OpenMP threads' worth) + upload
        print("device tree build refused (%s): falling back to the host build" % e, file=sys.stderr)
        t0 = time.perf_counter()
        host_tree = sq.force_tree_full(pman)
        tv = host_tree.view()
        capi.check(capi.hip.shq_tree_upload(ctx.h, C.byref(tv)))
        t_tree_build = time.perf_counter() - t0
        numnodes = int(host_tree.numnodes)
        tb = capi.TreeBuildStats(n, numnodes, 0, 0.0)

    def step(gp, one_call=None):
        """gravpm_force, then grav_short_tree for every particle with OldAcc = |FullTreeGravAccel of the last step + the new
        GravPM| / G (the reference's order on a PM step, run.cpp:518-563): as the library's one call (readout and OldAcc refresh
        ride in the walk's prologue) or, with --separate-calls, as its three separate ones.  Same results bit for bit
        (tests/test_gpu_gravity.py::test_treepm_step_equals_the_three_separate_calls)."""
        if one_call is None:
            one_call = not args.separate_calls and args.walk_mode == 0
        if one_call:
            capi.check(capi.hip.shq_treepm_step(ctx.h, C.byref(pmp), C.byref(gp), 1, sq.WALK_EXACT))
        else:
            capi.check(capi.hip.shq_pm_run(ctx.h, C.byref(pmp)))
            capi.check(capi.hip.shq_grav_refresh_oldacc(ctx.h, G))
            capi.check(capi.hip.shq_grav_short_run(ctx.h, C.byref(gp), None, 0, 1, args.walk_mode))

    # the timed steps run the production walk (no diagnostic counters); the wave-level figures under kernels.tree_* come from
    # one more, untimed step with the counters on (a separate kernel instantiation: its own row in a profile)
    capi.check(capi.hip.shq_set_walk_stats(ctx.h, 0))
    # seeding: PM + Barnes-Hut walk (theta = 0.175) so that the timed walks use the relative criterion
    t0 = time.perf_counter()
    step(gp_bh)
    ctx.synchronize()
    t_seed = time.perf_counter() - t0
    for _ in range(args.warmup):
        step(gp_rel)
    ctx.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()

    walk_ms, pm_ms = [], []
    barrier()
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.timer_begin(0)
    for _ in range(args.steps):
        step(gp_rel)
    ctx.timer_end(0)
    ctx.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    ev_ms = ctx.timer_ms(0)
    if dist is not None:
        import torch
        tt = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # per-kernel durations of the last step (HIP events recorded on the launch stream)
    st = sq.WalkStats()
    capi.check(capi.hip.shq_grav_short_download(ctx.h, None, None, None, C.byref(st)))
    ph = (C.c_double * 6)()
    capi.check(capi.hip.shq_pm_phase_ms(ctx.h, C.byref(ph)))
    ph = list(ph)
    fused = C.c_int(0)
    capi.check(capi.hip.shq_treepm_last_fused(ctx.h, C.byref(fused)))
    one_call_route = not args.separate_calls and args.walk_mode == 0
    step_route = ("shq_treepm_step: PM readout + OldAcc refresh in the walk's task prologue" if fused.value and one_call_route
                  else "shq_treepm_step: the PM's readout kernel forms OldAcc, then the walk" if one_call_route
                  else "shq_pm_run + shq_grav_refresh_oldacc + shq_grav_short_run")
    # the pair kernel's sticky status after the timed loop (no download inside it): launches the mop-up pass had to finish, the
    # deepest pair stack; an overflow in any of the timed steps comes back here as an error
    pair_rec, pair_high, pair_mop = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    capi.check(capi.hip.shq_walk_pair_status(ctx.h, C.byref(pair_rec), C.byref(pair_high), C.byref(pair_mop)))
    capi.check(capi.hip.shq_set_walk_stats(ctx.h, 1))
    step(gp_rel, one_call=False)
    capi.check(capi.hip.shq_grav_refresh_oldacc(ctx.h, G))   # the OldAcc of the NEXT step, for the checks below
    stw = sq.WalkStats()
    capi.check(capi.hip.shq_grav_short_download(ctx.h, None, None, None, C.byref(stw)))
    for k in ("nnodes_visited", "nwave_interactions", "nwave_node_interactions", "nnode_interactions"):
        setattr(st, k, getattr(stw, k))
    counted_walk_ms = float(stw.kernel_ms)

    ncells = float(nmesh) ** 3
    # ---- algorithmic bytes ------------------------------------------
    # tree walk: compulsory traffic only (targets in/out + node pool once); the walk is not HBM-bound
    tree_bytes = 68.0 * n + 76.0 * numnodes
    # PM as THIS pipeline runs it: zero (8 C write) + deposit (32 N read, 8 cells x 8 B RMW) +
    # 5 fused FFT passes (read + write of the padded mesh each) + readout (32 N read, 56 cells x 8 B
    # gathered, 32 N written)
    zp = 2 * (((nmesh // 2 + 1) + 3) // 4 * 4)
    mesh_bytes = 8.0 * nmesh * nmesh * zp
    fft_bytes = 5 * 2 * mesh_bytes
    deposit_bytes = (32.0 + 128.0) * n
    readout_bytes = (32.0 + 448.0 + 32.0) * n
    oldacc_bytes = (24.0 + 8.0) * n                       # the readout kernel also reads FullTreeGravAccel and writes OldAcc
    pm_bytes = mesh_bytes + deposit_bytes + fft_bytes + readout_bytes
    # PM with the reference's structure (1 r2c + 4 c2r, separate transfer sweeps), for comparison only
    pm_bytes_reference_structure = 468.0 * n + 240.0 * ncells
    walk_s = st.kernel_ms * 1e-3
    pm_s = ph[5] * 1e-3
    fft_s = (ph[1] + ph[2] + ph[3]) * 1e-3
    dominant = "grav_walk_exact_kernel" if walk_s >= pm_s else "pm (deposit + 5 fused FFT passes + readout)"
    ach = (tree_bytes / walk_s if walk_s >= pm_s else pm_bytes / pm_s) / 1e9
    # HBM traffic per launch from the committed PMC summary of this same command (profiles/, made with
    # tools/pmc_summary.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes), if present: a counter pass cannot
    # run inside this process, so these two numbers are the committed measurement of the same kernels on the same workload
    traffic = traffic_fft = traffic_pair = None
    valu_busy = None
    pj = load_profile("bench256_pmc_hbm.json")
    if pj and n1 == 256 and world == 1:
        # the production walk: potential on, no prefetch, leaf batch 2, no counters, relative criterion, primary walk
        key = production_walk_key(pj)
        if key:
            traffic = hbm_bytes(pj[key])
        fk = [k for k in pj if k.startswith("fft_t_")] or [k for k in pj if k.startswith("fft_pass_")]   # transposing / in-place pipeline
        if len(fk) == 5:
            traffic_fft = sum(hbm_bytes(pj[k]) for k in fk)
        pk = [k for k in pj if k.startswith("grav_pair_kernel_live<true>")]
        traffic_pair = hbm_bytes(pj[pk[0]]) if pk else None
    sj = load_profile("bench256_pmc_sq.json")
    if sj:
        key = production_walk_key(sj)
        if key and sj[key].get("SQ_BUSY_CYCLES"):
            e = sj[key]
            # SQ_INSTS_VALU x 4 cycles per wave instruction over the SIMD cycles the kernel was resident (SQ_BUSY_CYCLES counts per
            # shader engine: / 32 engines x 1024 SIMDs)
            valu_busy = e.get("SQ_INSTS_VALU", 0.0) * 4.0 / (e["SQ_BUSY_CYCLES"] / 32.0 * 1024.0)
    # The dominant kernel, the tree walk, is bound by FP64 issue, not by HBM (0.1 kB of compulsory traffic per
    # target against ~500 interactions of ~45 flop): its roofline is the FP64 vector peak (78.6
    # TFLOP/s on MI355X; no MFMA instruction is used or usable: bound "valu-f64").  The HBM view of the same launch
    # stays alongside: algorithmic bytes, measured traffic.
    walk_flops = 45.0 * st.ninteractions
    # what one launch of the production walk moves by design: the walk's compulsory bytes, the PM mesh it clears for the next deposit
    # (the `scrub` in the tasks' prologue), and - only on the fused route - the readout of its own targets (512 B each) + OldAcc
    pm_cov_bytes = deposit_bytes + fft_bytes + (0.0 if fused.value else readout_bytes + oldacc_bytes)
    walk_alg = {"tree_walk": tree_bytes, "pm_mesh_cleared_in_the_walk": mesh_bytes,
                "pm_readout_in_the_prologue": (readout_bytes + oldacc_bytes) if fused.value else 0.0}
    walk_alg_bytes = sum(walk_alg.values())
    walk_roofline = {"bound": "valu-f64", "kernel": "grav_walk_exact_kernel", "achieved": walk_flops / max(walk_s, 1e-12) / 1e12,
                     "peak": FP64_VECTOR_PEAK_TF, "unit": "TFLOP/s",
                     "frac": walk_flops / max(walk_s, 1e-12) / 1e12 / FP64_VECTOR_PEAK_TF, "traffic": traffic, "traffic_pair_kernel": traffic_pair,
                     "algorithmic_flops": walk_flops, "algorithmic_bytes": walk_alg_bytes, "algorithmic_bytes_parts": walk_alg,
                     "hbm_algorithmic_GBs": walk_alg_bytes / max(walk_s, 1e-12) / 1e9,
                     "valu_busy": valu_busy,
                     "note": "FP64-issue bound: 45 flop per interaction (SURVEY 8(d)) x interactions / kernel time against the FP64 "
                             "peak; valu_busy = SQ_INSTS_VALU x 4 / resident SIMD cycles from the committed SQ pass, in which the profiler "
                             "SERIALISES the main walk and the pair kernel (a counter pass cannot see them overlapped); `traffic` = HBM "
                             "bytes per launch of the production walk (main walk row only) from the committed FETCH_SIZE / WRITE_SIZE "
                             "passes, against algorithmic_bytes = the parts listed.  The HBM-bound part "
                             "of the step is the PM: roofline_pm_fft.  Kernel time = one HIP-event bracket on the walk's stream around "
                             "grav_walk_exact_kernel and grav_pair_kernel, which runs beside it on a second stream and is waited for "
                             "before the closing event (both start within 10 us, the pair kernel ends ~0.3 ms after the walk: "
                             "profiles/README.md); the interactions counted are those of both"}
    out = {
        "metric": "particle-steps/sec (grav+PM+SPH) at 256^3; rms force error vs ref",
        "value": n * world * args.steps / elapsed,
        "unit": "particle-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "dm-only %d^3 TreePM (S-%s, Nmesh %d, Asmth 1.5, Rcut 6, ErrTolForceAcc %g, exact window)"
                               % (n1, args.kind, nmesh, args.errtol),
                   "particles_per_gpu": n, "nmesh": nmesh, "parallelism": "replicas x%d" % world if world > 1 else "1 GPU",
                   "walk": "exact (per-target reference opening decisions)", "step_calls": step_route},
        "roofline": walk_roofline if walk_s >= pm_s else
                    {"bound": "hbm", "kernel": dominant, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": ach / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes": pm_bytes},
        "roofline_pm_fft": {"bound": "hbm", "kernel": "fft_t_z_fwd / fft_t_tile x3 / fft_t_z_inv (5 fused passes, transposing pipeline)",
                            "achieved": fft_bytes / max(fft_s, 1e-12) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": fft_bytes / max(fft_s, 1e-12) / 1e9 / HBM_PEAK_GBS, "traffic": traffic_fft,
                            "algorithmic_bytes": fft_bytes,
                            "note": "yardstick: tools/copy_probe.hip moves this mesh with 16-byte loads and stores at 5.8 TB/s in contiguous "
                                    "48 KB tiles (6.4 TB/s for a plain nontemporal copy), 4.6 TB/s with one side in 64-byte pieces, 3.0-3.9 TB/s "
                                    "with both sides in 64-byte pieces (the in-place passes of rounds 1-3); the transposing pipeline of round 4 "
                                    "has contiguous tiles on eight of its ten sides: its five passes at the probe's rates would take 6.9 ms"},
        "kernels": {
            "tree_walk_ms": st.kernel_ms, "tree_walk_with_counters_ms": counted_walk_ms,
            "pair_kernel_recovered_launches": int(pair_rec.value), "pair_kernel_stack_high_water": int(pair_high.value),
            "pair_kernel_lean_records": bool(capi.hip.shq_walk_pair_lean(ctx.h) == 1), "tree_interactions_per_target": st.ninteractions / max(1, st.ntargets),
            "tree_interactions_per_s": st.ninteractions / max(walk_s, 1e-12),
            "tree_fp64_frac_of_vector_peak": 45.0 * st.ninteractions / max(walk_s, 1e-12) / 1e12 / FP64_VECTOR_PEAK_TF,
            "tree_nodes_visited_per_wave": st.nnodes_visited / max(1.0, st.ntargets / 64.0),
            "tree_lane_efficiency": st.ninteractions / max(1.0, 64.0 * st.nwave_interactions),
            "tree_node_rounds_per_wave": st.nwave_node_interactions / max(1.0, st.ntargets / 64.0),
            "tree_particle_rounds_per_wave": (st.nwave_interactions - st.nwave_node_interactions) / max(1.0, st.ntargets / 64.0),
            "tree_node_interactions_per_target": st.nnode_interactions / max(1, st.ntargets),
            "tree_algorithmic_GBs": tree_bytes / max(walk_s, 1e-12) / 1e9,
            # the bespoke pipeline runs forward FFT, Green's function and inverse FFT as five fused passes: one phase
            "pm_ms": {"zero_and_deposit": ph[0], "fft_pipeline_5_passes_incl_greens_function": ph[1] + ph[2] + ph[3], "readout": ph[4],
                      "total": ph[5],
                      "note": "zero: the mesh is cleared in the shadow of the previous walk; readout: 0 when the walk's prologue does it (config.step_calls)"},
            "tree_wave_figures_from": "the counter launch (one task per wave, no leaf ring): lane efficiency and rounds per wave describe the union walk, not the ring's drain",
            # the bytes of what pm_ms.total covers: deposit + five passes + (unless the walk's prologue did it) readout and OldAcc; the
            # clearing of the mesh rides in the walk and is counted there (roofline.algorithmic_bytes_parts)
            "pm_algorithmic_GBs": pm_cov_bytes / max(pm_s, 1e-12) / 1e9,
            "pm_frac_of_hbm_peak": pm_cov_bytes / max(pm_s, 1e-12) / 1e9 / HBM_PEAK_GBS,
            "pm_reference_structure_GBs": pm_bytes_reference_structure / max(pm_s, 1e-12) / 1e9,
            "event_ms_per_step": ev_ms / args.steps,
        },
        "setup_s": {"synthetic_input": t_setup, "upload": t_upload, "tree_build_device": t_tree_build,
                    "tree_build_device_kernels_ms": float(tb.build_ms), "tree_nodes": numnodes, "tree_depth": int(tb.maxdepth),
                    "seed_step": t_seed},
    }
    if rank == 0 and not args.no_cpu_baseline:
        t0 = time.perf_counter()
        tree = sq.force_tree_full(pman)      # host build of the same tree, for the oracle only
        out["setup_s"]["tree_build_host_for_oracle"] = time.perf_counter() - t0
        oldacc = np.zeros(n)
        acc = np.zeros((n, 3))
        gpm = np.zeros((n, 3))
        capi.check(capi.hip.shq_grav_short_download(ctx.h, capi.ptr(acc), None, None, None))
        capi.check(capi.hip.shq_pm_download(ctx.h, capi.ptr(gpm), None))
        oldacc = np.linalg.norm(acc + gpm, axis=1) / G
        # one more device walk from exactly this OldAcc (the last timed step refreshed it from these same arrays), so that
        # the CPU walk and the device walk under comparison start from the same inputs
        capi.check(capi.hip.shq_grav_short_run(ctx.h, C.byref(gp_rel), None, 0, 1, args.walk_mode))
        capi.check(capi.hip.shq_grav_short_download(ctx.h, capi.ptr(acc), None, None, None))
        out["cpu_baseline"] = cpu_baseline(pos, P["Mass"], tree, gp_rel, oldacc, L, acc_gpu=acc)
        out["force_error"] = out["cpu_baseline"].pop("force_error")
        try:
            out["kernels"].update(distributed_walk_figures(ctx, sq, capi, tree, pos, oldacc, gp_rel, n))
        except Exception as e:  # an extra figure: never fatal
            out["kernels"]["dist_note"] = "distributed-walk figures skipped: %s" % e
    # Extra figure, outside `value`: a fully resident step with moving particles — drift, device tree
    # build, PM, walk, OldAcc, short-range and PM kicks — nothing crosses PCIe.
    try:
        P["Vel"] = np.random.default_rng(7).normal(size=(n, 3))
        sq.dynamics_upload(ctx, pman)
        # kick factors small enough that the S-cluster (accelerations up to ~1e16 in its core) keeps its shape over the passes:
        # with 1e-9 the core flew apart within a few passes and the loop timed another particle distribution (644 instead of 507
        # interactions per target)
        gk = np.full(capi.TIMEBINS + 1, 1e-24)
        nres = 8                                # one refresh period of the tree-order target list (tree_targets_refresh builds): exactly one
        #                                         of the timed passes rebuilds it, as one step in eight does in a run
        capi.check(capi.hip.shq_set_walk_stats(ctx.h, 0))
        # a sub-step of the hierarchical integrator: no potential update (update_potential = 0, its own kernel instantiation and
        # so its own row in a profile), kicks from the walk's Accel output (AccelStore, timestep.cpp:273)
        nwarm = 6                               # the GPU idled for seconds during the CPU baseline: let the clocks come back
        for it in range(nres + nwarm):
            if it == nwarm:
                ctx.synchronize()
                t0 = time.perf_counter()
            if it == nres + nwarm - 1:
                ctx.timer_begin(1)              # the last pass on one time line (kernels.resident_timeline_ms)
            sq.drift(ctx, 1e-4 * L / n1, L)
            if not (args.separate_calls or args.walk_mode != 0):
                # gravpm_force started early: the PM runs on the library's second stream while the tree is built; shq_treepm_step joins it
                capi.check(capi.hip.shq_pm_start(ctx.h, C.byref(pmp), G))
            sq.tree_build_device(ctx, L)
            # targets in tree order: the particle index order goes stale as the particles move
            if args.separate_calls or args.walk_mode != 0:
                capi.check(capi.hip.shq_pm_run(ctx.h, C.byref(pmp)))
                capi.check(capi.hip.shq_grav_refresh_oldacc(ctx.h, G))
                capi.check(capi.hip.shq_grav_short_run(ctx.h, C.byref(gp_rel), None, 0, 0, args.walk_mode | sq.WALK_TREE_ORDER))
            else:
                capi.check(capi.hip.shq_treepm_step(ctx.h, C.byref(pmp), C.byref(gp_rel), 0, sq.WALK_EXACT | sq.WALK_TREE_ORDER))
            sq.kick_short(ctx, gk, from_accel_store=True)
            sq.kick_pm(ctx, 1e-24)
        ctx.timer_end(1)
        ctx.synchronize()
        t_res = time.perf_counter() - t0
        def at(slot, which=0):
            ms = C.c_double(0)
            capi.check(capi.hip.shq_timer_between_ms(ctx.h, 1, 0, slot, which, C.byref(ms)))
            return round(ms.value, 3)
        out["kernels"]["resident_timeline_ms"] = {"pm_begin": at(8), "fft_begin": at(9), "fft_end": at(10), "pm_end": at(13), "tree_build_begin": at(16),
                                                  "tree_build_end": at(16, 1), "walk_begin": at(19), "walk_end": at(19, 1), "kicks_end": at(1, 1),
                                                  "note": "the last pass of the resident loop from its drift on (HIP events across the two streams)"}
        out["kernels"]["resident_full_step_ms"] = 1e3 * t_res / nres
        out["kernels"]["resident_full_step_particle_steps_per_s"] = n * nres / t_res
        rst = sq.WalkStats()
        capi.check(capi.hip.shq_grav_short_download(ctx.h, None, None, None, C.byref(rst)))
        out["kernels"]["resident_walk_ms"] = float(rst.kernel_ms)                 # the last pass's walk kernel
        out["kernels"]["resident_interactions_per_target"] = rst.ninteractions / max(rst.ntargets, 1)
    except sq.ShqError as e:  # e.g. a tree deeper than the device build supports: the extra figure is simply absent
        out["kernels"]["resident_full_step_ms"] = None
        out["kernels"]["resident_full_step_note"] = str(e)
    # Extra figures, outside `value`: the other particle loops of a step on the resident set — the
    # time-step assignment (find_timesteps particle loop) and a friends-of-friends pass over the same box.
    try:
        tp = capi.TimestepParams()
        tp.ErrTolIntAccuracy, tp.CourantFac, tp.MinSizeTimestep, tp.ForceSoftening = 0.02, 0.15, 1e-9, sq.FORCE_SOFTENING()
        tp.atime, tp.hubble, tp.fac3, tp.dti_max = 0.5, 0.28, 0.5 ** -1.0, 1 << 40
        tp.tl.Ti_Current, tp.tl.loga_now, tp.tl.Dloga_interval, tp.tl.nseg = 0, np.log(0.5), np.log(2.0) / (1 << 46), 1
        tp.tl.seg_snap[0], tp.tl.seg_loga[0], tp.tl.seg_loga[1] = 0, np.log(0.5), 0.0
        capi.check(capi.hip.shq_timebins_upload(ctx.h, None, None))
        tr = capi.TimestepResult()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            capi.check(capi.hip.shq_find_timesteps(ctx.h, C.byref(tp), None, 0, 0, -1, C.byref(tr)))
        out["kernels"]["find_timesteps_ms"] = 1e3 * (time.perf_counter() - t0) / 3
        out["kernels"]["find_timesteps_bins"] = [int(tr.mTimeBin), int(tr.maxTimeBin)]
        # A hierarchical step (BASELINE configs[4] "hierarchical timestepping"; timestep.cpp:305-480) on the resident
        # set, as a PM step runs it: gravity bins of all particles from the stored accelerations (shq_hier_gravity_bins), the
        # push-down rule, then the sub-step levels inside the library (shq_hier_gravity_levels: sub-list -> tree of the sub-list ->
        # walk -> bin refinement -> kick per level, SHQ_WALK_AUTO).  The top level of such a step is the full TreePM force of `value`.
        try:
            sq.tree_build_device(ctx, L)
            capi.check(capi.hip.shq_grav_short_run(ctx.h, C.byref(gp_rel), None, 0, 1, 0))
            capi.check(capi.hip.shq_grav_refresh_oldacc(ctx.h, G))
            top = 40
            tp.dti_max = 1 << top
            capi.check(capi.hip.shq_hier_gravity_bins(ctx.h, C.byref(tp), None, 0, 0, top, C.byref(tr)))
            counts = [int(c) for c in tr.timebincounts]
            largest = next((b for b in range(top, 0, -1) if counts[b] > 0), top)
            push = largest
            for b in range(largest, 0, -1):          # the push-down of a PM step (timestep.cpp:393-414)
                if counts[b] // 3 > counts[b - 1]:
                    break
                push = b - 1
                counts[b - 1] += counts[b]
            if 0 < push < largest:
                capi.check(capi.hip.shq_hier_push_down(ctx.h, None, 0, push))
                largest = push
            sq.build_active_particles(ctx, 0, True)      # a PM step: every particle is active (build_active_particles, timestep.cpp:1286-1349)
            gkl = np.full(capi.TIMEBINS + 1, 1e-24)
            levels = (capi.HierLevel * capi.TIMEBINS)()
            nlev, mingrav, bad = C.c_int(0), C.c_int(0), C.c_int64(0)
            ctx.synchronize()
            t0 = time.perf_counter()
            capi.check(capi.hip.shq_hier_gravity_levels(ctx.h, C.byref(tp), C.byref(gp_rel), L, 0x3f, 0, largest, capi.ptr(gkl), sq.WALK_AUTO,
                                                        C.byref(mingrav), C.byref(bad), levels, C.byref(nlev)))
            ctx.synchronize()
            t_lev = time.perf_counter() - t0
            lv = [{"bin": int(levels[i].timebin), "particles": int(levels[i].nparticles), "tree_nodes": int(levels[i].tree_nodes),
                   "tree_build_ms": float(levels[i].tree_build_ms), "walk_ms": float(levels[i].walk_ms),
                   "walk": "group" if levels[i].walk_mode == sq.WALK_GROUP else "exact"} for i in range(nlev.value)]
            out["kernels"]["hier_sublevels_ms"] = 1e3 * t_lev
            out["kernels"]["hier_step_ms"] = out["ms_per_step"] + 1e3 * t_lev
            out["kernels"]["hier_levels"] = lv
            out["kernels"]["hier_note"] = ("PM step of the hierarchical integrator: top level = the TreePM force of `value` (ms_per_step) + %d sub-step "
                                           "levels below bin %d run by shq_hier_gravity_levels on the resident set (wall time, host-inclusive); "
                                           "substeps of the particles in a level = sum of the deeper levels' lists" % (nlev.value, largest))
            out["kernels"]["hier_particle_substeps_per_s"] = (n + sum(x["particles"] for x in lv)) / (1e-3 * out["kernels"]["hier_step_ms"])
        except sq.ShqError as e:
            out["kernels"]["hier_note"] = "skipped: %s" % e
        fp = capi.FofParams(L, 0.2 * L / n1, 2, 1 + 16 + 32, 32, 0)
        ids = np.arange(1, n + 1, dtype=np.uint64)
        ng = C.c_int64()
        capi.check(capi.hip.shq_fof(ctx.h, C.byref(fp), capi.ptr(ids), None, None, C.byref(ng)))
        t0 = time.perf_counter()
        capi.check(capi.hip.shq_fof(ctx.h, C.byref(fp), capi.ptr(ids), None, None, C.byref(ng)))
        out["kernels"]["fof_ms"] = 1e3 * (time.perf_counter() - t0)
        out["kernels"]["fof_groups"] = int(ng.value)
        out["kernels"]["fof_note"] = "shq_fof end to end (tree of the primary type, linking, catalogue), linking length 0.2 mean separations, groups of >= 32"
    except sq.ShqError as e:
        out["kernels"]["fof_note"] = "skipped: %s" % e
    if not args.no_sph:
        figs = sph_figures(ctx, n1=args.sph_ngrid)
        for k in ("roofline_sph_density", "roofline_sph_hydro"):
            out[k] = figs.pop(k)
        out["kernels"].update(figs)
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)


if __name__ == "__main__":
    main()
