"""shenqi_amd — MI355X-native TreePM + SPH force engine behind shenqi's operator API.

Thin Python driver over two native libraries:
  lib/libshenqi_hip.so   hand-written HIP (gfx950) kernels + the C-ABI (include/shenqi_hip.h)
  lib/libshenqi_host.so  C++ host mirror of the reference operator interface
                         (force_tree_full, grav_short_tree, gravpm_force, ...)
Names follow the reference (libgadget/gravity.h, forcetree.h, partmanager.h).
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import (PARTICLE_DTYPE, SPH_DTYPE, NODE_DTYPE, WALK_EXACT, WALK_GROUP, WALK_AUTO, WALK_TREE_ORDER, WALK_DEFER_POSTPROCESS, GravParams, PMParams,
                   WalkStats, ShqError)

GASMASK, DMMASK, NUMASK, STARMASK, BHMASK = 1, 2, 4, 16, 32
ALLMASK = (1 << 6) - 1
SHORTRANGE_FORCE_WINDOW_TYPE_EXACT = 1
SHORTRANGE_FORCE_WINDOW_TYPE_ERFC = 2


class Context:
    """One library context per rank / GPU (shq_init)."""

    def __init__(self, device=0, stream=None):
        h = C.c_void_p()
        capi.check(capi.hip.shq_init(device, stream, C.byref(h)), "shq_init")
        self.h = h
        self.stream = stream        # the caller's stream the library works on (None: its own)

    def close(self):
        if self.h:
            capi.hip.shq_shutdown(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def synchronize(self):
        capi.check(capi.hip.shq_synchronize(self.h), "shq_synchronize")

    def timer_begin(self, slot):
        capi.check(capi.hip.shq_timer_begin(self.h, slot))

    def timer_end(self, slot):
        capi.check(capi.hip.shq_timer_end(self.h, slot))

    def timer_ms(self, slot):
        ms = C.c_double()
        capi.check(capi.hip.shq_timer_elapsed_ms(self.h, slot, C.byref(ms)))
        return ms.value


class PartManager:
    """PartManager[1] of the reference: a particle_data array plus NumPart / BoxSize."""

    def __init__(self, numpart, BoxSize):
        self.Base = np.zeros(int(numpart), dtype=PARTICLE_DTYPE)
        self.NumPart = int(numpart)
        self.BoxSize = float(BoxSize)
        self._h = capi.host.shqh_partmanager_create(capi.ptr(self.Base), self.NumPart, self.BoxSize)

    def __del__(self):
        if getattr(self, "_h", None) and capi is not None:   # capi is None when the interpreter is shutting down
            capi.host.shqh_partmanager_free(self._h)
            self._h = None

    def view(self):
        v = capi.PartView()
        capi.host.shqh_part_view(self._h, C.byref(v))
        return v


class ForceTree:
    def __init__(self, handle, pman):
        self._h = handle
        self._pman = pman  # keep the particles alive
        info = (C.c_int64 * 5)()
        capi.host.shqh_tree_info(handle, C.byref(info))
        self.firstnode, self.lastnode, self.numnodes, self.NumParticles, self.full_particle_tree_flag = [int(x) for x in info]
        self.BoxSize = pman.BoxSize

    @property
    def Nodes_base(self):
        p = capi.host.shqh_tree_nodes(self._h)
        buf = (C.c_char * (self.numnodes * 120)).from_address(p)
        return np.frombuffer(buf, dtype=NODE_DTYPE)

    def view(self):
        v = capi.TreeView()
        capi.host.shqh_tree_view(self._h, C.byref(v))
        return v

    def free(self):
        if self._h:
            capi.host.shqh_force_tree_free(self._h)
            self._h = None

    def __del__(self):
        self.free()


def force_tree_rebuild_mask(pman, mask, active=None, full=False):
    act = None if active is None else np.ascontiguousarray(active, dtype=np.int32)
    h = capi.host.shqh_force_tree_rebuild_mask(pman._h, mask, capi.ptr(act), 0 if act is None else len(act), int(full))
    if not h:
        raise ShqError(capi.host.shqh_last_error().decode())
    return ForceTree(h, pman)


def tree_build_device(ctx, BoxSize, mask=None, active=None):
    """shq_tree_build: build the tree of the uploaded particles on the device and install it in `ctx`.
    Returns capi.TreeBuildStats."""
    st = capi.TreeBuildStats()
    act, nact = _active_arg(active)
    capi.check(capi.hip.shq_tree_build(ctx.h, float(BoxSize), ALLMASK if mask is None else int(mask), act, nact, C.byref(st)),
               "shq_tree_build")
    return st


def tree_build_domain(ctx, BoxSize, geo, topleaves, ThisTask, firstnode, mask=None, active=None):
    """shq_tree_build_domain: the device tree under a domain decomposition.  geo: capi.TOPNODE_GEO_DTYPE array, topleaves:
    capi.TOPLEAF_DTYPE array (Task read, treenode written).  Returns (stats, local moments as capi.TOPLEAF_MOMENTS_DTYPE)."""
    st = capi.TreeBuildStats()
    act, nact = _active_arg(active)
    geo = np.ascontiguousarray(geo, dtype=capi.TOPNODE_GEO_DTYPE)
    assert topleaves.dtype == capi.TOPLEAF_DTYPE and topleaves.flags["C_CONTIGUOUS"]
    mom = np.zeros(len(topleaves), dtype=capi.TOPLEAF_MOMENTS_DTYPE)
    capi.check(capi.hip.shq_tree_build_domain(ctx.h, float(BoxSize), ALLMASK if mask is None else int(mask), act, nact, capi.ptr(geo), len(geo),
                                              capi.ptr(topleaves), len(topleaves), int(ThisTask), int(firstnode), capi.ptr(mom), C.byref(st)),
               "shq_tree_build_domain")
    return st, mom


def tree_set_topleaf_moments(ctx, moments):
    """shq_tree_set_topleaf_moments: the all-gathered top-leaf moments (capi.TOPLEAF_MOMENTS_DTYPE) into the pseudo nodes."""
    moments = np.ascontiguousarray(moments, dtype=capi.TOPLEAF_MOMENTS_DTYPE)
    capi.check(capi.hip.shq_tree_set_topleaf_moments(ctx.h, capi.ptr(moments), len(moments)), "shq_tree_set_topleaf_moments")


RESIDENT = "resident"        # the list of the last build_active_particles (SHQ_ACTIVE_RESIDENT)
RESIDENT_SUB = "resident-sub"  # the list of the last build_active_sublist (SHQ_SUBLIST_RESIDENT)


def _active_arg(active):
    """(pointer, count) for the `active` argument of the C-ABI: None, a host index list or a resident handle."""
    if active is None:
        return None, 0
    if isinstance(active, str):
        return {RESIDENT: capi.ACTIVE_RESIDENT, RESIDENT_SUB: capi.SUBLIST_RESIDENT}[active], 0
    act = np.ascontiguousarray(active, dtype=np.int32)
    _active_arg.keep = act
    return capi.ptr(act), len(act)


def toptree_upload(ctx, tree, topleaves):
    """shq_toptree_upload: the TopLevel nodes of the host tree + the domain's TopLeaves (capi.TOPLEAF_DTYPE)."""
    tl = np.ascontiguousarray(topleaves, dtype=capi.TOPLEAF_DTYPE)
    tv = tree.view()
    capi.check(capi.hip.shq_toptree_upload(ctx.h, C.byref(tv), capi.ptr(tl), len(tl)), "shq_toptree_upload")


def _toptree_exports(call, ntargets, active):
    """One call with a generous table; a second, exactly sized one only if that was too small."""
    act, nact = _active_arg(active)
    counts = np.zeros(ntargets, dtype=np.int32)
    n = C.c_int64()
    cap = max(1024, ntargets // 4)
    table = np.zeros(cap, dtype=capi.DATA_INDEX_DTYPE)
    rc = call(act, nact, capi.ptr(counts), capi.ptr(table), cap, C.byref(n))
    if rc != 0 and n.value > cap:
        table = np.zeros(n.value, dtype=capi.DATA_INDEX_DTYPE)
        rc = call(act, nact, capi.ptr(counts), capi.ptr(table), n.value, C.byref(n))
    capi.check(rc, "toptree exports")
    return counts, table[: n.value].copy()


def grav_toptree_exports(ctx, gp, ntargets, active=None):
    """shq_grav_toptree_exports: (exportcounts [ntargets] inclusive scan, DataIndexTable) of
    GravTopTreeWalk::toptree_visit (libgadget/gravshort2.hpp:362-438)."""
    return _toptree_exports(lambda a, na, cnt, tab, cap, n: capi.hip.shq_grav_toptree_exports(ctx.h, C.byref(gp), a, na, cnt, tab, cap, n),
                            ntargets, active)


def ngb_toptree_exports(ctx, symmetric, BoxSize, ntargets, active=None):
    """shq_ngb_toptree_exports: TopTreeWalk::toptree_visit with cull_node (libgadget/localtreewalk2.h:210-259)."""
    return _toptree_exports(lambda a, na, cnt, tab, cap, n: capi.hip.shq_ngb_toptree_exports(ctx.h, int(symmetric), float(BoxSize), a, na, cnt, tab,
                                                                                           cap, n), ntargets, active)


def timebins_upload(ctx, bin_gravity=None, bin_hydro=None):
    bg = None if bin_gravity is None else np.ascontiguousarray(bin_gravity, dtype=np.uint8)
    bh = None if bin_hydro is None else np.ascontiguousarray(bin_hydro, dtype=np.uint8)
    capi.check(capi.hip.shq_timebins_upload(ctx.h, capi.ptr(bg), capi.ptr(bh)), "shq_timebins_upload")


def build_active_particles(ctx, Ti_Current, is_pm_step=False):
    """shq_build_active_particles: build_active_particles (libgadget/timestep.cpp:1286-1349) on the device.
    Returns capi.ActiveInfo; the list stays resident (pass active=RESIDENT)."""
    info = capi.ActiveInfo()
    capi.check(capi.hip.shq_build_active_particles(ctx.h, int(Ti_Current), int(bool(is_pm_step)), C.byref(info)),
               "shq_build_active_particles")
    return info


def build_active_sublist(ctx, maxtimebin, Ti_Current):
    """shq_build_active_sublist: build_active_sublist (libgadget/timestep.cpp:1373-1399). Returns its length."""
    n = C.c_int64()
    capi.check(capi.hip.shq_build_active_sublist(ctx.h, int(maxtimebin), int(Ti_Current), C.byref(n)), "shq_build_active_sublist")
    return n.value


def active_download(ctx, sublist=False):
    n = C.c_int64()
    capi.check(capi.hip.shq_active_download(ctx.h, int(sublist), None, 0, C.byref(n)), "shq_active_download")
    out = np.zeros(n.value, dtype=np.int32)
    capi.check(capi.hip.shq_active_download(ctx.h, int(sublist), capi.ptr(out), n.value, C.byref(n)), "shq_active_download")
    return out


def tree_download(ctx, firstnode, numpart=0):
    """shq_tree_download: the device-built tree as a NODE array (numbered from `firstnode` in pre-order)
    and, if numpart > 0, the Father array."""
    nn = C.c_int64()
    capi.check(capi.hip.shq_tree_download(ctx.h, int(firstnode), None, 0, None, C.byref(nn)), "shq_tree_download")
    nodes = np.zeros(nn.value, dtype=NODE_DTYPE)
    father = np.full(int(numpart), -1, dtype=np.int32) if numpart > 0 else None
    capi.check(capi.hip.shq_tree_download(ctx.h, int(firstnode), capi.ptr(nodes), nn.value, capi.ptr(father), C.byref(nn)),
               "shq_tree_download")
    return nodes, father


def dynamics_upload(ctx, pman):
    pv = pman.view()
    capi.check(capi.hip.shq_dynamics_upload(ctx.h, C.byref(pv)), "shq_dynamics_upload")


def drift(ctx, ddrift, BoxSize, random_shift=None):
    """shq_drift: drift_all_particles (libgadget/drift.cpp:16-99) on the resident particles."""
    rs = None if random_shift is None else np.ascontiguousarray(random_shift, dtype=np.float64)
    capi.check(capi.hip.shq_drift(ctx.h, float(ddrift), float(BoxSize), capi.ptr(rs)), "shq_drift")


def kick_short(ctx, gravkick, active=None, from_accel_store=False):
    """shq_kick_short: gravity part of apply_half_kick (libgadget/timestep.cpp:838-872)."""
    gk = np.ascontiguousarray(gravkick, dtype=np.float64)
    assert gk.shape == (capi.TIMEBINS + 1,)
    act, nact = _active_arg(active)
    capi.check(capi.hip.shq_kick_short(ctx.h, capi.ptr(gk), act, nact, int(from_accel_store)), "shq_kick_short")


def kick_hydro(ctx, hydrokick, dt_entr, atime, MaxGasVel, active=None, from_hydro_output=True):
    """shq_kick_hydro: do_hydro_kick for gas (libgadget/timestep.cpp:970-1003). Returns the number of clamped particles."""
    hk = np.ascontiguousarray(hydrokick, dtype=np.float64)
    de = np.ascontiguousarray(dt_entr, dtype=np.float64)
    assert hk.shape == (capi.TIMEBINS + 1,) and de.shape == (capi.TIMEBINS + 1,)
    act, nact = _active_arg(active)
    nlim = C.c_int64()
    capi.check(capi.hip.shq_kick_hydro(ctx.h, capi.ptr(hk), capi.ptr(de), float(atime), float(MaxGasVel), act, nact, int(from_hydro_output),
                                       C.byref(nlim)), "shq_kick_hydro")
    return nlim.value


def entropy_download(ctx, n):
    out = np.zeros(n)
    capi.check(capi.hip.shq_entropy_download(ctx.h, capi.ptr(out)), "shq_entropy_download")
    return out


def kick_pm(ctx, Fgravkick):
    """shq_kick_pm: apply_PM_half_kick (libgadget/timestep.cpp:937-959)."""
    capi.check(capi.hip.shq_kick_pm(ctx.h, float(Fgravkick)), "shq_kick_pm")


def dynamics_download(ctx, pman):
    pv = pman.view()
    capi.check(capi.hip.shq_dynamics_download(ctx.h, C.byref(pv)), "shq_dynamics_download")


def force_tree_full(pman):
    return force_tree_rebuild_mask(pman, ALLMASK, None, full=True)


def set_gravshort_treepar(ErrTolForceAcc=0.002, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=2, Rcut=6.0,
                          FractionalGravitySoftening=1.0 / 30.0, ShortRangeForceWindowType=SHORTRANGE_FORCE_WINDOW_TYPE_EXACT):
    capi.host.shqh_set_gravshort_treepar(ErrTolForceAcc, BHOpeningAngle, MaxBHOpeningAngle, TreeUseBH, Rcut,
                                         FractionalGravitySoftening, ShortRangeForceWindowType)


def get_TreeUseBH():
    return capi.host.shqh_get_TreeUseBH()


def gravshort_set_softenings(MeanSeparation):
    capi.host.shqh_gravshort_set_softenings(MeanSeparation)


def FORCE_SOFTENING():
    return capi.host.shqh_FORCE_SOFTENING()


def make_grav_params(BoxSize, Asmth, Nmesh, G, rho0):
    gp = GravParams()
    capi.check_host(capi.host.shqh_make_grav_params(BoxSize, Asmth, Nmesh, G, rho0, C.byref(gp)), "make_grav_params")
    return gp


def grav_short_tree(ctx, act, pm, tree, AccelStore, rho0, Ti_Current=0, UseGPU=True, walk_mode=WALK_EXACT, pman=None):
    """grav_short_tree(act, pm, tree, AccelStore, rho0, Ti_Current, UseGPU), libgadget/gravity.h:89.
    act: int32 index array or None (all). pm: dict(Asmth, Nmesh, G). Returns WalkStats."""
    pman = pman or tree._pman
    stats = WalkStats()
    a = None if act is None else np.ascontiguousarray(act, dtype=np.int32)
    rc = capi.host.shqh_grav_short_tree(ctx.h, pman._h, tree._h, pm["Asmth"], pm["Nmesh"], pm["G"], capi.ptr(a),
                                        0 if a is None else len(a), capi.ptr(AccelStore), rho0, int(UseGPU), walk_mode,
                                        C.byref(stats))
    capi.check_host(rc, "grav_short_tree")
    return stats


def gravpm_force(ctx, pm, pman, UseGPU=True):
    capi.check_host(capi.host.shqh_gravpm_force(ctx.h, pman._h, pm["Asmth"], pm["Nmesh"], pm["G"], int(UseGPU)), "gravpm_force")


def synth_positions(kind, n, seed=20240601, L=1.0):
    """SURVEY §8(d) synthetic inputs: kind 'grid' | 'uniform' | 'cluster'."""
    k = {"grid": 0, "uniform": 1, "cluster": 2}[kind]
    pos = np.empty((int(n), 3), dtype=np.float64)
    capi.host.shqh_synth_positions(k, int(n), seed, L, capi.ptr(pos))
    return pos


def synth_positions_range(kind, nglobal, first, count, seed=20240601, L=1.0):
    """particles [first, first + count) of one global synthetic set of nglobal particles, the same for any split over ranks"""
    k = {"grid": 0, "uniform": 1, "cluster": 2}[kind]
    pos = np.empty((int(count), 3), dtype=np.float64)
    capi.host.shqh_synth_positions_range(k, int(nglobal), int(first), int(count), seed, L, capi.ptr(pos))
    return pos


def make_density_params(BoxSize, kernel=1, eta=1.0, MaxNumNgbDeviation=0.5, BlackHoleNgbFactor=2.0, update_hsml=1, DoEgyDensity=1, BlackHoleOn=0, MinGasHsml=0.006):
    """POD mirror of DensityPriv (densitytree2.hpp:10-52) for the C-ABI: DesNumNgb from DensityKrnl::desnumngb (densitykernel.hpp:36-41);
    kick factors all zero"""
    support = {1: 4, 2: 6, 4: 5}[kernel]
    des = 4.0 / 3 * np.pi * (support / 2.0 * eta) ** 3
    dp = capi.DensityParams()
    dp.BoxSize, dp.DesNumNgb, dp.DesNumNgbBH, dp.MinGasHsml = BoxSize, des, des * BlackHoleNgbFactor, MinGasHsml
    dp.MaxNumNgbDeviation = MaxNumNgbDeviation
    dp.update_hsml, dp.BlackHoleOn, dp.DoEgyDensity, dp.WindsDecouple = update_hsml, BlackHoleOn, DoEgyDensity, 0
    dp.DensityKernelType = kernel
    return dp


def make_hydro_params(BoxSize, atime=0.1, hubble=0.1, kernel=1, DensityIndependentSphOn=1, DensityContrastLimit=100.0, ArtBulkViscConst=0.75):
    """POD mirror of HydroPriv (hydratree2.hpp:83-119) for the C-ABI; kick factors and drifts all zero"""
    g = 5.0 / 3.0
    hp = capi.HydroParams()
    hp.BoxSize, hp.atime = BoxSize, atime
    hp.fac_mu = atime ** (3 * (g - 1) / 2) / atime
    hp.fac_vsic_fix = hubble * atime ** (3 * (g - 1))
    hp.hubble_a2 = hubble * atime * atime
    hp.ArtBulkViscConst, hp.DensityContrastLimit = ArtBulkViscConst, DensityContrastLimit
    hp.DensityIndependentSphOn, hp.DensityKernelType = DensityIndependentSphOn, kernel
    return hp


def morton_order(pos, L):
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    order = np.empty(len(pos), dtype=np.int32)
    capi.host.shqh_morton_order(capi.ptr(pos), len(pos), L, capi.ptr(order))
    return order


def hilbert_order(pos, L):
    """Peano-Hilbert order (what the reference keeps particles in, domain.cpp:268)."""
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    order = np.empty(len(pos), dtype=np.int32)
    capi.host.shqh_hilbert_order(capi.ptr(pos), len(pos), L, capi.ptr(order))
    return order


# ---- SPH operators (libgadget/density2.h, hydra2.h) ---------------------------------------------
from .capi import KickFactors, DensityParams, HydroParams, SphStats  # noqa: E402

DENSITY_KERNEL_CUBIC_SPLINE, DENSITY_KERNEL_QUINTIC_SPLINE, DENSITY_KERNEL_QUARTIC_SPLINE = 1, 2, 4
BH_SLOT_DTYPE = np.dtype([("Density", "<f8"), ("DivVel", "<f8")])


def set_densitypar(DensityResolutionEta=1.0, MaxNumNgbDeviation=0.5, DensityKernelType=DENSITY_KERNEL_CUBIC_SPLINE,
                   BlackHoleNgbFactor=2.0, MinGasHsml=0.006):
    capi.host.shqh_set_densitypar(DensityResolutionEta, MaxNumNgbDeviation, DensityKernelType, BlackHoleNgbFactor, MinGasHsml)


def GetNumNgb():
    return capi.host.shqh_GetNumNgb()


def set_hydropar(DensityIndependentSphOn=1, DensityContrastLimit=100.0, ArtBulkViscConst=0.75):
    capi.host.shqh_set_hydropar(DensityIndependentSphOn, DensityContrastLimit, ArtBulkViscConst)


def set_init_hsml(tree, MeanGasSeparation, pman):
    capi.check_host(capi.host.shqh_set_init_hsml(tree._h, MeanGasSeparation, pman._h), "set_init_hsml")


def force_tree_update_hmax(tree, pman):
    capi.host.shqh_force_tree_update_hmax(tree._h, pman._h)


def density(ctx, act, update_hsml, DoEgyDensity, BlackHoleOn, kick, tree, pman, SphP, BhP=None, GradRho_mag=None,
            UseGPU=True):
    """density(act, update_hsml, DoEgyDensity, BlackHoleOn, times..., &EntVarPred, GradRho_mag, tree, UseGPU),
    libgadget/density2.h:42.  Returns (EntVarPred, SphStats)."""
    a = None if act is None else np.ascontiguousarray(act, dtype=np.int32)
    evp = np.zeros(max(len(SphP), 1))
    st = SphStats()
    kick = kick if kick is not None else KickFactors()
    rc = capi.host.shqh_density(ctx.h, pman._h, tree._h, capi.ptr(SphP), len(SphP), capi.ptr(BhP), 0 if BhP is None else len(BhP),
                                capi.ptr(a), 0 if a is None else len(a), int(update_hsml), int(DoEgyDensity), int(BlackHoleOn),
                                C.byref(kick), capi.ptr(evp), capi.ptr(GradRho_mag), int(UseGPU), C.byref(st))
    capi.check_host(rc, "density")
    return evp, st


def hydro_force(ctx, act, atime, hubble, EntVarPred, kick, tree, pman, SphP, drifts=None, UseGPU=True):
    """hydro_force(act, atime, EntVarPred, times..., tree, UseGPU), libgadget/hydra2.h:9."""
    a = None if act is None else np.ascontiguousarray(act, dtype=np.int32)
    st = SphStats()
    kick = kick if kick is not None else KickFactors()
    d = None if drifts is None else np.ascontiguousarray(drifts, dtype=np.float64)
    rc = capi.host.shqh_hydro_force(ctx.h, pman._h, tree._h, capi.ptr(SphP), len(SphP), capi.ptr(a), 0 if a is None else len(a),
                                    atime, hubble, capi.ptr(EntVarPred), C.byref(kick), capi.ptr(d), int(UseGPU), C.byref(st))
    capi.check_host(rc, "hydro_force")
    return st
