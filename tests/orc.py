"""ctypes wrapper of the CPU oracle (oracle/liboracle.so).  Test infrastructure only."""
import ctypes as C
import os

import numpy as np

from shenqi_amd.capi import GravParams, PMParams, DensityParams, HydroParams, NODE_DTYPE, ptr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_path = os.path.join(ROOT, "oracle", "liboracle.so")
if not os.path.exists(_path):
    raise ImportError(f"{_path} missing: run `make -C oracle` (or __graft_entry__.build())")
lib = C.CDLL(_path)
_vp = C.c_void_p
lib.orc_tree_build.argtypes = [_vp, _vp, _vp, _vp, C.c_int64, C.c_int64, C.c_double, _vp, C.c_int64, _vp]
lib.orc_tree_build.restype = C.c_int64
lib.orc_grav_walk.argtypes = [_vp, C.c_int64, _vp, _vp, _vp, _vp, C.c_int64, C.POINTER(GravParams), _vp, _vp, _vp]
lib.orc_grav_walk.restype = None
lib.orc_grav_walk_secondary.argtypes = [_vp, C.c_int64, _vp, _vp, _vp, _vp, _vp, C.c_int64, C.POINTER(GravParams), _vp, _vp, _vp]
lib.orc_grav_walk_secondary.restype = None
lib.orc_grav_postprocess.argtypes = [_vp, _vp, C.c_int64, C.POINTER(GravParams), C.c_int, _vp, _vp]
lib.orc_grav_postprocess.restype = None
lib.orc_apply_accn.argtypes = [_vp, C.c_double, C.c_double, C.POINTER(GravParams), _vp, _vp]
lib.orc_force_direct.argtypes = [_vp, _vp, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_int, _vp]
lib.orc_force_direct.restype = None
lib.orc_pm_force.argtypes = [_vp, _vp, _vp, C.c_int64, C.POINTER(PMParams), C.c_int, C.c_int, _vp, _vp, _vp, _vp]
lib.orc_pm_force.restype = None
lib.orc_fft_r2c.argtypes = [C.c_int, _vp, _vp]
lib.orc_fft_r2c.restype = None
lib.orc_fft_c2r.argtypes = [C.c_int, _vp, _vp]
lib.orc_fft_c2r.restype = None
lib.orc_density_kernel.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, _vp]
lib.orc_num_threads.restype = C.c_int


class OrcSphArrays(C.Structure):
    _fields_ = [("n", C.c_int64), ("nsph", C.c_int64)] + [(k, _vp) for k in (
        "pos", "mass", "type", "flags", "pi", "hsml", "dthsml", "vel", "treeacc", "gravpm", "bin_grav", "bin_hydro",
        "density", "egywtdensity", "entropy", "dtentropy", "maxsignalvel", "hydroaccel", "dhsmlegydensityfactor",
        "divvel", "curlvel", "delaytime", "bh_density", "bh_divvel")]


lib.orc_set_init_hsml.argtypes = [_vp, C.c_int64, _vp, C.POINTER(OrcSphArrays), C.c_double, C.c_double]
lib.orc_set_init_hsml.restype = None
lib.orc_density.argtypes = [_vp, C.c_int64, _vp, C.POINTER(OrcSphArrays), _vp, C.c_int64, C.POINTER(DensityParams), _vp, _vp,
                            C.POINTER(C.c_int), C.POINTER(C.c_int64)]
lib.orc_update_hmax.argtypes = [_vp, C.c_int64, C.c_int64, C.POINTER(OrcSphArrays)]
lib.orc_update_hmax.restype = None
lib.orc_hydro.argtypes = [_vp, C.c_int64, C.POINTER(OrcSphArrays), _vp, C.c_int64, C.POINTER(HydroParams), _vp, C.POINTER(C.c_int64)]
lib.orc_hydro.restype = None


def stellar_density(nodes, firstnode, st, queue, BoxSize, DesNumNgb, MaxNgbDeviation, SPHWeighting, kernel):
    """stellar_density() (stellar_density2.cpp:306-341): returns (rc, StarVolumeSPH by particle, iterations, candidates)."""
    q = np.ascontiguousarray(queue, dtype=np.int32)
    vol = np.zeros(st.c.n)
    niter, nint = C.c_int(), C.c_int64()
    lib.orc_stellar_density.argtypes = [_vp, C.c_int64, C.POINTER(OrcSphArrays), _vp, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_int,
                                        C.c_int, _vp, C.POINTER(C.c_int), C.POINTER(C.c_int64)]
    rc = lib.orc_stellar_density(ptr(nodes), firstnode, C.byref(st.c), ptr(q), len(q), BoxSize, DesNumNgb, MaxNgbDeviation, int(SPHWeighting),
                                 int(kernel), ptr(vol), C.byref(niter), C.byref(nint))
    return rc, vol, niter.value, nint.value


def bh_veldisp(nodes, firstnode, st, queue, BoxSize, kf):
    """blackhole_veldisp() (veldisp2.cpp:164-199): (sums [nq, 5], vdisp [nq], NaN where the reference leaves VDisp alone)."""
    q = np.ascontiguousarray(queue, dtype=np.int32)
    out = np.zeros((len(q), 5))
    vd = np.full(len(q), np.nan)
    lib.orc_bh_veldisp.argtypes = [_vp, C.c_int64, C.POINTER(OrcSphArrays), _vp, C.c_int64, C.c_double, _vp, _vp, _vp]
    lib.orc_bh_veldisp.restype = None
    lib.orc_bh_veldisp(ptr(nodes), firstnode, C.byref(st.c), ptr(q), len(q), BoxSize, C.byref(kf), ptr(out), ptr(vd))
    return out, vd


def wind_veldisp(nodes, firstnode, st, queue, BoxSize, kf, Time, hubble):
    """winds_find_vel_disp wind part (veldisp2.cpp:203-528): (rc, vdisp [nq] NaN where unset, dmradius [nq], iterations)."""
    q = np.ascontiguousarray(queue, dtype=np.int32)
    vd = np.full(len(q), np.nan)
    dm = np.zeros(len(q))
    niter = C.c_int()
    lib.orc_wind_veldisp.argtypes = [_vp, C.c_int64, C.POINTER(OrcSphArrays), _vp, C.c_int64, C.c_double, _vp, C.c_double, C.c_double, _vp, _vp,
                                     C.POINTER(C.c_int)]
    rc = lib.orc_wind_veldisp(ptr(nodes), firstnode, C.byref(st.c), ptr(q), len(q), BoxSize, C.byref(kf), Time, hubble, ptr(vd), ptr(dm),
                              C.byref(niter))
    return rc, vd, dm, niter.value


def bh_dynfric(nodes, firstnode, st, potential, queue, BoxSize, kf, method, kernel, typemask):
    """bhdynfric.cpp treewalks: raw [nq, 12] results (MinPot, MinPotPos, MinPotVel, density, vel, rmsvel)."""
    q = np.ascontiguousarray(queue, dtype=np.int32)
    pot = np.ascontiguousarray(potential, dtype=np.float64)
    out = np.zeros((len(q), 12))
    lib.orc_bh_dynfric.argtypes = [_vp, C.c_int64, C.POINTER(OrcSphArrays), _vp, _vp, C.c_int64, C.c_double, _vp, C.c_int, C.c_int, C.c_int, _vp]
    lib.orc_bh_dynfric.restype = None
    lib.orc_bh_dynfric(ptr(nodes), firstnode, C.byref(st.c), ptr(pot), ptr(q), len(q), BoxSize, C.byref(kf), int(method), int(kernel), int(typemask),
                       ptr(out))
    return out


class SphState:
    """SoA copy of the particle / slot state the oracle's SPH functions work on."""

    def __init__(self, P, SphP, BhP=None):
        n, ns = len(P), len(SphP)
        f8 = lambda x: np.ascontiguousarray(x, dtype=np.float64)  # noqa: E731
        self.pos, self.mass = f8(P["Pos"]), np.ascontiguousarray(P["Mass"], dtype=np.float32)
        self.type = np.ascontiguousarray(P["Type"], dtype=np.uint8)
        self.flags = np.ascontiguousarray(P["Flags"] & 3, dtype=np.uint8)
        self.pi = np.ascontiguousarray(P["PI"], dtype=np.int32)
        self.hsml, self.dthsml = f8(P["Hsml"]), f8(P["DtHsml"])
        self.vel, self.treeacc, self.gravpm = f8(P["Vel"]), f8(P["FullTreeGravAccel"]), f8(P["GravPM"])
        self.bin_grav = np.ascontiguousarray(P["TimeBinGravity"], dtype=np.uint8)
        self.bin_hydro = np.ascontiguousarray(P["TimeBinHydro"], dtype=np.uint8)
        for k, name in (("density", "Density"), ("egywtdensity", "EgyWtDensity"), ("entropy", "Entropy"),
                        ("dtentropy", "DtEntropy"), ("maxsignalvel", "MaxSignalVel"), ("hydroaccel", "HydroAccel"),
                        ("dhsmlegydensityfactor", "DhsmlEgyDensityFactor"), ("divvel", "DivVel"), ("curlvel", "CurlVel"),
                        ("delaytime", "DelayTime")):
            setattr(self, k, f8(SphP[name]))
        nb = 0 if BhP is None else len(BhP)
        self.bh_density, self.bh_divvel = np.zeros(max(nb, 1)), np.zeros(max(nb, 1))
        self.c = OrcSphArrays(n, ns, *[ptr(getattr(self, k)) for k, _ in OrcSphArrays._fields_[2:]])


def set_init_hsml(nodes, firstnode, father, st, MeanGasSeparation, DesNumNgb):
    lib.orc_set_init_hsml(ptr(nodes), firstnode, ptr(father), C.byref(st.c), MeanGasSeparation, DesNumNgb)


def density(nodes, firstnode, father, st, dp, active=None, want_entvarpred=True, want_gradrho=False):
    evp = np.zeros(max(st.c.nsph, 1)) if want_entvarpred else None
    gr = np.zeros((max(st.c.nsph, 1), 3)) if want_gradrho else None
    niter, nint = C.c_int(0), C.c_int64(0)
    rc = lib.orc_density(ptr(nodes), firstnode, ptr(father), C.byref(st.c), ptr(active), 0 if active is None else len(active),
                         C.byref(dp), ptr(evp), ptr(gr), C.byref(niter), C.byref(nint))
    return rc, evp, gr, niter.value, nint.value


def update_hmax(nodes, firstnode, st):
    lib.orc_update_hmax(ptr(nodes), firstnode, len(nodes), C.byref(st.c))


def hydro(nodes, firstnode, st, hp, evp, active=None):
    nint = C.c_int64(0)
    lib.orc_hydro(ptr(nodes), firstnode, C.byref(st.c), ptr(active), 0 if active is None else len(active), C.byref(hp), ptr(evp), C.byref(nint))
    return nint.value


def tree_build(pos, mass, BoxSize, hsml=None, idx=None, numpart_total=None):
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    mass = np.ascontiguousarray(mass, dtype=np.float32)
    n = len(pos) if idx is None else len(idx)
    ntot = len(pos) if numpart_total is None else numpart_total
    maxnodes = int(2.0 * n) + 4096
    while True:
        nodes = np.zeros(maxnodes, dtype=NODE_DTYPE)
        father = np.full(ntot, -1, dtype=np.int32)
        nn = lib.orc_tree_build(ptr(pos), ptr(mass), ptr(hsml), ptr(idx), n, ntot, BoxSize, ptr(nodes), maxnodes, ptr(father))
        if nn >= 0:
            return nodes[:nn].copy(), ntot, father
        maxnodes *= 2


def grav_walk(nodes, firstnode, pos, mass, oldacc, gp, targets=None):
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    mass = np.ascontiguousarray(mass, dtype=np.float32)
    oldacc = np.ascontiguousarray(oldacc, dtype=np.float64)
    nt = len(pos) if targets is None else len(targets)
    acc = np.zeros((nt, 3))
    pot = np.zeros(nt)
    nint = np.zeros(nt, dtype=np.int64)
    lib.orc_grav_walk(ptr(nodes), firstnode, ptr(pos), ptr(mass), ptr(oldacc), ptr(targets), nt, C.byref(gp), ptr(acc), ptr(pot), ptr(nint))
    return acc, pot, nint


def grav_walk_secondary(nodes, firstnode, pos, mass, qpos, qnodelist, qoldacc, gp):
    """GravLocalTreeWalk::visit<TREEWALK_GHOSTS> (gravshort2.hpp:227-322) for imported queries."""
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    mass = np.ascontiguousarray(mass, dtype=np.float32)
    qpos = np.ascontiguousarray(qpos, dtype=np.float64)
    qnodelist = np.ascontiguousarray(qnodelist, dtype=np.int32)
    qoldacc = np.ascontiguousarray(qoldacc, dtype=np.float64)
    nq = len(qpos)
    acc = np.zeros((nq, 3)); pot = np.zeros(nq); nint = np.zeros(nq, dtype=np.int64)
    lib.orc_grav_walk_secondary(ptr(nodes), firstnode, ptr(pos), ptr(mass), ptr(qpos), ptr(qnodelist), ptr(qoldacc), nq,
                                C.byref(gp), ptr(acc), ptr(pot), ptr(nint))
    return acc, pot, nint


def _toptree(fn_call, ntargets):
    counts = np.zeros(ntargets, dtype=np.int32)
    total = fn_call(ptr(counts), None, 0)
    assert total >= 0
    from shenqi_amd import capi
    table = np.zeros(total, dtype=capi.DATA_INDEX_DTYPE)
    counts2 = np.zeros(ntargets, dtype=np.int32)
    assert fn_call(ptr(counts2), ptr(table), total) == total and np.array_equal(counts, counts2)
    return counts, table


def grav_toptree(nodes, firstnode, lastnode, topleaves, pos, oldacc, gp, targets=None):
    """GravTopTreeWalk::toptree_visit (gravshort2.hpp:362-438): (exports per target, DataIndexTable)."""
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    oldacc = np.ascontiguousarray(oldacc, dtype=np.float64)
    nt = len(pos) if targets is None else len(targets)
    lib.orc_grav_toptree.restype = C.c_int64
    return _toptree(lambda c, t, cap: lib.orc_grav_toptree(ptr(nodes), C.c_int64(firstnode), C.c_int64(lastnode), ptr(topleaves), ptr(pos),
                                                         ptr(oldacc), ptr(targets), C.c_int64(nt), C.byref(gp), c, t, C.c_int64(cap)), nt)


def ngb_toptree(nodes, firstnode, lastnode, topleaves, pos, hsml, symmetric, BoxSize, targets=None):
    """TopTreeWalk::toptree_visit with cull_node (localtreewalk2.h:210-259)."""
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    hsml = np.ascontiguousarray(hsml, dtype=np.float64)
    nt = len(pos) if targets is None else len(targets)
    lib.orc_ngb_toptree.restype = C.c_int64
    return _toptree(lambda c, t, cap: lib.orc_ngb_toptree(ptr(nodes), C.c_int64(firstnode), C.c_int64(lastnode), ptr(topleaves), ptr(pos),
                                                        ptr(hsml), int(symmetric), C.c_double(BoxSize), ptr(targets), C.c_int64(nt), c, t,
                                                        C.c_int64(cap)), nt)


def grav_postprocess(mass, gp, acc, pot, update_potential, targets=None):
    mass = np.ascontiguousarray(mass, dtype=np.float32)
    lib.orc_grav_postprocess(ptr(mass), ptr(targets), len(acc), C.byref(gp), int(update_potential), ptr(acc), ptr(pot))


def apply_accn(dx, r2, mass, gp):
    dx = np.ascontiguousarray(dx, dtype=np.float64)
    acc = np.zeros(3)
    pot = np.zeros(1)
    applied = lib.orc_apply_accn(ptr(dx), r2, mass, C.byref(gp), ptr(acc), ptr(pot))
    return applied, acc, pot[0]


def force_direct(pos, mass, BoxSize, G, h, repeat=1):
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    mass = np.ascontiguousarray(mass, dtype=np.float32)
    acc = np.zeros((len(pos), 3))
    lib.orc_force_direct(ptr(pos), ptr(mass), len(pos), BoxSize, G, h, repeat, ptr(acc))
    return acc


def pm_force(pos, mass, Nmesh, BoxSize, Asmth, G, skip=None, fixed_point_log2scale=-1, use_stencil=0, want_mesh=False):
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    mass = np.ascontiguousarray(mass, dtype=np.float32)
    pm = PMParams(Nmesh, 0, BoxSize, Asmth, G)
    g = np.zeros((len(pos), 3))
    pot = np.zeros(len(pos))
    rho = np.zeros((Nmesh, Nmesh, Nmesh)) if want_mesh else None
    phi = np.zeros((Nmesh, Nmesh, Nmesh)) if want_mesh else None
    lib.orc_pm_force(ptr(pos), ptr(mass), ptr(skip), len(pos), C.byref(pm), fixed_point_log2scale, use_stencil,
                     ptr(g), ptr(pot), ptr(rho), ptr(phi))
    return g, pot, rho, phi


_FFT_FN = C.CFUNCTYPE(None, C.c_int, C.c_void_p, C.c_void_p)
_fft_keep = []


def use_scipy_fft(workers):
    """orc_pm_force's transforms through scipy.fft (pocketfft, `workers` threads) instead of the oracle's own mixed-radix FFT; workers = 0
    restores the oracle's.  Same conventions (unscaled, half spectrum [x][y][z']).  For the CPU baseline of bench.py: the deposit,
    transfer functions and readout stay the oracle's, the five transforms get a production FFT."""
    lib.orc_set_fft.argtypes = [C.c_void_p, C.c_void_p]
    lib.orc_set_fft.restype = None
    if not workers:
        lib.orc_set_fft(None, None)
        del _fft_keep[:]
        return
    import scipy.fft

    def r2c(N, real, cplx):
        a = np.ctypeslib.as_array(C.cast(real, C.POINTER(C.c_double)), shape=(N, N, N))
        out = np.ctypeslib.as_array(C.cast(cplx, C.POINTER(C.c_double)), shape=(N, N, N // 2 + 1, 2)).view(np.complex128)[..., 0]
        out[...] = scipy.fft.rfftn(a, workers=workers)

    def c2r(N, cplx, real):
        a = np.ctypeslib.as_array(C.cast(cplx, C.POINTER(C.c_double)), shape=(N, N, N // 2 + 1, 2)).view(np.complex128)[..., 0]
        out = np.ctypeslib.as_array(C.cast(real, C.POINTER(C.c_double)), shape=(N, N, N))
        out[...] = scipy.fft.irfftn(a, s=(N, N, N), workers=workers, norm="forward")   # "forward": no 1/N^3 on the inverse = unscaled

    fr, fc = _FFT_FN(r2c), _FFT_FN(c2r)
    _fft_keep[:] = [fr, fc]
    lib.orc_set_fft(C.cast(fr, C.c_void_p), C.cast(fc, C.c_void_p))


def fft_r2c(real):
    N = real.shape[0]
    real = np.ascontiguousarray(real, dtype=np.float64)
    out = np.zeros((N, N, N // 2 + 1), dtype=np.complex128)
    lib.orc_fft_r2c(N, ptr(real), ptr(out))
    return out


def fft_c2r(cplx):
    N = cplx.shape[0]
    cplx = np.ascontiguousarray(cplx, dtype=np.complex128)
    out = np.zeros((N, N, N))
    lib.orc_fft_c2r(N, ptr(cplx), ptr(out))
    return out


def density_kernel(ktype, H, u, eta=1.0):
    out = np.zeros(5)
    rc = lib.orc_density_kernel(ktype, H, u, eta, ptr(out))
    assert rc == 0
    return dict(desnumngb=out[0], volume=out[1], wk=out[2], dwk=out[3], dW=out[4])


def boost_mt19937_uniform(seed, n, skip=0):
    """boost::random::mt19937(seed) through boost::random::uniform_real_distribution<double>(0,1):
    one 32-bit draw per variate, value = draw / 2^32 (tests/test_gravity.cpp:318, test_density.cpp:273)."""
    rs = np.random.RandomState()
    # init_genrand(seed), identical to boost/std mt19937(seed)
    mt = np.zeros(624, dtype=np.uint64)
    mt[0] = seed & 0xFFFFFFFF
    for i in range(1, 624):
        mt[i] = (1812433253 * (int(mt[i - 1]) ^ (int(mt[i - 1]) >> 30)) + i) & 0xFFFFFFFF
    rs.set_state(("MT19937", mt.astype(np.uint32), 624))
    raw = rs.randint(0, 2**32, size=skip + n, dtype=np.uint64)  # one tempered 32-bit output each
    return raw[skip:].astype(np.float64) / 4294967296.0


def power_spectrum(rho, Nmesh):
    """numpy restatement of what potential_transfer leaves in pm->ps for the density mesh `rho`
    (measure_power_spectrum / powerspectrum_add_mode, libgadget/gravpm.cpp:323-376, :430; bins as
    powerspectrum_alloc(pm->ps, pm->Nmesh, ...), gravpm.cpp:207): raw sums before powerspectrum_sum.
    Returns kk, power, nmodes (arrays of Nmesh bins) and norm."""
    N = int(Nmesh)
    dk = np.fft.rfftn(rho)                      # unnormalised forward transform, like petapm_fft_r2c
    k1 = np.fft.fftfreq(N, 1.0 / N).astype(np.int64)          # petapm_mesh_to_k: 0..N/2, -(N/2-1)..-1
    k1[N // 2] = N // 2
    kx, ky, kz = np.meshgrid(k1, k1, np.arange(N // 2 + 1, dtype=np.int64), indexing="ij")
    k2 = kx * kx + ky * ky + kz * kz
    m = dk.real**2 + dk.imag**2
    norm = float(m[0, 0, 0])

    def invsinc2(k):
        t = k * np.pi / N
        s = np.where(k == 0, 1.0, np.sin(t) / np.where(t == 0, 1.0, t))
        return 1.0 / (s * s)

    f = invsinc2(kx) * invsinc2(ky) * invsinc2(kz)
    size = N
    binsperunit = (size - 1) / np.log(np.sqrt(3) * N / 2.0)
    sel = k2 > 0
    kint = np.floor(binsperunit * np.log(k2[sel].astype(np.float64)) / 2.0).astype(np.int64)
    ok = kint < size
    w = np.where((kz[sel] == 0) | (kz[sel] == N // 2), 1, 2)[ok]
    kint = kint[ok]
    power = np.bincount(kint, weights=(w * m[sel][ok] * f[sel][ok] ** 2), minlength=size)
    kk = np.bincount(kint, weights=w * np.sqrt(k2[sel][ok].astype(np.float64)), minlength=size)
    nmodes = np.bincount(kint, weights=w, minlength=size).astype(np.int64)
    return kk, power, nmodes, norm
