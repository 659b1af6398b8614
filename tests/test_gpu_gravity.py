"""GPU parity tests of the short-range tree walk and the PM force, through the C-ABI."""
import ctypes as C

import numpy as np
import pytest

import shenqi_amd as sq
from shenqi_amd import capi
import orc
import common as cm

pytestmark = pytest.mark.gpu


def _positions(kind, n=16**3):
    if kind == "grid":
        return cm.grid_positions(round(n ** (1 / 3)))
    if kind == "close":
        return cm.close_positions(round(n ** (1 / 3)))
    if kind == "random":
        return cm.random_positions(orc.boost_mt19937_uniform(0, 3 * n), n)
    if kind == "cluster":
        return sq.synth_positions("cluster", n, L=cm.BOX)
    raise ValueError(kind)


def _gpu_walk(ctx, pman, tree, gp, oldacc_from, active=None, update_potential=True, mode=sq.WALK_EXACT):
    """resident API: upload, set OldAcc inputs, run, download"""
    P = pman.Base
    P["FullTreeGravAccel"] = oldacc_from[0]
    P["GravPM"] = oldacc_from[1]
    pv, tv = pman.view(), tree.view()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    capi.check(capi.hip.shq_tree_upload(ctx.h, C.byref(tv)))
    capi.check(capi.hip.shq_grav_refresh_oldacc(ctx.h, gp.G))
    a = None if active is None else np.ascontiguousarray(active, dtype=np.int32)
    capi.check(capi.hip.shq_grav_short_run(ctx.h, C.byref(gp), capi.ptr(a), 0 if a is None else len(a), int(update_potential), mode))
    n = pman.NumPart
    acc = np.zeros((n, 3)); pot = np.zeros(n); nint = np.zeros(n, dtype=np.int64)
    st = sq.WalkStats()
    capi.check(capi.hip.shq_grav_short_download(ctx.h, capi.ptr(acc), capi.ptr(pot), capi.ptr(nint), C.byref(st)))
    return acc, pot, nint, st


@pytest.mark.parametrize("kind", ["grid", "close", "random"])
@pytest.mark.parametrize("usebh", [1, 0])
def test_walk_exact_parity_16(ctx, kind, usebh):
    """L1 ladder: identical opening decisions => identical interaction counts (integers, exact)
    and forces within the reference's CPU<->GPU bar (runtests.cpp:441-443: max < 1e-5)."""
    pos = _positions(kind)
    n = len(pos)
    pman = cm.make_partmanager(pos)
    tree = sq.force_tree_full(pman)
    cm.reference_treepar(ErrTolForceAcc=0.002, MaxBHOpeningAngle=0.9, TreeUseBH=usebh)
    sq.gravshort_set_softenings(cm.BOX / np.cbrt(n))
    gp = sq.make_grav_params(cm.BOX, 1.5, 48, cm.G, cm.RHO0)
    rng = np.random.default_rng(5)
    told = rng.normal(size=(n, 3)) * 500.0      # arbitrary previous-step accelerations
    pold = rng.normal(size=(n, 3)) * 50.0
    acc, pot, nint, st = _gpu_walk(ctx, pman, tree, gp, (told, pold))
    oldacc = np.linalg.norm(told + pold, axis=1) / cm.G
    mass = pman.Base["Mass"]
    oacc, opot, onint = orc.grav_walk(tree.Nodes_base, tree.firstnode, pos, mass, oldacc, gp)
    orc.grav_postprocess(mass, gp, oacc, opot, True)
    assert np.array_equal(nint, onint)
    assert st.ninteractions == onint.sum() and st.min_interactions == onint.min() and st.max_interactions == onint.max()
    scale = np.abs(oacc).max()
    assert np.abs(acc - oacc).max() < 1e-11 * scale
    if kind != "grid":
        assert cm.force_err(acc, oacc).max() < 1e-5
    assert np.allclose(pot, opot, rtol=1e-10, atol=1e-10 * np.abs(opot).max())


@pytest.mark.parametrize("kind", ["random", "cluster"])
@pytest.mark.parametrize("usebh", [1, 0])
def test_walk_launch_modes_agree_and_match_the_oracle(ctx, kind, usebh):
    """The persistent launch (waves taking 64-target tasks from per-XCD counters) and its leaf ring (leaf particles evaluated out of a
    wave-private LDS ring, shq_set_walk_launch) against the one-task-per-wave launch and the oracle: interaction counts identical as
    integers in every mode; persistent without the ring is bit-identical to the plain launch (same order of the same operations);
    with the ring a target's leaf particles are summed after, not between, its node interactions: forces to 1e-13 of the largest,
    and within the same 1e-11 of the oracle as every other mode.  Ragged sizes: a last wave with idle lanes, fewer tasks than waves."""
    n = 20**3 + 37
    pos = _positions(kind, 20**3 + 37) if kind == "cluster" else cm.random_positions(orc.boost_mt19937_uniform(1, 3 * n), n)
    pman = cm.make_partmanager(pos)
    tree = sq.force_tree_full(pman)
    cm.reference_treepar(ErrTolForceAcc=0.005, MaxBHOpeningAngle=0.9, TreeUseBH=usebh)
    sq.gravshort_set_softenings(cm.BOX / 20)
    gp = sq.make_grav_params(cm.BOX, 1.5, 60, cm.G, cm.RHO0)
    rng = np.random.default_rng(9)
    told, pold = rng.normal(size=(n, 3)) * 500.0, rng.normal(size=(n, 3)) * 50.0
    mass = pman.Base["Mass"]
    oacc, opot, onint = orc.grav_walk(tree.Nodes_base, tree.firstnode, pos, mass, np.linalg.norm(told + pold, axis=1) / cm.G, gp)
    orc.grav_postprocess(mass, gp, oacc, opot, True)
    res = {}
    try:
        for mode in ((0, 0), (2, 0), (2, 1)):
            capi.check(capi.hip.shq_set_walk_launch(ctx.h, *mode))
            res[mode] = _gpu_walk(ctx, pman, tree, gp, (told, pold))
        sub = np.arange(5, n, 7, dtype=np.int32)        # an active list through the ring launch, no potential
        capi.check(capi.hip.shq_set_walk_launch(ctx.h, 2, 1))
        asub = _gpu_walk(ctx, pman, tree, gp, (told, pold), active=sub, update_potential=False)
        # the ring launch hands sparse subtrees to the pair kernel by default: the same launch entering every subtree itself
        capi.check(capi.hip.shq_set_walk_sparse(ctx.h, 0))
        res[(2, 1, "no pair kernel")] = _gpu_walk(ctx, pman, tree, gp, (told, pold))
        # the pair kernel fetches 80 of a record's 128 bytes and recomputes the rest (the pool passed the device's check that this
        # reproduces every record); 2 = whole records: the same decisions and the same bits
        lean = capi.hip.shq_walk_pair_lean(ctx.h)
        capi.check(capi.hip.shq_set_walk_sparse(ctx.h, 2))
        res[(2, 1, "whole records")] = _gpu_walk(ctx, pman, tree, gp, (told, pold))
        # the pair kernel beside the main walk (second stream, a flag per task) and behind it: the same sums in the same order
        capi.check(capi.hip.shq_set_walk_sparse(ctx.h, 1))
        capi.check(capi.hip.shq_set_walk_overlap(ctx.h, 2))
        res[(2, 1, "beside")] = _gpu_walk(ctx, pman, tree, gp, (told, pold))
        capi.check(capi.hip.shq_set_walk_overlap(ctx.h, 0))
        res[(2, 1, "behind")] = _gpu_walk(ctx, pman, tree, gp, (told, pold))
    finally:
        capi.check(capi.hip.shq_set_walk_overlap(ctx.h, 1))
        capi.check(capi.hip.shq_set_walk_launch(ctx.h, 1, 1))
        capi.check(capi.hip.shq_set_walk_sparse(ctx.h, 1))
    scale = np.abs(oacc).max()
    for mode, (acc, pot, nint, st) in res.items():
        assert np.array_equal(nint, onint), mode
        assert st.ninteractions == onint.sum() and st.min_interactions == onint.min() and st.max_interactions == onint.max()
        assert np.abs(acc - oacc).max() < 1e-11 * scale, mode
        assert np.allclose(pot, opot, rtol=1e-10, atol=1e-10 * np.abs(opot).max()), mode
    assert np.array_equal(res[(0, 0)][0], res[(2, 0)][0]) and np.array_equal(res[(0, 0)][1], res[(2, 0)][1])
    assert np.abs(res[(2, 1)][0] - res[(0, 0)][0]).max() < 1e-13 * scale
    assert np.abs(res[(2, 1, "no pair kernel")][0] - res[(0, 0)][0]).max() < 1e-13 * scale
    assert lean == 1
    for k in range(3):
        assert np.array_equal(res[(2, 1)][k], res[(2, 1, "whole records")][k]), k
        assert np.array_equal(res[(2, 1, "beside")][k], res[(2, 1, "behind")][k]), k
    assert np.array_equal(asub[2][sub], onint[sub])
    assert np.abs(asub[0][sub] - oacc[sub]).max() < 1e-11 * scale


def test_walk_exact_parity_64_cluster(ctx):
    """S-cluster 64^3, Nmesh 192, relative criterion at ErrTolForceAcc 0.005 (north-star setting)."""
    n = 64**3
    L = 1.0
    pos = sq.synth_positions("cluster", n, L=L)
    pos = pos[sq.morton_order(pos, L)]
    pman = cm.make_partmanager(pos, box=L)
    tree = sq.force_tree_full(pman)
    sq.set_gravshort_treepar(ErrTolForceAcc=0.005, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=1, Rcut=6.0)
    sq.gravshort_set_softenings(L / 64)
    gp = sq.make_grav_params(L, 1.5, 192, cm.G, cm.RHO0)
    mass = pman.Base["Mass"]
    z = np.zeros((n, 3))
    acc1, _, nint1, st1 = _gpu_walk(ctx, pman, tree, gp, (z, z))                    # BH seeding pass
    o1, op1, on1 = orc.grav_walk(tree.Nodes_base, tree.firstnode, pos, mass, np.zeros(n), gp)
    orc.grav_postprocess(mass, gp, o1, op1, True)
    assert np.array_equal(nint1, on1)
    sq.set_gravshort_treepar(ErrTolForceAcc=0.005, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=0, Rcut=6.0)
    gp = sq.make_grav_params(L, 1.5, 192, cm.G, cm.RHO0)
    acc2, pot2, nint2, st2 = _gpu_walk(ctx, pman, tree, gp, (o1, z))
    o2, op2, on2 = orc.grav_walk(tree.Nodes_base, tree.firstnode, pos, mass, np.linalg.norm(o1, axis=1) / cm.G, gp)
    orc.grav_postprocess(mass, gp, o2, op2, True)
    assert np.array_equal(nint2, on2)
    err = cm.force_err(acc2, o2)
    assert err.max() < 1e-5
    rms = np.sqrt(np.mean(np.sum((acc2 - o2) ** 2, axis=1) / np.sum(o2 ** 2, axis=1)))
    assert rms < 1e-3          # north-star tolerance (BASELINE.json); in practice ~1e-14
    print("64^3 cluster: interactions/target mean %.1f min %d max %d, visited/wave %.0f, kernel %.2f ms, rms %.2e"
          % (on2.mean(), on2.min(), on2.max(), st2.nnodes_visited / (n / 64), st2.kernel_ms, rms))


def test_walk_active_subset_and_edges(ctx):
    """ragged / empty active lists; accel rows of inactive particles untouched (reduce<PRIMARY>
    assigns only walked targets, localtreewalk2.h:39)."""
    pos = _positions("random")
    n = len(pos)
    pman = cm.make_partmanager(pos)
    tree = sq.force_tree_full(pman)
    cm.reference_treepar(MaxBHOpeningAngle=0.9, TreeUseBH=1)
    sq.gravshort_set_softenings(cm.BOX / np.cbrt(n))
    pm = dict(Asmth=1.5, Nmesh=48, G=cm.G)
    gp = sq.make_grav_params(cm.BOX, 1.5, 48, cm.G, cm.RHO0)
    mass = pman.Base["Mass"]
    full, _, _ = orc.grav_walk(tree.Nodes_base, tree.firstnode, pos, mass, np.zeros(n), gp)
    full *= cm.G
    for act in (np.arange(0, n, 7, dtype=np.int32), np.array([n - 1], dtype=np.int32), np.array([], dtype=np.int32),
                np.arange(0, 65, dtype=np.int32)):
        store = np.full((n, 3), 777.0)
        tree.full_particle_tree_flag = 1
        st = sq.grav_short_tree(ctx, act, pm, tree, store, cm.RHO0)
        assert st.ntargets == len(act)
        mask = np.zeros(n, dtype=bool)
        mask[act] = True
        assert np.all(store[~mask] == 777.0)
        if len(act):
            assert np.abs(store[mask] - full[mask]).max() < 1e-11 * np.abs(full).max()
        cm.reference_treepar(MaxBHOpeningAngle=0.9, TreeUseBH=1)


def test_error_behaviour(ctx):
    pos = _positions("grid")
    pman = cm.make_partmanager(pos)
    tree = sq.force_tree_full(pman)
    cm.reference_treepar()
    sq.gravshort_set_softenings(0.5)
    pm = dict(Asmth=1.5, Nmesh=48, G=cm.G)
    with pytest.raises(sq.ShqError):           # no CPU fallback in the product
        sq.grav_short_tree(ctx, None, pm, tree, None, cm.RHO0, UseGPU=False)
    with pytest.raises(sq.ShqError):
        sq.gravpm_force(ctx, pm, pman, UseGPU=False)
    with pytest.raises(sq.ShqError):           # exact window needs Asmth 1.5
        sq.grav_short_tree(ctx, None, dict(Asmth=1.25, Nmesh=48, G=cm.G), tree, None, cm.RHO0)
    gas_tree = sq.force_tree_rebuild_mask(pman, sq.GASMASK)
    with pytest.raises(sq.ShqError):           # gravshort2.hpp:208-210 mask check
        sq.grav_short_tree(ctx, None, pm, gas_tree, None, cm.RHO0)
    c2 = sq.Context(0)
    gp = sq.make_grav_params(cm.BOX, 1.5, 48, cm.G, cm.RHO0)
    rc = capi.hip.shq_grav_short_run(c2.h, C.byref(gp), None, 0, 1, 0)
    assert rc == 4 and b"upload" in capi.hip.shq_last_error()      # SHQ_ERR_STATE
    c2.close()


@pytest.mark.parametrize("kind", ["random", "close"])
def test_pm_parity_16(ctx, kind):
    """PM ladder: deposit mesh bit-exact against the fixed-point oracle, GravPM / potential
    within FFT-library rounding of the reference-structure oracle (5 FFTs, k-space differencing)."""
    pos = _positions(kind)
    n = len(pos)
    pman = cm.make_partmanager(pos)
    pv = pman.view()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    capi.check(capi.hip.shq_pm_set_debug(ctx.h, 1))
    pmp = sq.PMParams(48, 0, cm.BOX, 1.5, cm.G)
    capi.check(capi.hip.shq_pm_run(ctx.h, C.byref(pmp)))
    g = np.zeros((n, 3)); ppot = np.zeros(n)
    capi.check(capi.hip.shq_pm_download(ctx.h, capi.ptr(g), capi.ptr(ppot)))
    rho = np.zeros((48, 48, 48)); phi = np.zeros((48, 48, 48))
    capi.check(capi.hip.shq_pm_download_mesh(ctx.h, 0, capi.ptr(rho)))
    capi.check(capi.hip.shq_pm_download_mesh(ctx.h, 1, capi.ptr(phi)))
    capi.check(capi.hip.shq_pm_set_debug(ctx.h, 0))
    mass = pman.Base["Mass"]
    e = 61 - int(np.frexp(float(n))[1])       # the library's scale rule: 2^(61 - ex), msum < 2^ex
    og, opot, orho, ophi = orc.pm_force(pos, mass, 48, cm.BOX, 1.5, cm.G, fixed_point_log2scale=e, use_stencil=0, want_mesh=True)
    nbad = int((rho != orho).sum())
    assert nbad == 0, (nbad, float(np.abs(rho - orho).max()), float(rho.sum()), float(orho.sum()),
                       np.argwhere(rho != orho)[:4].tolist())    # integer deposit: bit-exact
    assert abs(rho.sum() - n) < 1e-6
    assert np.abs(phi - ophi).max() < 1e-11 * np.abs(ophi).max()
    assert np.abs(g - og).max() < 1e-10 * np.abs(og).max()
    assert np.abs(ppot - opot).max() < 1e-10 * np.abs(opot).max()
    # against the plain-f64 reference-structure oracle the only extra difference is the 2^-e quantum
    og2, _, _, _ = orc.pm_force(pos, mass, 48, cm.BOX, 1.5, cm.G)
    assert np.abs(g - og2).max() < 1e-9 * np.abs(og2).max()
    # bit-pattern reproducibility per FFT plan: same inputs, same plan => identical bits
    capi.check(capi.hip.shq_pm_run(ctx.h, C.byref(pmp)))
    g2 = np.zeros((n, 3)); p2 = np.zeros(n)
    capi.check(capi.hip.shq_pm_download(ctx.h, capi.ptr(g2), capi.ptr(p2)))
    assert np.array_equal(g, g2) and np.array_equal(ppot, p2)
    # ... and for any particle order (order-independent integer deposit)
    perm = np.random.default_rng(1).permutation(n)
    pman2 = cm.make_partmanager(pos[perm])
    pv2 = pman2.view()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv2)))
    capi.check(capi.hip.shq_pm_run(ctx.h, C.byref(pmp)))
    g3 = np.zeros((n, 3)); p3 = np.zeros(n)
    capi.check(capi.hip.shq_pm_download(ctx.h, capi.ptr(g3), capi.ptr(p3)))
    assert np.array_equal(g3, g[perm]) and np.array_equal(p3, ppot[perm])


@pytest.mark.parametrize("nmesh", [16, 24, 40, 48, 96, 192, 384, 512, 1024])
def test_pm_transposing_fft_pipeline_gives_the_in_place_pipelines_bits(ctx, nmesh):
    """The five FFT passes of the undivided PM as the transposing pipeline (mesh <-> scratch mesh, contiguous tiles on one side of
    every pass: fft3d.hip) against the in-place pipeline on the same deposit: potential mesh, GravPM and the PM potential bit for bit
    ("bit-pattern-reproducible per FFT plan": the two are the same plan, addressed differently); radix 16 / 4 / 2 / 3 / 5 stage
    mixes, meshes whose z pitch holds pad columns (16: Nc = 9 in 12) and ones without (24: Nc = 13 in 16)."""
    big = nmesh > 400                  # 512: the largest mesh on 256-thread workgroups; 1024 (C4's): 512-thread workgroups, > 2^32 bytes
    n = 64**3 + 11 if big else 20**3 + 11
    pos = cm.random_positions(orc.boost_mt19937_uniform(3, 3 * n), n)
    pman = cm.make_partmanager(pos)
    pv = pman.view()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    pmp = sq.PMParams(nmesh, 0, cm.BOX, 1.5, cm.G)
    got = {}
    try:
        capi.check(capi.hip.shq_pm_set_debug(ctx.h, 0 if big else 1))
        for mode in (0, 1, 1):                       # the second transposing run starts from the scratch mesh the first one left
            capi.check(capi.hip.shq_pm_set_fft_transposed(ctx.h, mode))
            capi.check(capi.hip.shq_pm_run(ctx.h, C.byref(pmp)))
            g = np.zeros((n, 3)); pp = np.zeros(n); phi = np.zeros((1 if big else nmesh,) * 3)
            capi.check(capi.hip.shq_pm_download(ctx.h, capi.ptr(g), capi.ptr(pp)))
            if not big:                # (the dense mesh copy of a big mesh is gigabytes: there the 2.6e5 particles' readouts sample it)
                capi.check(capi.hip.shq_pm_download_mesh(ctx.h, 1, capi.ptr(phi)))
            if mode in got:
                for a, b in zip(got[mode], (g, pp, phi)):
                    assert np.array_equal(a, b)
            got[mode] = (g, pp, phi)
    finally:
        capi.check(capi.hip.shq_pm_set_debug(ctx.h, 0))
        capi.check(capi.hip.shq_pm_set_fft_transposed(ctx.h, 1))
    for a, b, name in zip(got[0], got[1], ("GravPM", "PM potential", "potential mesh")):
        assert np.array_equal(a, b), (name, float(np.abs(a - b).max()))
    if not big:
        og, opot, _, _ = orc.pm_force(pos, pman.Base["Mass"], nmesh, cm.BOX, 1.5, cm.G)
        assert np.abs(got[1][0] - og).max() < 1e-9 * np.abs(og).max()
    else:
        assert np.all(np.isfinite(got[1][0])) and np.abs(got[1][0]).max() > 0


@pytest.mark.parametrize("N", [24, 40, 64, 80, 18])
def test_fft_dropins(ctx, N):
    """petapm_fft_r2c / petapm_fft_c2r drop-ins: unscaled, spectrum in the reference's Fourier layout [y][z'][x]
    (petapm.cpp:262-270), round trip = N^3 * identity.  24, 64: radix 16/4/2/3 stages of the bespoke pipeline; 40, 80:
    radix 5; 18: no bespoke transform (rocFFT).  The _xyz pair keeps [x][y][z']."""
    a = np.random.default_rng(2).normal(size=(N, N, N))
    ref = orc.fft_r2c(a)                                     # [x][y][z']
    out = np.zeros((N, N // 2 + 1, N), dtype=np.complex128)  # [y][z'][x]
    capi.check(capi.hip.shq_fft_r2c(ctx.h, N, capi.ptr(a), capi.ptr(out)))
    assert np.abs(out - ref.transpose(1, 2, 0)).max() < 1e-12 * np.abs(ref).max()
    back = np.zeros((N, N, N))
    capi.check(capi.hip.shq_fft_c2r(ctx.h, N, capi.ptr(out), capi.ptr(back)))
    assert np.abs(back - a * N**3).max() < 1e-11 * N**3
    out2 = np.zeros((N, N, N // 2 + 1), dtype=np.complex128)
    capi.check(capi.hip.shq_fft_r2c_xyz(ctx.h, N, capi.ptr(a), capi.ptr(out2)))
    assert np.abs(out2 - ref).max() < 1e-12 * np.abs(ref).max()
    capi.check(capi.hip.shq_fft_c2r_xyz(ctx.h, N, capi.ptr(out2), capi.ptr(back)))
    assert np.abs(back - a * N**3).max() < 1e-11 * N**3


@pytest.mark.parametrize("N", [24, 48, 18])
def test_pm_apply_other_petapm_clients(ctx, N):
    """shq_pm_apply = pm_apply_transfer_function + petapm_fft_c2r for the transfer functions of the other petapm clients, against the
    reference's mode enumeration restated in numpy on its Fourier layout [y][z'][x] (petapm.cpp:1258-1298: kpos = mesh_to_k of
    (y, z, x), handed to the transfer function as (kx, ky, kz)) and the oracle's c2r: the density and displacement transfers of
    libgenic/zeldovich.cpp:271-321 with a made-up DeltaSpec(k), the k2 = 0 mode left alone; the lensing planes' neutrino correction
    (plane.cpp:283-304: a radial factor, k2 = 0 set to zero); force_y_transfer (gravpm.cpp:464-488).  24, 48: bespoke transforms;
    18: rocFFT."""
    L = 7.0
    a = np.random.default_rng(4).normal(size=(N, N, N))
    spec = np.zeros((N, N // 2 + 1, N), dtype=np.complex128)           # [y][z'][x]
    capi.check(capi.hip.shq_fft_r2c(ctx.h, N, capi.ptr(a), capi.ptr(spec)))
    k1 = np.where(np.arange(N) <= N // 2, np.arange(N), np.arange(N) - N)
    ky, kz, kx = np.meshgrid(k1, k1[: N // 2 + 1], k1, indexing="ij")    # the enumeration's pos[] = (y, z, x) -> kpos = (kx, ky, kz)
    k2 = (kx.astype(np.int64) ** 2 + ky.astype(np.int64) ** 2 + kz.astype(np.int64) ** 2)
    nz = k2 > 0
    k2i = np.arange(3 * (N // 2) ** 2 + 1)
    kmag_i = np.sqrt(k2i) * 2 * np.pi / L
    delta = lambda k: 3.0 * k / (1.0 + (k / 2.0) ** 3)                   # a stand-in for curpower->DeltaSpec

    def check(tf_kind, axis, zero_mode, table, expect_spec, tol=1e-11):
        tab = np.ascontiguousarray(table, dtype=np.float64)
        tf = capi.PMTransfer(tf_kind, axis, zero_mode, 0, tab.ctypes.data)
        out = np.zeros((N, N, N))
        capi.check(capi.hip.shq_pm_apply(ctx.h, N, capi.ptr(spec), C.byref(tf), capi.ptr(out)))
        ref = orc.fft_c2r(np.ascontiguousarray(expect_spec.transpose(2, 0, 1)))    # [y][z'][x] -> [x][y][z']
        assert np.abs(out - ref).max() < tol * np.abs(ref).max(), (tf_kind, axis)

    # density_transfer: exp(-k2 / N^2) DeltaSpec(kmag) / sqrt(V), k2 = 0 untouched
    T = np.exp(-k2i / N**2) * delta(kmag_i) / np.sqrt(L**3)
    e = spec.copy()
    e[nz] *= T[k2[nz]]
    check(0, 0, 0, T, e)
    # disp_y_transfer: fac = 1 / (2 pi) / sqrt(L) * ky / k2 * DeltaSpec; value -> (-im fac, re fac) = i fac value
    with np.errstate(divide="ignore", invalid="ignore"):
        T = np.where(k2i > 0, 1.0 / (2 * np.pi) / np.sqrt(L) / np.maximum(k2i, 1) * delta(kmag_i), 0.0)
    e = spec.copy()
    e[nz] = 1j * (T[k2[nz]] * ky[nz]) * spec[nz]
    check(1, 1, 0, T, e)
    # plane_neutrino_correction_transfer: value *= nufac(log k), zero mode zeroed
    T = 0.05 * np.tanh(np.log(np.maximum(kmag_i, 1e-30)))
    e = spec * T[k2]
    e[~nz] = 0
    check(0, 0, 1, T, e)
    # uvbg.cpp:211-250: the real-space top-hat filter in k R (untouched below k R = 1e-4, so also at k2 = 0) times divide_by_ncell on EVERY mode
    R = 0.9
    kR = np.sqrt(k2i) * 2 * np.pi / L * R
    with np.errstate(divide="ignore", invalid="ignore"):
        T = np.where(kR > 1e-4, 3.0 * (np.sin(kR) / kR**3 - np.cos(kR) / kR**2), 1.0) / N**3
    check(0, 0, 2, T, spec * T[k2])
    # force_z_transfer: fac = -diff_kernel(kz 2 pi / N) N / L, value -> i fac value (every mode; k2 = 0 has kz = 0: fac = 0 either way)
    T = np.full(len(k2i), -(N / L))
    w = kz * (2 * np.pi / N)
    e = 1j * (T[k2] * (1 / 6.0 * (8 * np.sin(w) - np.sin(2 * w)))) * spec
    check(2, 2, 1, T, e)


def test_fft_seam_feeds_the_reference_transfer_function(ctx):
    """what a shenqi build would do with the drop-ins (petapm.cpp:403-452): r2c of the density mesh, potential_transfer applied
    to the [y][z'][x] spectrum exactly as pm_apply_transfer_function enumerates it (:1258-1298; restated in numpy on that
    layout in tests/cpu_ops.py), c2r: the potential mesh of the oracle's PM"""
    from cpu_ops import CpuOps
    import torch
    N, n = 48, 16**3
    pos = cm.random_positions(orc.boost_mt19937_uniform(0, 3 * n), n)
    mass = np.ones(n, dtype=np.float32)
    _, _, rho, phi = orc.pm_force(pos, mass, N, cm.BOX, 1.5, cm.G, want_mesh=True)
    spec = np.zeros((N, N // 2 + 1, N), dtype=np.complex128)
    capi.check(capi.hip.shq_fft_r2c(ctx.h, N, capi.ptr(np.ascontiguousarray(rho)), capi.ptr(spec)))
    t = torch.from_numpy(spec)
    CpuOps(N, cm.BOX, 1.5, cm.G).green(t, 0, N)              # in place, all y rows
    pot = np.zeros((N, N, N))
    capi.check(capi.hip.shq_fft_c2r(ctx.h, N, capi.ptr(np.ascontiguousarray(t.numpy())), capi.ptr(pot)))
    assert np.abs(pot - phi).max() < 1e-11 * np.abs(phi).max()


def _do_force_test(ctx, pos, Nmesh=48, ErrTol=0.002, direct=True):
    """do_force_test, tests/test_gravity.cpp:197-247, through the host mirror of the reference API."""
    n = len(pos)
    pman = cm.make_partmanager(pos)
    pm = dict(Asmth=1.5, Nmesh=Nmesh, G=cm.G)
    sq.gravpm_force(ctx, pm, pman)
    tree = sq.force_tree_full(pman)
    cm.reference_treepar(ErrTolForceAcc=ErrTol)                 # TreeUseBH = 2, MaxBHOpeningAngle = 0
    sq.gravshort_set_softenings(cm.BOX / np.cbrt(n))
    sq.grav_short_tree(ctx, None, pm, tree, None, cm.RHO0)
    assert sq.get_TreeUseBH() == 0                              # gravshort-tree2.cpp:168-171
    st = sq.grav_short_tree(ctx, None, pm, tree, None, cm.RHO0)
    P = pman.Base
    total = P["GravPM"] + P["FullTreeGravAccel"]
    if not direct:
        return total, st
    pair = orc.force_direct(pos, P["Mass"], cm.BOX, cm.G, sq.FORCE_SOFTENING(), 1)
    return cm.check_accns(pair, total), st


def test_reference_gate_force_flat_gpu(ctx):
    total, st = _do_force_test(ctx, cm.grid_positions(16), direct=False)
    assert np.abs(total).max() < 0.015          # tests/test_gravity.cpp:288-289
    assert np.abs(total).mean() < 0.005


@pytest.mark.parametrize("kind", ["close", "random0", "random1"])
def test_reference_gate_force_vs_direct_gpu(ctx, kind):
    n = 16**3
    if kind == "close":
        pos = cm.close_positions(16)
    else:
        k = int(kind[-1])
        pos = cm.random_positions(orc.boost_mt19937_uniform(0, 3 * n, skip=3 * n * k), n)
    (meanerr, maxerr), st = _do_force_test(ctx, pos)
    assert maxerr < 3 * 0.002                   # tests/test_gravity.cpp:165-166
    assert meanerr < 0.8 * 0.002


@pytest.mark.parametrize("nmesh", [48, 36])
def test_pm_power_spectrum(ctx, nmesh):
    """P(k) accumulated during the PM run (powerspectrum_add_mode, gravpm.cpp:323-356) against the numpy
    restatement applied to the oracle's density mesh; also for a mesh size without a bespoke FFT (36: rocFFT
    path).  Forces must not depend on whether the spectrum is measured."""
    n = 16**3
    pos = sq.synth_positions("cluster", n, L=cm.BOX)
    pman = cm.make_partmanager(pos)
    mass = pman.Base["Mass"]
    pmp = sq.PMParams(nmesh, 0, cm.BOX, 1.5, cm.G)
    pv = pman.view()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    capi.check(capi.hip.shq_pm_run(ctx.h, C.byref(pmp)))
    g0 = np.zeros((n, 3)); p0 = np.zeros(n)
    capi.check(capi.hip.shq_pm_download(ctx.h, capi.ptr(g0), capi.ptr(p0)))
    capi.check(capi.hip.shq_pm_measure_power(ctx.h, 1))
    capi.check(capi.hip.shq_pm_run(ctx.h, C.byref(pmp)))
    g1 = np.zeros((n, 3)); p1 = np.zeros(n)
    capi.check(capi.hip.shq_pm_download(ctx.h, capi.ptr(g1), capi.ptr(p1)))
    kk = np.zeros(nmesh); power = np.zeros(nmesh); nmodes = np.zeros(nmesh, dtype=np.int64); norm = C.c_double()
    capi.check(capi.hip.shq_pm_download_power(ctx.h, nmesh, capi.ptr(kk), capi.ptr(power), capi.ptr(nmodes), C.byref(norm)))
    capi.check(capi.hip.shq_pm_measure_power(ctx.h, 0))
    _, _, rho, _ = orc.pm_force(pos, mass, nmesh, cm.BOX, 1.5, cm.G, want_mesh=True)
    okk, opower, onmodes, onorm = orc.power_spectrum(rho, nmesh)
    assert np.array_equal(nmodes, onmodes) and nmodes.sum() > 0
    assert abs(norm.value - onorm) < 1e-10 * onorm and abs(onorm - float(mass.sum()) ** 2) < 1e-9 * onorm
    assert np.abs(kk - okk).max() < 1e-10 * okk.max()
    assert np.abs(power - opower).max() < 1e-9 * opower.max()
    # the separate forward / Green / inverse passes give the forces of the fused pipeline
    assert np.abs(g1 - g0).max() < 1e-12 * np.abs(g0).max() and np.abs(p1 - p0).max() < 1e-12 * np.abs(p0).max()


def test_secondary_walk_matches_oracle(ctx):
    """shq_grav_short_secondary = visit<TREEWALK_GHOSTS> (gravshort2.hpp:243-322): imported queries (positions
    that are not local particles) walk the branches under the top-level nodes of their NodeList.  The host tree
    gets a two-level top tree (root, its children and grandchildren flagged TopLevel, as a domain
    decomposition would), queries list 1-4 of the deepest top-level nodes in arbitrary order."""
    n = 20**3
    pos = sq.synth_positions("cluster", n, L=cm.BOX)
    pman = cm.make_partmanager(pos)
    tree = sq.force_tree_full(pman)
    nodes = tree.Nodes_base          # writable view of the host tree
    fn = tree.firstnode
    ctype = (nodes["flags"] >> 3) & 3
    root = nodes[0]
    lvl1 = [s for s in root["suns"] if s >= 0] if ctype[0] == 1 else []
    top = []
    for c in lvl1:
        nodes["flags"][c - fn] |= 2
        if ctype[c - fn] == 1:
            for g in nodes["suns"][c - fn]:
                if g >= 0:
                    nodes["flags"][g - fn] |= 2
                    top.append(int(g))
        else:
            top.append(int(c))
    assert len(top) >= 8
    cm.reference_treepar(ErrTolForceAcc=0.005, MaxBHOpeningAngle=0.9, Rcut=6.0, TreeUseBH=0)
    sq.gravshort_set_softenings(cm.BOX / 20)
    gp = sq.make_grav_params(cm.BOX, 1.5, 60, cm.G, cm.RHO0)
    rng = np.random.default_rng(12)
    nq = 3000
    q = np.zeros(nq, dtype=capi.GRAV_QUERY_DTYPE)
    q["Pos"] = rng.random((nq, 3)) * cm.BOX
    q["OldAcc"] = 10 ** rng.uniform(1, 4, size=nq)
    q["NodeList"] = -1
    for i in range(nq):
        k = rng.integers(1, 5)
        q["NodeList"][i, :k] = rng.choice(top, size=k, replace=False)
    pv, tv = pman.view(), tree.view()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    capi.check(capi.hip.shq_tree_upload(ctx.h, C.byref(tv)))
    res = np.zeros(nq, dtype=capi.GRAV_RESULT_DTYPE)
    nint = np.zeros(nq, dtype=np.int64)
    capi.check(capi.hip.shq_grav_short_secondary(ctx.h, C.byref(gp), capi.ptr(q), nq, capi.ptr(res), capi.ptr(nint), 1))
    # the reference visits the NodeList in the given order; the branches are disjoint, so only the summation
    # order can differ: compare with the oracle run on the list sorted the way the device walks it
    rank = {no: i for i, no in enumerate(cm_preorder(nodes, fn))}
    qs = q["NodeList"].copy()
    for i in range(nq):
        lst = sorted([x for x in qs[i] if x >= 0], key=lambda no: rank[no])
        qs[i] = lst + [-1] * (4 - len(lst))
    oacc, opot, onint = orc.grav_walk_secondary(nodes, fn, pos, pman.Base["Mass"], q["Pos"], qs, q["OldAcc"], gp)
    assert np.array_equal(nint, onint) and nint.max() > 0
    scale = np.abs(oacc).max()
    assert np.abs(res["Acc"] - oacc).max() < 1e-11 * scale
    assert np.allclose(res["Potential"], opot, rtol=1e-10, atol=1e-10 * np.abs(opot).max())


def cm_preorder(nodes, firstnode):
    out = []
    no = firstnode
    while firstnode <= no < firstnode + len(nodes):
        out.append(int(no))
        nd = nodes[no - firstnode]
        no = nd["suns"][0] if ((nd["flags"] >> 3) & 3) == 1 else nd["sibling"]
    return out


def test_walk_in_ranges_equals_one_launch(ctx):
    """shq_grav_short_run_range: the walk cut into ragged pieces (not multiples of the wave size, an empty one) gives the
    bits of the single launch; the interaction statistics of the pieces add up; bad ranges are refused."""
    pos = _positions("random")
    n = len(pos)
    pman = cm.make_partmanager(pos)
    tree = sq.force_tree_full(pman)
    cm.reference_treepar(ErrTolForceAcc=0.005, MaxBHOpeningAngle=0.9, TreeUseBH=0)
    sq.gravshort_set_softenings(cm.BOX / np.cbrt(n))
    gp = sq.make_grav_params(cm.BOX, 1.5, 48, cm.G, cm.RHO0)
    rng = np.random.default_rng(11)
    old = (rng.normal(size=(n, 3)), rng.normal(size=(n, 3)))
    acc0, pot0, nint0, st0 = _gpu_walk(ctx, pman, tree, gp, old)
    cuts = [0, 100, 100, 1357, 2048, n]
    for first, last in zip(cuts[:-1], cuts[1:]):
        capi.check(capi.hip.shq_grav_short_run_range(ctx.h, C.byref(gp), first, last - first, 1, sq.WALK_EXACT))
    acc = np.zeros((n, 3)); pot = np.zeros(n); nint = np.zeros(n, dtype=np.int64)
    st = sq.WalkStats()
    capi.check(capi.hip.shq_grav_short_download(ctx.h, capi.ptr(acc), capi.ptr(pot), capi.ptr(nint), C.byref(st)))
    assert np.array_equal(acc, acc0) and np.array_equal(pot, pot0) and np.array_equal(nint, nint0)
    assert st.ninteractions == st0.ninteractions == nint0.sum() and st.ntargets == n
    assert (st.min_interactions, st.max_interactions) == (st0.min_interactions, st0.max_interactions)
    assert capi.hip.shq_grav_short_run_range(ctx.h, C.byref(gp), n - 10, 11, 1, sq.WALK_EXACT) != 0
    assert capi.hip.shq_grav_short_run_range(ctx.h, C.byref(gp), -1, 5, 1, sq.WALK_EXACT) != 0


def test_reference_runtests_force_gates_gpu(ctx):
    """runtests.cpp:289-352 through the device path (host mirror of the reference API): default tree vs open tree (mean <= 1.2
    ErrTol, :314), a larger Rcut stays within ErrTol (:330), half the mesh is not more accurate (:351)."""
    from test_oracle_cpu import _runtests_sequence, check_runtests_gates
    n = 16**3
    pos = cm.random_positions(orc.boost_mt19937_uniform(0, 3 * n), n)
    pman = cm.make_partmanager(pos)
    tree = sq.force_tree_full(pman)
    P = pman.Base

    def pmforce(nmesh):
        sq.gravpm_force(ctx, dict(Asmth=1.5, Nmesh=nmesh, G=cm.G), pman)
        return P["GravPM"].copy()

    def walk(nmesh, par, treeacc, gpm):
        cm.reference_treepar(**par)
        sq.gravshort_set_softenings(cm.BOX / np.cbrt(n))
        P["FullTreeGravAccel"] = treeacc
        P["GravPM"] = gpm
        sq.grav_short_tree(ctx, None, dict(Asmth=1.5, Nmesh=nmesh, G=cm.G), tree, None, cm.RHO0)
        return P["FullTreeGravAccel"].copy()

    errtol, e_def, e_rcut, e_nmesh = _runtests_sequence(walk, pmforce, n)
    check_runtests_gates(errtol, e_def, e_rcut, e_nmesh)
    print("runtests gates on the device: default mean %.2e max %.2e, Rcut 9.5 mean %.2e, Nmesh/2 mean %.2e max %.2e"
          % (e_def.mean(), e_def.max(), e_rcut.mean(), e_nmesh.mean(), e_nmesh.max()))


def test_pm_excludes_swallowed_black_holes(ctx):
    """gravpm.cpp:176-178: a swallowed black hole stays in the particle table but does not gravitate (RegionInd = -2): it
    deposits nothing and reads nothing out (GravPM stays at the zero gravpm_force starts from, :88-92); the tree leaves it out
    as well (forcetree.cpp:805-806)"""
    n = 16**3
    pos = cm.random_positions(orc.boost_mt19937_uniform(4, 3 * n), n)
    pman = cm.make_partmanager(pos)
    P = pman.Base
    rng = np.random.default_rng(1)
    bh = rng.choice(n, size=60, replace=False)
    P["Type"][bh] = 5
    P["Mass"][bh] = 25.0                                   # heavy: leaving them in would change every force
    swallowed = bh[::2]
    P["Flags"][swallowed] |= 2
    P["GravPM"] = 7.0                                      # must be overwritten, zero for the swallowed
    skip = np.zeros(n, dtype=np.uint8)
    skip[swallowed] = 1
    sq.gravpm_force(ctx, dict(Asmth=1.5, Nmesh=48, G=cm.G), pman)
    ogpm, opot, rho, _ = orc.pm_force(pos, P["Mass"], 48, cm.BOX, 1.5, cm.G, skip=skip, want_mesh=True)
    assert abs(rho.sum() - P["Mass"][skip == 0].astype(np.float64).sum()) < 1e-6
    live = skip == 0
    assert np.abs(P["GravPM"][live] - ogpm[live]).max() < 1e-10 * np.abs(ogpm).max()
    assert np.all(P["GravPM"][swallowed] == 0)
    # with the holes gravitating the forces are different: the exclusion is what is being tested
    ogpm_all, _, _, _ = orc.pm_force(pos, P["Mass"], 48, cm.BOX, 1.5, cm.G)
    assert np.abs(ogpm_all[live] - ogpm[live]).max() > 1e-2 * np.abs(ogpm).max()
    # short-range side: the tree is built without them and they are no targets
    tree = sq.force_tree_full(pman)
    cm.reference_treepar(ErrTolForceAcc=0.002, MaxBHOpeningAngle=0.9, TreeUseBH=1)
    sq.gravshort_set_softenings(cm.BOX / 16)
    sq.grav_short_tree(ctx, None, dict(Asmth=1.5, Nmesh=48, G=cm.G), tree, None, cm.RHO0)
    assert abs(tree.Nodes_base["mass"][0] - P["Mass"][live].astype(np.float64).sum()) < 1e-6


@pytest.mark.parametrize("usebh", [1, 0])
def test_walk_erfc_window(ctx, usebh):
    """SHORTRANGE_FORCE_WINDOW_TYPE_ERFC (gravshort-tree2.cpp:55-60): the table is erfc(u) + 2u/sqrt(pi) exp(-u^2) and erfc(u)
    at u = x / (2 Asmth); the walk with it against the oracle's walk, and against the closed form evaluated per pair for a
    direct sum over a small particle set (no tree)"""
    from math import erfc, exp, pi, sqrt
    n = 16**3
    pos = cm.random_positions(orc.boost_mt19937_uniform(6, 3 * n), n)
    pman = cm.make_partmanager(pos)
    tree = sq.force_tree_full(pman)
    sq.set_gravshort_treepar(ErrTolForceAcc=0.002, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=usebh, Rcut=6.0,
                             FractionalGravitySoftening=1.0 / 30.0, ShortRangeForceWindowType=sq.SHORTRANGE_FORCE_WINDOW_TYPE_ERFC)
    sq.gravshort_set_softenings(cm.BOX / 16)
    asmth = 1.25                                            # the erfc window is not tied to the calibrated Asmth = 1.5
    gp = sq.make_grav_params(cm.BOX, asmth, 48, cm.G, cm.RHO0)
    x = gp.dx * np.arange(capi.NGRAVTAB)
    u = x * 0.5 / asmth
    want_f = np.array([erfc(v) + 2 * v / sqrt(pi) * exp(-v * v) for v in u])
    want_p = np.array([erfc(v) for v in u])
    assert np.abs(np.array(gp.shortrange_table) - want_f.astype(np.float32)).max() == 0
    assert np.abs(np.array(gp.shortrange_table_potential) - want_p.astype(np.float32)).max() == 0
    rng = np.random.default_rng(2)
    told = rng.normal(size=(n, 3)) * 300.0
    z = np.zeros((n, 3))
    acc, pot, nint, st = _gpu_walk(ctx, pman, tree, gp, (told, z))
    mass = pman.Base["Mass"]
    oacc, opot, onint = orc.grav_walk(tree.Nodes_base, tree.firstnode, pos, mass, np.linalg.norm(told, axis=1) / cm.G, gp)
    orc.grav_postprocess(mass, gp, oacc, opot, True)
    assert np.array_equal(nint, onint)
    assert np.abs(acc - oacc).max() < 1e-11 * np.abs(oacc).max()
    assert np.allclose(pot, opot, rtol=1e-10, atol=1e-10 * np.abs(opot).max())
    cm.reference_treepar()                                  # back to the exact window for the tests that follow


def test_pm_mesh_cleared_in_the_shadow_of_the_walk(ctx):
    """The first production-size walk after a PM run zeroes the PM mesh from inside the walk kernel and the next shq_pm_run skips its
    clearing kernel: GravPM / potential stay bit-identical, with the scrub on or off, after an intervening FFT seam call (another
    writer of the mesh) and for a walk too small to carry the scrub."""
    n, L, nmesh = 64**3, 1.0, 96
    pos = sq.synth_positions("cluster", n, L=L)
    pos = pos[sq.morton_order(pos, L)]
    pman = cm.make_partmanager(pos, box=L)
    tree = sq.force_tree_full(pman)
    sq.set_gravshort_treepar(ErrTolForceAcc=0.005, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=1, Rcut=6.0)
    sq.gravshort_set_softenings(L / 64)
    gp = sq.make_grav_params(L, 1.5, nmesh, cm.G, cm.RHO0)
    pmp = sq.PMParams(nmesh, 0, L, 1.5, cm.G)
    pv, tv = pman.view(), tree.view()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    capi.check(capi.hip.shq_tree_upload(ctx.h, C.byref(tv)))
    capi.check(capi.hip.shq_grav_refresh_oldacc(ctx.h, gp.G))

    def pm():
        capi.check(capi.hip.shq_pm_run(ctx.h, C.byref(pmp)))
        g = np.zeros((n, 3)); p = np.zeros(n)
        capi.check(capi.hip.shq_pm_download(ctx.h, capi.ptr(g), capi.ptr(p)))
        return g, p

    def walk(active=None):
        a = None if active is None else np.ascontiguousarray(active, dtype=np.int32)
        capi.check(capi.hip.shq_grav_short_run(ctx.h, C.byref(gp), capi.ptr(a), 0 if a is None else len(a), 1, sq.WALK_EXACT))
        acc = np.zeros((n, 3)); pot = np.zeros(n); nint = np.zeros(n, dtype=np.int64)
        capi.check(capi.hip.shq_grav_short_download(ctx.h, capi.ptr(acc), capi.ptr(pot), capi.ptr(nint), C.byref(sq.WalkStats())))
        return acc

    def prezeroed():
        z = C.c_int(-1)
        capi.check(capi.hip.shq_pm_mesh_prezeroed(ctx.h, C.byref(z)))
        return z.value

    capi.check(capi.hip.shq_pm_set_mesh_scrub(ctx.h, 0))
    g0, p0 = pm()
    a0 = walk()
    assert prezeroed() == 0
    capi.check(capi.hip.shq_pm_set_mesh_scrub(ctx.h, 1))
    g1, p1 = pm()
    assert np.array_equal(g1, g0) and np.array_equal(p1, p0)
    a1 = walk()
    assert prezeroed() == 1                       # this walk carried the scrub ...
    assert np.array_equal(a1, a0)                 # ... and its forces are what they were
    g2, p2 = pm()                                 # no clearing kernel in this run
    assert prezeroed() == 0
    assert np.array_equal(g2, g0) and np.array_equal(p2, p0)
    # a second walk before the next PM run has nothing left to do; a walk on 1/64 of the targets would have too large a share per task
    walk()
    assert prezeroed() == 1
    walk()
    assert prezeroed() == 1
    g3, p3 = pm()
    assert np.array_equal(g3, g0) and np.array_equal(p3, p0)
    walk(np.arange(0, n, 64))
    assert prezeroed() == 0
    g4, p4 = pm()
    assert np.array_equal(g4, g0) and np.array_equal(p4, p0)
    # another writer of the mesh between the scrub and the PM run: the FFT seam
    walk()
    assert prezeroed() == 1
    real = np.random.default_rng(5).standard_normal((nmesh, nmesh, nmesh))
    comp = np.zeros((nmesh, nmesh, nmesh // 2 + 1, 2))
    capi.check(capi.hip.shq_fft_r2c(ctx.h, nmesh, capi.ptr(real), capi.ptr(comp)))
    assert prezeroed() == 0
    g5, p5 = pm()
    assert np.array_equal(g5, g0) and np.array_equal(p5, p0)


def test_treepm_step_equals_the_three_separate_calls(ctx):
    """shq_treepm_step (gravpm_force then grav_short_tree in the reference's order, run.cpp:518-563) against shq_pm_run +
    shq_grav_refresh_oldacc + shq_grav_short_run from the same state: GravPM, PM potential, OldAcc-driven interaction counts, forces
    and potentials bit for bit, with the readout fused into the walk's prologue and with the fusion switched off; and the walk's
    result equals the oracle's walk from OldAcc = |FullTreeGravAccel_old + GravPM_new| / G."""
    n, L, nmesh = 64**3, 1.0, 96
    pos = sq.synth_positions("cluster", n, L=L)
    pos = pos[sq.morton_order(pos, L)]
    pman = cm.make_partmanager(pos, box=L)
    tree = sq.force_tree_full(pman)
    sq.set_gravshort_treepar(ErrTolForceAcc=0.005, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=0, Rcut=6.0)
    sq.gravshort_set_softenings(L / 64)
    gp = sq.make_grav_params(L, 1.5, nmesh, cm.G, cm.RHO0)
    pmp = sq.PMParams(nmesh, 0, L, 1.5, cm.G)
    rng = np.random.default_rng(11)
    P = pman.Base
    P["FullTreeGravAccel"] = rng.standard_normal((n, 3)) * 50.0       # "last step's" tree force
    P["GravPM"] = rng.standard_normal((n, 3))                         # stale: must not enter OldAcc
    pv, tv = pman.view(), tree.view()

    def start():
        capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
        capi.check(capi.hip.shq_tree_upload(ctx.h, C.byref(tv)))

    def results():
        g = np.zeros((n, 3)); pp = np.zeros(n)
        capi.check(capi.hip.shq_pm_download(ctx.h, capi.ptr(g), capi.ptr(pp)))
        acc = np.zeros((n, 3)); pot = np.zeros(n); nint = np.zeros(n, dtype=np.int64)
        capi.check(capi.hip.shq_grav_short_download(ctx.h, capi.ptr(acc), capi.ptr(pot), capi.ptr(nint), C.byref(sq.WalkStats())))
        return g, pp, acc, pot, nint

    start()
    capi.check(capi.hip.shq_pm_run(ctx.h, C.byref(pmp)))
    capi.check(capi.hip.shq_grav_refresh_oldacc(ctx.h, gp.G))
    capi.check(capi.hip.shq_grav_short_run(ctx.h, C.byref(gp), None, 0, 1, sq.WALK_EXACT))
    ref = results()
    fused = C.c_int(-1)
    for want in (1, 0):
        start()
        capi.check(capi.hip.shq_treepm_set_fuse(ctx.h, want))
        capi.check(capi.hip.shq_treepm_step(ctx.h, C.byref(pmp), C.byref(gp), 1, sq.WALK_EXACT))
        capi.check(capi.hip.shq_treepm_last_fused(ctx.h, C.byref(fused)))
        assert fused.value == want
        got = results()
        for a, b, name in zip(got, ref, ("GravPM", "PM potential", "acc", "pot", "ninteractions")):
            assert np.array_equal(a, b), (want, name, float(np.abs(a - b).max()))
    capi.check(capi.hip.shq_treepm_set_fuse(ctx.h, 1))
    # the same with the targets taken in tree (leaf) order, as a moving-particle step does
    start()
    capi.check(capi.hip.shq_pm_run(ctx.h, C.byref(pmp)))
    capi.check(capi.hip.shq_grav_refresh_oldacc(ctx.h, gp.G))
    capi.check(capi.hip.shq_grav_short_run(ctx.h, C.byref(gp), None, 0, 1, sq.WALK_EXACT | sq.WALK_TREE_ORDER))
    ref_t = results()
    start()
    capi.check(capi.hip.shq_treepm_step(ctx.h, C.byref(pmp), C.byref(gp), 1, sq.WALK_EXACT | sq.WALK_TREE_ORDER))
    capi.check(capi.hip.shq_treepm_last_fused(ctx.h, C.byref(fused)))
    assert fused.value == 1
    for a, b, name in zip(results(), ref_t, ("GravPM", "PM potential", "acc", "pot", "ninteractions")):
        assert np.array_equal(a, b), ("tree order", name, float(np.abs(a - b).max()))
    assert np.array_equal(ref_t[0], ref[0]) and np.array_equal(ref_t[4], ref[4])
    # the PM started early on the second stream (shq_pm_start), the tree installed meanwhile, shq_treepm_step joining it: the same bits;
    # with and without OldAcc formed by the early PM's readout
    for G_early in (gp.G, 0.0):
        capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
        capi.check(capi.hip.shq_pm_start(ctx.h, C.byref(pmp), G_early))
        capi.check(capi.hip.shq_tree_upload(ctx.h, C.byref(tv)))
        capi.check(capi.hip.shq_treepm_step(ctx.h, C.byref(pmp), C.byref(gp), 1, sq.WALK_EXACT))
        for a, b, name in zip(results(), ref, ("GravPM", "PM potential", "acc", "pot", "ninteractions")):
            assert np.array_equal(a, b), ("early PM", G_early, name, float(np.abs(a - b).max()))
    # the walk's opening criterion saw the NEW GravPM
    g, _, acc, pot, nint = ref
    oldacc = np.linalg.norm(P["FullTreeGravAccel"] + g, axis=1) / cm.G
    o, op, on = orc.grav_walk(tree.Nodes_base, tree.firstnode, pos, P["Mass"], oldacc, gp)
    orc.grav_postprocess(P["Mass"], gp, o, op, True)
    assert np.array_equal(nint, on)
    assert cm.force_err(acc, o).max() < 1e-5


def test_pair_kernel_failures_in_a_resident_loop(ctx):
    """The sparse-subtree pair kernel's two failure modes inside a resident loop of three shq_treepm_step calls WITHOUT a download
    in between (every launch checked, treewalk2.cuh:351-353).  (1) A live pair kernel whose waves give up waiting for the main walk
    (shq_set_walk_debug starves them: one poll) is finished by the mop-up pass behind the walk: forces, potentials and interaction
    counts of all three steps' end state equal the undisturbed loop's bit for bit, and the status says three launches were recovered.
    (2) A pair stack that runs full (shrunk to its minimum) drops pairs: the sticky error survives the launches that follow and comes
    back as SHQ_ERR_DEVICE from a later step's entry or, at the latest, from shq_synchronize; the report clears it."""
    n, L, nmesh = 64**3, 1.0, 96
    pos = sq.synth_positions("cluster", n, L=L)
    pos = pos[sq.morton_order(pos, L)]
    pman = cm.make_partmanager(pos, box=L)
    tree = sq.force_tree_full(pman)
    sq.set_gravshort_treepar(ErrTolForceAcc=0.005, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=0, Rcut=6.0)
    sq.gravshort_set_softenings(L / 64)
    gp = sq.make_grav_params(L, 1.5, nmesh, cm.G, cm.RHO0)
    pmp = sq.PMParams(nmesh, 0, L, 1.5, cm.G)
    P = pman.Base
    P["FullTreeGravAccel"] = np.random.default_rng(12).standard_normal((n, 3)) * 50.0
    pv, tv = pman.view(), tree.view()
    ERR_DEVICE = 2

    def start():
        capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
        capi.check(capi.hip.shq_tree_upload(ctx.h, C.byref(tv)))

    def status():
        rec, high, mop = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        rc = capi.hip.shq_walk_pair_status(ctx.h, C.byref(rec), C.byref(high), C.byref(mop))
        return rc, rec.value, high.value, mop.value

    def results():
        acc = np.zeros((n, 3)); pot = np.zeros(n); nint = np.zeros(n, dtype=np.int64)
        capi.check(capi.hip.shq_grav_short_download(ctx.h, capi.ptr(acc), capi.ptr(pot), capi.ptr(nint), C.byref(sq.WalkStats())))
        return acc, pot, nint

    def step():
        return capi.hip.shq_treepm_step(ctx.h, C.byref(pmp), C.byref(gp), 1, sq.WALK_EXACT)

    try:
        capi.check(capi.hip.shq_set_walk_launch(ctx.h, 2, 1))   # the production launch at this size: persistent waves, leaf ring,
        capi.check(capi.hip.shq_set_walk_overlap(ctx.h, 2))     # the pair kernel beside the main walk
        start()
        for _ in range(3):                                      # every step's OldAcc comes from the step before: a resident loop
            capi.check(step())
        ref = results()
        rc, rec0, high, mop = status()
        assert rc == 0 and mop == 0 and 0 < high <= 4096, (rc, high, mop)
        # (1) starved live pair kernel: the mop-up pass walks what the live waves left
        capi.check(capi.hip.shq_set_walk_debug(ctx.h, 1, 0))
        start()
        for _ in range(3):
            capi.check(step())
        rc, rec1, _, mop = status()
        got = results()
        assert rc == 0 and rec1 == rec0 + 3 and mop > 0, (rc, rec0, rec1, mop)
        for a, b, name in zip(got, ref, ("acc", "pot", "ninteractions")):
            assert np.array_equal(a, b), name
        # (2) pair stacks of 1344 entries overflow in the cluster: sticky, reported although two more launches follow the first failure
        capi.check(capi.hip.shq_set_walk_debug(ctx.h, 0, 1344))
        start()
        capi.check(step())                                      # queued; its failure is not known yet
        rcs = [step()]
        if rcs[-1] == 0:
            rcs.append(step())
        if rcs[-1] == 0:
            rcs.append(capi.hip.shq_synchronize(ctx.h))
        assert rcs[-1] == ERR_DEVICE and "pair stack" in capi.hip.shq_last_error().decode(), rcs
        # drained and cleared: the same loop with the stacks restored is the reference loop again
        capi.check(capi.hip.shq_set_walk_debug(ctx.h, 0, 0))
        while capi.hip.shq_synchronize(ctx.h) != 0:
            pass
        start()
        for _ in range(3):
            capi.check(step())
        got = results()
        for a, b, name in zip(got, ref, ("acc", "pot", "ninteractions")):
            assert np.array_equal(a, b), name
    finally:
        capi.hip.shq_set_walk_debug(ctx.h, 0, 0)
        capi.hip.shq_synchronize(ctx.h)
        capi.hip.shq_synchronize(ctx.h)
        capi.check(capi.hip.shq_set_walk_overlap(ctx.h, 1))
        capi.check(capi.hip.shq_set_walk_launch(ctx.h, 1, 1))
    assert np.isfinite(ref[0]).all() and ref[2].min() > 0


def test_treepm_step_ragged_boundary_and_swallowed(ctx):
    """The one-call step on a set whose size is no multiple of 64, spread over the whole box (waves with particles next to the faces
    take the readout's wrapped path) and holding swallowed black holes (no deposit, no readout: GravPM = 0, gravpm.cpp:176-178):
    bit-identical to the separate calls."""
    n = 5003
    pos = cm.random_positions(orc.boost_mt19937_uniform(9, 3 * n), n)
    pos[:40] *= 1e-3                                       # a few particles in the corner cell and on the faces
    pos[40:80, 0] = cm.BOX * (1 - 1e-9)
    pman = cm.make_partmanager(pos)
    P = pman.Base
    rng = np.random.default_rng(3)
    bh = rng.choice(n, size=50, replace=False)
    P["Type"][bh] = 5
    P["Flags"][bh[::2]] |= 2
    P["FullTreeGravAccel"] = rng.standard_normal((n, 3)) * 5.0
    tree = sq.force_tree_full(pman)
    cm.reference_treepar(ErrTolForceAcc=0.002, MaxBHOpeningAngle=0.9, TreeUseBH=0)
    sq.gravshort_set_softenings(cm.BOX / 17)
    gp = sq.make_grav_params(cm.BOX, 1.5, 48, cm.G, cm.RHO0)
    pmp = sq.PMParams(48, 0, cm.BOX, 1.5, cm.G)
    pv, tv = pman.view(), tree.view()

    def run(one_call):
        capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
        capi.check(capi.hip.shq_tree_upload(ctx.h, C.byref(tv)))
        if one_call:
            capi.check(capi.hip.shq_treepm_step(ctx.h, C.byref(pmp), C.byref(gp), 1, sq.WALK_EXACT))
            f = C.c_int(0)
            capi.check(capi.hip.shq_treepm_last_fused(ctx.h, C.byref(f)))
            assert f.value == 1
        else:
            capi.check(capi.hip.shq_pm_run(ctx.h, C.byref(pmp)))
            capi.check(capi.hip.shq_grav_refresh_oldacc(ctx.h, gp.G))
            capi.check(capi.hip.shq_grav_short_run(ctx.h, C.byref(gp), None, 0, 1, sq.WALK_EXACT))
        g = np.zeros((n, 3)); pp = np.zeros(n)
        capi.check(capi.hip.shq_pm_download(ctx.h, capi.ptr(g), capi.ptr(pp)))
        acc = np.zeros((n, 3)); pot = np.zeros(n); nint = np.zeros(n, dtype=np.int64)
        capi.check(capi.hip.shq_grav_short_download(ctx.h, capi.ptr(acc), capi.ptr(pot), capi.ptr(nint), C.byref(sq.WalkStats())))
        return g, pp, acc, pot, nint

    ref, got = run(False), run(True)
    for a, b, name in zip(got, ref, ("GravPM", "PM potential", "acc", "pot", "ninteractions")):
        assert np.array_equal(a, b), (name, float(np.abs(a - b).max()))
    assert np.all(ref[0][bh[::2]] == 0) and np.abs(ref[0]).max() > 0
