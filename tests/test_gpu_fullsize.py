"""BASELINE.json configs[1] at full size (dm-only 256^3, Nmesh 768, S-cluster) through properties that do not
need a full-size oracle run, plus an oracle comparison on a 1/256 sample of the targets:
  - the device-built tree is walked by the CPU oracle for the sampled targets: interaction counts equal as
    integers, forces at rounding level;
  - PM: two runs give identical bits (fixed-point deposit, fixed pass order); the mesh force conserves momentum
    (CIC deposit and readout with the same kernel, antisymmetric difference stencil);
  - tree: Newton's third law holds per accepted pair only statistically for a tree code, so the check is the
    reference's own gate: total momentum change small against the summed force magnitudes."""
import ctypes as C

import numpy as np
import pytest

import shenqi_amd as sq
from shenqi_amd import capi
import orc

pytestmark = pytest.mark.gpu

G = 43.0071
RHO0 = 0.3 * 3 * 0.1 * 0.1 / (8 * np.pi * G)


def test_baseline_256_properties(ctx):
    n1, L = 256, 1.0
    n, nmesh = n1**3, 3 * n1
    pos = sq.synth_positions("cluster", n, L=L)
    pos = pos[sq.hilbert_order(pos, L)]
    pman = sq.PartManager(n, L)
    P = pman.Base
    P["Pos"] = pos
    P["Type"] = 1
    P["Mass"] = 1.0
    _treepm_properties(ctx, pman, n1, nmesh, L, stride=256)


def test_c3_gravity_2x128_properties(ctx):
    """BASELINE.json configs[2], gravity half: 2 x 128^3 gas + dark matter (uniform, as the initial conditions of examples/small
    nearly are; masses Omega_b : Omega_m - Omega_b), Nmesh 384: PM + short-range walk over both species, same properties and
    the same sampled oracle comparison as the 256^3 case (bench.py's kernels.c3_* figures time exactly this)."""
    n1, L = 128, 1.0
    n = n1**3
    gas = sq.synth_positions("uniform", n, L=L)
    dm = sq.synth_positions("uniform", n, seed=77, L=L)
    pos = np.concatenate([gas, dm])
    order = sq.hilbert_order(pos, L)
    pman = sq.PartManager(2 * n, L)
    P = pman.Base
    P["Pos"] = pos[order]
    P["Type"] = np.where(order < n, 0, 1).astype(np.uint8)
    P["Mass"] = np.where(order < n, 0.16, 0.84)
    _treepm_properties(ctx, pman, n1, 3 * n1, L, stride=64)


def test_c4_512_properties(ctx):
    """BASELINE.json configs[3] (dm-50-512: 512^3 dark matter, Nmesh 1024 per benchmarks/dm-50-512/paramfile.gadget:8), the whole
    particle set on ONE card - what the 8-GPU run shards: 1.34e8 particles, ~5.9e7 tree nodes, an 8.7 GB mesh (beyond 2^32 bytes:
    64-bit mesh offsets everywhere, the 512-thread FFT workgroups of the transposing pipeline), 64-bit interaction totals.  Same
    properties as the 256^3 case, the oracle on every 2048th 64-target group over the device-built tree."""
    n1, L = 512, 1.0
    n = n1**3
    pos = sq.synth_positions("cluster", n, L=L)
    pos = pos[sq.hilbert_order(pos, L)]
    pman = sq.PartManager(n, L)
    P = pman.Base
    P["Pos"] = pos
    del pos
    P["Type"] = 1
    P["Mass"] = 1.0
    _treepm_properties(ctx, pman, n1, 1024, L, stride=2048)


def test_c5_gravity_2x256_properties(ctx):
    """BASELINE.json configs[4] (examples/hydro: 2 x 256^3 gas + dark matter, Nmesh 768), gravity half on one card: PM + short-range
    walk over all 3.4e7 particles of both species (masses Omega_b : Omega_m - Omega_b), the properties and the sampled oracle of the
    other sizes."""
    n1, L = 256, 1.0
    n = n1**3
    pos = np.concatenate([sq.synth_positions("uniform", n, L=L), sq.synth_positions("uniform", n, seed=77, L=L)])
    order = sq.hilbert_order(pos, L)
    pman = sq.PartManager(2 * n, L)
    P = pman.Base
    P["Pos"] = pos[order]
    P["Type"] = np.where(order < n, 0, 1).astype(np.uint8)
    P["Mass"] = np.where(order < n, 0.16, 0.84)
    del pos
    _treepm_properties(ctx, pman, n1, 3 * n1, L, stride=512)


def _treepm_properties(ctx, pman, n1, nmesh, L, stride):
    P = pman.Base
    n = pman.NumPart
    pos = P["Pos"].copy()
    pv = pman.view()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    tb = sq.tree_build_device(ctx, L)
    assert tb.nparticles == n and tb.numnodes > n // 8

    # ---- PM: bit reproducibility and momentum conservation
    pmp = sq.PMParams(nmesh, 0, L, 1.5, G)
    gpm = [np.zeros((n, 3)), np.zeros((n, 3))]
    ppot = [np.zeros(n), np.zeros(n)]
    for k in range(2):
        capi.check(capi.hip.shq_pm_run(ctx.h, C.byref(pmp)))
        capi.check(capi.hip.shq_pm_download(ctx.h, capi.ptr(gpm[k]), capi.ptr(ppot[k])))
    assert np.array_equal(gpm[0], gpm[1]) and np.array_equal(ppot[0], ppot[1])
    assert np.all(np.isfinite(gpm[0]))
    mg = P["Mass"][:, None].astype(np.float64) * gpm[0]
    assert np.abs(mg.sum(axis=0)).max() < 1e-9 * np.abs(mg).sum()

    # ---- walks: Barnes-Hut seed, then the relative criterion of the north-star setting
    sq.set_gravshort_treepar(ErrTolForceAcc=0.005, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=1, Rcut=6.0)
    sq.gravshort_set_softenings(L / n1)
    gp_bh = sq.make_grav_params(L, 1.5, nmesh, G, RHO0)
    sq.set_gravshort_treepar(ErrTolForceAcc=0.005, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=0, Rcut=6.0)
    gp = sq.make_grav_params(L, 1.5, nmesh, G, RHO0)
    capi.check(capi.hip.shq_grav_short_run(ctx.h, C.byref(gp_bh), None, 0, 1, sq.WALK_EXACT))
    seed = np.zeros((n, 3))
    capi.check(capi.hip.shq_grav_short_download(ctx.h, capi.ptr(seed), None, None, None))
    capi.check(capi.hip.shq_grav_refresh_oldacc(ctx.h, G))
    capi.check(capi.hip.shq_grav_short_run(ctx.h, C.byref(gp), None, 0, 1, sq.WALK_EXACT))
    acc = np.zeros((n, 3)); pot = np.zeros(n); nint = np.zeros(n, dtype=np.int64)
    st = sq.WalkStats()
    capi.check(capi.hip.shq_grav_short_download(ctx.h, capi.ptr(acc), capi.ptr(pot), capi.ptr(nint), C.byref(st)))
    assert st.ninteractions == int(nint.sum()) and st.min_interactions == nint.min() and st.max_interactions == nint.max()
    assert np.all(np.isfinite(acc)) and np.all(np.isfinite(pot))
    # tree code: momentum is conserved to the force accuracy, not exactly
    macc = P["Mass"][:, None].astype(np.float64) * acc
    assert np.abs(macc.sum(axis=0)).max() < 1e-3 * np.abs(macc).sum()

    # ---- oracle on every 256th 64-target group, walking the DEVICE-built tree
    nodes, _ = sq.tree_download(ctx, n, 0)
    groups = np.arange(0, n // 64, stride)
    targets = (groups[:, None] * 64 + np.arange(64)[None, :]).ravel().astype(np.int32)
    oldacc = np.linalg.norm(seed + gpm[0], axis=1) / G
    oacc, opot, onint = orc.grav_walk(nodes, n, pos, P["Mass"], oldacc, gp, targets=targets)
    orc.grav_postprocess(P["Mass"], gp, oacc, opot, True, targets=targets)
    assert np.array_equal(nint[targets], onint)
    scale = np.abs(oacc).max()
    assert np.abs(acc[targets] - oacc).max() < 1e-10 * scale
    assert np.allclose(pot[targets], opot, rtol=1e-9, atol=1e-10 * np.abs(opot).max())

    # ---- the walks above cleared the PM mesh in their shadow (persistent launch at this size): a third PM run, without its
    #      clearing kernel, reproduces the first two bit for bit
    z = C.c_int(-1)
    capi.check(capi.hip.shq_pm_mesh_prezeroed(ctx.h, C.byref(z)))
    assert z.value == 1
    g3 = np.zeros((n, 3)); p3 = np.zeros(n)
    capi.check(capi.hip.shq_pm_run(ctx.h, C.byref(pmp)))
    capi.check(capi.hip.shq_pm_download(ctx.h, capi.ptr(g3), capi.ptr(p3)))
    assert np.array_equal(g3, gpm[0]) and np.array_equal(p3, ppot[0])


def test_c3_sph_128_properties(ctx):
    """BASELINE.json configs[2] at its gas size (128^3 gas particles, quintic kernel, pressure-entropy SPH): properties that
    need no full-size oracle run, plus the oracle on a 1/512 sample of the targets over the same tree.
      - density: converged Hsml loop, every particle ends with a finite positive density; a second call changes nothing
        beyond the reference's own re-run bound (tests/test_density.cpp:203);
      - hydro: pair antisymmetry makes the total momentum change vanish: |sum m a| << sum |m a|;
      - oracle (fixed Hsml, sampled targets): densities and hydro accelerations to rounding."""
    _sph_properties(ctx, 128, 512)


def test_c5_sph_256_properties(ctx):
    """BASELINE.json configs[4] at its gas size on one card (256^3 = 1.7e7 gas particles): the properties of the 128^3 case and
    the oracle on every 4096th target."""
    _sph_properties(ctx, 256, 4096)


def _sph_properties(ctx, n1, sample):
    import common as cm
    L = cm.BOX
    n = n1**3
    pos = sq.synth_positions("uniform", n, L=L)
    pos = pos[sq.hilbert_order(pos, L)]
    pman, SphP, BhP = cm.make_gas(pos, np.full(n, 1.5 * L / n1), box=L)
    BhP = np.zeros(2, dtype=sq.BH_SLOT_DTYPE)
    P = pman.Base
    rng = np.random.default_rng(5)
    P["Vel"] = rng.normal(size=(n, 3)) * 0.01
    SphP["Entropy"] = rng.uniform(0.8, 1.2, size=n)
    kernel = 2
    sq.set_densitypar(DensityResolutionEta=1.0, MaxNumNgbDeviation=0.5, DensityKernelType=kernel, BlackHoleNgbFactor=2.0, MinGasHsml=1e-6)
    tree = sq.force_tree_rebuild_mask(pman, sq.GASMASK)
    evp, st0 = sq.density(ctx, None, 1, 1, 0, None, tree, pman, SphP, BhP)
    assert st0.ntargets == n and np.all(np.isfinite(SphP["Density"])) and SphP["Density"].min() > 0
    h0, rho0 = P["Hsml"].copy(), SphP["Density"].copy()
    assert 0.9 < rho0.mean() / (n / L**3) < 1.5                  # unit masses: the estimate at the particles' own places is biased high
    evp, st1 = sq.density(ctx, None, 1, 1, 0, None, tree, pman, SphP, BhP)
    assert np.all(np.abs(h0 / P["Hsml"] - 1) < 0.5 / sq.GetNumNgb())
    sq.force_tree_update_hmax(tree, pman)
    sq.set_hydropar(1, 100.0, 0.75)
    hs = sq.hydro_force(ctx, None, 0.1, cm.HUBBLE, evp, None, tree, pman, SphP)
    a = SphP["HydroAccel"]
    assert np.all(np.isfinite(a)) and np.abs(a).max() > 0
    assert np.abs(a.sum(axis=0)).max() < 1e-9 * np.abs(a).sum()   # unit masses: sum m a
    # oracle on every `sample`-th target, same tree, Hsml fixed at the converged values
    act = np.arange(0, n, sample, dtype=np.int32)
    dp = cm.density_params(BoxSize=L, kernel=kernel, update_hsml=0, DoEgyDensity=1, MinGasHsml=1e-6)
    S2 = SphP.copy()
    st = orc.SphState(P, S2, BhP)
    rc, oevp, _, _, _ = orc.density(tree.Nodes_base.copy(), tree.firstnode, None, st, dp, active=act)
    assert rc == 0
    assert np.abs(st.density[act] / SphP["Density"][act] - 1).max() < 1e-10
    assert np.abs(st.egywtdensity[act] / SphP["EgyWtDensity"][act] - 1).max() < 1e-10
    hp = cm.hydro_params(BoxSize=L, kernel=kernel, hubble=cm.HUBBLE)
    st = orc.SphState(P, SphP.copy(), BhP)
    orc.hydro(tree.Nodes_base, tree.firstnode, st, hp, evp, active=act)
    scale = np.abs(a).max()
    assert np.abs(st.hydroaccel[act] - a[act]).max() < 1e-9 * scale
