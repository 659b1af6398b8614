"""CPU tests pinning the SPH oracle against the reference's own density gates
(tests/test_density.cpp:73-90,134-236,208-303)."""
import numpy as np
import pytest

import shenqi_amd as sq
import orc
import common as cm


def _run_density_oracle(pos, hsml, lastisbh=False, kernel=1, DoEgyDensity=0):
    """do_density_test, tests/test_density.cpp:134-206, with the oracle."""
    pman, SphP, BhP = cm.make_gas(pos, hsml, lastisbh=lastisbh)
    P = pman.Base
    n = len(pos)
    dp = cm.density_params(kernel=kernel, BlackHoleOn=0, DoEgyDensity=DoEgyDensity)
    st = orc.SphState(P, SphP, BhP)
    # force_tree_rebuild_mask(GASMASK+BHMASK) -> set_init_hsml -> rebuild GASMASK tree (:144-165)
    idx_all = np.arange(n, dtype=np.int32)
    nodes, first, father = orc.tree_build(pos, P["Mass"], cm.BOX, idx=idx_all)
    orc.set_init_hsml(nodes, first, father, st, cm.BOX, dp.DesNumNgb)
    gas = np.nonzero(P["Type"] == 0)[0].astype(np.int32)
    nodes, first, father = orc.tree_build(pos, P["Mass"], cm.BOX, idx=gas, numpart_total=n)
    rc, evp, _, niter, nint = orc.density(nodes, first, father, st, dp)
    assert rc == 0
    return st, dp, (nodes, first, father), evp, niter


def _check_densities(st, MinGasHsml):
    """check_densities, tests/test_density.cpp:73-90"""
    gas = st.type == 0
    assert np.all(np.isfinite(st.hsml))
    assert np.all(np.isfinite(st.density)) and np.all(st.density > 0)
    assert st.hsml.min() >= MinGasHsml
    assert st.hsml.max() <= cm.BOX


def test_reference_gate_density_flat():
    """test_density_flat: 32^3 grid, expected mean Hsml 0.5 +- 5e-4 (tests/test_density.cpp:208-238)."""
    pos = cm.grid_positions(32)
    hsml = np.full(len(pos), 1.5 * cm.BOX / 32)
    st, dp, tree, evp, niter = _run_density_oracle(pos, hsml)
    assert abs(st.hsml.mean() - 0.5) < 5e-4, st.hsml.mean()
    _check_densities(st, 0.006)
    # the reference run gives mean Hsml 0.500387 and rho = 64.002 (= 32^3/8^3)
    assert abs(st.hsml.mean() - 0.500387) < 2e-6
    assert abs(st.density.mean() - 64.002) < 2e-3
    # second pass must reproduce Hsml within MaxNumNgbDeviation/DesNumNgb (:203)
    h0 = st.hsml.copy()
    rc, _, _, _, _ = orc.density(tree[0], tree[1], tree[2], st, dp)
    assert rc == 0
    assert np.all(np.abs(h0 / st.hsml - 1) < dp.MaxNumNgbDeviation / dp.DesNumNgb)


def test_reference_gate_density_close():
    """test_density_close: expected mean Hsml 0.131726 +- 1e-4, last particle a BH (:240-270)."""
    pos, hsml = cm.density_close_positions()
    st, dp, tree, evp, niter = _run_density_oracle(pos, hsml, lastisbh=True)
    assert abs(st.hsml.mean() - 0.131726) < 1e-4, st.hsml.mean()
    _check_densities(st, 0.006)


def test_reference_gate_density_random():
    """test_density_random: boost mt19937(0), expected mean Hsml 0.187515 +- 1e-3 (:272-320)."""
    n = 32**3
    for k in range(2):
        u = orc.boost_mt19937_uniform(0, 3 * n, skip=3 * n * k)
        pos = cm.random_positions(u, n)
        hsml = np.full(n, cm.BOX / 32)
        st, dp, tree, evp, niter = _run_density_oracle(pos, hsml)
        assert abs(st.hsml.mean() - 0.187515) < 1e-3, st.hsml.mean()
        _check_densities(st, 0.006)


def test_hydro_oracle_invariants():
    """No reference unit test exists for hydro; check pair antisymmetry
    (sum_i m_i a_i = 0 for the symmetric pressure force with all particles active) and that a
    uniform lattice with uniform entropy feels no net force."""
    n = 16**3
    u = orc.boost_mt19937_uniform(5, 3 * n)
    pos = cm.BOX * u.reshape(n, 3)
    hsml = np.full(n, cm.BOX / 16)
    pman, SphP, BhP = cm.make_gas(pos, hsml)
    P = pman.Base
    dp = cm.density_params(DoEgyDensity=1)
    st = orc.SphState(P, SphP, BhP)
    nodes, first, father = orc.tree_build(pos, P["Mass"], cm.BOX)
    orc.set_init_hsml(nodes, first, father, st, cm.BOX, dp.DesNumNgb)
    nodes, first, father = orc.tree_build(pos, P["Mass"], cm.BOX)
    rc, evp, _, niter, _ = orc.density(nodes, first, father, st, dp)
    assert rc == 0
    orc.update_hmax(nodes, first, st)
    hp = cm.hydro_params()
    nint = orc.hydro(nodes, first, st, hp, evp)
    acc = st.hydroaccel
    assert np.all(np.isfinite(acc)) and np.all(np.isfinite(st.dtentropy)) and np.all(st.maxsignalvel > 0)
    mom = (st.mass[:, None] * acc).sum(axis=0)
    assert np.abs(mom).max() < 1e-9 * np.abs(acc).sum()
    # lattice: forces cancel
    pos = cm.grid_positions(16)
    pman, SphP, BhP = cm.make_gas(pos, np.full(n, 1.5 * cm.BOX / 16))
    st = orc.SphState(pman.Base, SphP, BhP)
    nodes, first, father = orc.tree_build(pos, pman.Base["Mass"], cm.BOX)
    rc, evp, _, _, _ = orc.density(nodes, first, father, st, dp)
    orc.update_hmax(nodes, first, st)
    orc.hydro(nodes, first, st, hp, evp)
    scale = np.abs(acc).mean()
    assert np.abs(st.hydroaccel).max() < 1e-6 * scale
