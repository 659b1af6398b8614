"""Shared set-up of the reference's force tests (tests/test_gravity.cpp:25-247)."""
import numpy as np

import shenqi_amd as sq

G = 43.0071                      # tests/test_gravity.cpp:23
BOX = 8.0                        # :30
OMEGA0, HUBBLE = 0.3, 0.1        # CP.Omega0 (:211) and CP.Hubble in internal units (init_cosmology)
RHO0 = OMEGA0 * 3 * HUBBLE * HUBBLE / (8 * np.pi * G)   # :223


def grid_positions(ncbrt, box=BOX):
    i = np.arange(ncbrt**3)
    pos = np.empty((ncbrt**3, 3))
    pos[:, 0] = (box / ncbrt) * (i // ncbrt // ncbrt)
    pos[:, 1] = (box / ncbrt) * ((i // ncbrt) % ncbrt)
    pos[:, 2] = (box / ncbrt) * (i % ncbrt)
    return pos


def close_positions(ncbrt, close=5000.0):
    """tests/test_gravity.cpp:294-314 test_force_close"""
    i = np.arange(ncbrt**3)
    pos = np.empty((ncbrt**3, 3))
    pos[:, 0] = 4.0 + (i // ncbrt // ncbrt) / close
    pos[:, 1] = 4.0 + ((i // ncbrt) % ncbrt) / close
    pos[:, 2] = 4.0 + (i % ncbrt) / close
    return pos


def random_positions(u, numpart, box=BOX):
    """tests/test_gravity.cpp:316-341 do_random_test; u = 3*numpart uniform variates in draw order."""
    u = u.reshape(numpart, 3)
    pos = np.empty((numpart, 3))
    a, b = numpart // 4, 3 * numpart // 4
    pos[:a] = box * u[:a]
    pos[a:b] = box / 2 + box / 8 * np.exp((u[a:b] - 0.5) ** 2)
    pos[b:] = box * 0.1 + box / 32 * np.exp((u[b:] - 0.5) ** 2)
    return pos


def make_partmanager(pos, box=BOX, mass=1.0, ptype=1):
    """setup_particles, tests/test_gravity.cpp:174-195"""
    n = len(pos)
    pm = sq.PartManager(n, box)
    P = pm.Base
    P["Pos"] = pos
    P["Type"] = ptype
    P["Mass"] = mass
    P["ID"] = np.arange(1, n + 1)
    return pm


def reference_treepar(ErrTolForceAcc=0.002, MaxBHOpeningAngle=0.0, Rcut=7.0, TreeUseBH=2):
    """treeacc of tests/test_gravity.cpp:225-236 (MaxBHOpeningAngle is zero-initialised there)."""
    sq.set_gravshort_treepar(ErrTolForceAcc=ErrTolForceAcc, BHOpeningAngle=0.175, MaxBHOpeningAngle=MaxBHOpeningAngle,
                             TreeUseBH=TreeUseBH, Rcut=Rcut, FractionalGravitySoftening=1.0 / 30.0,
                             ShortRangeForceWindowType=sq.SHORTRANGE_FORCE_WINDOW_TYPE_EXACT)


def check_accns(pair, total):
    """check_accns + find_means, tests/test_gravity.cpp:78-119: errors in units of mean |a_direct|."""
    meanacc = np.abs(pair).mean()
    err = np.abs(pair - total) / meanacc
    return err.mean(), err.max()


def force_err(a, ref):
    """runtests.cpp:126-170: | |F|/|F_ref| - 1 | per particle."""
    na = np.linalg.norm(a, axis=1)
    nr = np.linalg.norm(ref, axis=1)
    ok = nr > 0
    return np.abs(na[ok] / nr[ok] - 1)
