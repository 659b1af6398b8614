#!/usr/bin/env python3
"""How far may a walk's opening decisions differ from the reference's before the north-star bar (rms |dF| / |F| < 1e-3) breaks?

The script behind DESIGN.md section 3.1(c) (checker infrastructure: it runs the oracle, so it lives under tests/).  On the
S-cluster at n^3 particles (default 64^3, Nmesh 3 n, ErrTolForceAcc 0.005, relative criterion after a Barnes-Hut seeding walk):
  * the reference walk (oracle) at ErrTolForceAcc 0.005 against the same walk at 0.004 and 0.0025 — a walk that opens a few per cent
    more nodes, which is what any conservative wave-shared criterion does;
  * the reference walk against the fully open tree (every leaf particle, ErrTolForceAcc -> 0 and theta -> 0), i.e. the walk's own
    truncation error.
Total force = short-range tree force + PM force (the criterion's OldAcc and the comparison both use the total).

  python tests/criterion_sensitivity.py [n]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import shenqi_amd as sq  # noqa: E402
import orc  # noqa: E402

G = 43.0071
RHO0 = 0.3 * 3 * 0.1 * 0.1 / (8 * np.pi * G)


def rms_rel(a, b):
    return float(np.sqrt(np.mean(np.sum((a - b) ** 2, axis=1) / np.sum(b**2, axis=1))))


def main():
    n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    n, L, nmesh = n1**3, 1.0, 3 * n1
    pos = sq.synth_positions("cluster", n, seed=20240601, L=L)
    pos = pos[sq.hilbert_order(pos, L)]
    pman = sq.PartManager(n, L)
    P = pman.Base
    P["Pos"], P["Type"], P["Mass"] = pos, 1, 1.0
    mass = np.ascontiguousarray(P["Mass"])
    tree = sq.force_tree_full(pman)

    def params(errtol, usebh, theta_max=0.9):
        sq.set_gravshort_treepar(ErrTolForceAcc=errtol, BHOpeningAngle=0.175, MaxBHOpeningAngle=theta_max, TreeUseBH=usebh, Rcut=6.0)
        sq.gravshort_set_softenings(L / n1)      # after the parameters: the softening is a fraction of them
        return sq.make_grav_params(L, 1.5, nmesh, G, RHO0)

    t0 = time.time()
    seed, _, _ = orc.grav_walk(tree.Nodes_base, tree.firstnode, pos, mass, np.zeros(n), params(0.005, 1))
    gpm = orc.pm_force(pos, mass, nmesh, L, 1.5, G)[0]
    oldacc = np.linalg.norm(seed * G + gpm, axis=1) / G
    print("seeding walk + PM: %.1f s" % (time.time() - t0), flush=True)
    res = {}
    for errtol in (0.005, 0.004, 0.0025):
        t0 = time.time()
        acc, _, nint = orc.grav_walk(tree.Nodes_base, tree.firstnode, pos, mass, oldacc, params(errtol, 0))
        res[errtol] = (acc * G + gpm, nint.mean())
        print("ErrTolForceAcc %.4f: %.1f interactions per target (%.1f s)" % (errtol, nint.mean(), time.time() - t0), flush=True)
    t0 = time.time()
    # fully open: the relative criterion opens everything for ErrTolForceAcc -> 0 (mass len^2 > r^4 * 0 for every node)
    full, _, nintf = orc.grav_walk(tree.Nodes_base, tree.firstnode, pos, mass, oldacc, params(1e-300, 0))
    full = full * G + gpm
    print("fully open tree: %.1f interactions per target (%.1f s)" % (nintf.mean(), time.time() - t0), flush=True)
    ref, nref = res[0.005]
    print("reference walk (0.005) vs fully open tree:           rms |dF|/|F| = %.2e   (the walk's own truncation error)" % rms_rel(ref, full))
    for errtol in (0.004, 0.0025):
        f, ni = res[errtol]
        print("walk at %.4f (+%.0f %% interactions) vs reference walk: rms |dF|/|F| = %.2e   (north-star bar 1e-3)"
              % (errtol, 100 * (ni / nref - 1), rms_rel(f, ref)))


if __name__ == "__main__":
    main()
