#!/usr/bin/env python3
"""Design probe (not product): evaluation rounds per wave of the exact union walk under deferred-evaluation policies
(tools/walk_defer_sim.cpp), on the bench's S-cluster at n^3 particles, CPU only.

  g++ -O2 -fopenmp -shared -fPIC -Iinclude tools/walk_defer_sim.cpp -o build/libwalk_defer_sim.so
  python tools/walk_defer_sim.py [n] [wave_stride]
"""
import ctypes as C
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


def main():
    n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    stride = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    n, L = n1**3, 1.0
    nmesh = 3 * n1
    cache = "/tmp/walk_defer_state_%d.npz" % n1
    pos = sq.synth_positions("cluster", n, seed=20240601, L=L)
    pos = pos[sq.hilbert_order(pos, L)]
    pman = sq.PartManager(n, L)
    P = pman.Base
    P["Pos"] = pos
    P["Type"] = 1
    P["Mass"] = 1.0
    sq.set_gravshort_treepar(ErrTolForceAcc=0.005, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=1, Rcut=6.0)
    sq.gravshort_set_softenings(L / n1)
    gp_bh = sq.make_grav_params(L, 1.5, nmesh, G, RHO0)
    sq.set_gravshort_treepar(ErrTolForceAcc=0.005, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=0, Rcut=6.0)
    gp_rel = sq.make_grav_params(L, 1.5, nmesh, G, RHO0)
    t0 = time.time()
    tree = sq.force_tree_full(pman)
    print("tree: %d nodes in %.1f s" % (tree.numnodes, time.time() - t0), flush=True)
    mass = np.ascontiguousarray(P["Mass"])
    if os.path.exists(cache):
        oldacc = np.load(cache)["oldacc"]
    else:
        t0 = time.time()
        acc, _, _ = orc.grav_walk(tree.Nodes_base, tree.firstnode, pos, mass, np.zeros(n), gp_bh)
        print("BH walk %.1f s" % (time.time() - t0), flush=True)
        t0 = time.time()
        gpm, _ = orc.pm_force(pos, mass, nmesh, L, 1.5, G)[:2]
        print("PM %.1f s" % (time.time() - t0), flush=True)
        oldacc = np.linalg.norm(acc * G + gpm, axis=1) / G
        np.savez(cache, oldacc=oldacc)
    lib = C.CDLL(os.path.join(ROOT, "build", "libwalk_defer_sim.so"))
    confs = [(32, -1, 65), (64, -1, 65), (64, -33, 65), (32, 0, 65), (64, 0, 65), (64, 32, 65), (128, 64, 65), (1 << 20, 1, 65)]
    for T in (8, 16, 24, 32, 48):
        confs += [(32, 0, T), (64, 0, T), (64, 32, T), (128, 64, T)]
    R = np.array([c[0] for c in confs], dtype=np.int32)
    Gn = np.array([c[1] for c in confs], dtype=np.int32)
    Tn = np.array([c[2] for c in confs], dtype=np.int32)
    nw = (n + 63) // 64
    ns = (nw + stride - 1) // stride
    nout = 5 + 3 * len(confs)
    out = np.zeros((ns, nout), dtype=np.int64)
    posc = np.ascontiguousarray(pos)
    t0 = time.time()
    lib.walk_defer_sim(C.c_void_p(tree.Nodes_base.ctypes.data), C.c_int64(tree.firstnode), C.c_void_p(posc.ctypes.data),
                       C.c_void_p(oldacc.ctypes.data), C.c_int64(n), C.byref(gp_rel), C.c_int64(stride), C.c_int(len(confs)),
                       C.c_void_p(R.ctypes.data), C.c_void_p(Gn.ctypes.data), C.c_void_p(Tn.ctypes.data), C.c_void_p(out.ctypes.data), C.c_int(nout))
    print("sim %.1f s, %d waves sampled" % (time.time() - t0, ns))
    m = out.mean(axis=0)
    print("per wave: visits %.0f  accepting visits %.0f  leaf rounds %.0f  => rounds now %.0f" % (m[0], m[1], m[2], m[1] + m[2]))
    print("interactions per target %.1f  (ideal rounds %.0f)  fullest lane %.0f" % (m[3] / 64, m[3] / 64, m[4]))
    now = m[0] * 84 + m[1] * 118 + m[2] * 140
    print("cost model (cycles per wave): visit 84, immediate node round 118, immediate leaf round 140, drained round 176 (leaf-only ring: 156)")
    print("now: %.0f cycles" % now)
    for c, (r, g, T) in enumerate(confs):
        kind = "leaf-only" if g < 0 else "all      "
        gg = -g - 1 if g < 0 else g
        rounds, imm = m[5 + 3 * c], m[7 + 3 * c]
        drained = rounds - imm
        cost = m[0] * 84 + imm * 118 + drained * (156 if g < 0 else 176)
        print("%s ring %7d gran %3d defer-below %2d: rounds %.0f = %.0f immediate + %.0f drained  drains %.0f  lane use %.2f  cost %.0f (%.3f of now)"
              % (kind, r, gg, T, rounds, imm, drained, m[6 + 3 * c], m[3] / (64 * rounds), cost, cost / now))


if __name__ == "__main__":
    main()
