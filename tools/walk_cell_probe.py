"""256^3 S-cluster: the exact walk with its 64-target groups aligned to tree cells.  The verdict of round 1 named it as an untried
lever: a wave that owns whole tree cells of <= cap targets (padded with idle lanes where the next cell does not fit) instead of 64
consecutive particles of the Peano-Hilbert order might take more coherent opening decisions.  Variants: the same particles in cell
(pre-order) order without padding; whole cells packed into waves, cap 64 / 32 / 16.  Needs SHQ_WALK_PADDING=1 (negative list entries
are idle lanes)."""
import ctypes as C
import os
import sys

import numpy as np

os.environ["SHQ_WALK_PADDING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", str(len(os.sched_getaffinity(0))))
import shenqi_amd as sq  # noqa: E402
from shenqi_amd import capi  # noqa: E402

n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 256
G = 43.0071
RHO0 = 0.3 * 3 * 0.1 * 0.1 / (8 * np.pi * G)
n = n1**3
L = 1.0
nmesh = 3 * n1
pos = sq.synth_positions("cluster", n, L=L)
pos = pos[sq.hilbert_order(pos, L)]
pman = sq.PartManager(n, L)
pman.Base["Pos"] = pos
pman.Base["Type"] = 1
pman.Base["Mass"] = 1.0
sq.set_gravshort_treepar(ErrTolForceAcc=0.005, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=1, Rcut=6.0)
sq.gravshort_set_softenings(L / n1)
gp_bh = sq.make_grav_params(L, 1.5, nmesh, G, RHO0)
sq.set_gravshort_treepar(ErrTolForceAcc=0.005, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=0, Rcut=6.0)
gp = sq.make_grav_params(L, 1.5, nmesh, G, RHO0)
pmp = sq.PMParams(nmesh, 0, L, 1.5, G)
pv = pman.view()
c = sq.Context(0)
capi.check(capi.hip.shq_particles_upload(c.h, C.byref(pv)))
sq.tree_build_device(c, L)
capi.check(capi.hip.shq_pm_run(c.h, C.byref(pmp)))
capi.check(capi.hip.shq_grav_short_run(c.h, C.byref(gp_bh), None, 0, 1, 0))
capi.check(capi.hip.shq_grav_refresh_oldacc(c.h, G))
capi.check(capi.hip.shq_grav_short_run(c.h, C.byref(gp), None, 0, 1, 0))
capi.check(capi.hip.shq_grav_refresh_oldacc(c.h, G))
s = sq.WalkStats()
acc0 = None


def t(label, active):
    ms = []
    act = None if active is None else np.ascontiguousarray(active, dtype=np.int32)
    for _ in range(3):
        capi.check(capi.hip.shq_grav_short_run(c.h, C.byref(gp), capi.ptr(act), 0 if act is None else len(act), 0, 0))
        capi.check(capi.hip.shq_grav_short_download(c.h, None, None, None, C.byref(s)))
        ms.append(s.kernel_ms)
    acc = np.zeros((n, 3))
    capi.check(capi.hip.shq_grav_short_download(c.h, capi.ptr(acc), None, None, None))
    global acc0
    if acc0 is None:
        acc0 = acc
    same = bool(np.array_equal(acc, acc0))
    lanes = n if act is None else len(act)
    print("%-58s %7.2f ms  %5.1f %% lanes used, %.1f interactions/target, forces identical: %s" %
          (label, min(ms), 100.0 * n / lanes, s.ninteractions / n, same), flush=True)


t("64 consecutive particles of the Hilbert order", None)
firstnode = n + 7
nodes, father = sq.tree_download(c, firstnode, n)
nn = len(nodes)
fa = nodes["father"].astype(np.int64) - firstnode
fa[0] = 0
sib = nodes["sibling"].astype(np.int64) - firstnode
end = np.where(nodes["sibling"] >= 0, sib, -1)
end[0] = nn
while (end < 0).any():
    miss = end < 0
    end[miss] = end[fa[miss]]
leafcount = np.bincount(father.astype(np.int64) - firstnode, minlength=nn)
cs = np.concatenate([[0], np.cumsum(leafcount)])
cnt = cs[end] - cs[np.arange(nn)]
assert cnt[0] == n
leaf_of = father.astype(np.int64) - firstnode
for cap in (64, 32, 16):
    top = (cnt <= cap) & ((cnt[fa] > cap) | (np.arange(nn) == 0)) & (cnt > 0)
    tops = np.flatnonzero(top)
    cell = tops[np.searchsorted(tops, leaf_of, side="right") - 1]
    assert (leaf_of < end[cell]).all() and (leaf_of >= cell).all()
    order = np.argsort(cell, kind="stable")
    sizes = cnt[tops]
    if cap == 64:
        t("cell (pre-order) order, no padding", order)
    # whole cells packed greedily into waves of 64 lanes
    wave_of = np.zeros(len(tops), dtype=np.int64)
    start = np.zeros(len(tops), dtype=np.int64)
    w, fill = 0, 0
    for k, sz in enumerate(sizes.tolist()):
        if fill + sz > 64:
            w += 1
            fill = 0
        wave_of[k] = w
        start[k] = fill
        fill += sz
    nw = w + 1
    act = np.full(nw * 64, -1, dtype=np.int32)
    # position of every particle inside its cell: rank among the sorted-by-cell order
    cell_sorted = cell[order]
    first_of_cell = np.searchsorted(cell_sorted, tops)
    kcell = np.searchsorted(tops, cell_sorted)
    within = np.arange(n) - first_of_cell[kcell]
    act[wave_of[kcell] * 64 + start[kcell] + within] = order
    t("whole cells of <= %d targets packed into waves (%d cells)" % (cap, len(tops)), act)
    # one cell per wave would be: report the lane use it would have
    print("    (one cell per wave would use %.1f %% of the lanes)" % (100.0 * n / (len(tops) * 64)), flush=True)
c.close()
