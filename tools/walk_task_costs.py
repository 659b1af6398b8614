#!/usr/bin/env python3
"""tools/walk_task_costs.py [n1] — per-task cost spread of the production walk and what task order does to the tail.

Runs the bench's 256^3 S-cluster through PM + Barnes-Hut seed + one relative-criterion walk, downloads the per-target interaction
counts, forms per-task (64 consecutive targets) costs — the wave walks the UNION of its lanes' trees, so a task's cost is between the
maximum and the sum of its lanes': both are listed — and replays the persistent launch (8192 waves taking tasks from per-XCD
counters in runs of 128) as list scheduling on the CPU, in the launch's order and longest-first.  Writes gpurun_out/walk_task_costs.json.
"""
import ctypes as C
import heapq
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import shenqi_amd as sq  # noqa: E402
from shenqi_amd import capi  # noqa: E402

G = 43.0071
RHO0 = 0.3 * 3 * 0.1 * 0.1 / (8 * np.pi * G)


def makespan(costs, order, workers):
    h = [0.0] * workers
    heapq.heapify(h)
    for t in order:
        heapq.heappush(h, heapq.heappop(h) + costs[t])
    return max(h)


def main():
    n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    n, L, nmesh = n1**3, 1.0, 3 * n1
    pos = sq.synth_positions("cluster", n, seed=20240601, L=L)
    pos = pos[sq.hilbert_order(pos, L)]
    pman = sq.PartManager(n, L)
    P = pman.Base
    P["Pos"], P["Type"], P["Mass"] = pos, 1, 1.0
    ctx = sq.Context()
    pv = pman.view()
    capi.check(capi.hip.shq_particles_upload(ctx.h, C.byref(pv)))
    sq.tree_build_device(ctx, L)
    pmp = sq.PMParams(nmesh, 0, L, 1.5, G)
    sq.set_gravshort_treepar(ErrTolForceAcc=0.005, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=1, Rcut=6.0)
    sq.gravshort_set_softenings(L / n1)
    gp_bh = sq.make_grav_params(L, 1.5, nmesh, G, RHO0)
    sq.set_gravshort_treepar(ErrTolForceAcc=0.005, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=0, Rcut=6.0)
    gp = sq.make_grav_params(L, 1.5, nmesh, G, RHO0)
    capi.check(capi.hip.shq_pm_run(ctx.h, C.byref(pmp)))
    capi.check(capi.hip.shq_grav_short_run(ctx.h, C.byref(gp_bh), None, 0, 1, 0))
    capi.check(capi.hip.shq_grav_refresh_oldacc(ctx.h, G))
    capi.check(capi.hip.shq_grav_short_run(ctx.h, C.byref(gp), None, 0, 1, 0))
    nint = np.zeros(n, dtype=np.int64)
    st = sq.WalkStats()
    capi.check(capi.hip.shq_grav_short_download(ctx.h, None, None, capi.ptr(nint), C.byref(st)))
    per = nint.reshape(-1, 64)
    out = {"n": n, "kernel_ms": st.kernel_ms, "interactions_per_target": float(nint.mean())}
    nt = per.shape[0]
    # the launch's order: region r = workgroup % 8 takes runs of 128 consecutive tasks, run k of region r = tasks [(8 k + r) 128, +128)
    launch_order = np.arange(nt)
    for name, c in (("sum", per.sum(axis=1).astype(np.float64)), ("max", per.max(axis=1).astype(np.float64))):
        mean_load = c.sum() / 8192
        res = {"task_cost_mean": float(c.mean()), "task_cost_max": float(c.max()), "task_cost_p99": float(np.percentile(c, 99)),
               "ideal_makespan": float(mean_load),
               "launch_order_makespan_over_ideal": makespan(c, launch_order, 8192) / mean_load,
               "longest_first_makespan_over_ideal": makespan(c, np.argsort(-c, kind="stable"), 8192) / mean_load,
               "largest_task_over_ideal_makespan": float(c.max() / mean_load)}
        out[name] = res
        print(name, json.dumps(res))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "walk_task_costs.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
