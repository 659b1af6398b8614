#!/bin/bash
# tools/readout_xcd_sweep.sh K... : the bench (no CPU baseline, no SPH figures) with SHQ_PM_XCD_K = each K (workgroups per XCD chunk of
# pm_readout_kernel; 0: plain order); prints ms/step and the PM phases.  Design probe, runs on the GPU box.
for k in "$@"; do
  SHQ_PM_XCD_K=$k python bench.py --no-cpu-baseline --no-sph --steps 5 > gpurun_out/ab.json 2> gpurun_out/ab.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/ab.json")); k=d["kernels"]
    print("K=%-5s step %.2f walk %.2f pm %s" % ("$k", d["ms_per_step"], k["tree_walk_ms"], {a: round(b,3) for a,b in k["pm_ms"].items() if not isinstance(b, str)}), flush=True)
except Exception as e:
    print("$k failed", e, flush=True)
PY
done
