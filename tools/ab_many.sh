#!/bin/bash
# tools/ab_many.sh DIR... : the bench (no CPU baseline, no SPH figures) with the in-tree libraries and with each alternative build
# (SHQ_LIBDIR); prints ms/step, walk ms and the PM phases.  Design probe, runs on the GPU box.
for which in tree "$@"; do
  if [ "$which" = tree ]; then unset SHQ_LIBDIR; else export SHQ_LIBDIR=$PWD/$which; fi
  python bench.py --no-cpu-baseline --no-sph --steps 3 > gpurun_out/ab.json 2> gpurun_out/ab.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/ab.json")); k=d["kernels"]
    print("%-14s step %.2f walk %.2f pm %s" % ("$which", d["ms_per_step"], k["tree_walk_ms"], {a: round(b,2) for a,b in k["pm_ms"].items() if not isinstance(b, str)}), flush=True)
except Exception as e:
    print("$which failed", e, flush=True)
PY
done
