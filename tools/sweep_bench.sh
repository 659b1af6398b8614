#!/bin/bash
# tools/sweep_bench.sh VAR v1 v2 ... : one bench.py run (no CPU baseline, no SPH figures) per value of the environment knob VAR;
# prints ms/step, walk ms, PM phases.  Design probe, runs on the GPU box.
var=$1; shift
for v in "$@"; do
  tag=$(echo "$v" | tr '/ ' '__' | tail -c 40)
  env $var=$v python bench.py --no-cpu-baseline --no-sph --steps 3 > gpurun_out/sweep_${var}_$tag.json 2> gpurun_out/sweep_${var}_$tag.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/sweep_${var}_$tag.json")); k=d["kernels"]
    print("$var=$v step %.2f walk %.2f pm %s" % (d["ms_per_step"], k["tree_walk_ms"], {a: round(b,2) for a,b in k["pm_ms"].items() if not isinstance(b, str)}), flush=True)
except Exception as e:
    print("$var=$v failed", e, flush=True)
PY
done
