#!/bin/bash
# tools/ab_env.sh VAR=VALUE [runs] : the bench (no CPU baseline, no SPH figures) alternately as built and with VAR=VALUE in the environment,
# `runs` times each (default 2); prints ms/step, walk ms and the PM phases.  Design probe, runs on the GPU box.
kv=$1; runs=${2:-2}
for i in $(seq $runs); do
  for which in default "$kv"; do
    if [ "$which" = default ]; then python bench.py --no-cpu-baseline --no-sph --steps 3 > gpurun_out/ab.json 2> gpurun_out/ab.err
    else env "$kv" python bench.py --no-cpu-baseline --no-sph --steps 3 > gpurun_out/ab.json 2> gpurun_out/ab.err; fi
    python - <<PY
import json
try:
    d=json.load(open("gpurun_out/ab.json")); k=d["kernels"]
    print("%-24s step %.2f walk %.2f pm %s" % ("$which", d["ms_per_step"], k["tree_walk_ms"], {a: round(b,2) for a,b in k["pm_ms"].items() if not isinstance(b, str)}), flush=True)
except Exception as e:
    print("$which failed", e, flush=True)
PY
  done
done
