#!/bin/bash
# tools/walk_counters.sh OUTDIR : SQ counter passes of the bench's production walk (what its waves wait for).  Run on the GPU box from the
# repo root; every counter group is its own rocprofv3 run (--pmc is never combined with a trace domain).
set -e -o pipefail
ROOT=$(pwd)
OUT=$ROOT/${1:-gpurun_out/walk_counters}
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
B="python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sph"
one() { find "$1" -name "*$2" | head -1; }
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY \
          --output-format csv -d $OUT/p1 -o q -- $B > $OUT/p1.log 2>&1
python3 $ROOT/tools/pmc_summary.py pmc "$(one $OUT/p1 counter_collection.csv)" > $OUT/sq1.json
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC \
          --output-format csv -d $OUT/p2 -o q -- $B > $OUT/p2.log 2>&1
python3 $ROOT/tools/pmc_summary.py pmc "$(one $OUT/p2 counter_collection.csv)" > $OUT/sq2.json
rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_IFETCH SQ_IFETCH_LEVEL \
          --output-format csv -d $OUT/p3 -o q -- $B > $OUT/p3.log 2>&1 || true
python3 $ROOT/tools/pmc_summary.py pmc "$(one $OUT/p3 counter_collection.csv)" > $OUT/sq3.json || true
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
python3 - <<PY
import json
for f in ("sq1","sq2","sq3"):
    try:
        d=json.load(open("$OUT/%s.json" % f))
    except Exception as e:
        print(f, "missing", e); continue
    for k,v in d.items():
        if k.startswith("grav_walk_exact_kernel<true, false, 2, 0, false, false"):
            print(f, k[:90]); print("   ", {a: b for a,b in v.items()})
PY
