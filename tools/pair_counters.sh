#!/bin/bash
# tools/pair_counters.sh : SQ counters of grav_pair_kernel (bench with SHQ_WALK_SPARSE=1).  GPU box, repo root.
set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pair_counters; mkdir -p $OUT; export TMPDIR=/tmp SHQ_WALK_SPARSE=1; cd /tmp
B="python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sph"
one() { find "$1" -name "*$2" | head -1; }
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o q -- $B > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; continue; }
  python3 $ROOT/tools/pmc_summary.py pmc "$(one $OUT/p$i counter_collection.csv)" > $OUT/c$i.json
  rm -rf $OUT/p$i
  python3 - <<PY
import json
d=json.load(open("$OUT/c$i.json"))
for k,v in d.items():
    if k.startswith("grav_pair_kernel<true"): print({a:(round(b/262144) if isinstance(b,float) else b) for a,b in v.items()})
PY
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ps -o s -- $B > $OUT/ps.log 2>&1
grep "grav_pair_kernel\|grav_walk_exact_kernel<true, false, 2, 0, false, false, true, true, true, true" "$(one $OUT/ps kernel_stats.csv)" | cut -d, -f1-4 | cut -c1-200
rm -rf $OUT/ps
