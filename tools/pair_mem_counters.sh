#!/bin/bash
# tools/pair_mem_counters.sh : texture-addresser / L1 / L2 counters of grav_pair_kernel (what its lanes' scattered record loads cost).
# A group the hardware cannot collect in one pass makes rocprofv3 abort and hang: two counters per block, and a timeout per pass.
# GPU box, repo root; every counter group is its own rocprofv3 run (--pmc is never combined with a trace domain).
set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pair_mem; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
B="python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sph"
one() { find "$1" -name "*$2" | head -1; }
i=0
for grp in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum"; do
  i=$((i+1))
  echo "pass $i: $grp"
  timeout -k 10 150 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o q -- $B > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; continue; }
  python3 $ROOT/tools/pmc_summary.py pmc "$(one $OUT/p$i counter_collection.csv)" > $OUT/c$i.json
  rm -rf $OUT/p$i
  python3 - <<PY
import json
d=json.load(open("$OUT/c$i.json"))
for k,v in d.items():
    if k.startswith("grav_pair_kernel<true"): print({a:(round(b) if isinstance(b,float) else b) for a,b in v.items()}, flush=True)
PY
done
