#!/bin/bash
# tools/fft_counters.sh : SQ counter passes over the FFT pass kernels (each group its own rocprofv3 run).  GPU box, repo root.
set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/fft_counters; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
B="python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sph"
one() { find "$1" -name "*$2" | head -1; }
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o q -- $B > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; continue; }
  python3 $ROOT/tools/pmc_summary.py pmc "$(one $OUT/p$i counter_collection.csv)" > $OUT/c$i.json
  rm -rf $OUT/p$i
  python3 - <<PY
import json
d=json.load(open("$OUT/c$i.json"))
for k,v in d.items():
    if k.startswith("fft_pass") and "768" in k: print(k[:40], {a:(round(b/1e6,1) if isinstance(b,float) else b) for a,b in v.items()})
PY
done
