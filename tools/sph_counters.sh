#!/bin/bash
# tools/sph_counters.sh [TAG] : kernel trace + SQ counter passes over the SPH kernels of tools/bench_sph.py 128 uniform 2 (density Hsml loop + hydro).
# Prints per kernel: average duration, waves, VALU / SALU / LDS instructions per wave, VALU busy, wait fractions.  GPU box, repo root.
set -o pipefail
TAG=${1:-sph}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/sph_counters_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
S="python3 $ROOT/tools/bench_sph.py 128 uniform 2"
one() { find "$1" -name "*$2" | head -1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_s -o s -- $S > $OUT/stats.log 2>&1
cp "$(one $OUT/p_s kernel_stats.csv)" $OUT/kernel_stats.csv
grep -E "density:|hydro:" $OUT/stats.log
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o q -- $S > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; continue; }
  python3 $ROOT/tools/pmc_summary.py pmc "$(one $OUT/p$i counter_collection.csv)" > $OUT/c$i.json
  rm -rf $OUT/p$i
done
rm -rf $OUT/p_s
python3 - $OUT <<'PY'
import csv, json, sys
out = sys.argv[1]
st = {r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]: r for r in csv.DictReader(open(out + "/kernel_stats.csv"))}
c1, c2 = json.load(open(out + "/c1.json")), json.load(open(out + "/c2.json"))
for k, v in c1.items():
    if not k.startswith("sph_") or k not in st:
        continue
    w = v.get("SQ_WAVES", 1.0)
    e2 = c2.get(k, {})
    busy = v.get("SQ_INSTS_VALU", 0) * 4.0 / max(1.0, v.get("SQ_BUSY_CYCLES", 0) / 32.0 * 1024.0)
    wc = max(1.0, v.get("SQ_WAVE_CYCLES", 1.0))
    print("%-60s %8.3f ms x %3s | waves %7d | per wave: VALU %7.0f SALU %7.0f LDS %6.0f VMEM_RD %5.0f | VALU busy %.2f | resident waves/SIMD %.1f | wait_any %.2f wait_inst %.2f active %.2f lds_conflict/lds_active %.2f"
          % (k[:60], float(st[k]["AverageNs"]) / 1e6, st[k]["Calls"], w, v.get("SQ_INSTS_VALU", 0) / w, v.get("SQ_INSTS_SALU", 0) / w, v.get("SQ_INSTS_LDS", 0) / w,
             v.get("SQ_INSTS_VMEM_RD", 0) / w, busy, wc / max(1.0, v.get("SQ_BUSY_CYCLES", 1) / 32.0 * 1024.0),
             e2.get("SQ_WAIT_ANY", 0) / wc, e2.get("SQ_WAIT_INST_ANY", 0) / wc, e2.get("SQ_ACTIVE_INST_ANY", 0) / wc,
             e2.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, e2.get("SQ_ACTIVE_INST_LDS", 1))))
PY
