#!/bin/bash
# tools/sph_cluster_stats.sh : kernel trace of tools/bench_sph.py 128 cluster 2 (S-cluster gas: the first-call Hsml loop, a steady-state density iteration, hydro):
# total time and calls per kernel.  GPU box, repo root.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/sph_cluster; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp; rm -rf $OUT/p
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o s -- python3 $ROOT/tools/bench_sph.py 128 cluster 2 > $OUT/log.txt 2>&1
grep -E "density:|hydro:" $OUT/log.txt
f=$(find $OUT/p -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:16]:
    print("%9.3f ms total %5s calls avg %8.3f max %8.3f  %s" % (float(r["TotalDurationNs"]) / 1e6, r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MaxNs"]) / 1e6,
          r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]))
PY
cp "$f" $OUT/kernel_stats.csv; rm -rf $OUT/p
