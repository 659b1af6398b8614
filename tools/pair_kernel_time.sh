#!/bin/bash
# tools/pair_kernel_time.sh [LIBDIR] : average duration of grav_pair_kernel and of the production walk kernel in a short bench run
# (rocprofv3 --kernel-trace --stats), with the in-tree libraries or with the build in LIBDIR.  GPU box, repo root.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pair_time; mkdir -p $OUT; export TMPDIR=/tmp
[ -n "$1" ] && export SHQ_LIBDIR=$ROOT/$1
cd /tmp
rm -rf $OUT/ps
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ps -o s -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sph > $OUT/ps.log 2>&1
f=$(find $OUT/ps -name "*kernel_stats.csv" | head -1)
echo "== ${1:-tree}"
python3 - "$f" <<PY
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if n.startswith("grav_pair_kernel") or n.startswith("grav_walk_exact_kernel"):
        print("%-95s %3s calls  %.3f ms" % (n[:n.find("(")] if "(" in n else n, r["Calls"], float(r["AverageNs"]) / 1e6))
PY
rm -rf $OUT/ps
