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
grep "grav_pair_kernel\|grav_walk_exact_kernel<true, false, 2, 0, false, false, true, true, true, true" "$f" | cut -d, -f1-4 | sed 's/(anonymous namespace):://; s/(.*)"/"/' | cut -c1-150
rm -rf $OUT/ps
