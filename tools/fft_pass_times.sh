#!/bin/bash
# tools/fft_pass_times.sh [ENV=VALUE ...] : rocprofv3 kernel trace of the bench (10 steps), average duration of every FFT pass kernel, the deposit
# and the walk, for the library as built and once more per ENV=VALUE given.  GPU box, repo root.
set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/fft_pass_times; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
run() {
  tag=$1; shift
  rm -rf $OUT/p_$tag
  if [ -n "$1" ]; then export "$@"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_$tag -o s -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-sph > $OUT/$tag.log 2>&1 || { echo "$tag failed"; tail -3 $OUT/$tag.log; }
  if [ -n "$1" ]; then for kv in "$@"; do unset "${kv%%=*}"; done; fi
  f=$(find $OUT/p_$tag -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$tag" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0.0
print("==", sys.argv[2])
for r in rows:
    name = r["Name"]
    if any(k in name for k in ("fft_", "pm_deposit", "grav_walk_exact", "grav_pair", "pm_zero", "pm_readout")):
        avg = float(r["AverageNs"]) / 1e6
        if "fft_" in name and int(r["Calls"]) >= 10:
            tot += avg
        print("%8.3f ms x %4s  %s" % (avg, r["Calls"], name[:110]))
print("sum of the FFT passes (rows with >= 10 calls): %.3f ms" % tot)
PY
  cp "$f" $OUT/${tag}_kernel_stats.csv
  rm -rf $OUT/p_$tag
}
run default
for kv in "$@"; do run "$(echo $kv | tr '=/' '__')" "$kv"; done
