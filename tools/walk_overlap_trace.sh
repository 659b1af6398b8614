#!/bin/bash
# tools/walk_overlap_trace.sh : start / end of the main walk and of the pair kernel in the last step of a short bench run (rocprofv3
# --kernel-trace): do the two run side by side?  GPU box, repo root.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/overlap_trace; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp; rm -rf $OUT/t
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o k -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sph > $OUT/t.log 2>&1
f=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<PY
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "grav_pair_kernel" in r["Kernel_Name"] or "grav_walk_exact_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print("%-78s queue %-4s start %9.3f ms  end %9.3f ms  grid %s wg %s" % (n[:78], r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))))
PY
rm -rf $OUT/t
