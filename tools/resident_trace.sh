#!/bin/bash
# tools/resident_trace.sh : kernel timeline (rocprofv3 --kernel-trace) of the LAST moving-particle step of the bench's resident loop: drift -> {PM on the
# second stream | tree build} -> walk + pair kernel -> kicks.  Prints every kernel longer than 50 us and the gaps.  GPU box, repo root.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/resident_trace; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp; rm -rf $OUT/t
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o k -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sph > $OUT/t.log 2>&1
f=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the resident loop's last step: from the last drift_kernel to the kick_pm_kernel after it
drifts = [i for i, r in enumerate(rows) if "drift_kernel" in r["Kernel_Name"]]
i0 = drifts[-1]
i1 = next(i for i in range(i0, len(rows)) if "kick_pm_kernel" in rows[i]["Kernel_Name"])
prev = [i for i in range(i0) if "kick_pm_kernel" in rows[i]["Kernel_Name"]]
t0 = int(rows[i0]["Start_Timestamp"])
if prev:
    print("previous step's kick_pm ended %.3f ms before this drift started" % ((t0 - int(rows[prev[-1]]["End_Timestamp"])) / 1e6))
small, last_end = 0.0, t0
for r in rows[i0:i1 + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if e - s > 50000:
        print("%-70s q%-3s start %8.3f dur %7.3f ms" % (n[:70], r.get("Queue_Id", "?"), (s - t0) / 1e6, (e - s) / 1e6))
    else:
        small += (e - s) / 1e6
print("kernels under 50 us: %.3f ms in all; step (drift start -> kick_pm end): %.3f ms" % (small, (int(rows[i1]["End_Timestamp"]) - t0) / 1e6))
PY
rm -rf $OUT/t
