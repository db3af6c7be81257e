#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-p8f}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in centroids frontend; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/$m -- python $R/bench.py --mode $m --no-cpu-baseline --steps 20 --warmup 5 > $O/$m.log 2>&1
find $O -name "*kernel_trace.csv" -delete
python - <<PY
import csv, glob
print("== $m")
for fn in glob.glob("$O/$m/**/*kernel_stats.csv", recursive=True):
    for r in sorted(csv.DictReader(open(fn)), key=lambda r: -float(r["TotalDurationNs"]))[:8]:
        print("%-90s calls %5s avg %8.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
