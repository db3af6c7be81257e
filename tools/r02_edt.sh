#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-edt}; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_weightmap.py -x -q -m gpu 2>&1 | tail -1
python bench.py --mode weightmap --no-cpu-baseline | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --mode weightmap --no-cpu-baseline --steps 20 --warmup 5 > $O/prof.log 2>&1
find $O -name "*kernel_trace.csv" -delete
python - <<PY
import csv, glob
for fn in glob.glob("$O/prof/**/*kernel_stats.csv", recursive=True):
    for r in sorted(csv.DictReader(open(fn)), key=lambda r: -float(r["TotalDurationNs"]))[:6]:
        print("%-90s calls %5s avg %8.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
