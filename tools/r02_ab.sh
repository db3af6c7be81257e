#!/bin/bash
# A/B on one box, back to back: the library in tools/_exp/$1 against the in-tree one (profiled, per-kernel averages)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in A B A B; do
  if [ $v = A ]; then lib=$R/tools/_exp/$1/libsequitr_hip.so; else lib=$R/sequitr_amd/_build/libsequitr_hip.so; fi
  python $R/tools/bench_with_lib.py $lib --steps 60 --warmup 20 --no-cpu-baseline --no-end-to-end | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['frac'])"
done
for v in A B; do
  if [ $v = A ]; then lib=$R/tools/_exp/$1/libsequitr_hip.so; else lib=$R/sequitr_amd/_build/libsequitr_hip.so; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -- python $R/tools/bench_with_lib.py $lib --steps 40 --warmup 10 --no-cpu-baseline --no-end-to-end > $O/prof_$v.log 2>&1
  find $O -name "*kernel_trace.csv" -delete
  python - <<PY
import csv, glob
for fn in glob.glob("$O/prof_$v/**/*kernel_stats.csv", recursive=True):
    rows = sorted(csv.DictReader(open(fn)), key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:6]:
        print("$v %-70s avg %9.1f us" % (r["Name"][24:94], float(r["AverageNs"]) / 1e3))
PY
done
