#!/bin/bash
# round 3 inference A/B: parity tests of the touched kernels, then per-kernel stats of the headline pass
set -o pipefail
TAG=${1:-r03i}; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py -x -q -m gpu > $O/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit 1
for v in "$@"; do
  name=$(echo "$v" | tr ' =' '__')
  ( export $v; timeout -k 10 200 python bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-end-to-end --no-side-lines > $O/bench_$name.json 2> $O/bench_$name.err ) ; echo "bench[$v] rc=$?"
  python -c "
import json; d=json.load(open('$O/bench_$name.json')); print('$v', d['ms_per_step'], d['value'], d['roofline']['frac'])"
done
cd /tmp && export TMPDIR=/tmp
v="$1"
( export $v; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-end-to-end --no-side-lines > $O/prof.log 2>&1 )
find $O -name "*kernel_trace.csv" -delete
python - <<PY
import csv, glob
for fn in glob.glob("$O/prof/**/*kernel_stats.csv", recursive=True):
    rows = sorted(csv.DictReader(open(fn)), key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("kernel time total ms", tot / 1e6)
    for r in rows[:10]:
        print("%-74s calls %5s avg %8.1f us %5.1f %%" % (r["Name"][:74], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
