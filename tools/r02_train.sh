#!/bin/bash
# training-step iteration on the GPU box: bf16 / train parity suites, then the training line (+ profile)
set -o pipefail
TAG=${1:-r02t}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_train.py tests/test_gpu_distributed.py -x -q -m gpu > $O/pytest_train.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest_train.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python bench.py --mode train --dtype bf16 --no-cpu-baseline > $O/bench_train_bf16.json 2> $O/bench_train.err; echo "bench rc=$?"; cut -c1-300 $O/bench_train_bf16.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -- python $R/bench.py --mode train --dtype bf16 --steps 30 --warmup 5 --no-cpu-baseline > $O/prof_train.log 2>&1
find $O -name "*kernel_trace.csv" -delete
python - <<PY
import csv, glob
for fn in glob.glob("$O/prof_train/**/*kernel_stats.csv", recursive=True):
    rows = sorted(csv.DictReader(open(fn)), key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("kernel time total ms", tot / 1e6)
    for r in rows[:22]:
        print("%-84s calls %5s avg %8.1f us %5.1f %%" % (r["Name"][:84], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
