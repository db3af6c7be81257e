#!/bin/bash
# GAN iteration on the GPU box: parity suites, the GAN line, kernel stats
set -o pipefail
TAG=${1:-r02g}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_gan.py tests/test_gpu_mixed.py tests/test_gpu_distributed.py tests/test_gpu_train.py -x -q -m gpu > $O/pytest_gan.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest_gan.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python bench.py --mode gan --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_gan_bf16.json 2> $O/bench_gan.err; echo "bench rc=$?"; cut -c1-330 $O/bench_gan_bf16.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_gan -- python $R/bench.py --mode gan --dtype bf16 --steps 4 --warmup 3 --no-cpu-baseline > $O/prof_gan.log 2>&1
find $O -name "*kernel_trace.csv" -delete
python - <<PY
import csv, glob
for fn in glob.glob("$O/prof_gan/**/*kernel_stats.csv", recursive=True):
    rows = sorted(csv.DictReader(open(fn)), key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    n = sum(int(r["Calls"]) for r in rows)
    print("kernel time total ms %.1f launches %d (7 iterations: %.0f per iteration)" % (tot / 1e6, n, n / 7.0))
    for r in rows[:24]:
        print("%-84s calls %5s avg %8.1f us %5.1f %%" % (r["Name"][:84], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    at = [r for r in rows if "at::native" in r["Name"]]
    print("ATen kernels: %d launches, %.2f %% of time" % (sum(int(r["Calls"]) for r in at), sum(float(r["Percentage"]) for r in at)))
PY
