#!/bin/bash
# conv-kernel iteration on the GPU box: the bit-exact suites that cover sq_conv_f32_v2.hip, then the inference line
set -o pipefail
TAG=${1:-r02c}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py tests/test_gpu_gan.py tests/test_gpu_jobs.py tests/test_gpu_frontend.py -x -q -m gpu > $O/pytest_conv.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest_conv.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --no-end-to-end > $O/bench_infer.json 2> $O/bench_infer.err; echo "bench rc=$?"; cut -c1-1400 $O/bench_infer.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_infer -- python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-end-to-end > $O/prof_infer.log 2>&1
find $O -name "*kernel_trace.csv" -delete
python - <<PY
import csv, glob
for fn in glob.glob("$O/prof_infer/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(fn)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:14]:
        print("%-90s calls %5s avg %9.1f us  %5.1f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
