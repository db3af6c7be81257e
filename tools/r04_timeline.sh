#!/bin/bash
# per-launch timeline of one bf16 training step (rocprofv3 kernel trace -> tools/trace_step.py)
set -o pipefail
TAG=${1:-r04tl}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/bench.py --mode train --dtype bf16 --steps 12 --warmup 4 --no-cpu-baseline > $O/trace.log 2>&1
echo "rocprof rc=$?"
python $R/tools/trace_step.py $O/trace > $O/step_timeline.txt
find $O -name "*kernel_trace.csv" -delete
tail -3 $O/step_timeline.txt
