#!/bin/bash
# per-launch timeline of one f32 inference step (rocprofv3 kernel trace -> tools/trace_step.py; the step ends with the head-fused launch)
set -o pipefail
TAG=${1:-r04itl}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-end-to-end --no-side-lines > $O/trace.log 2>&1
echo "rocprof rc=$?"
python $R/tools/trace_step.py $O/trace "conv_l0_kernelILi0ELi2E" > $O/infer_step_timeline.txt || python $R/tools/trace_step.py $O/trace "conv_l0_kernel<0, 2" > $O/infer_step_timeline.txt
find $O -name "*kernel_trace.csv" -delete
cat $O/infer_step_timeline.txt
