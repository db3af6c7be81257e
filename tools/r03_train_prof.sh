#!/bin/bash
O=gpurun_out/r03trainprof; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python $R/bench.py --mode train --dtype bf16 --steps 6 --warmup 3 --no-cpu-baseline > $R/$O/prof.log 2>&1; echo "prof rc=$?"
cd $R
f=$(ls $O/prof/*/*kernel_trace.csv | head -1)
python tools/r03_iter_kernels.py $f adam_kernel 1 > $O/iter.txt; head -40 $O/iter.txt
