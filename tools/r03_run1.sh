#!/bin/bash
# round 3, GPU call 1: changed test modules, the learning-rate probe, the default bench line
set -o pipefail
TAG=${1:-r03a}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_bf16.py tests/test_gpu_distributed.py tests/test_gpu_jobs.py -x -q -m gpu > $O/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 500 python tools/r03_lr_probe.py > $O/lr_probe.log 2>&1; echo "probe rc=$?"; tail -12 $O/lr_probe.log | cut -c1-400
