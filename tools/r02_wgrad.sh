#!/bin/bash
# wgrad iteration: parity tests, micro-benchmark, SQ counters
set -o pipefail
TAG=${1:-wg}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_train.py tests/test_gpu_gan.py -x -q -m gpu -k "wgrad or training or mixed or config5" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 120 python tools/wgrad_bf16_bench.py 2>&1 | tee $O/bench.txt
bash tools/pmc_micro.sh $TAG/pmc tools/wgrad_one.py | grep -A2 "wgrad_bf16_kernel"
