#!/bin/bash
# round 4: bf16 conv kernel diet -- parity tests of everything that goes through sq_conv_bf16.hip, then the train / GAN lines
set -o pipefail
TAG=${1:-r04bf}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_train.py tests/test_gpu_gan_bf16.py tests/test_gpu_gan.py -x -q -m gpu > $O/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit 1
for m in "train bf16" "gan bf16"; do
  set -- $m
  timeout -k 10 300 python bench.py --mode $1 --dtype $2 --no-cpu-baseline > $O/bench_$1.json 2> $O/bench_$1.err; echo "bench $1 rc=$?"
  python -c "
import json; d=json.load(open('$O/bench_$1.json')); print('$1', d['ms_per_step'], d['value'])"
done
