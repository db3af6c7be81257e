#!/bin/bash
# round 4: the level-0 f32 kernel family (sq_conv_f32_l0.hip) -- parity tests, then per-operator warm timing A/B
set -o pipefail
TAG=${1:-r04l0}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py -x -q -m gpu > $O/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -ne 0 ] && exit 1
echo "--- L0 kernel"; timeout -k 10 200 python tools/r03_op_bench.py level0 2>&1 | tee $O/op_l0.txt
echo "--- generic kernel (SQ_CONV_L0=0)"; SQ_CONV_L0=0 timeout -k 10 200 python tools/r03_op_bench.py level0 2>&1 | tee $O/op_v2.txt
for v in SQ_CONV_L0=1 SQ_CONV_L0=0; do
  ( export $v; timeout -k 10 200 python bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-end-to-end --no-side-lines > $O/bench_$v.json 2> $O/bench_$v.err ); echo "bench[$v] rc=$?"
  python -c "
import json; d=json.load(open('$O/bench_$v.json')); print('$v', d['ms_per_step'], d['value'], d['roofline']['frac'])"
done
