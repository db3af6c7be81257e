#!/bin/bash
# Round-2 check on the GPU box: parity suite, smoke, every bench line.  Usage: bash tools/r02_check.sh TAG
set -o pipefail
TAG=${1:-r02a}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest_gpu.log
[ $rc -eq 124 ] && exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout -k 10 300 python bench.py > $O/bench_infer.json 2> $O/bench_infer.err; echo "bench rc=$?"; cat $O/bench_infer.json
SQ_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_infer_2rank_gloo.json 2> $O/bench_infer_2rank.err; echo "2rank rc=$?"; cat $O/bench_infer_2rank_gloo.json
timeout -k 10 300 python bench.py --mode train --dtype bf16 > $O/bench_train_bf16.json 2> $O/bench_train_bf16.err; echo "train rc=$?"; cat $O/bench_train_bf16.json
timeout -k 10 300 python bench.py --mode train --dtype bf16 --scaling strong --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_train_bf16_strong.json 2> $O/bench_train_bf16_strong.err; echo "train strong rc=$?"; cat $O/bench_train_bf16_strong.json
timeout -k 10 400 python bench.py --mode gan --dtype bf16 --steps 20 --warmup 5 > $O/bench_gan_bf16.json 2> $O/bench_gan_bf16.err; echo "gan rc=$?"; cat $O/bench_gan_bf16.json
