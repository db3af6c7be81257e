#!/bin/bash
# Round-end validation on the GPU box: parity suite, smoke, every bench line and the rocprof evidence.
# Usage (from the repo root on the box): bash tools/round_validate.sh rNN      -> gpurun_out/rNN/
set -o pipefail
TAG=${1:-r00}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_gpu.log
[ $rc -eq 124 ] && exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"
( time timeout -k 10 900 python bench.py > $O/bench_infer.json 2> $O/bench_infer.err ) 2> $O/bench_infer.time; echo "bench rc=$?"; grep real $O/bench_infer.time
SQ_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_infer_2rank_gloo.json 2>/dev/null; echo "2-rank rc=$?"
timeout -k 10 300 python bench.py --mode infer-bf16 > $O/bench_infer_bf16.json 2>/dev/null
timeout -k 10 300 python bench.py --mode train --dtype bf16 > $O/bench_train_bf16.json 2>/dev/null; echo "train rc=$?"
timeout -k 10 300 python bench.py --mode train --dtype bf16 --scaling strong --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_train_bf16_strong.json 2>/dev/null
timeout -k 10 300 python bench.py --mode train --dtype f32 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_train_f32.json 2>/dev/null
timeout -k 10 400 python bench.py --mode gan --dtype bf16 --steps 20 --warmup 5 > $O/bench_gan_bf16.json 2>/dev/null; echo "gan rc=$?"
timeout -k 10 300 python bench.py --mode gan --dtype f32 --steps 10 --warmup 4 --no-cpu-baseline > $O/bench_gan_f32.json 2>/dev/null
for m in centroids weightmap weightmap2 frontend; do timeout -k 10 300 python bench.py --mode $m > $O/bench_$m.json 2>/dev/null; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_infer -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-end-to-end --no-side-lines > $O/prof_infer.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -- python $R/bench.py --mode train --dtype bf16 --steps 30 --warmup 5 --no-cpu-baseline > $O/prof_train.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_gan -- python $R/bench.py --mode gan --dtype bf16 --steps 4 --warmup 3 --no-cpu-baseline > $O/prof_gan.log 2>&1
find $O -name "*kernel_trace.csv" -delete
ls $O
