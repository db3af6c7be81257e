#!/bin/bash
# Round-end validation on the GPU box: parity suite, smoke, the bench lines and the rocprof evidence.
# Usage (from the repo root on the box): bash tools/round_validate.sh rNN
set -o pipefail
TAG=${1:-r00}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout -k 10 300 python bench.py > $O/bench_infer.json 2> $O/bench_infer.err; echo "bench rc=$?"
timeout -k 10 200 python bench.py --mode train --dtype bf16 > $O/bench_train_bf16.json 2>/dev/null
timeout -k 10 200 python bench.py --mode train --dtype f32 > $O/bench_train_f32.json 2>/dev/null
timeout -k 10 300 python bench.py --mode gan --dtype f32 --steps 20 --warmup 5 > $O/bench_gan.json 2>/dev/null
timeout -k 10 300 python bench.py --mode gan --dtype bf16 --steps 20 --warmup 5 > $O/bench_gan_bf16.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_infer -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-end-to-end > $O/prof_infer.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_gan_bf16 -- python $R/bench.py --mode gan --dtype bf16 --steps 5 --warmup 2 > $O/prof_gan_bf16.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/pmc_write.log 2>&1
find $O -name "*kernel_trace.csv" -delete
ls $O
# SQ counters (MFMA-pipe busy per kernel): four more --pmc passes, summarised by tools/pmc_sq_summary.py
cd $R && bash tools/pmc_sq.sh ${TAG}_sq --no-end-to-end > $O/pmc_sq.log 2>&1 && python tools/pmc_sq_summary.py $R/gpurun_out/${TAG}_sq > $O/pmc_sq_summary.json
ls $O
