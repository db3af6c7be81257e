#!/bin/bash
# re-emit the three judged bench lines after the PMC files they cite were refreshed
O=gpurun_out/r03v3; mkdir -p $O
( time timeout -k 10 900 python bench.py > $O/bench_infer.json 2> $O/bench_infer.err ) 2> $O/bench_infer.time; echo "bench rc=$?"; grep real $O/bench_infer.time
timeout -k 10 300 python bench.py --mode train --dtype bf16 > $O/bench_train_bf16.json 2>/dev/null; echo "train rc=$?"
timeout -k 10 400 python bench.py --mode gan --dtype bf16 --steps 20 --warmup 5 > $O/bench_gan_bf16.json 2>/dev/null; echo "gan rc=$?"
