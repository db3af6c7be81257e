#!/bin/bash
O=gpurun_out/r03train; mkdir -p $O
run() { SQ_WGRAD_PAIR_MAJOR=$1 timeout -k 10 300 python bench.py --mode $2 --dtype bf16 --steps 40 --warmup 5 --no-cpu-baseline > $O/t.json 2>$O/t.err
    python -c "
import json; d=json.loads(open('$O/t.json').read().strip().splitlines()[-1]); print('pair_major=$1 $2', d['ms_per_step'])"; }
run 0 train; run 1 train; run 0 train; run 1 train
run 0 gan; run 1 gan
