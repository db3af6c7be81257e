#!/bin/bash
O=gpurun_out/r03train; mkdir -p $O
run() { SQ_CONV_BF16_MINBN=$1 timeout -k 10 300 python bench.py --mode gan --dtype bf16 --steps 30 --warmup 5 --no-cpu-baseline > $O/t.json 2>$O/t.err
    python -c "
import json; d=json.loads(open('$O/t.json').read().strip().splitlines()[-1]); print('minbn=$1', d['ms_per_step'])"; }
run 16; run 32; run 64; run 16; run 32; run 64
