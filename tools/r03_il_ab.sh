#!/bin/bash
O=gpurun_out/r03train; mkdir -p $O
run() { timeout -k 10 300 python tools/bench_with_lib.py sequitr_amd/$1/libsequitr_hip.so --mode $2 --dtype bf16 --steps 40 --warmup 5 --no-cpu-baseline > $O/t.json 2>$O/t.err
    python -c "
import json; d=json.loads(open('$O/t.json').read().strip().splitlines()[-1]); print('$1 $2', d['ms_per_step'])"; }
run _build_c2 train; run _build train; run _build_c2 train; run _build train
run _build_c2 gan; run _build gan
