#!/bin/bash
O=gpurun_out/r03train; mkdir -p $O
run() { SQ_CONV_SPLITK_BELOW=$1 SQ_CONV_SPLITK_TARGET=$2 timeout -k 10 300 python bench.py --mode gan --dtype bf16 --steps 30 --warmup 5 --no-cpu-baseline > $O/t.json 2>$O/t.err
    python -c "
import json; d=json.loads(open('$O/t.json').read().strip().splitlines()[-1]); print('below=$1 target=$2', d['ms_per_step'])"; }
run 256 512; run 1024 1536; run 1024 3072; run 256 1024; run 256 2048; run 256 512; run 1024 1536
