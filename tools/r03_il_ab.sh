#!/bin/bash
O=gpurun_out/r03train; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_gan_bf16.py tests/test_gpu_gan.py -x -q -m gpu 2>&1 | tail -3 || exit 1
run() { SQ_CONV_SPLITK=$1 timeout -k 10 300 python bench.py --mode gan --dtype bf16 --steps 30 --warmup 5 --no-cpu-baseline > $O/t.json 2>$O/t.err
    python -c "
import json; d=json.loads(open('$O/t.json').read().strip().splitlines()[-1]); print('splitk=$1', d['ms_per_step'])"; }
run 1; run 2; run 1; run 2
