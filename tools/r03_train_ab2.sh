#!/bin/bash
O=gpurun_out/r03train; mkdir -p $O
run() { SQ_WGRAD_GROUP_SHRINK=$3 SQ_WGRAD_GROUP_BLOCKS=$1 SQ_WGRAD_GROUP=$2 timeout -k 10 300 python bench.py --mode train --dtype bf16 --steps 30 --warmup 5 --no-cpu-baseline > $O/t.json 2>$O/t.err
    python -c "
import json; d=json.loads(open('$O/t.json').read().strip().splitlines()[-1]); print('blocks $1 group $2 shrink $3', d['ms_per_step'])"; }
run 0 0 1
run 0 67108864 4
run 1536 67108864 1
run 0 67108864 4
run 1536 67108864 1
SQ_WGRAD_GROUP_DEBUG=1 SQ_WGRAD_GROUP_BLOCKS=1536 SQ_WGRAD_GROUP=67108864 timeout -k 10 300 python bench.py --mode train --dtype bf16 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep "^group" | sort | uniq -c | head -30
SQ_WGRAD_GROUP_DEBUG=1 SQ_WGRAD_GROUP_SHRINK=4 SQ_WGRAD_GROUP_BLOCKS=0 SQ_WGRAD_GROUP=67108864 timeout -k 10 300 python bench.py --mode train --dtype bf16 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep "^group" | sort | uniq -c | head -30
