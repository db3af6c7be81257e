#!/bin/bash
# training step: grouped deep-layer weight gradients on / off on one box
O=gpurun_out/r03train; mkdir -p $O
for g in 67108864 0 1000000000; do
  SQ_WGRAD_GROUP=$g timeout -k 10 300 python bench.py --mode train --dtype bf16 --steps 30 --warmup 5 --no-cpu-baseline > $O/train_$g.json 2>$O/train_$g.err; echo "group=$g rc=$?"
done
python - <<'PY'
import json
for g in (67108864, 0, 1000000000):
    try:
        d = json.loads(open("gpurun_out/r03train/train_%d.json" % g).read().strip().splitlines()[-1])
        print("SQ_WGRAD_GROUP=%d" % g, d["ms_per_step"], "ms", d["value"], d["unit"])
    except Exception as e:
        print(g, "failed", e)
PY
