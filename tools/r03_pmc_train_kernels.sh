#!/bin/bash
# per-kernel HBM traffic of the bf16 training step (FETCH_SIZE / WRITE_SIZE passes) -> per-instantiation table
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03pmck; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/train_$c -- python $R/bench.py --mode train --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline > $O/train_$c.log 2>&1 || { echo "$c failed"; tail -3 $O/train_$c.log; exit 1; }
  find $O/train_$c -name "*kernel_trace.csv" -delete
done
cd $R
python tools/pmc_traffic.py $(ls $O/train_FETCH_SIZE/*/*_counter_collection.csv) $(ls $O/train_WRITE_SIZE/*/*_counter_collection.csv) --steps 3 --kernel-regex conv_mfma_bf16 > $O/train_kernels_traffic.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r03pmck/train_kernels_traffic.json'))
for k, v in sorted(d.items(), key=lambda kv: -(kv[1].get('hbm_bytes_per_launch_corrected', 0) * kv[1].get('launches', 0)) if isinstance(kv[1], dict) and 'launches' in kv[1] else 0)[:40]:
    if isinstance(v, dict) and 'launches' in v:
        print("%4d x %8.1f MB (fetch %7.1f x2, write %7.1f)  %s" % (v['launches'], v['hbm_bytes_per_launch_corrected'] / 1e6, v['fetch_kb_raw'] * 1024 / 1e6, v['write_kb'] * 1024 / 1e6, k[:110]))
PY
find $O -name "*counter_collection.csv" -delete
