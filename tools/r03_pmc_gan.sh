#!/bin/bash
# PMC traffic passes (FETCH_SIZE / WRITE_SIZE separately, kernel-trace only) for the GAN line alone
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03pmcgan; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/gan_$c -- python $R/bench.py --mode gan --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline > $O/gan_$c.log 2>&1 || { echo "gan $c failed"; tail -3 $O/gan_$c.log; exit 1; }
  find $O/gan_$c -name "*kernel_trace.csv" -delete
done
cd $R
python tools/pmc_step_traffic.py $O/gan_FETCH_SIZE $O/gan_WRITE_SIZE --marker adam_prepare --segments 2 --command "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --mode gan --dtype bf16 --steps 2 --warmup 1" > $O/r03_pmc_gan_bf16_traffic.json
python -c "
import json
d=json.load(open('$O/r03_pmc_gan_bf16_traffic.json'))['_summary']; print({k:(round(v/1e9,3) if isinstance(v,float) and v>1e6 else v) for k,v in d.items() if k not in ('command','correction','note')})
"
find $O -name "*counter_collection.csv" -size +30M -delete
