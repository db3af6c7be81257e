#!/bin/bash
# PMC traffic passes (FETCH_SIZE / WRITE_SIZE separately, kernel-trace only) for the training and GAN lines + inference
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in train gan; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${mode}_$c -- python $R/bench.py --mode $mode --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline > $O/${mode}_$c.log 2>&1 || { echo "$mode $c failed"; tail -3 $O/${mode}_$c.log; exit 1; }
    find $O/${mode}_$c -name "*kernel_trace.csv" -delete
  done
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/infer_$c -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-side-lines > $O/infer_$c.log 2>&1 || { echo "infer $c failed"; exit 1; }
  find $O/infer_$c -name "*kernel_trace.csv" -delete
done
cd $R
python tools/pmc_step_traffic.py $O/train_FETCH_SIZE $O/train_WRITE_SIZE --marker adam_kernel --segments 1 --algorithmic-bytes 10368319488 --command "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --mode train --dtype bf16 --steps 2 --warmup 1" > $O/r04_pmc_train_bf16_traffic.json
python tools/pmc_step_traffic.py $O/gan_FETCH_SIZE $O/gan_WRITE_SIZE --marker adam_prepare --segments 2 --command "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --mode gan --dtype bf16 --steps 2 --warmup 1" > $O/r04_pmc_gan_bf16_traffic.json
python tools/pmc_traffic.py $(ls $O/infer_FETCH_SIZE/*/*_counter_collection.csv) $(ls $O/infer_WRITE_SIZE/*/*_counter_collection.csv) --steps 4 --algorithmic-bytes-per-step 6.28e9 --command "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-side-lines" > $O/r04_pmc_hbm_traffic.json
python -c "
import json
for f in ('r04_pmc_train_bf16_traffic','r04_pmc_gan_bf16_traffic','r04_pmc_hbm_traffic'):
    d=json.load(open('$O/'+f+'.json'))['_summary']; print(f, {k:(round(v/1e9,3) if isinstance(v,float) and v>1e6 else v) for k,v in d.items() if k not in ('command','correction','note')})
"
find $O -name "*counter_collection.csv" -size +30M -delete
