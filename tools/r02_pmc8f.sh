#!/bin/bash
# PMC traffic passes of the SURVEY 8f bench modes (FETCH_SIZE / WRITE_SIZE separately, kernel-trace only)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02pmc8f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in weightmap centroids frontend; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${mode}_$c -- python $R/bench.py --mode $mode --steps 20 --warmup 5 --no-cpu-baseline > $O/${mode}_$c.log 2>&1 || { echo "$mode $c failed"; tail -3 $O/${mode}_$c.log; exit 1; }
    find $O/${mode}_$c -name "*kernel_trace.csv" -delete
  done
done
cd $R
python tools/pmc_mode_traffic.py $O/weightmap_FETCH_SIZE $O/weightmap_WRITE_SIZE --steps 25 --kernel-regex "edt_" --algorithmic-bytes 83886080 --command "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --mode weightmap --steps 20 --warmup 5" > $O/r02_pmc_weightmap_traffic.json
python tools/pmc_mode_traffic.py $O/centroids_FETCH_SIZE $O/centroids_WRITE_SIZE --steps 25 --kernel-regex "cc_" --command "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --mode centroids --steps 20 --warmup 5" > $O/r02_pmc_centroids_traffic.json
python tools/pmc_mode_traffic.py $O/frontend_FETCH_SIZE $O/frontend_WRITE_SIZE --steps 25 --kernel-regex "frame_|tiles_norm|stitch_" --command "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --mode frontend --steps 20 --warmup 5" > $O/r02_pmc_frontend_traffic.json
python -c "
import json
for m in ('weightmap','centroids','frontend'):
    d=json.load(open('$O/r02_pmc_%s_traffic.json'%m)); print(m, round(d['_summary']['hbm_bytes_per_step']/1e6,2),'MB/step', {k:round(v['hbm_bytes_per_step']/1e6,2) for k,v in d.items() if k!='_summary'})
"
find $O -name "*counter_collection.csv" -size +30M -delete
