#!/bin/bash
# per-layer HBM-side traffic and time of the grouped weight-gradient kernel (one layer per launch)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03wgl; mkdir -p $O
export SQ_WGRAD_GROUP_SHRINK=${SHRINK:-4}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -- python $R/tools/r03_wgrad_layers.py > $O/$c.log 2>&1 || { echo "$c failed"; tail -3 $O/$c.log; exit 1; }
done
cd $R
python - <<'PY'
import csv, glob, collections
SH = [(256, 16, 32), (256, 32, 32), (128, 32, 64), (128, 64, 64), (64, 64, 128), (64, 128, 128), (32, 128, 256), (32, 256, 256)]
def load(c):
    fn = glob.glob('gpurun_out/r03wgl/%s/*/*counter_collection.csv' % c)[0]
    rows = [r for r in csv.DictReader(open(fn)) if r['Counter_Name'] == c and 'wgrad_bf16_group' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    return rows
fe, wr = load('FETCH_SIZE'), load('WRITE_SIZE')
tr = glob.glob('gpurun_out/r03wgl/FETCH_SIZE/*/*kernel_trace.csv')[0]
dur = {}
for r in csv.DictReader(open(tr)):
    if 'wgrad_bf16_group' in r['Kernel_Name']:
        dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for i, (h, ci, co) in enumerate(SH):
    f = [float(r['Counter_Value']) for r in fe[4 * i: 4 * i + 4]][1:]
    w = [float(r['Counter_Value']) for r in wr[4 * i: 4 * i + 4]][1:]
    t = [dur[r['Dispatch_Id']] for r in fe[4 * i: 4 * i + 4]][1:]
    alg = 16 * h * h * (ci + co) * 2 / 1e6
    fm = 2 * sum(f) / len(f) * 1024 / 1e6
    print("%3d->%3d @%3d^2: algorithmic %6.1f MB  fetched %7.1f MB (%.2fx)  written %5.1f MB  %6.1f us (under the counter pass)  grid %s" % (
        ci, co, h, alg, fm, fm / alg, sum(w) / len(w) * 1024 / 1e6, sum(t) / len(t), fe[4 * i]['Grid_Size']))
PY
find $O -name "*.csv" -size +5M -delete
